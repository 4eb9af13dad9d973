"""scratch: MobileNet-v2 96x96 with the block fusion off, step-by-step sync (SMELTER_DEBUG_SYNC=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import zoo
from smelter_amd.graph import Configuration, ONNXGraph
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 96
model = zoo.mobilenet_v2(hw=(hw, hw))
x = zoo.synthetic_images(2, 3, (hw, hw))
ex = ONNXGraph(model, Configuration(useHipGraph=False)).hipGraph(0, 2)
for i in range(ex.numOps): print(i, ex.opDesc(i), file=sys.stderr)
ex.encode(torch.from_numpy(x).cuda())
print(ex.toFloatArray().ravel()[:4])
