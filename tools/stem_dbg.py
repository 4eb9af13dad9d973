"""scratch: full-size model with the loader-wave stem, step-by-step sync"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import zoo
from smelter_amd.graph import Configuration, ONNXGraph
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
at = len(sys.argv) > 2
model = zoo.resnet50()
x = torch.from_numpy(zoo.synthetic_images(bs, 3, (224, 224))).cuda()
ex = ONNXGraph(model, Configuration(useHipGraph=False, autotune=at)).hipGraph(0, bs)
print("compiled", file=sys.stderr)
ex.encode(x)
print(ex.toFloatArray().ravel()[:4])
