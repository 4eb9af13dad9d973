import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smelter_amd import zoo
from smelter_amd.graph import Configuration, ONNXGraph
m = zoo.toy()
x = zoo.synthetic_images(2, 3, (64, 64))
xd = torch.from_numpy(x).cuda()
outs = []
for keep in (True, False):
    ex = ONNXGraph(m, Configuration(useHipGraph=False, keepIntermediates=keep)).hipGraph(0, 2)
    ex.encode(xd); torch.cuda.synchronize()
    y = ex.toFloatArray()
    print("keep", keep, [ex.opDesc(i) for i in range(ex.numOps)][:2], y.shape, float(np.abs(y).max()), np.isfinite(y).all())
    outs.append(y)
d = np.abs(outs[0] - outs[1])
print("max diff", d.max(), "of", np.abs(outs[0]).max(), "argmax", np.unravel_index(d.argmax(), d.shape))
print(outs[0][0, 0, :3, :6]); print(outs[1][0, 0, :3, :6])
