import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smelter_amd import zoo
from smelter_amd.graph import Configuration, ONNXGraph
m = zoo.resnet50()
x = torch.from_numpy(zoo.synthetic_images(64)).cuda()
outs = {}
for bc in (False, True):
    g = ONNXGraph(m, Configuration(autotune=True, branchConcurrency=bc)); ex = g.hipGraph(0, 64)
    for _ in range(10): ex.encode(x)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ex.encode(x)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30)
    outs[bc] = ex.toFloatArray()
    side = [ex.opDesc(i) for i in range(ex.numOps) if "side-stream" in ex.opDesc(i)]
    print("branchConcurrency=%s: %.3f ms/step, %.0f img/s, act arena %.0f MB, side steps %d" % (bc, best, 64 / best * 1e3, ex.activationBytes / 1e6, len(side)))
    for s in side: print("   ", s)
print("max abs diff between the two schedules:", float(np.abs(outs[True] - outs[False]).max()))
