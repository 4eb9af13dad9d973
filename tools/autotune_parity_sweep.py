"""Autotuned executors at several batch sizes / image sizes against the CPU oracle (first images only): catches tile-config-specific
bugs that the fixed-shape tests cannot see because the autotuner picks other configs there.   python tools/autotune_parity_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle as O
from smelter_amd import zoo
from smelter_amd.graph import Configuration, ONNXGraph

CASES = [("r50", lambda hw: zoo.resnet50(hw=hw), (224, 224), [1, 7, 32]), ("r50", lambda hw: zoo.resnet50(hw=hw), (96, 160), [5]),
         ("mbv2", lambda hw: zoo.mobilenet_v2(hw=hw), (224, 224), [3, 64]), ("mbv2", lambda hw: zoo.mobilenet_v2(hw=hw), (128, 96), [16]),
         ("unet", lambda hw: zoo.unet(hw=hw), (256, 256), [2]), ("unet", lambda hw: zoo.unet(hw=hw), (96, 64), [9])]
bad = 0
for name, make, hw, batches in CASES:
    m = make(hw)
    og = O.Graph(m)
    for bs in batches:
        x = zoo.synthetic_images(bs, 3, hw, seed=bs)
        ex = ONNXGraph(m, Configuration(autotune=True)).hipGraph(0, bs)
        ex.encode(torch.from_numpy(x).cuda())
        got = ex.toFloatArray()
        k = min(bs, 2)
        ref = og.run(x[:k])
        e = float(np.abs(got[:k] - ref).max() / np.abs(ref).max())
        cfgs = sorted({ex.opDesc(i).split("cfg=")[1] for i in range(ex.numOps) if "cfg=" in ex.opDesc(i)}, key=int)
        ok = e < 5e-3
        bad += not ok
        print("%-5s %3dx%-3d bs=%-3d nerr %.2e %s  configs used: %s" % (name, hw[0], hw[1], bs, e, "ok" if ok else "FAIL", ",".join(cfgs)))
sys.exit(1 if bad else 0)
