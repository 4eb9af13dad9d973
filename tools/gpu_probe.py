"""Development probe for a GPU box: ResNet-50 (or another zoo model) step time and per-op timings.
    python tools/gpu_probe.py [--model r50|mbv2|unet] [--batch N] [--ops] [--no-autotune]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import oracle as O  # noqa: E402
from smelter_amd import zoo  # noqa: E402
from smelter_amd.graph import Configuration, ONNXGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="r50")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--ops", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--insitu", action="store_true")
    a = ap.parse_args()
    make, hw, bs = {"r50": (zoo.resnet50, (224, 224), 64), "mbv2": (zoo.mobilenet_v2, (224, 224), 256), "unet": (zoo.unet, (512, 512), 16)}[a.model]
    bs = a.batch or bs
    t = time.time(); m = make(); print("gen %.1fs" % (time.time() - t))
    t = time.time(); g = ONNXGraph(m, Configuration(autotune=not a.no_autotune)); ex = g.hipGraph(0, bs); print("compile %.2fs" % (time.time() - t))
    x = torch.from_numpy(zoo.synthetic_images(bs, 3, hw)).cuda()
    for _ in range(5):
        ex.encode(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ex.encode(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s bs=%d: %.3f ms/iter  %.0f img/s  %.1f TFLOP/s (%.1f%% of 2.5PF)  alg bytes %.2f GB -> %.2f TB/s  act arena %.0f MB launches %d" % (
        a.model, bs, ms, bs / ms * 1e3, ex.flops / ms / 1e9, ex.flops / ms / 1e9 / 2500 * 100, ex.algorithmicBytes / 1e9, ex.algorithmicBytes / ms / 1e9,
        ex.activationBytes / 1e6, ex.numLaunches))
    if a.check:
        got = ex.toFloatArray()[:1]
        ref = O.Graph(m).run(x[:1].cpu().numpy())
        print("parity (first image) nerr %.2e" % (np.abs(got - ref).max() / np.abs(ref).max()))
    if a.insitu:
        import re
        prof = ex.profile(x, 20)
        print("  in-graph (cold) per-op: us | roofline-bound us (max of bytes/6.5TB/s, flops/1.6PF) | ratio")
        tot = tb = 0
        for i in range(ex.numOps):
            d = ex.opDesc(i); us = prof[i + 1] * 1e3; tot += us
            mm = re.search(r"M=(\d+) N=(\d+) K=(\d+) k=(\d+)x\d+ s=(\d+) g=(\d+)", d)
            if mm:
                M, Nn, K, k, s, gr = map(int, mm.groups())
                cin = K // (k * k) * gr
                by = 2 * (M * s * s * cin + M * Nn * (2 if "+res" in d else 1) + K * Nn)
                fl = 2.0 * M * Nn * K
                bound = max(by / 6.5e12, fl / 1.6e15) * 1e6
                tb += bound
                print("  %7.1f %7.1f %5.2fx  %6.0f TF/s %5.2f TB/s  %s" % (us, bound, us / bound, fl / us / 1e6, by / us / 1e6, d))
            else:
                print("  %7.1f                 %s" % (us, d))
        print("  sum %.1f us, conv bound sum %.1f us; input conversion %.1f us" % (tot, tb, prof[0] * 1e3))
    if a.ops:
        tot = 0
        for i in range(ex.numOps):
            t_ms = ex.timeOp(i, 10); tot += t_ms
            print("  %7.1f us  %s" % (t_ms * 1e3, ex.opDesc(i)))
        print("  sum of ops %.3f ms" % tot)


if __name__ == "__main__":
    main()
