"""Development probe for a GPU box: kernel parity vs the CPU oracle, graph parity, first ResNet-50 timings.
Run:  python tools/gpu_probe.py [--r50] [--ops]      (writes nothing; prints)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import oracle as O  # noqa: E402
from smelter_amd import ops, zoo  # noqa: E402
from smelter_amd.graph import Configuration, ONNXGraph  # noqa: E402


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def conv_case(n, h, w, cin, cout, k, s, p, res=False, act=1, cfg=None, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    xh = O.round_to_half(x); wh = O.round_to_half(wt)
    ref = O.conv2d(xh, wh, b, stride=(s, s), pads=(p, p, p, p))
    r = None
    if res:
        r = O.round_to_half(rng.standard_normal(ref.shape).astype(np.float32)); ref = ref + r
    if act:
        ref = np.maximum(ref, 0)
    xd = torch.from_numpy(xh.transpose(0, 2, 3, 1).copy()).cuda().half()
    wd = torch.from_numpy(wh).cuda().half()
    bd = torch.from_numpy(b).cuda()
    rd = torch.from_numpy(r.transpose(0, 2, 3, 1).copy()).cuda().half() if res else None
    y = ops.conv2d(xd, wd, bd, residual=rd, stride=(s, s), pads=(p, p, p, p), act=act, cfg=cfg)
    torch.cuda.synchronize()
    got = y.float().cpu().numpy().transpose(0, 3, 1, 2)
    return relerr(got, ref)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--r50", action="store_true")
    ap.add_argument("--ops", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    print("device", torch.cuda.get_device_name(0))
    for cfg in [None] + list(range(ops.num_conv_configs())):
        errs = [conv_case(2, 14, 14, 64, 96, 1, 1, 0, cfg=cfg), conv_case(2, 15, 13, 32, 72, 3, 1, 1, res=True, cfg=cfg),
                conv_case(1, 17, 17, 16, 40, 3, 2, 1, cfg=cfg), conv_case(3, 9, 9, 8, 8, 7, 2, 3, cfg=cfg),
                conv_case(2, 28, 28, 128, 256, 1, 2, 0, cfg=cfg)]
        print("conv cfg", cfg, ["%.2e" % e for e in errs])
    print("direct (cin=3)", "%.2e" % conv_case(2, 16, 16, 3, 8, 3, 1, 1))

    for name, m, shape in (("toy", zoo.toy(), (2, 3, 64, 64)), ("unet", zoo.unet(hw=(64, 64)), (2, 3, 64, 64)),
                           ("r50-small", zoo.resnet50(hw=(64, 64)), (2, 3, 64, 64)), ("mbv2-small", zoo.mobilenet_v2(hw=(64, 64)), (2, 3, 64, 64))):
        x = zoo.synthetic_images(shape[0], 3, shape[2:])
        ref32 = O.Graph(m).run(x)
        ref16 = O.Graph(m, act_fp16=True).run(x)
        for fuse in (True, False):
            g = ONNXGraph(m, Configuration(fuse=fuse, useHipGraph=fuse))
            ex = g.hipGraph(0, shape[0])
            xd = torch.from_numpy(x).cuda()
            ex.encode(xd); got = ex.toFloatArray()
            print(name, "fuse", fuse, "ops", ex.numOps, "launches", ex.numLaunches, "err vs fp32 oracle %.2e" % relerr(got, ref32),
                  "vs fp16-storage oracle %.2e" % relerr(got, ref16))

    if a.r50:
        t = time.time(); m = zoo.resnet50(); print("gen %.1fs" % (time.time() - t))
        for autotune in (False, True):
            t = time.time(); g = ONNXGraph(m, Configuration(autotune=autotune)); ex = g.hipGraph(0, a.batch); print("compile %.2fs" % (time.time() - t))
            x = torch.from_numpy(zoo.synthetic_images(a.batch)).cuda()
            for _ in range(5):
                ex.encode(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ex.encode(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print("R50 bs=%d autotune=%s: %.3f ms/iter  %.0f img/s  %.1f TFLOP/s (%.1f%% of 2.5PF)  act arena %.0f MB launches %d" % (
                a.batch, autotune, ms, a.batch / ms * 1e3, ex.flops / ms / 1e9, ex.flops / ms / 1e9 / 2500 * 100, ex.activationBytes / 1e6, ex.numLaunches))
            if a.ops:
                tot = 0
                for i in range(ex.numOps):
                    t_ms = ex.timeOp(i, 10); tot += t_ms
                    print("  %7.1f us  %s" % (t_ms * 1e3, ex.opDesc(i)))
                print("  sum of ops %.3f ms" % tot)
        ref = O.Graph(m).run(zoo.synthetic_images(a.batch)[:2])
        got = ex.toFloatArray()[:2]
        print("R50 parity (first 2 images) rel err %.2e" % relerr(got, ref))


if __name__ == "__main__":
    main()
