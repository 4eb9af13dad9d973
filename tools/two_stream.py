"""Experiment: the bs-64 batch as TWO half-batch executors replayed concurrently on two streams (each kernel of a half batch fills at most half of the CUs of
the 1-block-per-CU layers, so the other stream's kernel runs beside it and one stream's launch boundaries / prologues / tails overlap the other's work).
    python tools/two_stream.py [--model r50] [--batch 64] [--ways 2] [--steps 50]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smelter_amd import zoo
from smelter_amd.graph import Configuration, ONNXGraph

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="r50")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ways", type=int, default=2)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
make, hw = {"r50": (zoo.resnet50, (224, 224)), "mbv2": (zoo.mobilenet_v2, (224, 224)), "unet": (zoo.unet, (512, 512))}[a.model]
model = make()


def timed(fn, steps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    out.sort()
    return out[len(out) // 2], out[0]


x = torch.from_numpy(zoo.synthetic_images(a.batch, 3, hw)).cuda()
ex = ONNXGraph(model, Configuration(autotune=True)).hipGraph(0, a.batch)
s0 = torch.cuda.Stream()
med, mn = timed(lambda: ex.encode(x, s0), a.steps)
print("one executor  bs=%d            : %.4f ms (min %.4f)  %.0f img/s" % (a.batch, med, mn, a.batch / med * 1e3))
ref = None
ex.encode(x, s0); torch.cuda.synchronize(); ref = ex.toFloatArray()
del ex
sub = a.batch // a.ways
g0 = ONNXGraph(model, Configuration(autotune=True))
e0 = g0.hipGraph(0, sub)
plan = e0.plan
execs = [e0] + [ONNXGraph(model, Configuration(autotune=True)).hipGraph(0, sub, plan=plan) for _ in range(a.ways - 1)]
streams = [torch.cuda.Stream() for _ in range(a.ways)]
xs = [x[i * sub:(i + 1) * sub].contiguous() for i in range(a.ways)]
med1, mn1 = timed(lambda: execs[0].encode(xs[0], streams[0]), a.steps)
print("one executor  bs=%d alone      : %.4f ms (min %.4f)" % (sub, med1, mn1))


def both():
    for e, xx, s in zip(execs, xs, streams):
        e.encode(xx, s)


med2, mn2 = timed(both, a.steps)
print("%d executors x bs=%d concurrent : %.4f ms (min %.4f)  %.0f img/s   (%.1f %% vs one executor)" % (a.ways, sub, med2, mn2, a.batch / med2 * 1e3, (med2 / med - 1) * 100))
both(); torch.cuda.synchronize()
got = np.concatenate([e.toFloatArray() for e in execs])
print("max |two-stream - one-executor| / max|ref| = %.2e" % (np.abs(got - ref).max() / np.abs(ref).max()))
