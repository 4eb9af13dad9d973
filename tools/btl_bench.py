"""whole bottleneck block in one launch (sm_bottleneck_fwd) vs its three launches, ResNet-50 bs=64 shapes (cold caches between timed launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
def timed(fn, reps=7, cold=True):
    ts = []
    for i in range(reps + 1):
        if cold: flush.fill_(i & 255)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
n = 64
for hw, cm in ((56, 64), (28, 128)):
    c = 4 * cm
    x = torch.randn((n, hw, hw, c), device="cuda").half()
    w1 = (torch.randn((cm, 1, 1, c), device="cuda") * (2.0 / c) ** 0.5).half(); b1 = torch.randn(cm, device="cuda") * 0.1
    w2 = (torch.randn((cm, 3, 3, cm), device="cuda") * (2.0 / (9 * cm)) ** 0.5).half(); b2 = torch.randn(cm, device="cuda") * 0.1
    w3 = (torch.randn((c, 1, 1, cm), device="cuda") * (1.0 / cm) ** 0.5).half(); b3 = torch.randn(c, device="cuda") * 0.1
    def three():
        t1 = ops.conv2d(x, w1, b1, act=1)
        t2 = ops.conv2d(t1, w2, b2, pads=(1, 1, 1, 1), act=1)
        return ops.conv2d(t2, w3, b3, residual=x, act=1)
    y = torch.empty_like(x)
    f = lambda: ops.bottleneck(x, w1, b1, w2, b2, w3, b3, out=y)
    print("%dx%d C=%d Cm=%d: three launches (default configs) %7.1f us cold ; fused %7.1f us cold, %7.1f us warm (%.2f TB/s on x + y = %.0f MB)" % (
        hw, hw, c, cm, timed(three), timed(f), timed(f, cold=False), 2.0 * x.numel() * 2 / timed(f) / 1e6, 2.0 * x.numel() * 2 / 1e6))
