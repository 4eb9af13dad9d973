"""conv1x1_pair vs two launches on ResNet-50 stage-1 shapes (cold caches between launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
n, hh = 64, 56
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
for n2 in (64, 128):
    x = torch.randn((n, hh, hh, 64), device="cuda").half(); r = torch.randn((n, hh, hh, 256), device="cuda").half()
    w1 = (torch.randn((256, 1, 1, 64), device="cuda") * 0.17).half(); b1 = torch.randn(256, device="cuda")
    w2 = (torch.randn((n2, 1, 1, 256), device="cuda") * 0.09).half(); b2 = torch.randn(n2, device="cuda")
    y = torch.empty((n, hh, hh, 256), device="cuda", dtype=torch.float16); z = torch.empty((n, hh, hh, n2), device="cuda", dtype=torch.float16)
    def timed(fn, reps=8):
        ts = []
        for i in range(reps + 1):
            flush.fill_(i & 255)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            if i: ts.append(a.elapsed_time(b) * 1e3)
        ts.sort(); return ts[len(ts) // 2]
    best1 = min(timed(lambda c=c: ops.conv2d(x, w1, b1, residual=r, act=1, cfg=c, out=y)) for c in (6, 7, 8, 9, 13, 14, 18, 24, 25, 26, 27, 29))
    best2 = min(timed(lambda c=c: ops.conv2d(y, w2, b2, act=1, cfg=c, out=z)) for c in (6, 7, 9, 12, 13, 14, 15, 18, 20))
    tp = timed(lambda: ops.conv1x1_pair(x, w1, b1, r, w2, b2, out_y=y, out_z=z))
    byts = 2.0 * (x.numel() + r.numel() + y.numel() + z.numel())
    print("N2=%d: separate %.1f + %.1f = %.1f us ; pair %.1f us (%.2f TB/s on %.0f MB)" % (n2, best1, best2, best1 + best2, tp, byts / tp / 1e6, byts / 1e6))
