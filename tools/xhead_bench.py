"""expand head (3x3 64->64 -> 1x1 64->256 + residual + ReLU in one launch) vs the two launches, ResNet-50 stage-1 shape, cold caches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
n, hh = 64, 56
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
x = torch.randn((n, hh, hh, 64), device="cuda").half(); r = torch.randn((n, hh, hh, 256), device="cuda").half()
w = (torch.randn((64, 3, 3, 64), device="cuda") * 0.06).half(); b = torch.randn(64, device="cuda")
w3 = (torch.randn((256, 1, 1, 64), device="cuda") * 0.12).half(); b3 = torch.randn(256, device="cuda")
y = torch.empty((n, hh, hh, 64), device="cuda", dtype=torch.float16); z = torch.empty((n, hh, hh, 256), device="cuda", dtype=torch.float16)
COLD = os.environ.get("WARM") is None
def timed(fn, reps=8):
    ts = []
    for i in range(reps + 1):
        if COLD: flush.fill_(i & 255)
        else: y.zero_()                                     # (something between the launches, small enough to leave the operands in the Infinity Cache)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        if i: ts.append(a.elapsed_time(e) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
t3 = {c: timed(lambda c=c: ops.conv2d(x, w, b, pads=(1, 1, 1, 1), act=1, cfg=c, out=y)) for c in (61, 65)}
t1 = {c: timed(lambda c=c: ops.conv2d(y, w3, b3, residual=r, act=1, cfg=c, out=z)) for c in (6, 7, 9, 13, 14, 18, 24, 25, 26, 27, 29)}
XC = int(os.environ.get("XCFG", "94"))
tx = timed(lambda: ops.conv2d_xhead(x, w, b, w3, b3, residual=r, cfg=XC))
tn = timed(lambda: ops.conv2d_xhead(x, w, b, w3, b3, residual=None, cfg=XC))
print("3x3:", {k: round(v, 1) for k, v in t3.items()})
print("1x1+res:", {k: round(v, 1) for k, v in t1.items()})
byts = 2.0 * (x.numel() + r.numel() + z.numel())
print("cfg", XC, "separate %.1f + %.1f = %.1f us ; xhead %.1f us (%.2f TB/s on %.0f MB) ; xhead without residual %.1f us" % (min(t3.values()), min(t1.values()), min(t3.values()) + min(t1.values()), tx, byts / tx / 1e6, byts / 1e6, tn))
