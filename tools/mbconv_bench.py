"""fused inverted-residual block vs its three layers, MobileNet-v2 bs=256 shapes (cold caches between timed launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
def timed(fn, reps=5):
    ts = []
    for i in range(reps + 1):
        flush.fill_(i & 255)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
n = 256
for hw, cin, ch, cout, s, res in ((112, 16, 96, 24, 2, False), (56, 24, 144, 24, 1, True), (56, 24, 144, 32, 2, False), (28, 32, 192, 32, 1, True), (28, 32, 192, 64, 2, False), (112, 32, 32, 16, 1, False),
                                    (14, 64, 384, 64, 1, True), (14, 64, 384, 96, 1, False), (14, 96, 576, 96, 1, True), (14, 96, 576, 160, 2, False), (7, 160, 960, 160, 1, True), (7, 160, 960, 320, 1, False))[int(os.environ.get("MB_FROM", "0")):]:
    from smelter_amd._native import MbConvDesc
    import ctypes
    if not ops.lib().sm_mbconv_supported(ctypes.byref(MbConvDesc(n, hw, hw, (hw + 2 - 3) // s + 1, (hw + 2 - 3) // s + 1, s, cin, ch, cout, 1 if ch != cin else 0, 1 if res else 0, 1, 0.0, 1, 0.0, 0, 0.0, cin, cout))):
        print("%3dx%-3d %2d->%3d->%2d s%d : not supported" % (hw, hw, cin, ch, cout, s)); continue
    expand = ch != cin
    x = torch.randn((n, hw, hw, cin), device="cuda").half()
    w1 = (torch.randn((ch, 1, 1, cin), device="cuda") * (2.0 / cin) ** 0.5).half(); b1 = torch.randn(ch, device="cuda") * 0.1
    wd = (torch.randn((ch, 3, 3, 1), device="cuda") * 0.47).half(); bd = torch.randn(ch, device="cuda") * 0.1
    w2 = (torch.randn((cout, 1, 1, ch), device="cuda") * (1.0 / ch) ** 0.5).half(); b2 = torch.randn(cout, device="cuda") * 0.1
    def three():
        hdn = ops.conv2d(x, w1, b1, act=1) if expand else x
        h2 = ops.conv2d(hdn, wd, bd, stride=(s, s), pads=(1, 1, 1, 1), groups=ch, act=1)
        return ops.conv2d(h2, w2, b2, residual=x if res else None)
    tf = timed(lambda: ops.mbconv(x, w1 if expand else None, b1 if expand else None, wd.reshape(ch, 3, 3), bd, w2, b2, stride=s, residual=res))
    t3 = timed(three)
    ho = (hw + 2 - 3) // s + 1
    byts = 2.0 * n * (hw * hw * cin * (2 if res else 1) + ho * ho * cout)
    print("%3dx%-3d %2d->%3d->%2d s%d%s : three launches %7.1f us ; fused %7.1f us (%.2f TB/s on the block's own %.0f MB)" % (hw, hw, cin, ch, cout, s, " +res" if res else "", t3, tf, byts / tf / 1e6, byts / 1e6))
