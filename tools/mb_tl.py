"""SM_TIMELINE build only: clock64() per phase of the fused inverted-residual kernel (thread 0 of block 0), MobileNet-v2 bs=256 shapes"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
raw = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smelter_amd", "libsmelter_hip.so"))
n = 256
SHAPES = ((112, 16, 96, 24, 2, False), (56, 24, 144, 24, 1, True), (28, 32, 192, 32, 1, True), (112, 32, 32, 16, 1, False), (14, 64, 384, 64, 1, True), (14, 96, 576, 96, 1, True), (7, 160, 960, 160, 1, True))
for hw, cin, ch, cout, s, res in SHAPES[int(os.environ.get('MB_FROM', '0')):]:
    expand = ch != cin
    x = torch.randn((n, hw, hw, cin), device="cuda").half()
    w1 = (torch.randn((ch, 1, 1, cin), device="cuda") * (2.0 / cin) ** 0.5).half(); b1 = torch.randn(ch, device="cuda") * 0.1
    wd = (torch.randn((ch, 3, 3), device="cuda") * 0.47).half(); bd = torch.randn(ch, device="cuda") * 0.1
    w2 = (torch.randn((cout, 1, 1, ch), device="cuda") * (1.0 / ch) ** 0.5).half(); b2 = torch.randn(cout, device="cuda") * 0.1
    f = lambda: ops.mbconv(x, w1 if expand else None, b1 if expand else None, wd, bd, w2, b2, stride=s, residual=res)
    f(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    raw.sm_mbconv_debug_timeline(out, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    raw.sm_mbconv_debug_timeline(out, 0)
    v = list(out)
    print("%dx%d %d->%d->%d s%d: %.1f us; clk: top+patch %d | A %d | barrier %d | B %d | barrier %d | C %d | epilogue %d | [7] %d | total %d" % (
        hw, hw, cin, ch, cout, s, a.elapsed_time(b) * 1e3, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], sum(v[:8])))
