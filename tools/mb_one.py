"""one fused inverted-residual launch pair (warm-up + measured) for rocprofv3: python tools/mb_one.py [shape index]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
SH = ((112, 16, 96, 24, 2, False), (56, 24, 144, 24, 1, True), (28, 32, 192, 32, 1, True), (112, 32, 32, 16, 1, False))
hw, cin, ch, cout, s, res = SH[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
n = 256; expand = ch != cin
x = torch.randn((n, hw, hw, cin), device="cuda").half()
w1 = (torch.randn((ch, 1, 1, cin), device="cuda") * (2.0 / cin) ** 0.5).half(); b1 = torch.randn(ch, device="cuda") * 0.1
wd = (torch.randn((ch, 3, 3), device="cuda") * 0.47).half(); bd = torch.randn(ch, device="cuda") * 0.1
w2 = (torch.randn((cout, 1, 1, ch), device="cuda") * (1.0 / ch) ** 0.5).half(); b2 = torch.randn(cout, device="cuda") * 0.1
for _ in range(3):
    ops.mbconv(x, w1 if expand else None, b1 if expand else None, wd, bd, w2, b2, stride=s, residual=res)
torch.cuda.synchronize()
