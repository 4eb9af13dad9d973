import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from smelter_amd import ops, _native as N
import ctypes as C
torch.manual_seed(0)
n, c, h, w, cout, k, s, p = 2, 3, 20, 24, 8, 3, 1, 1
img = torch.randn(n, c, h, w, device="cuda")
wt = torch.randn(cout, c, k, k, device="cuda") * 0.2
ref = F.conv2d(img.half().float(), wt.half().float(), stride=s, padding=p)
ho, wo = ref.shape[2], ref.shape[3]
P = (k + 1) // 2
Hs, Ws = (ho - 1) * s + 2 * (P - 1) + 1, (wo - 1) * s + 8
x2 = torch.zeros(n * Hs * Ws * 8 + 64, dtype=torch.float16, device="cuda")
N.check(N.lib().sm_nchw_f32_to_rowpair_f16(C.c_void_p(img.data_ptr()), C.c_void_p(x2.data_ptr()), n, c, h, w, Hs, Ws, p, p, None))
torch.cuda.synchronize()
X2 = x2[:n * Hs * Ws * 8].view(n, Hs, Ws, 8).float().cpu()
# check the layout against a direct construction
imgp = F.pad(img.half().float().cpu(), (p, Ws, p, Hs))
ok = True
for half in (0, 1):
    for cc in range(c):
        exp = imgp[:, cc, half:half + Hs, 0:Ws]
        if not torch.equal(exp, X2[..., half * 4 + cc]): ok = False
print("layout ok", ok)
wst = torch.zeros(cout, P, 8, 2, 4)
for kh in range(k):
    for kw in range(k):
        wst[:, kh // 2, kw, kh & 1, :c] = wt[:, :, kh, kw].cpu()
wst = wst.reshape(cout, P, 1, 64).half().cuda().contiguous()
xv = torch.as_strided(x2, (n, Hs, Ws, 64), (Hs * Ws * 8, Ws * 8, 8, 1))
out = torch.empty((n, ho, wo, cout), dtype=torch.float16, device="cuda")
d = N.ConvDesc(n, Hs, Ws, 64, cout, P, 1, s, s, 2, 1, 0, 0, 0, 0, 1, ho, wo, 0, 0.0, 8, cout, 0)
for cfg in (-1, 9, 3):
    N.check(N.lib().sm_conv2d_fwd_cfg(C.byref(d), cfg, C.c_void_p(x2.data_ptr()), C.c_void_p(wst.data_ptr()), None, None, C.c_void_p(out.data_ptr()), None, 0, None))
    torch.cuda.synchronize()
    got = out.float().permute(0, 3, 1, 2)
    print("cfg", cfg, "max err", float((got - ref).abs().max()), "ref max", float(ref.abs().max()))
