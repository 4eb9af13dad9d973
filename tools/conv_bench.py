"""Single-layer convolution micro-benchmark (tile config sweep / ablations / rocprofv3 --pmc target).
    python tools/conv_bench.py --shape 64,56,56,64,64,3,1,1 [--cfg 7] [--iters 50]      shape = N,H,W,Cin,Cout,k,s,p
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,56,56,64,64,3,1,1")
ap.add_argument("--cfg", default="all")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--res", action="store_true")
ap.add_argument("--ws", action="store_true", help="give the launcher a split-K workspace")
ap.add_argument("--lib", default="", help="another build of the library, e.g. build/libsmelter_hip_tl.so (make ... EXTRA=-DSM_TIMELINE): sm_conv2d_time then prints per-block phase stamps to stderr")
a = ap.parse_args()
if a.lib:
    _native.LIB_PATH = os.path.abspath(a.lib)
from smelter_amd import ops  # noqa: E402
n, h, w, cin, cout, k, s, p = map(int, a.shape.split(","))
x = torch.randn((n, h, w, cin), device="cuda").half()
wt = (torch.randn((cout, k, k, cin), device="cuda") * (2.0 / (cin * k * k)) ** 0.5).half()
b = torch.randn(cout, device="cuda")
ho, wo = ops.conv_out_size(h, w, k, k, s, s, (p, p, p, p))
y = torch.empty((n, ho, wo, cout), device="cuda", dtype=torch.float16)
r = torch.randn((n, ho, wo, cout), device="cuda").half() if a.res else None
flops = 2.0 * n * ho * wo * cout * cin * k * k
byts = 2.0 * (x.numel() + y.numel() * (2 if a.res else 1) + wt.numel())
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda") if a.ws else None
def pick(tok):      # an index, or a substring of the configuration's label ("kq:", "p9:128x128")
    return [int(tok)] if tok.isdigit() else [c for c in range(ops.num_conv_configs()) if tok in ops.config_name(c)]
cfgs = range(ops.num_conv_configs()) if a.cfg == "all" else [c for tok in a.cfg.split(",") for c in pick(tok)]
for cfg in cfgs:
    try:
        ms = ops.conv2d_time(x, wt, b, residual=r, stride=(s, s), pads=(p, p, p, p), act=1, cfg=cfg, iters=a.iters, out=y, workspace=ws)
    except Exception as e:
        continue
    us = ms * 1e3
    print("cfg %3d  %8.1f us  %7.1f TFLOP/s  %6.2f TB/s (compulsory)  %s" % (cfg, us, flops / us / 1e6, byts / us / 1e6, ops.config_name(cfg)))
