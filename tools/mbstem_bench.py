"""MobileNet-v2's first three layers (stem 3x3/2 3->32, depthwise 3x3, project 32->16) at bs = 256: fused launch vs stem + block"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
n = 256
x = torch.randn((n, 3, 224, 224), device="cuda")
ws = (torch.randn((32, 3, 3, 3), device="cuda") * 0.27); bs = torch.randn(32, device="cuda") * 0.1
wd = (torch.randn((32, 3, 3), device="cuda") * 0.47).half(); bd = torch.randn(32, device="cuda") * 0.1
w2 = (torch.randn((16, 1, 1, 32), device="cuda") * 0.18).half(); b2 = torch.randn(16, device="cuda") * 0.1
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
def timed(fn, reps=6):
    ts = []
    for i in range(reps + 1):
        flush.fill_(i & 255)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        if i: ts.append(a.elapsed_time(e) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
L = ops.lib()
# (the separate stem: the first loader-wave configuration for 32 output channels, by its label — indices move when configurations are added)
lw = [c for c in range(L.sm_conv2d_num_configs()) if L.sm_conv2d_config_loader_waves(c) > 0 and "->32" in ops.config_name(c)][0]
t_stem = timed(lambda: ops.stem_conv(x, ws.cpu(), bs, stride=2, pads=(1, 1, 1, 1), act=1, cfg=lw))
h1 = ops.stem_conv(x, ws.cpu(), bs, stride=2, pads=(1, 1, 1, 1), act=1, cfg=lw)
t_blk = timed(lambda: ops.mbconv(h1, None, None, wd, bd, w2, b2, stride=1, residual=False))
t_fused = timed(lambda: ops.mbconv_stem(x, ws.half(), bs, wd, bd, w2, b2))
print("stem (%s) %.1f us + block %.1f us = %.1f us ; fused %.1f us" % (ops.config_name(lw), t_stem, t_blk, t_stem + t_blk, t_fused))
