"""Timeline of ONE sm_bottleneck_fwd launch (diagnostic build with -DSM_BN_TL, tools/bneck_tl.sh builds it): s_memtime stamps of wave 0 (3x3 + expand) and wave 4 (reducer)
of every block; prints the median over blocks of every stamp relative to the block's own first stamp, in clocks (s_memtime ticks at 100 MHz on gfx950: x 24 = core clocks).
    SMELTER_LIB_PATH=build/libsmelter_hip_bntl.so python tools/bneck_tl.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smelter_amd import ops
n, hw, c0 = 64, 56, 256
torch.manual_seed(0)
x = torch.randn((n, hw, hw, c0), device="cuda").half()
w1 = (torch.randn((64, c0), device="cuda") * (2.0 / c0) ** 0.5).half(); b1 = torch.randn(64, device="cuda") * 0.1
w2 = (torch.randn((64, 3, 3, 64), device="cuda") * (2.0 / 576) ** 0.5).half(); b2 = torch.randn(64, device="cuda") * 0.1
w3 = (torch.randn((c0, 64), device="cuda") * (2.0 / 64) ** 0.5).half(); b3 = torch.randn(c0, device="cuda") * 0.1
bufs = [x.clone(), torch.empty_like(x)]
L = ops.lib()
for i in range(6): ops.bottleneck(bufs[i & 1], w1, b1, w2, b2, w3, b3, residual=bufs[i & 1], out=bufs[(i + 1) & 1])
torch.cuda.synchronize()
L.sm_bneck_timeline_clear()
ops.bottleneck(bufs[0], w1, b1, w2, b2, w3, b3, residual=bufs[0], out=bufs[1])
torch.cuda.synchronize()
tl = np.zeros(256 * 2 * 64, np.uint64)
assert L.sm_bneck_timeline_read(tl.ctypes.data_as(C.c_void_p)) == 0
tl = tl.reshape(256, 2, 64).astype(np.int64)
t0 = np.where(tl > 0, tl, np.iinfo(np.int64).max).min(axis=(1, 2), keepdims=True)      # the block's first stamp
rel = np.where(tl > 0, tl - t0, -1)
span = (tl.max() - np.where(tl > 0, tl, np.iinfo(np.int64).max).min())
print("whole launch (first stamp of any block -> last stamp of any block): %d ticks" % span)
names_pc = ["start", "3x3 loop done", "3x3 epilogue done", "expand group 0 done", "expand group 1 done"]
names_rd = ["start", "unit 0 MFMAs done", "unit 0 written", "unit 1 MFMAs done", "unit 1 written"]
for q in range(9):
    for role, names in ((1, names_rd), (0, names_pc)):
        for k in range(5):
            v = rel[:, role, q * 5 + k]; v = v[v >= 0]
            if v.size: print("period %d %-8s %-22s median %6d  min %6d  max %6d  (blocks %d)" % (q, "reducer" if role else "pc", names[k], np.median(v), v.min(), v.max(), v.size))
