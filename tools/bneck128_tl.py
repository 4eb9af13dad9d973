"""Timeline of ONE bneck128 launch (diagnostic build -DSM_BN_TL): s_memtime stamps of the four waves of every block; median over blocks relative to the block's first stamp.
    SMELTER_LIB_PATH=build/libsmelter_hip_bntl.so python tools/bneck128_tl.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smelter_amd import ops
n, hw, c0, cm = 64, 28, 512, 128
torch.manual_seed(0)
x = torch.randn((n, hw, hw, c0), device="cuda").half()
w1 = (torch.randn((cm, c0), device="cuda") * (2.0 / c0) ** 0.5).half(); b1 = torch.randn(cm, device="cuda") * 0.1
w2 = (torch.randn((cm, 3, 3, cm), device="cuda") * (2.0 / (9 * cm)) ** 0.5).half(); b2 = torch.randn(cm, device="cuda") * 0.1
w3 = (torch.randn((c0, cm), device="cuda") * (2.0 / cm) ** 0.5).half(); b3 = torch.randn(c0, device="cuda") * 0.1
bufs = [x.clone(), torch.empty_like(x)]
L = ops.lib()
for i in range(6): ops.bottleneck(bufs[i & 1], w1, b1, w2, b2, w3, b3, residual=bufs[i & 1], out=bufs[(i + 1) & 1])
torch.cuda.synchronize()
L.sm_bneck_timeline_clear()
ops.bottleneck(bufs[0], w1, b1, w2, b2, w3, b3, residual=bufs[0], out=bufs[1])
torch.cuda.synchronize()
tl = np.zeros(256 * 2 * 64, np.uint64)
assert L.sm_bneck_timeline_read(tl.ctypes.data_as(C.c_void_p)) == 0
tl = tl.reshape(256, 2, 2, 32).astype(np.int64)       # [block][wave & 1][wave >> 1][slot]
t0 = np.where(tl > 0, tl, np.iinfo(np.int64).max).min(axis=(1, 2, 3), keepdims=True)
rel = np.where(tl > 0, tl - t0, -1)
names = ["start", "set-up done (patch cleared, DMA table)", "reduce stages done", "T1 written", "patch barrier passed", "3x3 loop done", "T2 written", "T2 barrier passed"] + ["expand pass %d done" % i for i in range(8)] + ["pass 1: MFMAs of half 0 done", "pass 1: epilogue of half 0 done", "pass 1: MFMAs of half 1 done", "pass 1: epilogue of half 1 done"]
for k, nm in enumerate(names):
    v = rel[:, :, :, k]; v = v[v >= 0]
    if v.size: print("%-42s median %6d  min %6d  max %6d" % (nm, np.median(v), v.min(), v.max()))
