"""Per-phase timeline of the wave-split 3x3 kernel (conv3x3_kq.hpp).  Needs the diagnostic library (tools/kq_tl.sh builds it, no GPU needed):
    python tools/kq_tl.py --lib build/libsmelter_hip_kqtl.so --shape 64,14,14,256,256 --cfg "kq:k4n1 196x64"
Stamps per wave: entry | setup done | patch DMA issued | own DMA landed | barrier passed | unit 1 | unit 2 | unit 9 | main loop done | waits + barrier | round 0 done | end."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smelter_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,14,14,256,256")
ap.add_argument("--cfg", default="kq:k4n1 196x64")
ap.add_argument("--stride", type=int, default=1, help="2: the stride-2 instantiations (labels kq/2:...)")
ap.add_argument("--flush", action="store_true", help="evict L2 / Infinity Cache (640 MB memset) before the measured launch")
ap.add_argument("--lib", default="build/libsmelter_hip_kqtl.so")
a = ap.parse_args()
_native.LIB_PATH = os.path.abspath(a.lib)
from smelter_amd import ops  # noqa: E402
n, h, w, cin, cout = map(int, a.shape.split(","))
cfg = [c for c in range(ops.num_conv_configs()) if ops.config_name(c) == a.cfg][0]
x = torch.randn((n, h, w, cin), device="cuda").half()
wt = (torch.randn((cout, 3, 3, cin), device="cuda") * (2.0 / (cin * 9)) ** 0.5).half()
b = torch.randn(cout, device="cuda")
y = torch.empty((n, (h - 1) // a.stride + 1, (w - 1) // a.stride + 1, cout), device="cuda", dtype=torch.float16)
junk = torch.empty(640 << 20, dtype=torch.uint8, device="cuda") if a.flush else None
for it in range(3):
    if junk is not None:
        junk.fill_(it)
    ops.conv2d(x, wt, b, stride=(a.stride, a.stride), pads=(1, 1, 1, 1), act=1, cfg=cfg, out=y)
torch.cuda.synchronize()
L = ops.lib()
nb = 1024
buf = np.zeros((nb, 4, 16), np.uint64)
assert L.sm_kq_timeline_read(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
live = buf[:, 0, 15] > 0
t = buf[live].astype(np.int64)
print("%d live blocks of %s on %s" % (live.sum(), a.cfg, a.shape))
w0 = t[:, :, 0].min()
print("block entry  (us after the first): min %.2f med %.2f max %.2f" % tuple(np.percentile((t[:, 0, 0] - w0) / 100.0, [0, 50, 100])))
print("block exit   (us after the first entry): min %.2f med %.2f max %.2f" % tuple(np.percentile((t[:, :, 15].max(1) - w0) / 100.0, [0, 50, 100])))
names = ["setup", "issue patch DMA", "own DMA landed", "barrier", "unit 0", "unit 1", "units 2-8", "units 9-17", "drain waits + barrier", "reduce round 0", "reduce round 1 + stores"]
d = t[:, :, 2:13] - t[:, :, 1:12]
print("%-28s %9s %9s %9s   (shader clocks, over blocks and waves)" % ("phase", "min", "median", "max"))
for i, nm in enumerate(names):
    print("%-28s %9.0f %9.0f %9.0f" % (nm, d[:, :, i].min(), np.median(d[:, :, i]), d[:, :, i].max()))
tot = t[:, :, 12] - t[:, :, 1]
print("%-28s %9.0f %9.0f %9.0f" % ("whole wave", tot.min(), np.median(tot), tot.max()))
print("wall per block: median %.2f us ; clock ~ %.2f GHz" % (np.median(t[:, 0, 15] - t[:, 0, 0]) / 100.0, np.median(tot) / np.median((t[:, 0, 15] - t[:, 0, 0]) * 10.0)))
