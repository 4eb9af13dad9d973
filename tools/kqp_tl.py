"""Per-phase timeline of the persistent wave-split 3x3 kernel (conv3x3_kqp.hpp); diagnostic library as for tools/kq_tl.py (tools/kq_tl.sh).
    python tools/kqp_tl.py --lib build/libsmelter_hip_kqtl.so --shape 16,256,256,128,64 --cfg "kqp:k2n2 128->64"
Stamps per wave: entry | weights + tables + first patch issued | first patch landed | barrier | main loop of tile 0 | epilogue of tile 0 | table toggle |
(tile 1:) top wait + barrier | main loop | epilogue | ... | end."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smelter_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16,256,256,128,64")
ap.add_argument("--cfg", default="kqp:k2n2 128->64")
ap.add_argument("--lib", default="build/libsmelter_hip_kqtl.so")
a = ap.parse_args()
_native.LIB_PATH = os.path.abspath(a.lib)
from smelter_amd import ops  # noqa: E402
n, h, w, cin, cout = map(int, a.shape.split(","))
cfg = [c for c in range(ops.num_conv_configs()) if ops.config_name(c) == a.cfg][0]
x = torch.randn((n, h, w, cin), device="cuda").half()
wt = (torch.randn((cout, 3, 3, cin), device="cuda") * (2.0 / (cin * 9)) ** 0.5).half()
b = torch.randn(cout, device="cuda")
y = torch.empty((n, h, w, cout), device="cuda", dtype=torch.float16)
for it in range(3):
    ops.conv2d(x, wt, b, stride=(1, 1), pads=(1, 1, 1, 1), act=1, cfg=cfg, out=y)
torch.cuda.synchronize()
buf = np.zeros((1024, 4, 16), np.uint64)
assert ops.lib().sm_kq_timeline_read(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
live = buf[:, 0, 15] > 0
t = buf[live].astype(np.int64)
print("%d live blocks of %s on %s" % (live.sum(), a.cfg, a.shape))
names = ["weights, tables, first patch issued", "first patch landed", "barrier", "tile 0: main loop", "tile 0: epilogue", "tile 0: table toggle", "tile 1: top wait + barrier",
         "tile 1: main loop", "tile 1: epilogue"]
d = t[:, :, 2:11] - t[:, :, 1:10]
print("%-40s %9s %9s %9s   (shader clocks, over blocks and waves)" % ("phase", "min", "median", "max"))
for i, nm in enumerate(names):
    print("%-40s %9.0f %9.0f %9.0f" % (nm, d[:, :, i].min(), np.median(d[:, :, i]), d[:, :, i].max()))
tot = t[:, :, 12] - t[:, :, 1]
print("%-40s %9.0f %9.0f %9.0f" % ("whole wave", tot.min(), np.median(tot), tot.max()))
print("wall per block: median %.2f us ; clock ~ %.2f GHz" % (np.median(t[:, 0, 15] - t[:, 0, 0]) / 100.0, np.median(tot) / np.median((t[:, 0, 15] - t[:, 0, 0]) * 10.0)))
