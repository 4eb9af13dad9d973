"""SM_TIMELINE build only: s_memtime per phase of the bottleneck kernel (wave 0 of block 0)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops
raw = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smelter_amd", "libsmelter_hip.so"))
n = 64
for hw, cm in ((56, 64), (28, 128)):
    c = 4 * cm
    x = torch.randn((n, hw, hw, c), device="cuda").half()
    w1 = (torch.randn((cm, 1, 1, c), device="cuda") * 0.05).half(); b1 = torch.randn(cm, device="cuda") * 0.1
    w2 = (torch.randn((cm, 3, 3, cm), device="cuda") * 0.03).half(); b2 = torch.randn(cm, device="cuda") * 0.1
    w3 = (torch.randn((c, 1, 1, cm), device="cuda") * 0.05).half(); b3 = torch.randn(c, device="cuda") * 0.1
    for _ in range(3): ops.bottleneck(x, w1, b1, w2, b2, w3, b3)
    out = (ctypes.c_ulonglong * 8)()
    raw.sm_bottleneck_debug_timeline(out)
    v = list(out)
    print("%dx%d Cm=%d clk: tile setup %d | A wait W1/X %d | A loop %d | A epilogue+B issue %d | B loop %d | B epilogue, h2 -> regs %d | C wait W3 %d | C loop %d | total %d" % (hw, hw, cm, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], sum(v)))
