"""Measuring stick (NOT on the product path): what the vendor GEMM (hipBLASLt through torch.mm) reaches on the ResNet-50
1x1-conv GEMM shapes, cold (L2 flushed by a 512 MB memset between launches) and warm.  python tools/gemm_ref.py"""
import torch
shapes = [(200704, 64, 64), (200704, 256, 64), (200704, 64, 256), (200704, 128, 256), (50176, 512, 128), (50176, 128, 512),
          (50176, 256, 512), (12544, 1024, 256), (12544, 256, 1024), (12544, 512, 1024), (3136, 2048, 512), (3136, 512, 2048),
          (200704, 64, 576), (50176, 128, 1152), (12544, 256, 2304), (3136, 512, 4608)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").half(); b = torch.randn(N, K, device="cuda").half(); c = torch.empty(M, N, device="cuda", dtype=torch.half)
    for _ in range(3): torch.mm(a, b.t(), out=c)
    res = []
    for cold in (1, 0):
        ts = []
        for _ in range(10):
            if cold: flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.mm(a, b.t(), out=c); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); res.append(ts[len(ts) // 2])
    fl = 2.0 * M * N * K
    print("M=%6d N=%4d K=%4d  cold %6.1f us %6.0f TF/s   warm %6.1f us %6.0f TF/s" % (M, N, K, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6))
