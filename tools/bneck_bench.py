"""Time sm_bottleneck_fwd against the two launches it replaces (1x1 reduce, then 3x3 + expand head) at ResNet-50's stage-1 shape, same box, same buffers.
    python tools/bneck_bench.py [--n 64] [--hw 56] [--c0 256] [--iters 50]
With a diagnostic build (make EXTRA=-DSM_ABLATION) SM_BN_DEBUG=<bits> switches parts of the fused kernel off (bneck.hip BnParams.dbg)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smelter_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64); ap.add_argument("--hw", type=int, default=56); ap.add_argument("--c0", type=int, default=256); ap.add_argument("--cm", type=int, default=64); ap.add_argument("--iters", type=int, default=50); ap.add_argument("--only", default="", help="one | two: launch only that form (PMC passes)")
a = ap.parse_args()
n, hw, c0, cm = a.n, a.hw, a.c0, a.cm
torch.manual_seed(0)
x = torch.randn((n, hw, hw, c0), device="cuda").half()
x2 = torch.randn((n, hw, hw, c0), device="cuda").half()
w1 = (torch.randn((cm, 1, 1, c0), device="cuda") * (2.0 / c0) ** 0.5).half(); b1 = torch.randn(cm, device="cuda") * 0.1
w2 = (torch.randn((cm, 3, 3, cm), device="cuda") * (2.0 / (9 * cm)) ** 0.5).half(); b2 = torch.randn(cm, device="cuda") * 0.1
w3 = (torch.randn((c0, cm), device="cuda") * (2.0 / cm) ** 0.5).half(); b3 = torch.randn(c0, device="cuda") * 0.1
out = torch.empty_like(x)

def two(src, dst):
    t1 = ops.conv2d(src, w1, b1, act=ops.ACT_RELU)
    if cm == 64:
        _, y = ops.conv2d_xhead(t1, w2, b2, w3, b3, residual=src, act3=ops.ACT_RELU)
        return y
    t2 = ops.conv2d(t1, w2, b2, pads=(1, 1, 1, 1), act=ops.ACT_RELU)       # (the 128-channel middle: three launches, as the graph runs them)
    return ops.conv2d(t2, w3.reshape(c0, 1, 1, cm), b3, residual=src, act=ops.ACT_RELU)
def one(src, dst):
    return ops.bottleneck(src, w1.reshape(cm, c0), b1, w2, b2, w3, b3, residual=src, out=dst)

def timeit(fn, chain):
    # chain: the block's output feeds the next launch (as two identical blocks in a row do in the network: the input was written by the previous launch).
    # The launches are captured into ONE hipGraph and replayed (as the executor replays its launch sequence): the host's per-call cost (ctypes, descriptors) is outside the clock.
    bufs = [x.clone(), torch.empty_like(x)]
    for _ in range(3): fn(bufs[0], bufs[1])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2): fn(bufs[0], bufs[1])
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for i in range(a.iters):
                if chain: fn(bufs[i & 1], bufs[(i + 1) & 1])
                else: fn(bufs[0], bufs[1])
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3

ya, yb = two(x, None), one(x, out)
torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(ya, yb)), " differing:", int((ya != yb).sum()))
flops = 2.0 * n * hw * hw * (c0 * cm + 9 * cm * cm + cm * c0)
for chain in (False, True):
    t2, t1 = timeit(two, chain), timeit(one, chain)
    mb2 = n * hw * hw * (c0 * 2 + cm * 2 * 2 * (1 if cm == 64 else 2) + c0 * 2 * 2) / 1e6; mb1 = n * hw * hw * (c0 * 2 * 2) / 1e6
    print("%s: two launches %.1f us (%.0f MB, %.2f TB/s)   one launch %.1f us (%.0f MB compulsory, %.2f TB/s, %.0f TFLOP/s)" % (
        "chained" if chain else "same buffers", t2, mb2, mb2 / t2, t1, mb1, mb1 / t1, flops / t1 / 1e6))
