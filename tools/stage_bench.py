"""Time sm_stage_fwd on the ResNet-50 bs-64 stage shapes: the run as ONE launch against the same phases launched one at a time and against the per-layer kernels
(sm_conv2d_fwd / sm_conv2d_dual_fwd_cfg, heuristic configuration).  Synthetic He-normal weights, |N(0,1)| activations.  Usage: python tools/stage_bench.py [stage4|stage3] [iters]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from smelter_amd import ops


def build(n, hw, cin, mid, blocks, seed=0, pack=True, four_waves=False):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    st = ops.Stage(four_waves=four_waves); layers = []
    def phase(x, cout, k=1, res=None, x2=None):
        kk = x.shape[3] + (x2.shape[3] if x2 is not None else 0)
        w = (rn(cout, k, k, kk) * (2.0 / (k * k * kk)) ** 0.5).half(); b = (rn(cout) * 0.1).float()
        y = st.add(x, w, b, residual=res, act=1, x2=x2, stride2=2)
        layers.append((x, w, b, res, x2, y, k))
        return y
    t0 = rn(n, hw, hw, mid).abs().half(); xin = rn(n, 2 * hw, 2 * hw, cin // 2).abs().half()
    cur = phase(t0, cin, 1, x2=xin)
    for _ in range(blocks):
        r = phase(cur, mid, 1); t = phase(r, mid, 3); cur = phase(t, cin, 1, res=cur)
    if pack:
        st.pack()
    return st, layers


def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "stage4"; iters = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 50
    hw, cin, mid, blocks = (7, 2048, 512, 2) if which == "stage4" else (14, 1024, 256, 5)
    st, layers = build(64, hw, cin, mid, blocks, pack="nopack" not in sys.argv, four_waves="4w" in sys.argv)
    print(which, "packed weights:", "nopack" not in sys.argv, "kinds", st.kinds(), flush=True)
    st.run(); torch.cuda.synchronize(); print("status (error bits, mode)", st.status(), flush=True)
    fused = [l[5].clone() for l in layers]
    t_fused = timeit(lambda: st.run(), iters)
    print("%s: ONE launch            %8.1f us" % (which, t_fused), flush=True)
    tot = 0.0
    for i in range(len(layers)):
        t = timeit(lambda: st.run([i]), iters); tot += t
        x, w, b, res, x2, y, k = layers[i]
        flops = 2.0 * 64 * hw * hw * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        print("   phase %2d kind %d  %dx%d %4d->%4d  alone %7.1f us  %6.0f TFLOP/s" % (i, st.kinds()[i], k, k, w.shape[3], w.shape[0], t, flops / t * 1e-6), flush=True)
    print("%s: one launch per phase  %8.1f us" % (which, tot), flush=True)
    def per_layer():
        for (x, w, b, res, x2, y, k) in layers:
            if x2 is not None: ops.conv2d_dual(x, x2, w, b, stride2=2, act=1, out=y)
            else: ops.conv2d(x, w, b, residual=res, pads=((k - 1) // 2,) * 4, act=1, out=y)
    per_layer(); torch.cuda.synchronize()
    bad = [i for i, l in enumerate(layers) if not torch.allclose(l[5].float(), fused[i].float(), rtol=2e-2, atol=2e-2)]
    print("per-layer kernels vs the run: phases beyond 2e-2:", bad)
    print("%s: per-layer kernels     %8.1f us (heuristic configurations, not autotuned)" % (which, timeit(per_layer, iters)), flush=True)
    print("status", st.status())


if __name__ == "__main__":
    main()
