"""Kernel-level parity on a real MI355X: every `sm_*` entry point, called through the C ABI, against the CPU oracle
on identical seeded inputs.  Inputs/weights are pre-rounded to fp16 so both sides see the same numbers; the oracle
computes in fp32 and its result is rounded to fp16 once, as the kernel's store does.

Tolerance (fp16 output, fp32 accumulate): |got - ref| <= 1 fp16 ulp of the value (2^-10 relative) plus an absolute
floor of 2^-11 * max|ref| * sqrt(K)*1e-2 covering accumulation-order differences that flip a rounding.  Byte-exact
for the pure data-movement kernels (copy, nearest upsample, max-pool, layout)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle as O
from smelter_amd import ops

pytestmark = pytest.mark.gpu

# every tile configuration the built library holds (the C ABI answers without a GPU): a config added to the pool is swept automatically
ALL_CFGS = list(range(ops.num_conv_configs()))


def cfgs_named(*subs, without=()):
    """configurations whose label contains every string of `subs` and none of `without` (kernel families are selected by label, not by index)"""
    return [c for c in ALL_CFGS if all(s in ops.config_name(c) for s in subs) and not any(w in ops.config_name(c) for w in without)]


def cfg_named(label):
    """the configuration with exactly this label"""
    hit = [c for c in ALL_CFGS if ops.config_name(c) == label]
    assert len(hit) == 1, "tile configuration %r: %d matches" % (label, len(hit))
    return hit[0]


# every *_cfg call that returns SM_OK records its configuration here (ops._ran); the last test of this file reads it
ops.CFG_LOG = CFGS_RUN = set()
REF_K64 = cfg_named("k64:128x128")                         # conv_igemm64 two-stage 128x128: the reference tiling the specialised kernels are compared with
WS_PLAIN = cfgs_named("ws:", without=("head",))           # weight-stationary 3x3, no fused head
assert WS_PLAIN and cfgs_named("p9:") and cfgs_named("kq:") and cfgs_named("wst:") and cfgs_named("ps:") and cfgs_named("ws:", " head") and cfgs_named("xhead", without=("side",))


def h(a):
    return O.round_to_half(np.asarray(a, np.float32))


def to_dev(a_nchw):
    return torch.from_numpy(np.ascontiguousarray(a_nchw.transpose(0, 2, 3, 1))).cuda().half()


def to_host(t_nhwc):
    torch.cuda.synchronize()
    return t_nhwc.float().cpu().numpy().transpose(0, 3, 1, 2)


def assert_close(got, ref, k=1):
    ref16 = h(ref)
    tol = np.abs(ref16) * 2.0 ** -10 + np.abs(ref).max() * 2.0 ** -11 * max(1.0, np.sqrt(k)) * 1e-2 + 1e-7
    bad = np.abs(got - ref16) > tol
    assert not bad.any(), "max err %g at %d elements" % (np.abs(got - ref16).max(), bad.sum())


def assert_pointwise_close_on_sampled_pixels(got, x, w, b, r, act, slope, stride=1, samples=4096, seed=0):
    """A 1x1 convolution is per pixel: for tensors too large for the CPU oracle, `samples` random output pixels are gathered into a 1 x P image and THAT goes
    through O.conv2d (same oracle code, same tolerance) — the full-size shapes meet the oracle at kernel level, not only another GPU kernel (VERDICT r3 item 4 i)."""
    n, cout, ho, wo = got.shape
    rng = np.random.default_rng(seed)
    P = min(samples, n * ho * wo)
    flat = rng.choice(n * ho * wo, size=P, replace=False)
    ni, rem = np.divmod(flat, ho * wo); oy, ox = np.divmod(rem, wo)
    xs = np.ascontiguousarray(x[ni, :, oy * stride, ox * stride].T[None, :, None, :])          # [1, cin, 1, P]
    ref = O.conv2d(xs, w, b)[0, :, 0, :]                                                           # [cout, P]
    if r is not None:
        ref = ref + r[ni, :, oy, ox].T
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    assert_close(got[ni, :, oy, ox].T, ref, x.shape[1])


CONV_CASES = [
    # n, h, w, cin, cout, k, s, p, res, act            -- comments: which path / edge is exercised
    (2, 14, 14, 64, 96, 1, 1, 0, False, 1),            # 1x1, Cout not a tile multiple
    (2, 15, 13, 32, 72, 3, 1, 1, True, 1),             # 3x3 ragged M, residual + relu
    (1, 17, 17, 16, 40, 3, 2, 1, False, 0),            # stride 2, no activation
    (3, 9, 9, 8, 8, 7, 2, 3, False, 1),                # 7x7 s2 p3 (stem class), K = 392
    (2, 28, 28, 128, 256, 1, 2, 0, True, 1),           # 1x1 s2 downsample + residual
    (1, 7, 7, 512, 64, 3, 1, 1, False, 1),             # long K (4608)
    (3, 9, 7, 64, 72, 3, 1, 1, True, 1),               # 3x3 s1 p1, Cin % 64 == 0: a tile spans three images (strip kernel: vertical padding across images)
    (2, 14, 14, 128, 128, 3, 1, 1, False, 2),          # strip kernel: two cin blocks, PRelu
    (1, 33, 5, 64, 40, 3, 1, 1, True, 0),              # strip kernel: narrow image (every third pixel touches a horizontal border), ragged M and N
    (5, 3, 3, 192, 256, 3, 1, 1, False, 1),            # strip kernel: 3x3 images (all taps but the centre cross a border somewhere)
    (2, 5, 5, 64, 1000, 1, 1, 0, False, 0),            # Cout = 1000 (tail in N)
    (1, 1, 1, 2048, 1000, 1, 1, 0, False, 0),          # fully connected shape, M = 1
    (2, 6, 6, 24, 20, 3, 1, 1, True, 2),               # Cin % 8 == 0 but not 32; PRelu epilogue
    (1, 8, 8, 3, 8, 3, 1, 1, False, 1),                # Cin = 3 -> direct kernel
    (1, 6, 7, 12, 6, 3, 1, 1, True, 1),                # Cin % 8 != 0, odd Cout -> direct kernel, scalar stores
]


FC_CASES = [
    # rows, K, N, act       -- fc_small.hpp: 16 outputs x 16 rows per block, K over the block's four waves
    (64, 2048, 1000, 0),    # ResNet-50's classifier at bs 64: 63 x 4 blocks, the last output tile holds 8 outputs
    (70, 2048, 1000, 1),    # ragged rows (five row tiles, the last holds 6), ReLU
    (1, 2048, 1000, 0),     # one row
    (256, 1280, 1000, 0),   # MobileNet-v2's classifier at bs 256 (40 k steps): four row groups per block
    (150, 512, 1000, 1),    # two row groups per block, the last block's second group holds 6 rows
    (250, 320, 1000, 2),    # four row groups per block, the last group ragged (10 rows)
    (17, 288, 36, 2),       # 9 k steps over four waves (2 + 2 + 2 + 3), N % 16 != 0, leaky
    (33, 256, 8, 0),        # the smallest K the kernel takes, one output tile of 8
]


@pytest.mark.parametrize("m,k,n,act", FC_CASES)
def test_fully_connected_on_few_rows(m, k, n, act):
    """MPSCNNFullyConnectedNode (Converters.swift:297) on a batch of vectors: the dedicated kernel behind sm_conv2d_fwd for 1 x 1 maps with M <= 256 (no tile configuration, no
    split-K reduce launch), against the oracle's fp32 dot products"""
    d = ops.ConvDesc(m, 1, 1, k, n, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, act, 0.25, k, n, 0)
    assert not ops.lib().sm_conv2d_uses_tile_config(ctypes.byref(d))          # a dedicated kernel takes the shape
    rng = np.random.default_rng(m * 7 + k + n)
    x = h(rng.standard_normal((m, k, 1, 1)))
    w = h(rng.standard_normal((n, 1, 1, k)) * np.sqrt(2.0 / k))
    b = rng.standard_normal(n).astype(np.float32)
    ref = O.conv2d(x, w, b)
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), act=act, slope=slope)
    assert_close(to_host(y), ref, k)
    # the same bits on every launch (the four K partials are added in a fixed order)
    y2 = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), act=act, slope=slope)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("cfg", [None] + ALL_CFGS)
def test_conv2d(case, cfg):
    n, hh, ww, cin, cout, k, s, p, res, act = case
    if cfg is not None and (cin % 8 or cfg >= ops.num_conv_configs() or (cfg >= 6 and cin % 64)):
        pytest.skip("tile configs only apply to the MFMA path (6..11: the Cin % 64 == 0 LDS-DMA kernel)")
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (cin * k * k)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, w, b, stride=(s, s), pads=(p, p, p, p))
    r = None
    if res:
        r = h(rng.standard_normal(ref.shape))
        ref = ref + r
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), residual=to_dev(r) if res else None,
                       stride=(s, s), pads=(p, p, p, p), act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if cfg is not None and "does not apply" in e.detail:      # e.g. the single-stage configs need K == 64
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * k * k)


@pytest.mark.parametrize("cin,cout,res,act", [
    (16, 96, False, "relu"),       # MobileNet-v2 expand: one k step, 1.5 passes of 64 couts
    (24, 144, False, "relu"),      # Cin not a multiple of 16: the second k step is half empty
    (32, 16, False, "none"),       # projection: Cout < 64, no activation
    (144, 24, True, "none"),       # nine k steps, residual
    (176, 64, True, "prelu"),      # eleven of the twelve instantiated k steps, one full pass
    (96, 24, False, "tanh"),       # activation outside the ReLU family
    (144, 128, False, "relu"),     # largest weight image the kernel accepts: 74 KB of dynamic LDS (> 64 KB)
])
def test_pointwise_thin_kernel(cin, cout, res, act, monkeypatch):
    """pw_thin.hpp: untiled streaming kernel for 1x1 / stride 1 layers with Cin % 64 != 0, Cin <= 192 and M >= 8192 (ragged M here)"""
    # (with a residual operand the launcher keeps the layer on a tiled kernel: measured 95 vs 83 us on MobileNet-v2's 144 -> 24 + res layer; the thin kernel's residual
    #  form is reachable in -DSM_ABLATION builds only, where SMELTER_PW_THIN_RES=1 selects it — the product library reads no environment variable at kernel level)
    rng = np.random.default_rng(cin * 1000 + cout)
    n, hh, ww = 3, 61, 47                                  # M = 8601 = 268 * 32 + 25
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 1, 1, cin)) * np.sqrt(2.0 / cin))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, w, b)
    r = h(rng.standard_normal(ref.shape)) if res else None
    if res:
        ref = ref + r
    code = ops.ACT_CODES[act]
    slope = 0.25 if act == "prelu" else 0.0
    if act in ("relu", "prelu"):
        ref = np.where(ref >= 0, ref, ref * slope)
    elif act == "tanh":
        ref = np.tanh(ref)
    d = ops.ConvDesc(n, hh, ww, cin, cout, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, hh, ww, code, slope, 0, 0, 0)
    assert ops.lib().sm_conv2d_uses_tile_config(__import__("ctypes").byref(d)) == 0      # the dedicated kernel takes this shape
    y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), residual=to_dev(r) if res else None, act=code, slope=slope)
    assert_close(to_host(y), ref, cin)


@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_conv2d_generic_epilogue_every_config(cfg):
    """Tanh is not a ReLU-family activation, so every kernel takes its GENERIC epilogue (bias-initialised accumulators, per-element
    activation) instead of the straight-line fast path the ResNet cases exercise; Cin % 64 == 0 so that all BK=64 configs apply.
    (Caught: K-group configs added the bias once per group on this path.)"""
    if cfg >= ops.num_conv_configs():
        pytest.skip("no such config")
    rng = np.random.default_rng(77)
    n, hh, ww, cin, cout, k = 2, 10, 9, 128, 72, 3
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, k, k, cin)) * np.sqrt(1.0 / (cin * k * k)))
    b = rng.standard_normal(cout).astype(np.float32)
    r = h(rng.standard_normal((n, cout, hh, ww)))
    ref = np.tanh(O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1)) + r)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), residual=to_dev(r), stride=(1, 1), pads=(1, 1, 1, 1),
                       act=ops.ACT_CODES["tanh"], cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * k * k)


def test_conv2d_split_k_workspace():
    n, cin, cout = 64, 2048, 1000                       # the ResNet-50 FC: M = 64 cannot fill 256 CUs without split-K
    rng = np.random.default_rng(5)
    x = h(rng.standard_normal((n, cin, 1, 1)))
    w = h(rng.standard_normal((cout, 1, 1, cin)) * 0.02)
    b = rng.standard_normal(cout).astype(np.float32)
    nbytes = ops.conv2d_workspace_bytes((n, 1, 1, cin), (cout, 1, 1, cin))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), workspace=ws)
    assert_close(to_host(y), O.conv2d(x, w, b), cin)


@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_conv2d_split_k_every_config(cfg):
    """few tiles + long K + a workspace: the launcher splits K across blocks (fp32 partials + reduce kernel) for every tile config
    that has a split-K form; residual + ReLU are then applied by the reduce kernel"""
    if cfg >= ops.num_conv_configs():
        pytest.skip("no such config")
    rng = np.random.default_rng(91)
    n, hh, ww, cin, cout, k = 1, 6, 5, 512, 64, 3            # M = 30: one pixel tile; K = 4608
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (cin * k * k)))
    b = rng.standard_normal(cout).astype(np.float32)
    r = h(rng.standard_normal((n, cout, hh, ww)))
    ref = np.maximum(O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1)) + r, 0)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), residual=to_dev(r), stride=(1, 1), pads=(1, 1, 1, 1),
                       act=1, cfg=cfg, workspace=ws)
    except SmelterError as e:
        if "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * k * k)


def test_conv2d_writes_into_concat_slice_and_reads_from_one():
    rng = np.random.default_rng(7)
    x = h(rng.standard_normal((2, 16, 9, 9)))
    w = h(rng.standard_normal((24, 3, 3, 16)) * 0.1)
    big_in = torch.zeros((2, 9, 9, 48), dtype=torch.float16, device="cuda")
    big_in[..., 16:32] = to_dev(x)
    big_out = torch.full((2, 9, 9, 64), 7.0, dtype=torch.float16, device="cuda")
    ops.conv2d(big_in[..., 16:32], torch.from_numpy(w).cuda().half(), None, pads=(1, 1, 1, 1), out=big_out[..., 8:32])
    ref = O.conv2d(x, w, None, pads=(1, 1, 1, 1))
    assert_close(to_host(big_out[..., 8:32]), ref, 144)
    assert (big_out[..., :8] == 7).all() and (big_out[..., 32:] == 7).all()      # neighbours of the slice untouched


@pytest.mark.parametrize("c,hw,s,res,act", [
    (144, (28, 28), 1, False, 1),      # MobileNet-v2 widths: 18 channel vectors -> 9 per block
    (192, (14, 14), 2, False, 1),      # stride 2: 8 vectors per block, ragged 7x7 output
    (960, (7, 7), 1, True, 2),         # small map (8x8 tile), 15 vectors per block, residual + PRelu
    (32, (33, 17), 2, False, 0),       # odd sizes, tiles ragged in both directions, no activation
    (8, (5, 40), 1, True, 1),          # one channel vector, wide image
])
def test_depthwise3x3_lds_tiles(c, hw, s, res, act):
    """the LDS-tiled 3x3 depthwise kernel (MobileNet-v2 path) over its tile / channel-block variants; bit-level tolerance as test_depthwise"""
    rng = np.random.default_rng(c + hw[0])
    x = h(rng.standard_normal((3, c, hw[0], hw[1])))
    w = h(rng.standard_normal((c, 3, 3, 1)) * 0.3)
    b = rng.standard_normal(c).astype(np.float32)
    ref = O.conv2d(x, w, b, stride=(s, s), pads=(1, 1, 1, 1), groups=c)
    r = h(rng.standard_normal(ref.shape)) if res else None
    if res:
        ref = ref + r
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), residual=to_dev(r) if res else None,
                   stride=(s, s), pads=(1, 1, 1, 1), groups=c, act=act, slope=slope)
    assert_close(to_host(y), ref, 9)


@pytest.mark.parametrize("c,k,s", [(16, 3, 1), (32, 3, 2), (8, 5, 1)])
def test_depthwise(c, k, s):
    rng = np.random.default_rng(c)
    x = h(rng.standard_normal((2, c, 11, 10)))
    w = h(rng.standard_normal((c, k, k, 1)) * 0.3)
    b = rng.standard_normal(c).astype(np.float32)
    p = k // 2
    ref = np.maximum(O.conv2d(x, w, b, stride=(s, s), pads=(p, p, p, p), groups=c), 0)
    y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(s, s), pads=(p, p, p, p), groups=c, act=1)
    assert_close(to_host(y), ref, k * k)


@pytest.mark.parametrize("c", [8, 64, 5])
def test_add_relu_batchnorm(c):
    rng = np.random.default_rng(c)
    a, b = h(rng.standard_normal((2, c, 7, 9))), h(rng.standard_normal((2, c, 7, 9)))
    assert_close(to_host(ops.add(to_dev(a), to_dev(b))), a + b)
    assert_close(to_host(ops.add(to_dev(a), to_dev(b), act=1)), np.maximum(a + b, 0))
    assert np.array_equal(to_host(ops.relu(to_dev(a))), np.maximum(a, 0))
    assert_close(to_host(ops.relu(to_dev(a), slope=0.1)), np.where(a >= 0, a, a * np.float32(0.1)))
    g, be, m = (rng.standard_normal(c).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.5, 1.5, c).astype(np.float32)
    s = g / np.sqrt(v + np.float32(1e-5)); t = be - m * s
    y = ops.batchnorm(to_dev(a), torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda())
    assert_close(to_host(y), O.batchnorm(a, g, be, m, v, 1e-5), 4)


@pytest.mark.parametrize("c,hw,k,s,p", [(64, 112, 3, 2, 1), (16, 13, 3, 2, 1), (8, 12, 2, 2, 0), (5, 9, 3, 1, 1),
                                        (64, 256, 2, 2, 0), (128, 70, 3, 2, 1), (24, 50, 2, 2, 0), (40, 101, 3, 2, 1)])   # the unrolled max-pool variants (1 / 2 / 4 outputs per thread), ragged rows
def test_pools(c, hw, k, s, p):
    rng = np.random.default_rng(hw)
    x = h(rng.standard_normal((2, c, hw, hw)))
    assert np.array_equal(to_host(ops.maxpool2d(to_dev(x), (k, k), (s, s), (p, p))), O.maxpool(x, (k, k), (s, s), (p, p)))
    assert_close(to_host(ops.avgpool2d(to_dev(x), (k, k), (s, s), (p, p))), O.avgpool(x, (k, k), (s, s), (p, p)), k * k)


@pytest.mark.parametrize("c,hw", [(2048, 7), (1280, 7), (10, 5), (64, 1)])
def test_global_avgpool(c, hw):
    x = h(np.random.default_rng(c).standard_normal((3, c, hw, hw)))
    assert_close(to_host(ops.global_avgpool(to_dev(x))), O.global_avgpool(x), hw * hw)


@pytest.mark.parametrize("c", [8, 32, 3])
def test_upsample(c):
    x = h(np.random.default_rng(c).standard_normal((2, c, 5, 6)))
    assert np.array_equal(to_host(ops.upsample(to_dev(x), 2, 2)), O.upsample(x, 2, 2))
    assert np.array_equal(to_host(ops.upsample(to_dev(x), 3, 2)), O.upsample(x, 3, 2))
    for ac in (True, False):
        assert_close(to_host(ops.upsample(to_dev(x), 2, 2, True, ac)), O.upsample(x, 2, 2, True, ac), 4)


def test_copy_channels_is_concat():
    rng = np.random.default_rng(0)
    a, b = h(rng.standard_normal((2, 16, 6, 6))), h(rng.standard_normal((2, 24, 6, 6)))
    out = torch.empty((2, 6, 6, 40), dtype=torch.float16, device="cuda")
    ops.copy_channels(to_dev(a), out[..., :16]); ops.copy_channels(to_dev(b), out[..., 16:])
    assert np.array_equal(to_host(out), np.concatenate([a, b], 1))


@pytest.mark.parametrize("c,ld", [(3, 8), (3, 4), (3, 3), (16, 16), (5, 8)])
def test_layout_round_trip_and_float_array(c, ld):
    x = np.random.default_rng(c).random((2, c, 9, 7), dtype=np.float32)
    d = ops.nchw_f32_to_nhwc_f16(torch.from_numpy(x).cuda(), ld=ld)
    torch.cuda.synchronize()
    assert d.shape == (2, 9, 7, ld)
    assert np.array_equal(d[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2), h(x))
    assert (d[..., c:] == 0).all()                       # channel padding is zero-filled
    back = ops.nhwc_f16_to_nchw_f32(d[..., :c])
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy(), h(x))
    fa = ops.nhwc_f16_to_float_array(d[..., :c])
    torch.cuda.synchronize()
    assert np.array_equal(fa.cpu().numpy(), O.pack_float_array(h(x)))        # MPSImage.toFloatArray order


def test_layout_halo():
    x = np.random.default_rng(1).random((1, 3, 5, 6), dtype=np.float32)
    d = ops.nchw_f32_to_nhwc_f16(torch.from_numpy(x).cuda(), ld=4, halo=(3, 3, 2, 5))
    torch.cuda.synchronize()
    assert d.shape == (1, 10, 14, 4)
    inner = d[:, 3:8, 3:9, :3].float().cpu().numpy().transpose(0, 3, 1, 2)
    assert np.array_equal(inner, h(x))
    assert float(d.float().abs().sum().cpu()) == pytest.approx(float(np.abs(h(x)).sum()), rel=1e-3)


def test_bad_arguments_are_errors_not_crashes():
    from smelter_amd._native import SmelterError
    x = torch.zeros((1, 4, 4, 8), dtype=torch.float16, device="cuda")
    w = torch.zeros((8, 3, 3, 8), dtype=torch.float16, device="cuda")
    with pytest.raises(SmelterError) as e:
        ops.conv2d(x, w, dilation=(0, 1), pads=(1, 1, 1, 1))
    assert e.value.kind == "invalidArgument"


# ---- ConvTranspose: MPSCNNConvolutionTransposeNode, Converters.swift:265-287 ---------------------
CONVT_CASES = [
    # n, h, w, cin, cout, k, s, p, out_pad, dil, act
    (2, 5, 6, 16, 8, 2, 2, 0, 0, 1, 1),       # the U-Net up-convolution: every phase is a 1x1 conv
    (2, 6, 5, 8, 12, 3, 2, 1, 1, 1, 0),       # PyTorch's "same x2" ConvTranspose (k3 s2 p1 op1)
    (1, 7, 7, 24, 8, 4, 2, 1, 0, 1, 1),       # k4 s2 p1
    (1, 5, 5, 3, 5, 3, 1, 1, 0, 1, 0),        # stride 1 = correlation with the flipped kernel; Cin % 8 != 0
    (1, 4, 5, 8, 4, 3, 3, 0, 2, 1, 2),        # stride 3 > taps per phase, output_padding 2, PRelu
    (2, 6, 6, 8, 6, 3, 2, 2, 1, 2, 1),        # dilation 2 (size rule ignores it, like the reference)
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose2d_direct(case):
    n, hh, ww, cin, cout, k, s, p, op, dil, act = case
    rng = np.random.default_rng(31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w_iohw = h(rng.standard_normal((cin, cout, k, k)) * 0.3)
    b = rng.standard_normal(cout).astype(np.float32)
    w_flip = O.reformat_conv_weight(w_iohw, cout, cin, k, k, is_transpose=True).reshape(cout, k, k, cin)
    ref = O.conv_transpose2d(x, w_flip, b, stride=(s, s), pads=(p, p, p, p), dil=(dil, dil), out_pad=(op, op))
    if act:
        ref = O.relu(ref, 0.0 if act == 1 else 0.25)
    got = ops.conv_transpose2d(to_dev(x), torch.from_numpy(w_flip).cuda().half().contiguous(), torch.from_numpy(b).cuda(), stride=(s, s),
                               pads=(p, p, p, p), dilation=(dil, dil), out_pad=(op, op), act=act, slope=0.25 if act == 2 else 0.0)
    assert_close(to_host(got), ref, k=cin * k * k)


@pytest.mark.parametrize("label", [None, "64x64", "k64:64x64", "k64w8:128x128(64x32)", "k64w8s3i:128x128(64x32)"])   # generic / BK=64 / 8-wave / interleaved-issue kernels
def test_conv2d_output_placement(label):
    """sm_conv_desc.y_sH...: the conv's Ho x Wo grid lands on a strided, offset pixel lattice of a larger image"""
    cfg = None if label is None else cfg_named(label)
    import ctypes as C
    from smelter_amd import _native as N
    rng = np.random.default_rng(32)
    n, hh, ww, cin, cout = 2, 9, 7, 64, 24
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * 0.1)
    b = rng.standard_normal(cout).astype(np.float32)
    dense = h(O.conv2d(x, w, b, pads=(1, 1, 1, 1)))
    xd, wd, bd = to_dev(x), torch.from_numpy(w).cuda().half().contiguous(), torch.from_numpy(b).cuda()
    YH, YW, sH, sW, oH, oW = 2 * hh + 1, 3 * ww, 2, 3, 1, 2
    y = torch.full((n, YH, YW, cout), -7.0, dtype=torch.float16, device="cuda")
    d = N.ConvDesc(n, hh, ww, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, hh, ww, 0, 0.0, cin, cout, 0, sH, sW, oH, oW, YH, YW)
    N.check(N.lib().sm_conv2d_fwd_cfg(C.byref(d), -1 if cfg is None else cfg, C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                      None, C.c_void_p(y.data_ptr()), None, 0, None))
    got = to_host(y)
    expect = np.full((n, cout, YH, YW), -7.0, np.float32)
    expect[:, :, oH::sH, oW::sW][:, :, :hh, :ww] = dense
    placed = got[:, :, oH:oH + sH * hh:sH, oW:oW + sW * ww:sW]
    assert_close(placed, dense, k=cin * 9)
    mask = np.ones_like(got, bool); mask[:, :, oH:oH + sH * hh:sH, oW:oW + sW * ww:sW] = False
    assert (got[mask] == -7.0).all()                                   # nothing else was touched
    # a residual operand or a lattice that leaves the image is rejected
    bad = N.ConvDesc(n, hh, ww, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, hh, ww, 0, 0.0, cin, cout, 0, sH, sW, oH, oW, YH - 2, YW)
    with pytest.raises(N.SmelterError):
        N.check(N.lib().sm_conv2d_fwd_cfg(C.byref(bad), -1, C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()), None, None, C.c_void_p(y.data_ptr()), None, 0, None))


# ---- the remaining MPSCNNNeuron* / MPSNN arithmetic / normalisation / padding nodes -------------------
def assert_close_f(got, ref, rel=2.0 ** -10, floor=1e-3):
    """fp16 output of an fp32 transcendental: 1 fp16 ulp + the libm difference between device and host expf/logf/tanhf"""
    ref16 = h(ref)
    tol = np.abs(ref16) * rel * 2 + floor * 2.0 ** -10 * max(1.0, np.abs(ref).max())
    assert (np.abs(got - ref16) <= tol).all(), float(np.abs(got - ref16).max())


@pytest.mark.parametrize("kind", ["sigmoid", "hardsigmoid", "tanh", "elu", "abs", "softplus", "softsign", "exp", "log", "clip"])
@pytest.mark.parametrize("c", [16, 6])
def test_neuron_family(kind, c):
    rng = np.random.default_rng(51)
    x = h(rng.standard_normal((2, c, 7, 9)) * 2)
    if kind == "log":
        x = h(np.abs(x) + 0.1)
    a = 0.7 if kind == "elu" else 1.25 if kind == "clip" else 0.0
    assert_close_f(to_host(ops.neuron(kind, to_dev(x), a)), O.neuron(kind, x, a))


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div"])
@pytest.mark.parametrize("c", [24, 5])
def test_binary_family(op, c):
    rng = np.random.default_rng(52)
    a = h(rng.standard_normal((2, c, 6, 5)))
    b = h(rng.standard_normal((2, c, 6, 5)) + 3.0)
    assert_close(to_host(ops.binary(op, to_dev(a), to_dev(b))), O.binary(op, a, b))


@pytest.mark.parametrize("c", [8, 10, 1000, 3])
@pytest.mark.parametrize("log", [False, True])
def test_softmax_over_channels(c, log):
    rng = np.random.default_rng(53)
    x = h(rng.standard_normal((2, c, 3, 4)) * 3)
    assert_close_f(to_host(ops.softmax(to_dev(x), log)), O.softmax(x, log), floor=2e-3)


@pytest.mark.parametrize("c,groups,hw", [(16, 16, (9, 7)), (16, 4, (8, 8)), (6, 3, (5, 5)), (8, 1, (13, 3))])
def test_instance_and_group_norm(c, groups, hw):
    rng = np.random.default_rng(54)
    x = h(rng.standard_normal((3, c, hw[0], hw[1])) * 2 + 0.5)
    g, b = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    got = to_host(ops.group_norm(to_dev(x), groups, torch.from_numpy(g).cuda(), torch.from_numpy(b).cuda()))
    assert_close_f(got, O.group_norm(x, groups, g, b), floor=4e-3)


@pytest.mark.parametrize("mode", ["constant", "reflect", "edge"])
@pytest.mark.parametrize("c", [8, 3])
def test_pad2d_is_byte_exact(mode, c):
    rng = np.random.default_rng(55)
    x = h(rng.standard_normal((2, c, 6, 7)))
    got = to_host(ops.pad2d(to_dev(x), (1, 2, 3, 1), mode))
    assert np.array_equal(got, O.pad2d(x, (1, 2, 3, 1), mode))


@pytest.mark.parametrize("mode", ["bilinear", "lanczos"])
@pytest.mark.parametrize("size", [(8, 20), (38, 26), (19, 13), (64, 64)])
def test_resize_planes_for_force_input_scale(mode, size):
    import ctypes as C
    from smelter_amd import _native as N
    x = np.random.default_rng(61).standard_normal((2, 3, 19, 13)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    y = torch.empty((2, 3, size[0], size[1]), dtype=torch.float32, device="cuda")
    N.check(N.lib().sm_resize_nchw_f32(C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), 6, 19, 13, size[0], size[1], {"lanczos": 1, "bilinear": 2}[mode], None))
    torch.cuda.synchronize()
    ref = O.resize(x, size, mode)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("src,dst", [((8, 4, 6), (8, 2, 12)), ((8, 1, 1), (2, 2, 2)), ((3, 5, 4), (60, 1, 1)), ((4, 3, 3), (1, 36, 1))])
def test_reshape_nchw_order_is_byte_exact(src, dst):
    """MPSNNReshapeNode when the reshape is not an alias in NHWC: == numpy's C-order reshape of the NCHW array"""
    x = h(np.random.default_rng(71).standard_normal((3,) + src))
    got = to_host(ops.reshape_nchw_order(to_dev(x), *dst))
    assert np.array_equal(got, x.reshape((3,) + dst))


WS_CASES = [
    # n, h, w, cin, cout, act   -- conv3x3_ws.hpp: weight-stationary persistent 3x3 (tile configs 61..67)
    (2, 56, 56, 64, 64, 1),      # ResNet-50 stage 1: 28-pixel column bands, fragments straddle tile rows
    (1, 40, 72, 64, 32, 0),      # three 24-pixel bands, no activation
    (3, 9, 7, 64, 64, 1),        # images smaller than a tile: one ragged tile per image, every border case
    (2, 33, 50, 32, 32, 2),      # Cin = 32 (four chunks per pixel), PRelu, ragged rows and columns
    (1, 64, 64, 32, 64, 1),      # 32-pixel bands, exact tiles
    (1, 16, 130, 64, 32, 1),     # five 26-pixel bands
    (5, 20, 32, 64, 64, 1),      # more tiles than fit one image: tile ranges cross image boundaries
]


@pytest.mark.parametrize("case", WS_CASES)
@pytest.mark.parametrize("cfg", WS_PLAIN)
def test_conv3x3_weight_stationary(case, cfg):
    n, hh, ww, cin, cout, act = case
    if cfg >= ops.num_conv_configs():
        pytest.skip("no such config")
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:                      # the config's Cin / Cout is another one
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * 9)


def test_conv3x3_weight_stationary_writes_into_a_concat_slice():
    """pixel strides larger than the channel count on both sides (zero-copy Concat placement)"""
    rng = np.random.default_rng(5)
    x = h(rng.standard_normal((2, 64, 24, 40))); w = h(rng.standard_normal((32, 3, 3, 64)) * 0.06); b = rng.standard_normal(32).astype(np.float32)
    xin = torch.zeros((2, 24, 40, 96), dtype=torch.float16, device="cuda"); xin[..., 16:80] = to_dev(x)
    out = torch.full((2, 24, 40, 64), 7.0, dtype=torch.float16, device="cuda")
    ops.conv2d(xin[..., 16:80], torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=1, cfg=cfg_named("ws:64->32 w4x2"), out=out[..., 32:])
    ref = np.maximum(O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1)), 0)
    assert_close(to_host(out[..., 32:]), ref, 64 * 9)
    assert float(out[..., :32].min()) == 7.0 and float(out[..., :32].max()) == 7.0


P9_CASES = [
    # n, h, w, cin, cout, act   -- conv3x3_p9.hpp: one patch per cin block serves all nine taps (tile configs 68..73)
    (2, 28, 28, 128, 128, 1),    # ResNet-50 stage 2: a tile spans ~9 rows of a 28-wide image, ragged first / last row
    (3, 14, 14, 256, 256, 1),    # stage 3: a 256-pixel tile spans two images (two halo-row pairs in one patch)
    (6, 7, 7, 512, 64, 2),       # stage-4 geometry: 49-pixel images, five to six images per tile, PRelu
    (1, 64, 64, 64, 128, 0),     # wide image: 32-pixel column bands, exact row-aligned tiles, no activation
    (2, 9, 7, 64, 72, 1),        # Cout % 32 != 0: partial cout block, tiny images
    (1, 33, 40, 128, 200, 1),    # two 20-pixel bands, ragged everything
    (2, 40, 72, 192, 40, 1),     # three bands of 24, three cin blocks (patch buffers alternate more than once)
]


@pytest.mark.parametrize("case", P9_CASES)
@pytest.mark.parametrize("cfg", cfgs_named("p9:"))
def test_conv3x3_nine_tap_patch(case, cfg):
    n, hh, ww, cin, cout, act = case
    if cfg >= ops.num_conv_configs():
        pytest.skip("no such config")
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * 9)


KQ_CASES = [
    # n, h, w, cin, cout, act   -- conv3x3_kq.hpp: 196-pixel tiles, K split over the four waves, weights straight into registers
    (3, 14, 14, 256, 256, 1),    # ResNet-50 stage 3: one image per tile, four cout tiles of 64 (or two of 128)
    (2, 28, 28, 128, 128, 1),    # stage 2: bands of 7 rows, real halo rows above and below the inner bands
    (9, 7, 7, 512, 64, 2),       # stage 4: four images per tile (shared zero rows between them), a ragged last tile of one image, two super-blocks, PRelu
    (2, 13, 11, 256, 72, 0),     # 143-pixel images (three idle fragments), Cout % 16 != 0: clamped weight rows, partial cout tile, no activation
    (1, 30, 9, 256, 64, 1),      # bands of 23 rows, ragged last band (7 rows)
    (5, 8, 7, 512, 40, 1),       # three 56-pixel images per tile, last tile holds two; narrowest width the slot scheme takes
    (1, 17, 40, 128, 136, 1),    # bands of 5 rows of a wide image, ragged last band of 2
]


@pytest.mark.parametrize("case", KQ_CASES)
@pytest.mark.parametrize("cfg", cfgs_named("kq:"))
def test_conv3x3_wave_split(case, cfg):
    n, hh, ww, cin, cout, act = case
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:                      # Cin is not a multiple of the config's super-block
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * 9)


KQ2_CASES = [
    # n, h, w, cin, cout, act, pads   -- conv3x3_kq.hpp S2: 3x3 / stride 2, the stride in the patch layout (columns de-interleaved), 196- or 112-pixel output tiles
    (3, 56, 56, 128, 128, 1, (1, 1, 1, 1)),    # ResNet-50 conv_101: bands of 7 (4) output rows, two super-blocks of 64 channels (one of 128)
    (3, 28, 28, 256, 256, 1, (1, 1, 1, 1)),    # conv_208: one image (a band of 7 rows) per tile
    (9, 14, 14, 512, 512, 1, (1, 1, 1, 1)),    # conv_365: four (two) images per tile, ragged last tile of one image
    (2, 15, 13, 128, 72, 2, (1, 1, 1, 1)),     # odd sizes: the last tap row / column is the bottom / right padding; Cout % 16 != 0, PRelu
    (2, 16, 20, 128, 64, 0, (1, 1, 0, 0)),     # no padding behind (ONNX pads [1, 1, 0, 0]): same output size at even sizes, no activation
    (1, 37, 41, 256, 136, 1, (1, 1, 1, 1)),    # 19 x 21 outputs: bands of 9 (5) rows, ragged last band, three cout tiles (one partial)
    (5, 9, 8, 128, 64, 1, (1, 1, 1, 1)),       # 5 x 4 outputs: ten (five) images per tile, narrowest width
]


@pytest.mark.parametrize("case", KQ2_CASES)
@pytest.mark.parametrize("cfg", cfgs_named("kq/2:"))
def test_conv3x3_stride2_wave_split(case, cfg):
    """the first 3x3 of a ResNet stage (ONNXConvolutionPadding.swift:91-113: explicit pads, stride 2) on the wave-split kernel, against the oracle"""
    n, hh, ww, cin, cout, act, pads = case
    rng = np.random.default_rng(hash(case[:6]) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(x, w, b, stride=(2, 2), pads=pads)
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(2, 2), pads=pads, act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:                      # Cin is not a multiple of the config's super-block, or the patch of a tile does not fit the LDS
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * 9)


def test_conv3x3_stride2_takes_the_three_resnet_layers():
    """every stride-2 3x3 of ResNet-50 at batch 64 has at least one wave-split configuration (the autotuner decides among them and the tiled kernels); full size: the
    configurations against the default tiled kernel (same fp32 accumulation up to the order of the channel slices)"""
    from smelter_amd._native import SmelterError
    rng = np.random.default_rng(11)
    for hw, c in ((56, 128), (28, 256), (14, 512)):
        x = torch.from_numpy(rng.standard_normal((64, hw, hw, c)).astype(np.float16)).cuda()
        w = torch.from_numpy((rng.standard_normal((c, 3, 3, c)) * np.sqrt(2.0 / (c * 9))).astype(np.float16)).cuda()
        b = torch.from_numpy(rng.standard_normal(c).astype(np.float32)).cuda()
        ref = ops.conv2d(x, w, b, stride=(2, 2), pads=(1, 1, 1, 1), act=1).float()
        took = 0
        for cfg in cfgs_named("kq/2:"):
            try:
                y = ops.conv2d(x, w, b, stride=(2, 2), pads=(1, 1, 1, 1), act=1, cfg=cfg).float()
            except SmelterError as e:
                assert "does not apply" in e.detail
                continue
            took += 1
            assert y.shape == ref.shape and float((y - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -9, (hw, c, ops.config_name(cfg))
        assert took >= 1, (hw, c)


KQP_CASES = [
    # n, h, w, cin, cout, act, ups   -- conv3x3_kqp.hpp: persistent, whole weight tensor in registers, 14 x 14 tiles (h, w = the OUTPUT size; ups: the input is half of it)
    (2, 28, 28, 128, 64, 1, False),    # exact tiles, two channel slices (the halves trade fragments through the released patch buffer)
    (1, 33, 45, 128, 64, 2, False),    # ragged right / bottom tiles (5 rows, 3 columns), PRelu
    (2, 30, 26, 64, 128, 1, False),    # one slice, four cout groups; ragged both ways
    (3, 9, 7, 64, 128, 0, False),      # images smaller than a tile, no activation
    (2, 28, 44, 128, 64, 1, True),     # fused nearest 2x upsampling of the input (14 x 22 source), ragged columns
    (1, 36, 20, 64, 128, 1, True),     # ... one slice
    (1, 150, 130, 128, 64, 1, False),  # 11 x 10 tiles: several tiles per block, buffers alternate
]


@pytest.mark.parametrize("case", KQP_CASES)
@pytest.mark.parametrize("cfg", cfgs_named("kqp:"))
def test_conv3x3_persistent_weights_in_registers(case, cfg):
    n, hh, ww, cin, cout, act, ups = case
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh // 2, ww // 2) if ups else (n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(O.upsample(x, 2, 2) if ups else x, w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=act, slope=slope, cfg=cfg, ups=ups)
    except SmelterError as e:
        if "does not apply" in e.detail:                      # the other instantiation's channel counts
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * 9)


def test_conv3x3_persistent_weights_in_registers_between_concat_slices():
    """pixel strides larger than the channel count on both sides, through the fused upsampling too, and nothing written outside the output slice"""
    rng = np.random.default_rng(8)
    for cin, cout, label in ((128, 64, "kqp:k2n2 128->64"), (64, 128, "kqp:k1n4 64->128")):
        for ups in (False, True):
            hh, ww = (18, 30)
            x = h(rng.standard_normal((2, cin, hh // 2, ww // 2) if ups else (2, cin, hh, ww)))
            w = h(rng.standard_normal((cout, 3, 3, cin)) * 0.04); b = rng.standard_normal(cout).astype(np.float32)
            xin = torch.zeros(x.shape[:1] + x.shape[2:] + (cin + 48,), dtype=torch.float16, device="cuda"); xin[..., 16:16 + cin] = to_dev(x)
            out = torch.full((2, hh, ww, cout + 24), 7.0, dtype=torch.float16, device="cuda")
            ops.conv2d(xin[..., 16:16 + cin], torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=1, cfg=cfg_named(label),
                       out=out[..., 8:8 + cout], ups=ups)
            ref = np.maximum(O.conv2d(O.upsample(x, 2, 2) if ups else x, w, b, stride=(1, 1), pads=(1, 1, 1, 1)), 0)
            assert_close(to_host(out[..., 8:8 + cout]), ref, cin * 9)
            assert float(out[..., :8].min()) == 7.0 and float(out[..., 8 + cout:].max()) == 7.0


def test_conv3x3_wave_split_between_concat_slices():
    """pixel strides larger than the channel count on both sides, and nothing written outside the output slice"""
    rng = np.random.default_rng(6)
    x = h(rng.standard_normal((2, 256, 14, 14))); w = h(rng.standard_normal((64, 3, 3, 256)) * 0.03); b = rng.standard_normal(64).astype(np.float32)
    xin = torch.zeros((2, 14, 14, 320), dtype=torch.float16, device="cuda"); xin[..., 32:288] = to_dev(x)
    for cfg in cfgs_named("kq:"):
        out = torch.full((2, 14, 14, 96), 7.0, dtype=torch.float16, device="cuda")
        ops.conv2d(xin[..., 32:288], torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=1, cfg=cfg, out=out[..., 16:80])
        ref = np.maximum(O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1)), 0)
        assert_close(to_host(out[..., 16:80]), ref, 256 * 9)
        assert float(out[..., :16].min()) == 7.0 and float(out[..., 80:].max()) == 7.0


@pytest.mark.parametrize("case", [(2, 16, 20, 64, 64, 1), (1, 32, 32, 128, 40, 0), (3, 7, 9, 64, 136, 2), (2, 21, 33, 64, 32, 1), (1, 40, 24, 32, 32, 1)])
@pytest.mark.parametrize("cfg", [None] + WS_PLAIN + cfgs_named("p9:"))
def test_conv3x3_reads_through_fused_nearest_upsampling(case, cfg):
    """sm_conv_desc.x_ups: conv(upsample2x_nearest(x)) computed from the half-resolution x (the executor fuses Upsample -> Conv this way)"""
    n, hh, ww, cin, cout, act = case                                  # hh x ww = the LOW resolution
    rng = np.random.default_rng(hash(case) % 2 ** 31)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv2d(O.upsample(x, 2, 2), w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    slope = 0.25 if act == 2 else 0.0
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), act=act, slope=slope, cfg=cfg, ups=True)
    except SmelterError as e:
        if cfg is not None and "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, cin * 9)
    with pytest.raises(SmelterError):                                 # every other kernel family refuses an upsampled input
        ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), stride=(1, 1), pads=(1, 1, 1, 1), cfg=cfg_named("k64:128x64"), ups=True)


@pytest.mark.parametrize("n,hw,cin,ch,cout,stride,res,expand", [
    (2, (56, 56), 24, 144, 24, 1, True, True),     # MobileNet-v2 block 3: + residual
    (2, (56, 48), 16, 96, 24, 2, False, True),     # block 2: stride 2, Cin = 16 (one k step)
    (1, (28, 28), 32, 192, 32, 1, True, True),     # block 5: widest hidden tensor (192)
    (3, (29, 27), 32, 192, 64, 2, False, True),    # stride 2 on odd sizes, Cout = 64 (two cout blocks)
    (2, (19, 21), 32, 32, 16, 1, False, False),    # block 1: no expand (depthwise reads the input), ragged tiles
    (1, (9, 9), 24, 144, 32, 2, False, True),      # image barely larger than a tile
    (1, (14, 14), 64, 384, 64, 1, True, True),     # 14x14 blocks: W1 / W2 read as fragments from global memory (five k steps, two cout blocks)
    (2, (14, 14), 96, 576, 96, 1, True, True),     # seven k steps, three cout blocks, 18 chunks
    (1, (15, 13), 64, 384, 96, 2, False, True),    # the same path at stride 2, odd sizes
    (3, (14, 14), 64, 384, 96, 1, False, True),    # image per block (kernel 4): no residual, three cout fragments, more images than one block walks alone
    (2, (7, 7), 160, 960, 160, 1, True, True),     # 7x7 blocks: two pixel fragments, cout fragments dealt to four wave groups, 15 chunks
    (2, (7, 7), 160, 960, 320, 1, False, True),    # ten cout fragments
    (2, (6, 9), 96, 200, 72, 1, False, True),      # ragged everything: 54 pixels, Ch not a multiple of the chunk, Cout not of 32
    (1, (13, 15), 48, 136, 48, 1, True, True),     # 195 pixels (6.1 fragments), Cin = 48 on the four-k-step instantiation
    (2, (21, 43), 24, 72, 48, 1, False, True),     # register-window kernel (mbconv5): two cout blocks, ragged tiles, Ch not a multiple of 32
    (1, (37, 19), 16, 40, 56, 2, False, True),     # ... at stride 2 on odd sizes
    (2, (40, 36), 40, 80, 40, 1, True, True),      # ... with three k steps (Cin = 40) + residual
    (1, (33, 70), 32, 64, 24, 2, False, True),     # ... three k steps at stride 2, image larger than four 14 x 14 tiles
])
@pytest.mark.parametrize("kernel", [None, 1, 2, 3, 4])
def test_mbconv_fused_block_equals_the_three_layers(n, hw, cin, ch, cout, stride, res, expand, kernel, monkeypatch):
    """sm_mbconv_fwd against the oracle's three convolutions with the hidden tensors rounded to fp16 where the separate launches store them;
    kernel = the launcher's own choice, or one of the three implementations forced (mbconv.hip: all-in-LDS blocks, chunked blocks, wave pipelines)"""
    from smelter_amd._native import MbConvDesc
    import ctypes
    so = lambda v: (v + 2 - 3) // stride + 1
    dsc = MbConvDesc(n, hw[0], hw[1], so(hw[0]), so(hw[1]), stride, cin, ch, cout, 1 if expand else 0, 1 if res else 0, 1, 0.0, 1, 0.0, 0, 0.0, cin, cout)
    if kernel is not None:      # (forced through sm_mbconv_fwd_kind: the kernel-level launchers read no environment variable)
        v1 = cin <= 32 and ch <= 192 and cout <= 64
        v2 = (not expand or cin <= 64) and cout <= (64 if stride == 1 else 128) and (not res or expand)
        v3 = (cin <= 46 and cout <= 64 and ch <= 192) if expand else True
        v3 = v3 or (expand and ch >= 384 and cin <= 110 and cout <= 96)
        v4 = ops.lib().sm_mbconv_kernel_kind(ctypes.byref(dsc)) == 4       # (the image-per-block kernel's domain includes an LDS budget: asked, not restated)
        if not (v1, v2, v3, v4)[kernel - 1]:
            pytest.skip("outside this kernel's domain (the launcher falls back to another one)")
    elif not ops.lib().sm_mbconv_supported(ctypes.byref(dsc)):
        pytest.skip("no fused kernel takes this geometry (the executor keeps the three launches)")
    rng = np.random.default_rng(cin * 7 + ch + stride)
    x = h(rng.standard_normal((n, cin, hw[0], hw[1])))
    w1 = h(rng.standard_normal((ch, 1, 1, cin)) * np.sqrt(2.0 / cin)) if expand else None
    b1 = rng.standard_normal(ch).astype(np.float32) * 0.1 if expand else None
    wd = h(rng.standard_normal((ch, 3, 3, 1)) * np.sqrt(2.0 / 9)); bd = rng.standard_normal(ch).astype(np.float32) * 0.1
    w2 = h(rng.standard_normal((cout, 1, 1, ch)) * np.sqrt(1.0 / ch)); b2 = rng.standard_normal(cout).astype(np.float32) * 0.1
    hid = h(np.maximum(O.conv2d(x, w1, b1), 0)) if expand else x
    hid2 = h(np.maximum(O.conv2d(hid, wd, bd, stride=(stride, stride), pads=(1, 1, 1, 1), groups=ch), 0))
    ref = O.conv2d(hid2, w2, b2) + (x if res else 0)
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    y = ops.mbconv(to_dev(x), dev(w1).half() if expand else None, dev(b1), dev(wd.reshape(ch, 3, 3)).half(), dev(bd), dev(w2).half(), dev(b2), stride=stride, residual=res, kernel=kernel or 0)
    got, ref16 = to_host(y), h(ref)
    # three chained layers: a different fp32 summation order can flip the rounding of a hidden value, which moves an output by up to an
    # fp16 ulp of the LARGEST magnitude in play -> per-element bound of 2 ulp + one ulp of max|ref|
    assert (np.abs(got - ref16) <= np.abs(ref16) * 2.0 ** -9 + np.abs(ref16).max() * 2.0 ** -10).all(), np.abs(got - ref16).max()
    assert np.mean(got != ref16) < 0.2                                  # the vast majority of the elements is bit-identical


@pytest.mark.parametrize("hw,cin,ch,cout,stride,acts", [
    ((30, 26), 24, 96, 24, 1, (ops.ACT_NONE, ops.ACT_NONE, ops.ACT_NONE)),     # purely linear block
    ((30, 26), 24, 96, 40, 2, (ops.ACT_RELU, ops.ACT_NONE, ops.ACT_RELU)),     # mixed: the generic (slope) instantiation, ReLU behind the projection
    ((22, 34), 40, 64, 16, 1, (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_NONE)),
])
def test_mbconv_register_window_kernel_other_activations(hw, cin, ch, cout, stride, acts):
    """mbconv5_kernel's RELU = false instantiations (act1 / act2 by slope) and act3 != none: same oracle as above with the activations as configured"""
    rng = np.random.default_rng(cin + ch + cout)
    act = lambda a, v: np.maximum(v, 0) if a == ops.ACT_RELU else v
    x = h(rng.standard_normal((2, cin, hw[0], hw[1])))
    w1 = h(rng.standard_normal((ch, 1, 1, cin)) * np.sqrt(2.0 / cin)); b1 = rng.standard_normal(ch).astype(np.float32) * 0.1
    wd = h(rng.standard_normal((ch, 3, 3, 1)) * np.sqrt(2.0 / 9)); bd = rng.standard_normal(ch).astype(np.float32) * 0.1
    w2 = h(rng.standard_normal((cout, 1, 1, ch)) * np.sqrt(1.0 / ch)); b2 = rng.standard_normal(cout).astype(np.float32) * 0.1
    hid = h(act(acts[0], O.conv2d(x, w1, b1)))
    hid2 = h(act(acts[1], O.conv2d(hid, wd, bd, stride=(stride, stride), pads=(1, 1, 1, 1), groups=ch)))
    ref16 = h(act(acts[2], O.conv2d(hid2, w2, b2)))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = to_host(ops.mbconv(to_dev(x), dev(w1).half(), dev(b1), dev(wd.reshape(ch, 3, 3)).half(), dev(bd), dev(w2).half(), dev(b2), stride=stride, residual=False, acts=acts))
    assert (np.abs(got - ref16) <= np.abs(ref16) * 2.0 ** -9 + np.abs(ref16).max() * 2.0 ** -10).all(), np.abs(got - ref16).max()
    assert np.mean(got != ref16) < 0.2


@pytest.mark.parametrize("hw,cin,ch,cout,stride,res", [
    ((56, 56), 24, 144, 24, 1, True),      # MobileNet-v2 block 3: register-window kernel (kind 5)
    ((56, 56), 24, 144, 32, 2, False),     # ... stride 2
    ((28, 28), 32, 192, 32, 1, True),      # 28 x 28
    ((14, 14), 64, 384, 64, 1, True),      # one image per block (kind 4)
    ((14, 14), 96, 576, 160, 2, False),    # stride 2 into 7 x 7
    ((7, 7), 160, 960, 160, 1, True),      # 7 x 7
    ((20, 24), 16, 96, 24, 1, False),      # all-in-LDS / chunked domain
    ((112, 112), 32, 32, 16, 1, False),    # MobileNet-v2's first block: NO expand layer (depthwise -> project), the wave-pipeline kernel's CLIP instantiations
    ((40, 36), 32, 32, 24, 2, False),      # ... stride 2, ragged tiles
])
def test_mbconv_relu6_clip(hw, cin, ch, cout, stride, res):
    """a torchvision MobileNet-v2 export holds ONNX Clip(0, 6) where the stand-in model has Relu.  The fused block kernels take SM_ACT_CLIP (upper bound in
    `slope`): min(max(x, 0), 6) behind the expand and the depthwise layer, on inputs scaled so that the bound bites"""
    import ctypes
    from smelter_amd._native import MbConvDesc
    so = lambda v: (v + 2 - 3) // stride + 1
    expand = ch != cin
    dsc = MbConvDesc(2, hw[0], hw[1], so(hw[0]), so(hw[1]), stride, cin, ch, cout, 1 if expand else 0, 1 if res else 0, 12, 6.0, 12, 6.0, 0, 0.0, cin, cout)
    if not ops.lib().sm_mbconv_supported(ctypes.byref(dsc)):
        pytest.skip("no fused kernel takes this geometry (the executor keeps the three launches)")
    rng = np.random.default_rng(cin + ch + cout + stride)
    x = h(rng.standard_normal((2, cin, hw[0], hw[1])) * 3)
    if not expand:
        x = h(np.clip(x * 2, 0, 6))                      # (the block input IS the hidden tensor: what a clipped stem would hand over)
    w1 = h(rng.standard_normal((ch, 1, 1, cin)) * np.sqrt(2.0 / cin)); b1 = rng.standard_normal(ch).astype(np.float32)
    wd = h(rng.standard_normal((ch, 3, 3, 1)) * np.sqrt(2.0 / 9) * 2); bd = rng.standard_normal(ch).astype(np.float32)
    w2 = h(rng.standard_normal((cout, 1, 1, ch)) * np.sqrt(1.0 / ch)); b2 = rng.standard_normal(cout).astype(np.float32) * 0.1
    hid = h(np.clip(O.conv2d(x, w1, b1), 0, 6)) if expand else x
    hid2 = h(np.clip(O.conv2d(hid, wd, bd, stride=(stride, stride), pads=(1, 1, 1, 1), groups=ch), 0, 6))
    assert (hid == 6).mean() > 0.02 and (hid2 == 6).mean() > 0.02, "the test data must reach the upper bound"
    ref = O.conv2d(hid2, w2, b2) + (x if res else 0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = to_host(ops.mbconv(to_dev(x), dev(w1).half() if expand else None, dev(b1) if expand else None, dev(wd.reshape(ch, 3, 3)).half(), dev(bd), dev(w2).half(), dev(b2), stride=stride, residual=res,
                             acts=(ops.ACT_CLIP, ops.ACT_CLIP, ops.ACT_NONE), slopes=(6.0, 6.0, 0.0)))
    ref16 = h(ref)
    assert (np.abs(got - ref16) <= np.abs(ref16) * 2.0 ** -9 + np.abs(ref16).max() * 2.0 ** -10).all(), np.abs(got - ref16).max()


@pytest.mark.parametrize("acts,slopes", [
    ((12, 12, 12), (6.0, 6.0, 2.0)),      # Clip behind all three layers: the bound on the PROJECT layer was dropped by the register-window kernel
    ((12, 0, 0), (6.0, 0.0, 0.0)),        # Clip next to a linear depthwise layer: the kernel's general-activation branch (RELU == 0) must keep the bound
    ((12, 2, 12), (6.0, 0.1, 1.5)),       # Clip, leaky ReLU, Clip
    ((1, 12, 12), (0.0, 4.0, 3.0)),       # plain ReLU, then two different bounds
])
@pytest.mark.parametrize("hw,cin,ch,cout,stride,res", [
    ((56, 56), 24, 144, 24, 1, True),      # register-window kernel (kind 5)
    ((56, 56), 24, 144, 32, 2, False),     # ... stride 2
    ((14, 14), 64, 384, 64, 1, True),      # one image per block (kind 4)
    ((20, 24), 16, 96, 24, 1, False),      # all-in-LDS / chunked domain
])
def test_mbconv_clip_every_layer(hw, cin, ch, cout, stride, res, acts, slopes):
    """SM_ACT_CLIP on any of the three layers of a fused inverted-residual block, mixed with the other members of the ReLU family — every block kernel the
    descriptor can land on must apply each layer's own bound (12 = SM_ACT_CLIP with the bound in `slope`, 1 = ReLU, 2 = leaky ReLU, 0 = none)"""
    import ctypes
    from smelter_amd._native import MbConvDesc
    so = lambda v: (v + 2 - 3) // stride + 1
    dsc = MbConvDesc(2, hw[0], hw[1], so(hw[0]), so(hw[1]), stride, cin, ch, cout, 1, 1 if res else 0, acts[0], slopes[0], acts[1], slopes[1], acts[2], slopes[2], cin, cout)
    if not ops.lib().sm_mbconv_supported(ctypes.byref(dsc)):
        pytest.skip("no fused kernel takes this geometry / activation triple (the executor keeps the three launches)")
    def act(v, a, sl):
        return np.clip(v, 0, sl) if a == 12 else np.maximum(v, 0) if a == 1 else np.where(v < 0, v * sl, v) if a == 2 else v
    rng = np.random.default_rng(cin + ch + cout + stride + sum(acts))
    x = h(rng.standard_normal((2, cin, hw[0], hw[1])) * 3)
    w1 = h(rng.standard_normal((ch, 1, 1, cin)) * np.sqrt(2.0 / cin)); b1 = rng.standard_normal(ch).astype(np.float32)
    wd = h(rng.standard_normal((ch, 3, 3, 1)) * np.sqrt(2.0 / 9) * 2); bd = rng.standard_normal(ch).astype(np.float32)
    w2 = h(rng.standard_normal((cout, 1, 1, ch)) * np.sqrt(4.0 / ch)); b2 = rng.standard_normal(cout).astype(np.float32)
    hid = h(act(O.conv2d(x, w1, b1), acts[0], slopes[0]))
    hid2 = h(act(O.conv2d(hid, wd, bd, stride=(stride, stride), pads=(1, 1, 1, 1), groups=ch), acts[1], slopes[1]))
    pre = O.conv2d(hid2, w2, b2) + (x if res else 0)
    ref = act(pre, acts[2], slopes[2])
    if acts[2] == 12:
        assert (pre > slopes[2]).mean() > 0.02, "the test data must reach the project layer's bound"
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = to_host(ops.mbconv(to_dev(x), dev(w1).half(), dev(b1), dev(wd.reshape(ch, 3, 3)).half(), dev(bd), dev(w2).half(), dev(b2), stride=stride, residual=res, acts=acts, slopes=slopes))
    ref16 = h(ref)
    # (the floor follows the UNCLIPPED project output: a bound on the last layer shrinks max|ref| but not the noise the hidden tensors' roundings carry into it)
    assert (np.abs(got - ref16) <= np.abs(ref16) * 2.0 ** -9 + np.abs(pre).max() * 2.0 ** -10).all(), np.abs(got - ref16).max()


@pytest.mark.parametrize("n,c,hw,cout,stride,pad,act,slope", [
    (2, 3, (64, 64), 32, 2, 1, 12, 6.0),               # ReLU6 behind the stem (SM_ACT_CLIP), as MobileNet-v2's first layer holds it
    (2, 1, (70, 45), 32, 1, 1, ops.ACT_RELU, 0.0),     # U-Net's first layer (one plane), ragged block tiles
    (1, 1, (512, 512), 32, 1, 1, ops.ACT_RELU, 0.0),   # ... at its real size
    (2, 3, (64, 64), 32, 2, 1, ops.ACT_RELU, 0.0),     # RGB, stride 2
    (1, 3, (37, 83), 64, 1, 0, ops.ACT_NONE, 0.0),     # no padding, two cout blocks, no activation
    (3, 2, (33, 31), 16, 2, 0, ops.ACT_PRELU, 0.2),    # two planes, stride 2 without padding, leaky, 16 outputs
    (1, 3, (18, 130), 24, 2, 1, ops.ACT_RELU, 0.0),    # three 64-column staging passes
])
def test_image_stem_3x3_from_the_fp32_batch(n, c, hw, cout, stride, pad, act, slope):
    """sm_stem3x3_f32_fwd against the oracle's convolution of the fp16-rounded image (what the converted copy held)"""
    rng = np.random.default_rng(c * 100 + cout + stride)
    x = rng.standard_normal((n, c, hw[0], hw[1])).astype(np.float32) * (8 if act == ops.ACT_CLIP else 1)      # (clipped: scaled so that the upper bound bites)
    w = h(rng.standard_normal((cout, 3, 3, c)) * np.sqrt(2.0 / (9 * c))); b = rng.standard_normal(cout).astype(np.float32) * 0.3
    ref = O.conv2d(h(x), w, b, stride=(stride, stride), pads=(pad, pad, pad, pad))
    ref = np.maximum(ref, 0) if act == ops.ACT_RELU else np.where(ref < 0, ref * slope, ref) if act == ops.ACT_PRELU else np.clip(ref, 0, slope) if act == ops.ACT_CLIP else ref
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    y = ops.stem3x3(dev(x), dev(w).half(), dev(b), stride=stride, pad=pad, act=act, slope=slope)
    assert_close(to_host(y), ref, 9 * c)


@pytest.mark.parametrize("n,hw,ch,cout,st_stride,stride", [
    (2, (64, 64), 32, 16, 2, 1),      # MobileNet-v2's first three layers: 3x3/2 stem 3 -> 32, depthwise 3x3, project 32 -> 16
    (1, (45, 51), 32, 16, 2, 1),      # odd image, ragged tiles
    (2, (40, 24), 32, 24, 1, 2),      # stride-1 stem, stride-2 depthwise
    (1, (30, 38), 16, 64, 2, 1),      # 16 stem channels, 64 outputs (two cout blocks)
    (2, (40, 24), 32, 24, 1, 1),      # stride-1 stem, 24 outputs
    (3, (224, 224), 32, 16, 2, 1),    # the real size
    (1, (33, 70), 32, 32, 2, 1),      # 32 outputs, ragged tiles
    (1, (20, 20), 32, 8, 2, 1),       # 8 outputs
])
def test_mbconv_with_the_image_stem_inside(n, hw, ch, cout, st_stride, stride):
    """sm_mbconv_stem_fwd: stem (3x3 on the fp32 NCHW batch, padding 1, ReLU) -> depthwise 3x3 -> project in one launch, against the oracle's three
    convolutions with the hidden tensors rounded to fp16 where the separate launches store them (the stem reads the fp16-rounded image, as
    the conversion kernel would have produced it)"""
    rng = np.random.default_rng(ch + cout + hw[0] + st_stride)
    x = rng.standard_normal((n, 3, hw[0], hw[1])).astype(np.float32)
    ws = h(rng.standard_normal((ch, 3, 3, 3)) * np.sqrt(2.0 / 27)); bs = rng.standard_normal(ch).astype(np.float32) * 0.1
    wd = h(rng.standard_normal((ch, 3, 3, 1)) * np.sqrt(2.0 / 9)); bd = rng.standard_normal(ch).astype(np.float32) * 0.1
    w2 = h(rng.standard_normal((cout, 1, 1, ch)) * np.sqrt(1.0 / ch)); b2 = rng.standard_normal(cout).astype(np.float32) * 0.1
    hid = h(np.maximum(O.conv2d(h(x), ws, bs, stride=(st_stride, st_stride), pads=(1, 1, 1, 1)), 0))
    hid2 = h(np.maximum(O.conv2d(hid, wd, bd, stride=(stride, stride), pads=(1, 1, 1, 1), groups=ch), 0))
    ref = O.conv2d(hid2, w2, b2)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    y = ops.mbconv_stem(dev(x), dev(ws).half(), dev(bs), dev(wd.reshape(ch, 3, 3)).half(), dev(bd), dev(w2).half(), dev(b2), st_stride=st_stride, st_pad=1, stride=stride)
    got, ref16 = to_host(y), h(ref)
    assert (np.abs(got - ref16) <= np.abs(ref16) * 2.0 ** -9 + np.abs(ref16).max() * 2.0 ** -10).all(), np.abs(got - ref16).max()
    assert np.mean(got != ref16) < 0.2


@pytest.mark.parametrize("n,hw,cin,cout,act", [(2, (64, 64), 32, 32, 1), (1, (32, 96), 64, 32, 2), (2, (20, 32), 64, 64, 1), (1, (16, 128), 32, 64, 0)])
@pytest.mark.parametrize("cfg", [-1] + [c for c in WS_PLAIN if "x2" in ops.config_name(c)])
def test_conv3x3_with_fused_maxpool(n, hw, cin, cout, act, cfg):
    """sm_conv2d_pool_fwd_cfg: y bit-identical to the plain launch of the same config, pooled == max-pool of y (exact)"""
    rng = np.random.default_rng(cin + cout + hw[1])
    x = h(rng.standard_normal((n, cin, hw[0], hw[1])))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9))); b = rng.standard_normal(cout).astype(np.float32)
    slope = 0.25 if act == 2 else 0.0
    from smelter_amd._native import SmelterError
    try:
        y, yp = ops.conv2d_pool(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if cfg >= 0 and "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    ref = O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    assert_close(to_host(y), ref, cin * 9)
    assert np.array_equal(to_host(yp), O.maxpool(to_host(y), (2, 2), (2, 2), (0, 0)))


@pytest.mark.parametrize("n,hw,cin,cout,act", [(2, (64, 64), 64, 128, 1), (1, (16, 96), 128, 128, 2), (3, (8, 32), 64, 72, 1), (1, (128, 128), 64, 128, 1)])
def test_conv3x3_patch_kernel_with_fused_maxpool(n, hw, cin, cout, act):
    """the pooling form of conv3x3_p9.hpp (wider layers than the weight-stationary kernel takes: the U-Net's 128-channel level): y bit-identical to the plain launch of
    the same tile shape, pooled == max-pool of y (exact); shapes it cannot take (W % 32, H % 8) are refused, not mis-pooled"""
    cfg = [c for c in ALL_CFGS if ops.config_name(c) == "p9:256x128 s3 pool"][0]
    plain = [c for c in ALL_CFGS if ops.config_name(c) == "p9:256x128 s3"][0]
    rng = np.random.default_rng(cin + cout + hw[1])
    x = h(rng.standard_normal((n, cin, hw[0], hw[1])))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9))); b = rng.standard_normal(cout).astype(np.float32)
    slope = 0.25 if act == 2 else 0.0
    wd, bd = torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda()
    y, yp = ops.conv2d_pool(to_dev(x), wd, bd, act=act, slope=slope, cfg=cfg)
    ref = O.conv2d(x, w, b, stride=(1, 1), pads=(1, 1, 1, 1))
    ref = np.where(ref >= 0, ref, ref * slope)
    assert_close(to_host(y), ref, cin * 9)
    assert np.array_equal(to_host(yp), O.maxpool(to_host(y), (2, 2), (2, 2), (0, 0)))
    y0 = ops.conv2d(to_dev(x), wd, bd, pads=(1, 1, 1, 1), act=act, slope=slope, cfg=plain)
    assert torch.equal(y0, y)
    from smelter_amd._native import SmelterError
    for bad in ((12, 32), (16, 48)):                                  # H % 8 != 0; W % 32 != 0
        xb = to_dev(h(rng.standard_normal((1, cin, bad[0], bad[1]))))
        with pytest.raises(SmelterError):
            ops.conv2d_pool(xb, wd, bd, act=act, slope=slope, cfg=cfg)


@pytest.mark.parametrize("shape", [
    # n, C, H, W, cout, k, stride, pads
    (2, 3, 64, 64, 64, 7, 2, (3, 3, 3, 3)),      # ResNet-50 class
    (1, 3, 75, 91, 64, 7, 2, (3, 3, 3, 3)),      # ragged tiles on both axes
    (2, 3, 40, 72, 64, 3, 1, (1, 1, 1, 1)),      # U-Net class
    (2, 3, 66, 66, 32, 3, 2, (1, 1, 1, 1)),      # MobileNet-v2 class
    (1, 1, 33, 47, 32, 3, 1, (1, 1, 1, 1)),      # one channel, ragged
    (1, 4, 50, 38, 32, 7, 2, (2, 3, 3, 2)),      # asymmetric padding, 4 channels
    (1, 3, 9, 9, 64, 3, 1, (0, 0, 0, 0)),        # image smaller than a tile
])
@pytest.mark.parametrize("cfg", [None] + cfgs_named("wst:"))
def test_stem_on_the_weight_stationary_kernel(shape, cfg):
    """Image stem through the row-pair image (ops.stem_conv = executor's lowering): every stem configuration of conv3x3_ws.hpp that takes
    the shape agrees with the oracle convolution, and with the default pick.  Configurations 97+ have loader waves that build the patches from
    the fp32 batch inside the launch (sm_conv2d_stem_f32_fwd_cfg): bit-identical to the row-pair path"""
    n, C, H, W, cout, k, s, pads = shape
    rng = np.random.default_rng(H * W + cout + k)
    x = rng.standard_normal((n, C, H, W)).astype(np.float32)
    w = h(rng.standard_normal((cout, k, k, C)) * np.sqrt(2.0 / (C * k * k))); b = rng.standard_normal(cout).astype(np.float32)
    from smelter_amd._native import SmelterError
    try:
        y = ops.stem_conv(torch.from_numpy(x).cuda(), torch.from_numpy(w), torch.from_numpy(b).cuda(), stride=s, pads=pads, act=1, cfg=cfg)
    except SmelterError as e:
        if cfg is not None and "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    ref = np.maximum(O.conv2d(h(x), w, b, stride=(s, s), pads=pads), 0)
    assert_close(to_host(y), ref, C * k * k)
    if cfg is not None and ops.lib().sm_conv2d_config_loader_waves(cfg) > 0:
        assert torch.equal(y, ops.stem_conv(torch.from_numpy(x).cuda(), torch.from_numpy(w), torch.from_numpy(b).cuda(), stride=s, pads=pads, act=1))


@pytest.mark.parametrize("shape", [
    # n, C, H, W, cout, k, stride, pads
    (2, 3, 64, 64, 64, 7, 2, (3, 3, 3, 3)),      # ResNet-50 class: 32x32 map -> 16x16 pooled, last window row / column clamped
    (1, 3, 224, 224, 64, 7, 2, (3, 3, 3, 3)),    # the real size: 8 x 7 tiles of 7 x 8 pooled pixels
    (3, 3, 76, 92, 64, 7, 2, (3, 3, 3, 3)),      # ragged tiles on both axes, three images
    (2, 3, 40, 72, 64, 3, 1, (1, 1, 1, 1)),      # 3x3 / stride-1 stem (two taps) behind the same pool
    (1, 4, 52, 36, 64, 7, 2, (2, 3, 3, 2)),      # asymmetric padding, 4 channels
    (1, 1, 12, 12, 64, 3, 1, (1, 1, 1, 1)),      # one plane, image smaller than a tile
])
def test_stem_with_fused_maxpool3x3(shape):
    """sm_conv2d_stem_f32_pool_fwd (stem_pool.hpp): pooled output BIT-IDENTICAL to max-pool(stem) through the two separate launches (max of fp16 values is exact),
    and to the oracle's pooling of the stem the GPU produced"""
    n, C, H, W, cout, k, s, pads = shape
    rng = np.random.default_rng(C * 31 + H + W)
    x = rng.random((n, C, H, W), dtype=np.float32)
    w = h(rng.standard_normal((cout, k, k, C)) * np.sqrt(2.0 / (C * k * k))); b = rng.standard_normal(cout).astype(np.float32)
    xd, wd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(w), torch.from_numpy(b).cuda()
    y = ops.stem_conv(xd, wd, bd, stride=s, pads=pads, act=1)
    want = ops.maxpool2d(y, (3, 3), (2, 2), (1, 1))
    sentinel = torch.full((n, want.shape[1], want.shape[2], cout + 8), 3.0, dtype=torch.float16, device="cuda")
    got = ops.stem_conv_pool(xd, wd, bd, stride=s, pads=pads, act=1, out=sentinel[..., :cout])
    torch.cuda.synchronize()
    assert (sentinel[..., cout:] == 3).all()
    assert torch.equal(got, want)
    assert np.array_equal(to_host(got), O.maxpool(to_host(y), (3, 3), (2, 2), (1, 1)))
    ref = O.maxpool(np.maximum(O.conv2d(h(x), w, b, stride=(s, s), pads=pads), 0), (3, 3), (2, 2), (1, 1))
    assert_close(to_host(got), ref, C * k * k)


@pytest.mark.parametrize("case", [
    # n, ho, wo, K1, K2, cout, stride2, act
    (2, 14, 14, 128, 256, 512, 2, 1),                  # ResNet stage 2, first block: expand 128 -> 512 + shortcut 256 -> 512 / 2 on the 28x28 block input
    (3, 7, 9, 256, 512, 1024, 2, 2),                   # stage 3 class, ragged M (189), PReLU
    (2, 5, 5, 64, 64, 200, 1, 0),                      # stride 1 second source, ragged N, no activation
    (1, 7, 7, 512, 1024, 2048, 2, 1),                  # stage 4 class: 24 K tiles
    (3, 9, 7, 128, 256, 256, 2, 1),                    # K = 384 (the weights-in-registers ring form), ragged M (189), one N slice
    (2, 6, 6, 64, 320, 512, 1, 2),                     # K = 384 split 64 + 320, stride 1 second source, PReLU
    (16, 28, 28, 128, 256, 512, 2, 1),                 # the real stage-2 shape at bs 16: 392 stages of 32 pixels
])
@pytest.mark.parametrize("cfg", [-1] + cfgs_named("ps:") + cfgs_named("rwl:", "dual"))
def test_pointwise_over_two_sources(case, cfg):
    """sm_conv2d_dual_fwd_cfg (conv1x1_ps.hpp, second source): equals the convolution over the channel-concatenated, subsampled operand (oracle), and is
    bit-identical to that convolution run by the same configuration on a materialised concatenation"""
    n, ho, wo, k1, k2, cout, s2, act = case
    rng = np.random.default_rng(k1 + k2 + cout)
    x = h(rng.standard_normal((n, k1, ho, wo))); x2 = h(rng.standard_normal((n, k2, (ho - 1) * s2 + 1 + (s2 - 1), (wo - 1) * s2 + 1)))
    w = h(rng.standard_normal((cout, 1, 1, k1 + k2)) * np.sqrt(2.0 / (k1 + k2))); b = rng.standard_normal(cout).astype(np.float32)
    slope = 0.25 if act == 2 else 0.0
    cat = np.concatenate([x, x2[:, :, ::s2, ::s2][:, :, :ho, :wo]], axis=1)
    ref = O.conv2d(cat, w, b)
    if act:
        ref = np.where(ref >= 0, ref, ref * slope)
    wd, bd = torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda()
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d_dual(to_dev(x), to_dev(x2), wd, bd, stride2=s2, act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if cfg >= 0 and "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    assert_close(to_host(y), ref, k1 + k2)
    if cfg >= 0 and "dual" in ops.config_name(cfg):   # (takes launches with a second source only: compared with the persistent tiled kernel instead)
        y0 = ops.conv2d_dual(to_dev(x), to_dev(x2), wd, bd, stride2=s2, act=act, slope=slope, cfg=cfgs_named("ps:")[0])
        assert (y == y0).float().mean().item() > 0.995
    elif cfg >= 0:
        assert torch.equal(y, ops.conv2d(to_dev(cat), wd, bd, act=act, slope=slope, cfg=cfg))


RW_CASES = [
    # n, h, w, cin, cout, residual, act                -- what the case exercises
    (2, 14, 14, 256, 1024, True, 1),                   # ResNet layer-3 expand: four N slices, 6.1 steps of 64 pixels
    (3, 7, 9, 256, 512, True, 2),                      # ragged M (189 = 2 steps + 61 pixels), PReLU, two N slices
    (1, 40, 40, 256, 64, False, 1),                    # 256 -> 64 reduce: one slice, 12.5 steps of 128 pixels
    (2, 9, 9, 256, 128, False, 0),                     # 256 -> 128, no activation, M = 162 (1.3 steps)
    (1, 3, 3, 256, 256, False, 1),                     # M = 9: a single ragged step, most ranges empty
    (2, 28, 28, 128, 512, True, 1),                    # ResNet layer-2 expand
    (1, 5, 5, 128, 256, False, 1),                     # K = 128 without residual
    (2, 20, 20, 64, 64, False, 1),                     # 64 -> 64, steps of 256 pixels, ragged (800 = 3.1 steps)
    (1, 13, 11, 64, 256, True, 1),                     # 64 -> 256 + residual
    (64, 14, 14, 256, 1024, True, 1),                  # full size: against the reference tiling everywhere, against the oracle on 4096 sampled pixels
    (2, 28, 28, 512, 128, False, 1),                   # ResNet layer-2 reduce on the four-consumer ring form (K = 512: 128 W registers per consumer)
    (3, 7, 9, 512, 256, False, 2),                     # two slices of 128, ragged M (189), PReLU
    (1, 3, 3, 512, 128, False, 0),                     # a single ragged stage
    (4, 7, 7, 512, 2048, True, 1),                     # ResNet layer-4 expand + block input: sixteen slices of 128
    (2, 5, 9, 512, 384, True, 2),                      # three slices, ragged M (90), PReLU
    (5, 7, 7, 320, 1280, False, 1),                    # MobileNet-v2's last pointwise layer: five slabs of K, ten slices
]


@pytest.mark.parametrize("case", RW_CASES)
@pytest.mark.parametrize("cfg", cfgs_named("rw:") + cfgs_named("rwl:"))
def test_pointwise_weights_in_registers(case, cfg):
    """conv1x1_rw.hpp (tile-config kind 8): every instantiation that takes the shape agrees with the oracle; against conv_igemm64 (REF_K64) all
    values agree to the parity tolerance (the residual enters through the MFMA accumulator here: a rounding may flip) and nearly all bit for bit"""
    n, hh, ww, cin, cout, res, act = case
    rng = np.random.default_rng(cin * 11 + cout + hh)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 1, 1, cin)) * np.sqrt(2.0 / cin)); b = rng.standard_normal(cout).astype(np.float32)
    r = h(rng.standard_normal((n, cout, hh, ww))) if res else None
    slope = 0.25 if act == 2 else 0.0
    xd, wd, bd, rd = to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), to_dev(r) if res else None
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(xd, wd, bd, residual=rd, act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    y6 = ops.conv2d(xd, wd, bd, residual=rd, act=act, slope=slope, cfg=REF_K64)
    got, g6 = to_host(y), to_host(y6)
    assert_close(got, g6, cin)
    assert np.mean(got != g6) < (2e-3 if res else 1e-9)
    if n * hh * ww <= 4096:
        ref = O.conv2d(x, w, b)
        if res:
            ref = ref + r
        if act:
            ref = np.where(ref >= 0, ref, ref * slope)
        assert_close(got, ref, cin)
    else:
        assert_pointwise_close_on_sampled_pixels(got, x, w, b, r, act, slope)


def test_pointwise_weights_in_registers_inside_a_concat_buffer():
    """pixel strides larger than the channel count on all three operands (zero-copy concat placement / pass 3h's K-concatenated operand)"""
    rng = np.random.default_rng(5)
    n, hh, ww, cin, cout = 2, 12, 10, 128, 256
    x = h(rng.standard_normal((n, cin, hh, ww))); w = h(rng.standard_normal((cout, 1, 1, cin)) * np.sqrt(2.0 / cin)); b = rng.standard_normal(cout).astype(np.float32)
    r = h(rng.standard_normal((n, cout, hh, ww)))
    xb = torch.zeros((n, hh, ww, cin + 64), dtype=torch.float16, device="cuda"); xb[..., 64:] = to_dev(x)
    rb = torch.zeros((n, hh, ww, cout + 8), dtype=torch.float16, device="cuda"); rb[..., :cout] = to_dev(r)
    yb = torch.full((n, hh, ww, cout + 32), 7.0, dtype=torch.float16, device="cuda")
    cfg = cfgs_named("rw:k128", "+res")[0]
    ops.conv2d(xb[..., 64:], torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), residual=rb[..., :cout], act=1, cfg=cfg, out=yb[..., 16:16 + cout])
    torch.cuda.synchronize()
    assert (yb[..., :16] == 7).all() and (yb[..., 16 + cout:] == 7).all()          # nothing written beside the slice
    ref = np.maximum(O.conv2d(x, w, b) + r, 0)
    assert_close(yb[..., 16:16 + cout].float().cpu().numpy().transpose(0, 3, 1, 2), ref, cin)


PS_CASES = [
    # n, h, w, cin, cout, stride, residual, act        -- what the case exercises
    (2, 14, 14, 256, 1024, 1, True, 1),                # ResNet layer-3 expand: 4 K tiles, residual
    (1, 28, 28, 512, 128, 1, False, 1),                # reduce: 8 K tiles, one n tile
    (3, 9, 11, 64, 200, 1, True, 2),                   # one K tile per output tile, ragged M (297) and ragged N (200), PReLU
    (2, 14, 14, 512, 1024, 2, True, 1),                # stride-2 projection shortcut (+ residual)
    (1, 7, 7, 2048, 512, 1, False, 0),                 # 32 K tiles, fewer tiles than CUs, no activation
    (64, 14, 14, 256, 1024, 1, True, 1),               # full size: 784 tiles of 128x128 on 256 CUs (3..4 tiles per block)
    (4, 56, 56, 64, 64, 1, False, 1),                  # 98 m tiles x 1 n tile, K = 64
]


@pytest.mark.parametrize("case", PS_CASES)
@pytest.mark.parametrize("cfg", cfgs_named("ps:") + cfgs_named("256x256"))
def test_persistent_pointwise_kernel(case, cfg):
    """conv1x1_ps.hpp: every configuration that takes the shape agrees with the oracle and is bit-identical to conv_igemm64 (REF_K64)"""
    if cfg >= ops.num_conv_configs():
        pytest.skip("no such config")
    n, hh, ww, cin, cout, s, res, act = case
    rng = np.random.default_rng(cin * 7 + cout + hh)
    x = h(rng.standard_normal((n, cin, hh, ww)))
    w = h(rng.standard_normal((cout, 1, 1, cin)) * np.sqrt(2.0 / cin)); b = rng.standard_normal(cout).astype(np.float32)
    ho, wo = (hh - 1) // s + 1, (ww - 1) // s + 1
    r = h(rng.standard_normal((n, cout, ho, wo))) if res else None
    slope = 0.25 if act == 2 else 0.0
    xd, wd, bd, rd = to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), to_dev(r) if res else None
    from smelter_amd._native import SmelterError
    try:
        y = ops.conv2d(xd, wd, bd, residual=rd, stride=(s, s), act=act, slope=slope, cfg=cfg)
    except SmelterError as e:
        if "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    y6 = ops.conv2d(xd, wd, bd, residual=rd, stride=(s, s), act=act, slope=slope, cfg=REF_K64)
    assert torch.equal(y, y6)
    if n * ho * wo <= 4096:
        ref = O.conv2d(x, w, b, stride=(s, s))
        if res:
            ref = ref + r
        if act:
            ref = np.where(ref >= 0, ref, ref * slope)
        assert_close(to_host(y), ref, cin)
    else:
        assert_pointwise_close_on_sampled_pixels(to_host(y), x, w, b, r, act, slope, stride=s)


def test_persistent_pointwise_kernel_into_a_channel_slice():
    """output and residual rows wider than Cout (concat placement): only the slice is written"""
    rng = np.random.default_rng(5)
    n, hh, ww, cin, cout = 2, 13, 17, 128, 64
    x = h(rng.standard_normal((n, cin, hh, ww))); w = h(rng.standard_normal((cout, 1, 1, cin)) * 0.1); b = rng.standard_normal(cout).astype(np.float32)
    big = torch.full((n, hh, ww, 160), 7.0, dtype=torch.float16, device="cuda")
    out = big[..., 32:96]
    ops.conv2d(to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), act=1, out=out, cfg=cfg_named("ps:128x128(32x64) s4"))
    ref = np.maximum(O.conv2d(x, w, b), 0)
    assert_close(to_host(out.contiguous()), ref, cin)
    assert (big[..., :32] == 7).all() and (big[..., 96:] == 7).all()


@pytest.mark.parametrize("n,hw,cin,cout,n3,act3,store_y", [
    (2, (64, 64), 32, 32, 8, 0, False),      # U-Net's last two layers: 3x3 32->32 + ReLU -> 1x1 32->8, the 3x3 output not stored
    (1, (40, 96), 64, 32, 16, 1, True),      # both outputs stored, ReLU on the head
    (2, (24, 36), 64, 64, 32, 2, False),     # four k steps of head input, 32 head channels, PReLU, ragged tiles
    (1, (16, 32), 32, 64, 24, 0, True),      # 24 head channels (three groups of 8)
])
@pytest.mark.parametrize("cfg", [-1] + cfgs_named("ws:", " head"))
def test_conv3x3_with_fused_pointwise_head(n, hw, cin, cout, n3, act3, store_y, cfg):
    """sm_conv2d_head_fwd_cfg: y (when stored) bit-identical to the plain launch, y3 == the 1x1 convolution of the ROUNDED y (the stored tensor's
    values), i.e. what the two separate launches compute"""
    rng = np.random.default_rng(cin + cout + n3)
    x = h(rng.standard_normal((n, cin, hw[0], hw[1])))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9))); b = rng.standard_normal(cout).astype(np.float32)
    w3 = h(rng.standard_normal((n3, 1, 1, cout)) * np.sqrt(1.0 / cout)); b3 = rng.standard_normal(n3).astype(np.float32)
    slope3 = 0.25 if act3 == 2 else 0.0
    from smelter_amd._native import SmelterError
    xd, wd, bd = to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda()
    try:
        y, y3 = ops.conv2d_head(xd, wd, bd, torch.from_numpy(w3).cuda().half(), torch.from_numpy(b3).cuda(), act3=act3, slope3=slope3, cfg=cfg, store_y=store_y)
    except SmelterError as e:
        if cfg >= 0 and "does not apply" in e.detail:
            pytest.skip(e.detail)
        raise
    yplain = ops.conv2d(xd, wd, bd, pads=(1, 1, 1, 1), act=1)
    if store_y:
        assert torch.equal(y, yplain)
    z = ops.conv2d(yplain, torch.from_numpy(w3).cuda().half(), torch.from_numpy(b3).cuda(), act=act3, slope=slope3)      # the separate second launch
    assert torch.equal(y3, z)
    ref = O.conv2d(to_host(yplain), w3, b3)
    if act3:
        ref = np.where(ref >= 0, ref, ref * slope3)
    assert_close(to_host(y3), ref, cout)


@pytest.mark.parametrize("n,hw,n3,act3,with_res,store_y", [
    (2, (56, 56), 256, 1, True, False),      # ResNet-50 stage 1: 3x3 64->64 + ReLU -> 1x1 64->256 + block input + ReLU, the 3x3 output not stored
    (1, (56, 56), 256, 1, False, False),     # no residual (a projection-shortcut block keeps its Add elsewhere)
    (3, (20, 37), 128, 2, True, False),      # ragged tiles, four chunks of head channels, PReLU
    (1, (9, 70), 32, 0, True, False),        # one chunk, no activation, a partial column band
    (1, (24, 28), 64, 1, True, True),        # storing the 3x3 output as well: only the single-role kernel (an SM_ABLATION build) does, the product refuses
    (2, (16, 16), 64, 1, False, False),      # wide head without residual
])
@pytest.mark.parametrize("cfg", [-1] + cfgs_named("xhead", without=("side",)))
def test_conv3x3_with_fused_expand_head(n, hw, n3, act3, with_res, store_y, cfg):
    """sm_conv2d_xhead_fwd_cfg (producer / consumer waves, conv3x3_wsx.hpp): y3 = the two separate launches (3x3, then 1x1 + residual + activation on the ROUNDED 3x3 output) —
    bit-identical without a residual, within one fp16 ulp at a few values per million with one; y (when stored) bit-identical to the plain 3x3 launch"""
    cin = cout = 64
    rng = np.random.default_rng(n3 + hw[0] + hw[1])
    x = h(rng.standard_normal((n, cin, hw[0], hw[1])))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9))); b = rng.standard_normal(cout).astype(np.float32)
    w3 = h(rng.standard_normal((n3, 1, 1, cout)) * np.sqrt(1.0 / cout)); b3 = rng.standard_normal(n3).astype(np.float32)
    res = h(rng.standard_normal((n, n3, hw[0], hw[1]))) if with_res else None
    slope3 = 0.25 if act3 == 2 else 0.0
    xd, wd, bd = to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda()
    w3d, b3d = torch.from_numpy(w3).cuda().half(), torch.from_numpy(b3).cuda()
    resd = to_dev(res) if with_res else None
    from smelter_amd._native import SmelterError
    try:
        y, y3 = ops.conv2d_xhead(xd, wd, bd, w3d, b3d, residual=resd, act3=act3, slope3=slope3, store_y=store_y, cfg=cfg)
    except SmelterError as e:
        if store_y and ("does not apply" in e.detail or "no kernel takes" in e.detail):      # the two-role kernel never stores the 3x3 output
            pytest.skip(e.detail)
        raise
    yplain = ops.conv2d(xd, wd, bd, pads=(1, 1, 1, 1), act=1)
    if store_y:
        assert torch.equal(y, yplain)
    z = ops.conv2d(yplain, w3d, b3d, act=act3, slope=slope3, residual=resd)      # the separate second launch
    if with_res:
        # the residual is added by the matrix pipe (one-hot MFMA) BEFORE the bias: (acc + res) + b3 instead of (acc + b3) + res, and the MFMA's
        # accumulate step is not the VALU's fp32 add bit for bit — a few values per ten thousand land on the neighbouring fp16
        bad = y3 != z
        assert int(bad.sum()) <= max(4, int(1e-3 * z.numel())), (int(bad.sum()), z.numel())
        assert bool(((y3.float() - z.float()).abs() <= z.float().abs() * 2.0 ** -10 + 1e-7).all())
    else:
        assert torch.equal(y3, z), "max |diff| %g at %d values" % (float((y3.float() - z.float()).abs().max()), int((y3 != z).sum()))
    ref = O.conv2d(to_host(yplain), w3, b3)
    if with_res:
        ref = ref + res
    if act3:
        ref = np.where(ref >= 0, ref, ref * slope3)
    assert_close(to_host(y3), ref, cout)


@pytest.mark.parametrize("n,hw,n3,act3", [
    (2, (56, 56), 256, 1),       # ResNet-50's first bottleneck block: 3x3 64->64 -> 1x1 64->256, + projection 64->256 of the block input, ReLU
    (3, (20, 37), 128, 2),       # ragged tiles, PReLU
    (1, (9, 70), 64, 0),         # two chunks (consumers 2 and 3 idle), no activation
])
def test_conv3x3_with_fused_expand_head_and_projection(n, hw, n3, act3):
    """sm_conv2d_xhead_proj_fwd_cfg ("ws:64->64 w4+4 xhead side"): y3 = act3(w3 . y + ws . xs + b3 + bs).  The expand layer's output is never rounded, so the result is
    compared with the fp32 evaluation on the ROUNDED 3x3 output (what the three separate launches compute, minus one rounding)"""
    cin = cout = 64
    rng = np.random.default_rng(n3 + hw[0] + hw[1] + 7)
    x = h(rng.standard_normal((n, cin, hw[0], hw[1])))
    xs = h(rng.standard_normal((n, 64, hw[0], hw[1])))
    w = h(rng.standard_normal((cout, 3, 3, cin)) * np.sqrt(2.0 / (cin * 9))); b = rng.standard_normal(cout).astype(np.float32)
    w3 = h(rng.standard_normal((n3, 1, 1, cout)) * np.sqrt(1.0 / cout)); b3 = rng.standard_normal(n3).astype(np.float32)
    ws = h(rng.standard_normal((n3, 1, 1, 64)) * np.sqrt(1.0 / 64)); bs = rng.standard_normal(n3).astype(np.float32)
    slope3 = 0.25 if act3 == 2 else 0.0
    xd, wd, bd = to_dev(x), torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda()
    w3d, wsd, xsd = torch.from_numpy(w3).cuda().half(), torch.from_numpy(ws).cuda().half(), to_dev(xs)
    y3 = ops.conv2d_xhead_proj(xd, wd, bd, w3d, torch.from_numpy(b3 + bs).cuda(), xsd, wsd, act3=act3, slope3=slope3, cfg=cfg_named("ws:64->64 w4+4 xhead side"))
    assert torch.equal(y3, ops.conv2d_xhead_proj(xd, wd, bd, w3d, torch.from_numpy(b3 + bs).cuda(), xsd, wsd, act3=act3, slope3=slope3))   # (the launcher's own pick is that configuration)
    yplain = ops.conv2d(xd, wd, bd, pads=(1, 1, 1, 1), act=1)
    y27 = ops.conv2d(yplain, w3d, torch.from_numpy(b3).cuda())                                          # the three separate launches
    z = ops.conv2d(xsd, wsd, torch.from_numpy(bs).cuda(), act=act3, slope=slope3, residual=y27)
    ref = O.conv2d(to_host(yplain), w3, b3) + O.conv2d(xs, ws, bs)
    if act3:
        ref = np.where(ref >= 0, ref, ref * slope3)
    assert_close(to_host(y3), ref, 2 * cout)
    assert float((y3.float() - z.float()).abs().max()) <= 2e-2 * float(z.float().abs().max())


def test_zz_every_tile_configuration_ran_unskipped():
    """1880 of this file's parametrisations skip with "does not apply": a configuration whose predicate rejected EVERY shape of every test would be green
    forever (VERDICT r3 item 4 ii).  ops.CFG_LOG holds the configurations that some *_cfg call in this process ran to SM_OK; after a complete run of this file
    that must be all of them.  (A filtered run — `-k`, a single test — has not swept the pool: skipped, not judged.)"""
    if len(CFGS_RUN) < len(ALL_CFGS) // 2:
        pytest.skip("partial run of tests/test_gpu_ops.py: %d of %d configurations exercised" % (len(CFGS_RUN), len(ALL_CFGS)))
    never = [(c, ops.config_name(c)) for c in ALL_CFGS if c not in CFGS_RUN]
    assert not never, "tile configurations no test ran: %s" % never


def test_kernel_families_shape_sweeps_against_the_default_kernels():
    """Seeded sweeps (round 6; the same method found the missing zero slot of the 128-channel block kernel): the classifier kernel against an fp32 matrix product over
    M = 1 .. 256 rows, every K / N step; every stride-2 and stride-1 wave-split configuration against the launcher's default kernel on random image sizes (rows with
    and without a spare zero column, bands and whole images, one-sided padding), channel counts and activations"""
    from smelter_amd._native import SmelterError
    rng = np.random.default_rng(5); torch.manual_seed(5)
    for it in range(100):
        m = int(rng.integers(1, 257)); k = 32 * int(rng.integers(8, 80)); n = 4 * int(rng.integers(1, 300)); act = int(rng.integers(0, 3))
        x = torch.randn((m, 1, 1, k), device="cuda").half(); w = (torch.randn((n, 1, 1, k), device="cuda") * (2.0 / k) ** 0.5).half(); b = torch.randn(n, device="cuda")
        y = ops.conv2d(x, w, b, act=act, slope=0.25).float().reshape(m, n)
        ref = x.reshape(m, k).float() @ w.reshape(n, k).float().t() + b
        ref = torch.relu(ref) if act == 1 else torch.where(ref < 0, ref * 0.25, ref) if act == 2 else ref
        assert not bool(((y - ref).abs() > ref.abs() * 2.0 ** -9 + float(ref.abs().max()) * 2.0 ** -10).any()), ("fc", m, k, n, act)
    for label, stride, sizes in (("kq/2:", 2, (3, 60)), ("kq:", 1, (1, 40))):
        cfgs = cfgs_named(label)
        took = 0
        for it in range(60):
            n = int(rng.integers(1, 7)); hh = int(rng.integers(*sizes)); ww = int(rng.integers(7, 60)); cin = 64 * int(rng.integers(1, 7)); cout = 8 * int(rng.integers(1, 40))
            act = int(rng.integers(0, 3)); pads = (1, 1, int(rng.integers(0, 2)) if stride == 2 else 1, int(rng.integers(0, 2)) if stride == 2 else 1)
            x = torch.randn((n, hh, ww, cin), device="cuda").half(); w = (torch.randn((cout, 3, 3, cin), device="cuda") * (2.0 / (9 * cin)) ** 0.5).half(); b = torch.randn(cout, device="cuda")
            ref = ops.conv2d(x, w, b, stride=(stride, stride), pads=pads, act=act, slope=0.25).float()
            for c in cfgs:
                try:
                    y = ops.conv2d(x, w, b, stride=(stride, stride), pads=pads, act=act, slope=0.25, cfg=c).float()
                except SmelterError as e:
                    assert "does not apply" in e.detail
                    continue
                took += 1
                assert y.shape == ref.shape and not bool(((y - ref).abs() > ref.abs() * 2.0 ** -9 + float(ref.abs().max()) * 2.0 ** -10).any()), (ops.config_name(c), n, hh, ww, cin, cout, act, pads)
        assert took >= 60, label


def test_fused_inverted_residual_shape_sweep_against_the_three_launches():
    """Seeded sweep over sm_mbconv_fwd_kind's domain (round 6) — image sizes 2 .. 59, Cin 8 .. 192, hidden widths 8 .. 960, Cout 8 .. 320, both strides, with and without the
    expand layer and the residual, every kernel by name and the launcher's choice — against the same block as three sm_conv2d_fwd launches.  What it found: the
    register-window kernel's LDS size was computed with K = Cin + 2 rounded up where its instantiations use two or three whole k steps (Cin = 8: the last wave's
    buffer lay behind the allocation); kernel 1 took a depthwise-first block WITH a residual and dropped the residual; naming a kernel that does not take the block fell
    through to the older kernels instead of the launcher's choice."""
    from smelter_amd._native import SmelterError
    rng = np.random.default_rng(7); torch.manual_seed(2)
    took, kinds = 0, set()
    for it in range(260):
        n = int(rng.integers(1, 6)); hh = int(rng.integers(2, 60)); ww = int(rng.integers(2, 60)); stride = int(rng.choice([1, 2]))
        cin = 8 * int(rng.integers(1, 25)); ch = cin * int(rng.choice([1, 3, 6])) if rng.integers(0, 4) else 32 * int(rng.integers(1, 31)); cout = 8 * int(rng.integers(1, 41))
        ch = min(ch, 960)
        expand = bool(rng.integers(0, 5))
        if not expand:
            ch = cin
        residual = bool(rng.integers(0, 2)) and stride == 1 and cin == cout
        if rng.integers(0, 3) == 0 and stride == 1:
            cout = cin; residual = True
        acts = (int(rng.choice([1, 2])), int(rng.choice([1, 2])), int(rng.choice([0, 1]))); slopes = (0.2, 0.3, 0.0)
        x = torch.randn((n, hh, ww, cin), device="cuda").half()
        w1 = (torch.randn((ch, 1, 1, cin), device="cuda") * (2.0 / cin) ** 0.5).half() if expand else None; b1 = torch.randn(ch, device="cuda") * 0.2 if expand else None
        wd = (torch.randn((ch, 3, 3), device="cuda") * (2.0 / 9) ** 0.5).half(); bd = torch.randn(ch, device="cuda") * 0.2
        w2 = (torch.randn((cout, 1, 1, ch), device="cuda") * (2.0 / ch) ** 0.5).half(); b2 = torch.randn(cout, device="cuda") * 0.2
        ref = None
        for kern in (0, 1, 2, 3, 4):
            try:
                y = ops.mbconv(x, w1.reshape(ch, cin) if expand else None, b1, wd, bd, w2.reshape(cout, ch), b2, stride=stride, residual=residual, acts=acts, slopes=slopes, kernel=kern).float()
            except SmelterError as e:
                assert "unsupported block geometry" in e.detail, e.detail
                break                                                       # (no kernel takes the block: the executor keeps its layers)
            if ref is None:
                t = ops.conv2d(x, w1, b1, act=acts[0], slope=slopes[0]) if expand else x
                t = ops.conv2d(t, wd.reshape(ch, 3, 3, 1), bd, stride=(stride, stride), pads=(1, 1, 1, 1), groups=ch, act=acts[1], slope=slopes[1])
                ref = ops.conv2d(t, w2, b2, residual=x if residual else None, act=acts[2], slope=slopes[2]).float()
            took += 1; kinds.add(kern)
            what = (kern, n, hh, ww, cin, ch, cout, stride, expand, residual, acts)
            assert y.shape == ref.shape and not bool(torch.isnan(y).any()), what
            assert not bool(((y - ref).abs() > ref.abs() * 2.0 ** -7 + float(ref.abs().max()) * 2.0 ** -8).any()), what
    assert took >= 300 and kinds == {0, 1, 2, 3, 4}
    # the three defects by name
    from smelter_amd._native import MbConvDesc
    d = MbConvDesc(2, 18, 30, 18, 30, 1, 24, 24, 24, 0, 1, 1, 0.0, 1, 0.0, 0, 0.0, 24, 24)
    assert not ops.lib().sm_mbconv_supported(ctypes.byref(d))              # depthwise-first block with a residual: no kernel's
