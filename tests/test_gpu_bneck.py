"""sm_bottleneck_fwd (csrc/kernels/bneck.hip) on a real MI355X against the CPU oracle: a whole bottleneck block — 1x1 reduce, 3x3 / stride 1 / pad 1, 1x1 expand + residual,
each with its activation (Converters.swift:187-338, :430-451, :342-359) — as ONE launch, the two hidden tensors in LDS only.

The oracle runs the three layers one after the other and rounds both hidden tensors to fp16 (the executor's rounding list: what the layer-by-layer launches store), so the
comparison is the kernel-level tolerance of test_gpu_ops.py: one fp16 ulp of the value plus an accumulation-order floor.  Besides the oracle the fused launch is compared
bit for bit with the two launches it replaces (reduce, then 3x3 + expand head) where those exist (C3 <= 256 with a residual)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle as O
from smelter_amd import ops
from smelter_amd._native import BneckDesc

pytestmark = pytest.mark.gpu


def h(a):
    return O.round_to_half(np.asarray(a, np.float32))


def to_dev(a_nchw):
    return torch.from_numpy(np.ascontiguousarray(a_nchw.transpose(0, 2, 3, 1))).cuda().half()


def to_host(t_nhwc):
    torch.cuda.synchronize()
    return t_nhwc.float().cpu().numpy().transpose(0, 3, 1, 2)


def act(v, a, sl):
    return np.maximum(v, 0) if a == ops.ACT_RELU else np.where(v < 0, v * sl, v) if a == ops.ACT_PRELU else v


def make_block(rng, n, hh, ww, c0, c3, cm=64):
    x = h(rng.standard_normal((n, c0, hh, ww)))
    w1 = h(rng.standard_normal((cm, 1, 1, c0)) * np.sqrt(2.0 / c0)); b1 = rng.standard_normal(cm).astype(np.float32) * 0.2
    w2 = h(rng.standard_normal((cm, 3, 3, cm)) * np.sqrt(2.0 / (9 * cm))); b2 = rng.standard_normal(cm).astype(np.float32) * 0.2
    w3 = h(rng.standard_normal((c3, 1, 1, cm)) * np.sqrt(2.0 / cm)); b3 = rng.standard_normal(c3).astype(np.float32) * 0.2
    return x, w1, b1, w2, b2, w3, b3


def oracle_block(x, w1, b1, w2, b2, w3, b3, res, acts, slopes):
    t1 = h(act(O.conv2d(x, w1, b1), acts[0], slopes[0]))
    t2 = h(act(O.conv2d(t1, w2, b2, pads=(1, 1, 1, 1)), acts[1], slopes[1]))
    y = O.conv2d(t2, w3, b3)
    if res is not None:
        y = y + res
    return act(y, acts[2], slopes[2]), t1, t2


def assert_close(got, ref, k):
    """Three chained layers with two hidden fp16 roundings: where the kernel's accumulation order flips the rounding of a hidden element the output moves by a few 1e-4,
    which in turn may flip ITS rounding — two fp16 ulps of the value plus an absolute floor (the tolerance of the fused inverted-residual block tests), and the mismatches
    must stay rare.  The sharp statement is test_bottleneck_bit_identical_to_the_two_launches: each of those launches meets the oracle at the one-ulp kernel tolerance."""
    ref16 = h(ref)
    tol = np.abs(ref16) * 2.0 ** -9 + np.abs(ref16).max() * 2.0 ** -10
    bad = np.abs(got - ref16) > tol
    assert not bad.any(), "max err %g at %d of %d elements" % (np.abs(got - ref16).max(), bad.sum(), bad.size)
    assert np.mean(got != ref16) < 0.10, "too many elements differ from the oracle's rounding: %g" % np.mean(got != ref16)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CASES = [
    # n, h, w, c0, c3, residual, acts, slopes                      -- what is exercised
    (2, 56, 56, 256, 256, True, (1, 1, 1), (0, 0, 0)),             # ResNet-50 stage 1, blocks 2 / 3: 4 column bands x 7 row bands, whole tiles
    (3, 28, 28, 256, 256, True, (1, 1, 1), (0, 0, 0)),             # TH = 7 (4 bands of 7 rows): the fifth patch fragment is half empty
    (2, 14, 14, 128, 128, True, (1, 1, 1), (0, 0, 0)),             # C0 = 128 instantiation, one column band
    (2, 20, 17, 64, 96, False, (1, 1, 0), (0, 0, 0)),              # C0 = 64, no residual, ragged in both directions, C3 = 96 (the fourth consumer wave has no channels)
    (1, 9, 30, 256, 32, False, (2, 2, 2), (0.1, 0.2, 0.3)),        # leaky activations on all three layers, C3 = 32, three column bands (the last holds 2 columns)
    (5, 7, 7, 64, 64, True, (0, 1, 1), (0, 0, 0)),                 # tiny images: one tile each; no activation behind the reduce (the padding must still be ZERO, not b1)
    (1, 3, 3, 128, 256, False, (1, 0, 1), (0, 0, 0)),              # 3 x 3 image: every tap but the centre crosses a border; linear 3x3
    (300, 8, 14, 64, 32, False, (1, 1, 1), (0, 0, 0)),             # 300 tiles on 256 blocks: ranges of one and of two tiles
]


@pytest.mark.parametrize("n,hh,ww,c0,c3,use_res,acts,slopes", CASES)
def test_bottleneck_against_oracle(n, hh, ww, c0, c3, use_res, acts, slopes):
    rng = np.random.default_rng(n * 1000 + hh * 31 + ww + c0 + c3)
    x, w1, b1, w2, b2, w3, b3 = make_block(rng, n, hh, ww, c0, c3)
    res = x if (use_res and c0 == c3) else (h(rng.standard_normal((n, c3, hh, ww))) if use_res else None)
    d = BneckDesc(n, hh, ww, c0, 64, c3, c0, c3, c3 if res is not None else 0, acts[0], slopes[0], acts[1], slopes[1], acts[2], slopes[2])
    assert ops.lib().sm_bottleneck_supported(ctypes.byref(d))
    ref, _, _ = oracle_block(x, w1, b1, w2, b2, w3, b3, res, acts, slopes)
    xd = to_dev(x)
    resd = xd if (res is x) else (to_dev(res) if res is not None else None)
    got = to_host(ops.bottleneck(xd, dev(w1.reshape(64, c0)).half(), dev(b1), dev(w2).half(), dev(b2), dev(w3.reshape(c3, 64)).half(), dev(b3), residual=resd, acts=acts, slopes=slopes))
    assert_close(got, ref, 576)


CASES128 = [
    # n, h, w, c0, c3, residual, acts, slopes                      -- the 128-channel middle (bneck128.hpp): one band tile per block, three phases
    (2, 28, 28, 512, 512, True, (1, 1, 1), (0, 0, 0)),             # ResNet-50 stage 2, blocks 2 - 4: four bands of 7 rows, 252 patch pixels, 196 output pixels per tile
    (3, 14, 14, 256, 256, True, (1, 1, 1), (0, 0, 0)),             # whole images per tile (TH = 14 = H: no halo rows inside the image)
    (2, 20, 17, 128, 384, False, (1, 1, 0), (0, 0, 0)),            # ragged: bands of 12 rows (the last holds 8), RS = 24, C0 = 128 (two x stages), no residual
    (1, 9, 30, 64, 128, False, (2, 2, 2), (0.1, 0.2, 0.3)),        # one x stage, leaky activations, 30-pixel rows (RS = 32), one pass of 32 couts per wave
    (5, 7, 7, 192, 256, True, (0, 1, 1), (0, 0, 0)),               # tiny images, an odd number of x stages, no activation behind the reduce (padding still ZERO)
    (1, 4, 56, 64, 128, False, (1, 0, 1), (0, 0, 0)),              # 56-pixel rows: three rows per tile (RS = 64), linear 3x3
    (2, 13, 31, 64, 128, False, (1, 1, 1), (0, 0, 0)),             # RS == W + 1 (no spare zero column): the bottom-right tap of a band's last pixel is one slot BEHIND the patch rows
    (2, 9, 55, 128, 256, True, (0, 2, 1), (0, 0.2, 0)),            # ... at 55-pixel rows (RS = 56, bands of 2 rows): every other row ends a band
]


@pytest.mark.parametrize("n,hh,ww,c0,c3,use_res,acts,slopes", CASES128)
def test_bottleneck128_against_oracle(n, hh, ww, c0, c3, use_res, acts, slopes):
    rng = np.random.default_rng(n * 1000 + hh * 31 + ww + c0 + c3 + 128)
    x, w1, b1, w2, b2, w3, b3 = make_block(rng, n, hh, ww, c0, c3, cm=128)
    res = x if (use_res and c0 == c3) else (h(rng.standard_normal((n, c3, hh, ww))) if use_res else None)
    d = BneckDesc(n, hh, ww, c0, 128, c3, c0, c3, c3 if res is not None else 0, acts[0], slopes[0], acts[1], slopes[1], acts[2], slopes[2])
    assert ops.lib().sm_bottleneck_supported(ctypes.byref(d))
    ref, _, _ = oracle_block(x, w1, b1, w2, b2, w3, b3, res, acts, slopes)
    xd = to_dev(x)
    resd = xd if (res is x) else (to_dev(res) if res is not None else None)
    got = to_host(ops.bottleneck(xd, dev(w1.reshape(128, c0)).half(), dev(b1), dev(w2).half(), dev(b2), dev(w3.reshape(c3, 128)).half(), dev(b3), residual=resd, acts=acts, slopes=slopes))
    assert_close(got, ref, 1152)


def test_bottleneck128_full_size_properties():
    """BASELINE.json's full size (ResNet-50 stage 2 at bs 64: 64 x 28 x 28 x 512): image independence (first, middle, last image alone = inside the batch, bit for bit), those
    three images against the oracle, and the permutation property"""
    rng = np.random.default_rng(12)
    n, hh, ww, c0 = 64, 28, 28, 512
    x1, w1, b1, w2, b2, w3, b3 = make_block(rng, 3, hh, ww, c0, c0, cm=128)
    xd = torch.randn((n, hh, ww, c0), device="cuda").half()
    pick = [0, 31, 63]
    xd[pick] = to_dev(x1)
    w1d, w2d, w3d = dev(w1.reshape(128, c0)).half(), dev(w2).half(), dev(w3.reshape(c0, 128)).half()
    b1d, b2d, b3d = dev(b1), dev(b2), dev(b3)
    full = ops.bottleneck(xd, w1d, b1d, w2d, b2d, w3d, b3d, residual=xd)
    for i in pick:
        xi = xd[i:i + 1].contiguous()
        alone = ops.bottleneck(xi, w1d, b1d, w2d, b2d, w3d, b3d, residual=xi)
        assert torch.equal(alone[0], full[i]), "image %d depends on its neighbours" % i
    ref, _, _ = oracle_block(x1, w1, b1, w2, b2, w3, b3, x1, (1, 1, 1), (0, 0, 0))
    assert_close(to_host(full[pick]), ref, 1152)
    perm = torch.randperm(n, device="cuda")
    xp = xd[perm].contiguous()
    assert torch.equal(ops.bottleneck(xp, w1d, b1d, w2d, b2d, w3d, b3d, residual=xp), full[perm])


def test_bottleneck_padding_is_zero_not_bias():
    """Outside the image T1 is the 3x3's zero padding (ONNXConvolutionPadding.swift:73-79) — not act1(b1), which is what the reduce GEMM leaves on a zero pixel.  A large positive
    b1 makes the difference visible at every border pixel."""
    rng = np.random.default_rng(7)
    n, hh, ww, c0, c3 = 2, 10, 16, 64, 64
    x, w1, b1, w2, b2, w3, b3 = make_block(rng, n, hh, ww, c0, c3)
    b1 = np.full(64, 3.0, np.float32)
    ref, _, _ = oracle_block(x, w1, b1, w2, b2, w3, b3, None, (1, 1, 1), (0, 0, 0))
    got = to_host(ops.bottleneck(to_dev(x), dev(w1.reshape(64, c0)).half(), dev(b1), dev(w2).half(), dev(b2), dev(w3.reshape(c3, 64)).half(), dev(b3)))
    assert_close(got, ref, 576)


@pytest.mark.parametrize("n,hh,ww,c0", [(2, 56, 56, 256), (3, 21, 33, 256), (2, 28, 28, 128), (4, 14, 14, 64)])
def test_bottleneck_bit_identical_to_the_two_launches(n, hh, ww, c0):
    """The fused launch = the 1x1 reduce launch + the 3x3 -> expand-head launch it replaces, bit for bit: same MFMA steps of 16 in the same order from zero, the same
    fp32 bias / activation / single rounding of both hidden tensors (what lets the executor's flip test of SM_FUSE_BNECK compare with torch.equal)."""
    c3 = c0
    d = ops.ConvDesc(n, hh, ww, 64, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, hh, ww, 1, 0.0, 64, 64, 0)
    if not ops.lib().sm_conv2d_xhead_supported(ctypes.byref(d), c3, 1):
        pytest.skip("no expand-head kernel for this shape")
    rng = np.random.default_rng(n + hh + ww + c0)
    x, w1, b1, w2, b2, w3, b3 = make_block(rng, n, hh, ww, c0, c3)
    xd = to_dev(x)
    w1d, w2d, w3d = dev(w1).half(), dev(w2).half(), dev(w3.reshape(c3, 64)).half()
    b1d, b2d, b3d = dev(b1), dev(b2), dev(b3)
    t1 = ops.conv2d(xd, w1d, b1d, act=ops.ACT_RELU)
    _, two = ops.conv2d_xhead(t1, w2d, b2d, w3d, b3d, residual=xd, act3=ops.ACT_RELU)
    one = ops.bottleneck(xd, w1d.reshape(64, c0), b1d, w2d, b2d, w3d, b3d, residual=xd)
    torch.cuda.synchronize()
    assert torch.equal(one, two), "differs at %d elements, max %g" % ((one != two).sum().item(), (one.float() - two.float()).abs().max().item())


def test_bottleneck_full_size_properties():
    """BASELINE.json's full size (ResNet-50 stage 1 at bs 64: 64 x 56 x 56 x 256) — too large for the CPU oracle in seconds, so size-independent properties instead:
    (a) the batch is 64 independent images: image i of the full launch = a launch of image i alone (first, middle, last), bit for bit;
    (b) those three images meet the oracle;
    (c) permuting the batch permutes the output."""
    rng = np.random.default_rng(11)
    n, hh, ww, c0 = 64, 56, 56, 256
    x1, w1, b1, w2, b2, w3, b3 = make_block(rng, 3, hh, ww, c0, c0)
    xd = torch.randn((n, hh, ww, c0), device="cuda").half()
    pick = [0, 31, 63]
    xd[pick] = to_dev(x1)
    w1d, w2d, w3d = dev(w1.reshape(64, c0)).half(), dev(w2).half(), dev(w3.reshape(c0, 64)).half()
    b1d, b2d, b3d = dev(b1), dev(b2), dev(b3)
    full = ops.bottleneck(xd, w1d, b1d, w2d, b2d, w3d, b3d, residual=xd)
    for k, i in enumerate(pick):
        xi = xd[i:i + 1].contiguous()
        alone = ops.bottleneck(xi, w1d, b1d, w2d, b2d, w3d, b3d, residual=xi)
        assert torch.equal(alone[0], full[i]), "image %d depends on its neighbours" % i
    ref, _, _ = oracle_block(x1, w1, b1, w2, b2, w3, b3, x1, (1, 1, 1), (0, 0, 0))
    assert_close(to_host(full[pick]), ref, 576)
    perm = torch.randperm(n, device="cuda")
    xp = xd[perm].contiguous()
    assert torch.equal(ops.bottleneck(xp, w1d, b1d, w2d, b2d, w3d, b3d, residual=xp), full[perm])


def test_bottleneck_refusals():
    """shapes the kernel does not take are refused by sm_bottleneck_supported and sm_bottleneck_fwd answers SM_ERR_INVALID_ARGUMENT (no silent other path)"""
    L = ops.lib()
    ok = BneckDesc(1, 8, 8, 256, 64, 256, 256, 256, 256, 1, 0.0, 1, 0.0, 1, 0.0)
    assert L.sm_bottleneck_supported(ctypes.byref(ok))
    for field, val in (("Cm", 96), ("C0", 512), ("C0", 96), ("C3", 288), ("C3", 40), ("act2", 3), ("x_ld", 128), ("y_ld", 250)):
        bad = BneckDesc(1, 8, 8, 256, 64, 256, 256, 256, 256, 1, 0.0, 1, 0.0, 1, 0.0)
        setattr(bad, field, val)
        assert not L.sm_bottleneck_supported(ctypes.byref(bad)), field
    x = torch.zeros((1, 8, 8, 512), device="cuda").half()
    bad = BneckDesc(1, 8, 8, 512, 64, 256, 512, 256, 0, 1, 0.0, 1, 0.0, 1, 0.0)
    w = torch.zeros((64 * 576,), device="cuda").half()
    y = torch.zeros((1, 8, 8, 256), device="cuda").half()
    rc = L.sm_bottleneck_fwd(ctypes.byref(bad), x.data_ptr(), w.data_ptr(), None, w.data_ptr(), None, w.data_ptr(), None, None, y.data_ptr(), None)
    assert rc != 0


def test_bottleneck_shape_sweep_against_the_separate_launches():
    """Seeded sweep over the domain of both kernels — image sizes from 1 x 1 to 39 x 62, every C0 / C3 step, batch 1 - 5, all activation triples, with and without a residual,
    operands that are channel slices of wider buffers (pixel strides above the channel count on all three tensors) — against the SAME block as three sm_conv2d_fwd launches.
    Found in round 6: rows without a spare zero column (W + 1 a multiple of 8) read the bottom-right tap of a band's last pixel from the neighbouring LDS plane."""
    rng = np.random.default_rng(20261001)
    torch.manual_seed(7)
    tested = 0
    for it in range(160):
        cm = int(rng.choice([64, 128]))
        if cm == 64:
            c0 = int(rng.choice([64, 128, 256])); c3 = 32 * int(rng.integers(1, 9)); hh = int(rng.integers(1, 40)); ww = int(rng.integers(1, 45))
        else:
            c0 = 64 * int(rng.integers(1, 9)); c3 = 128 * int(rng.integers(1, 5)); hh = int(rng.integers(1, 40)); ww = int(rng.integers(4, 63))
        n = int(rng.integers(1, 6))
        use_res = bool(rng.integers(0, 2))
        acts = tuple(int(a) for a in rng.integers(0, 3, 3)); slopes = tuple(float(v) for v in rng.uniform(0.05, 0.4, 3))
        xpad, ypad, rpad = (int(8 * rng.integers(0, 3)) for _ in range(3))
        d = BneckDesc(n, hh, ww, c0, cm, c3, c0 + xpad, c3 + ypad, c3 + rpad if use_res else 0, acts[0], slopes[0], acts[1], slopes[1], acts[2], slopes[2])
        if not ops.lib().sm_bottleneck_supported(ctypes.byref(d)):
            continue                                                        # (a band of this width does not fit the LDS: the executor keeps the layers)
        x = torch.randn((n, hh, ww, c0 + xpad), device="cuda").half()[..., :c0]
        w1 = (torch.randn((cm, 1, 1, c0), device="cuda") * (2.0 / c0) ** 0.5).half(); b1 = torch.randn(cm, device="cuda") * 0.2
        w2 = (torch.randn((cm, 3, 3, cm), device="cuda") * (2.0 / (9 * cm)) ** 0.5).half(); b2 = torch.randn(cm, device="cuda") * 0.2
        w3 = (torch.randn((c3, 1, 1, cm), device="cuda") * (2.0 / cm) ** 0.5).half(); b3 = torch.randn(c3, device="cuda") * 0.2
        res = torch.randn((n, hh, ww, c3 + rpad), device="cuda").half()[..., :c3] if use_res else None
        yfull = torch.full((n, hh, ww, c3 + ypad), 7.0, device="cuda").half()
        y = ops.bottleneck(x, w1.reshape(cm, c0), b1, w2, b2, w3.reshape(c3, cm), b3, residual=res, acts=acts, slopes=slopes, out=yfull[..., :c3])
        t1 = ops.conv2d(x, w1, b1, act=acts[0], slope=slopes[0])
        t2 = ops.conv2d(t1, w2, b2, pads=(1, 1, 1, 1), act=acts[1], slope=slopes[1])
        ref = ops.conv2d(t2, w3, b3, residual=res, act=acts[2], slope=slopes[2]).float()
        torch.cuda.synchronize()
        tested += 1
        what = (cm, c0, c3, n, hh, ww, use_res, acts, xpad, ypad, rpad)
        # same roundings; the separate 3x3 may add its K slices in another order: two fp16 ulps of the value + an accumulation floor, and few elements may differ at all
        tol = ref.abs() * 2.0 ** -8 + float(ref.abs().max()) * 2.0 ** -9
        assert not bool(((y.float() - ref).abs() > tol).any()), what
        assert float((y.float() != ref).float().mean()) < 0.05, what
        assert ypad == 0 or (float(yfull[..., c3:].min()) == 7.0 and float(yfull[..., c3:].max()) == 7.0), what       # nothing written beside the output slice
    assert tested >= 120
