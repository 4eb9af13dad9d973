"""Known-answer tables that pin the oracle (and, side by side, the product's host helpers) for everything Smelter
computes ITSELF — each expected value is derived by hand from the cited reference lines, not from either
implementation.  The reference has no tests of its own, so these are the pins."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from smelter_amd import _native as N
from smelter_amd import onnx_writer as ow


# ---- Float16.swift:17-77 (IEEE binary16, vImage semantics) --------------------------------------------------
def test_half_to_float_all_65536_patterns():
    bits = np.arange(65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)          # numpy's IEEE widening: third-party, not ours
    got_o = O.half_to_float(bits)
    got_p = np.empty(65536, np.float32)
    N.lib().sm_half_to_float(bits.ctypes.data, got_p.ctypes.data, 65536)
    for got in (got_o, got_p):
        same = got.view(np.uint32) == ref.view(np.uint32)
        assert (same | (np.isnan(got) & np.isnan(ref))).all()


HALF_KAT = [  # (float32 value, expected half bits) — hand-derived boundary cases of round-to-nearest-even
    (0.0, 0x0000), (-0.0, 0x8000), (1.0, 0x3C00), (-2.0, 0xC000), (65504.0, 0x7BFF),
    (65519.996, 0x7BFF),      # just below the halfway point to 2^16 -> largest finite
    (65520.0, 0x7C00),        # exactly halfway between 65504 and 65536: ties to even -> inf
    (1e9, 0x7C00), (-1e9, 0xFC00), (float("inf"), 0x7C00),
    (2.0 ** -24, 0x0001),     # smallest subnormal
    (2.0 ** -25, 0x0000),     # halfway between 0 and 2^-24: ties to even -> 0
    (2.0 ** -25 * 1.0000001, 0x0001),
    (2.0 ** -14, 0x0400),     # smallest normal
    (2.0 ** -14 - 2.0 ** -25, 0x0400),  # halfway between largest subnormal (0x3FF) and 0x400: even is 0x400
    (1.0 + 2.0 ** -11, 0x3C00),          # halfway 1.0 .. 1.0+2^-10: ties to even (mantissa 0)
    (1.0 + 3 * 2.0 ** -11, 0x3C02),      # halfway between 0x3C01 and 0x3C02 -> even 0x3C02
    (1.0 + 2.0 ** -11 + 2.0 ** -20, 0x3C01),
    (0.1, 0x2E66), (1.0 / 3.0, 0x3555), (3.14159265, 0x4248),
]


@pytest.mark.parametrize("val,bits", HALF_KAT)
def test_float_to_half_kat(val, bits):
    x = np.array([val], np.float32)
    assert int(O.float_to_half(x)[0]) == bits
    h = np.empty(1, np.uint16)
    N.lib().sm_float_to_half(x.ctypes.data, h.ctypes.data, 1)
    assert int(h[0]) == bits


def test_float_to_half_matches_numpy_rne_on_a_dense_sweep():
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.standard_normal(100000).astype(np.float32) * s for s in (1e-8, 1e-6, 6e-5, 1e-3, 1, 100, 65000)])
    fin = np.arange(0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = ((fin[:-1] + fin[1:]) / 2).astype(np.float32)                     # every tie point of the format
    xs = np.concatenate([xs, mids, -mids, np.nextafter(mids, np.float32(np.inf)), np.nextafter(mids, np.float32(-np.inf))])
    with np.errstate(over="ignore"):
        ref = xs.astype(np.float16).view(np.uint16)
    assert np.array_equal(O.float_to_half(xs), ref)
    h = np.empty(xs.shape, np.uint16)
    N.lib().sm_float_to_half(xs.ctypes.data, h.ctypes.data, xs.size)
    assert np.array_equal(h, ref)


# ---- Array+Extensions.swift:52-93 ---------------------------------------------------------------------------
def test_reformat_conv_weight_iota_2x3x2x2():
    # in[oc][ic][kh][kw] = linear index; out[oc][kh][kw][ic] (:78-83).  Expected table written out by hand:
    # oc=0: (kh,kw)=(0,0) -> ic 0,1,2 = 0,4,8 ; (0,1) -> 1,5,9 ; (1,0) -> 2,6,10 ; (1,1) -> 3,7,11 ; oc=1: +12
    expect = [0, 4, 8, 1, 5, 9, 2, 6, 10, 3, 7, 11, 12, 16, 20, 13, 17, 21, 14, 18, 22, 15, 19, 23]
    for dt in (np.float32, np.uint16):
        w = np.arange(24).astype(dt)
        assert O.reformat_conv_weight(w, 2, 3, 2, 2).tolist() == expect
        out = np.empty_like(w)
        assert N.lib().sm_reformat_conv_weight(w.ctypes.data, out.ctypes.data, w.itemsize, 2, 3, 2, 2, 0) == 0
        assert out.tolist() == expect


def test_reformat_conv_transpose_weight_flip_2x2x2x2():
    # in[ic][oc][kh][kw]; out[oc][KH-1-kh][KW-1-kw][ic] (:71-76).  in index = ic*8 + oc*4 + kh*2 + kw.
    # out (oc=0): position (0,0) takes kh=1,kw=1 -> ic0: 3, ic1: 11 ; (0,1) kh=1,kw=0 -> 2,10 ; (1,0) -> 1,9 ; (1,1) -> 0,8
    expect = [3, 11, 2, 10, 1, 9, 0, 8, 7, 15, 6, 14, 5, 13, 4, 12]
    w = np.arange(16, dtype=np.float32)
    assert O.reformat_conv_weight(w, 2, 2, 2, 2, True).tolist() == expect
    out = np.empty_like(w)
    N.lib().sm_reformat_conv_weight(w.ctypes.data, out.ctypes.data, 4, 2, 2, 2, 2, 1)
    assert out.tolist() == expect
    # and it equals the reference's own numpy recipe transpose([1,2,3,0])[:, ::-1, ::-1, :] (ONNX2MPS.py:58-62)
    assert np.array_equal(np.asarray(expect, np.float32).reshape(2, 2, 2, 2), w.reshape(2, 2, 2, 2).transpose(1, 2, 3, 0)[:, ::-1, ::-1, :])


# ---- ONNXConvolutionPadding.swift:91-113, PyTorchPoolPadding.swift:56-75,94-103 ----------------------------------
CONV_SIZE_KAT = [  # (in, k, s, pad_begin, pad_end) -> out = (in + pb + pe - k) / s + 1   (integer division)
    ((224, 7, 2, 3, 3), 112), ((56, 3, 1, 1, 1), 56), ((56, 1, 1, 0, 0), 56), ((56, 3, 2, 1, 1), 28), ((56, 1, 2, 0, 0), 28),
    ((28, 3, 2, 1, 1), 14), ((14, 3, 2, 1, 1), 7), ((7, 3, 1, 1, 1), 7), ((112, 3, 2, 1, 1), 56), ((512, 3, 1, 1, 1), 512),
    ((5, 3, 2, 0, 0), 2), ((6, 3, 2, 0, 1), 3), ((10, 3, 1, 2, 0), 10),
]


@pytest.mark.parametrize("args,out", CONV_SIZE_KAT)
def test_conv_padded_size(args, out):
    n, k, s, pb, pe = args
    assert O.conv_padded_size(n, n, (k, k), (s, s), (pb, pb, pe, pe)) == (out, out)
    ow_, oh_ = C.c_int32(), C.c_int32()
    N.lib().sm_conv_padded_size(n, n, k, k, s, s, pb, pb, pe, pe, 0, 0, 0, C.byref(ow_), C.byref(oh_))
    assert (ow_.value, oh_.value) == (out, out)


def test_conv_transpose_padded_size():
    # (in-1)*s - pt - pb + k + output_padding (:97-103): in=7,s=2,k=3,p=1,op=1 -> 6*2-2+3+1 = 14
    assert O.conv_padded_size(7, 7, (3, 3), (2, 2), (1, 1, 1, 1), (1, 1), True) == (14, 14)
    ow_, oh_ = C.c_int32(), C.c_int32()
    N.lib().sm_conv_padded_size(7, 5, 3, 3, 2, 2, 1, 1, 1, 1, 1, 0, 1, C.byref(ow_), C.byref(oh_))
    assert (ow_.value, oh_.value) == (13, 10)      # W: 12-2+3+0 = 13 ; H: 8-2+3+1 = 10


POOL_KAT = [  # (in, k, s, p) -> (out = Int(Float(in+2p-k)/Float(s) + 1), window start = -((out-1)*s + k - in)/2)
    ((112, 3, 2, 1), (56, 0)),     # read = 55*2+3 = 113, extra 1 -> 1/2 = 0  (the reference's offset rule, NOT -p)
    ((56, 2, 2, 0), (28, 0)), ((7, 7, 1, 0), (1, 0)), ((8, 3, 2, 1), (4, 0)), ((9, 3, 2, 1), (5, -1)),
    ((6, 3, 1, 1), (6, -1)), ((5, 2, 2, 0), (2, 0)),
]


@pytest.mark.parametrize("args,exp", POOL_KAT)
def test_pool_size_and_window_start(args, exp):
    n, k, s, p = args
    out, start = exp
    assert O.pool_padded_size(n, n, (k, k), (s, s), (p, p)) == (out, out)
    assert O.pool_window_start(n, out, k, s) == start
    ow_, oh_ = C.c_int32(), C.c_int32()
    N.lib().sm_pool_padded_size(n, n, k, k, s, s, p, p, C.byref(ow_), C.byref(oh_))
    assert (ow_.value, oh_.value) == (out, out)
    assert N.lib().sm_pool_window_start(n, out, k, s) == start


# ---- MPSImage+Extensions.swift:26-59 -----------------------------------------------------------------------------
@pytest.mark.parametrize("c", [1, 2, 3, 4, 5, 8])
def test_to_float_array_packing(c):
    n, h, w = 2, 2, 3
    x = np.arange(n * c * h * w, dtype=np.float32).reshape(n, c, h, w)
    got = O.pack_float_array(x)
    slices, comps = (c + 3) // 4, (c if c < 3 else 4)
    assert got.size == w * h * (c if c < 3 else slices * 4) * n == N.lib().sm_float_array_count(n, c, h, w)
    # element (image i, slice s, y, x, k) sits at ((i*slices + s)*h*w + y*w + x)*comps + k and holds channel 4s+k (0 if padding)
    for i in range(n):
        for s in range(slices):
            for y in range(h):
                for xx in range(w):
                    for k in range(comps):
                        ch = 4 * s + k
                        exp = x[i, ch, y, xx] if ch < c else 0.0
                        assert got[((i * slices + s) * h * w + y * w + xx) * comps + k] == exp


def test_resnet_logits_packing_is_plain_row_major():
    x = np.random.default_rng(0).standard_normal((3, 1000, 1, 1)).astype(np.float32)
    assert np.array_equal(O.pack_float_array(x), x.reshape(-1))


# ---- Onnx_TensorProto+Extensions.swift:2-66 via Constant -> Upsample scales (.integers) and PRelu / weights (.floats) ----
def _upsample_model(scales_tensor_bytes):
    b = ow.GraphBuilder("g")
    b.add_input("input", [1, 1, 2, 2])
    b.inits.append(scales_tensor_bytes)
    y = b.add_node("Upsample", ["input", "scales"], mode="nearest")
    b.add_output(y, [1, 1, 4, 6])
    return b.to_model()


@pytest.mark.parametrize("kind", ["f32_raw", "f32_typed", "f16_raw", "i64_raw", "i64_typed", "f64_typed", "i32_typed"])
def test_integers_accessor_every_dtype_branch(kind):
    vals = [1, 1, 2, 3]
    t = {"f32_raw": lambda: ow.tensor("scales", np.array([1, 1, 2.9, 3.5], np.float32)),        # Int(Float) truncates (:21-24)
         "f32_typed": lambda: ow.tensor("scales", np.array([1, 1, 2.9, 3.5], np.float32), raw=False),
         "f16_raw": lambda: ow.tensor("scales", np.array([1, 1, 2.5, 3.75], np.float16)),       # :29-30
         "i64_raw": lambda: ow.tensor("scales", np.array(vals, np.int64)),                      # :11-14
         "i64_typed": lambda: ow.tensor("scales", np.array(vals, np.int64), raw=False),
         "f64_typed": lambda: ow.tensor("scales", np.array([1, 1, 2.2, 3.9], np.float64), raw=False),
         "i32_typed": lambda: ow.tensor("scales", np.array(vals, np.int32), raw=False)}[kind]()
    x = np.arange(4, dtype=np.float32).reshape(1, 1, 2, 2)
    y = O.Graph(_upsample_model(t)).run(x)
    assert y.shape == (1, 1, 4, 6)                                     # sy=2, sx=3
    assert np.array_equal(y[0, 0], np.repeat(np.repeat(x[0, 0], 2, 0), 3, 1))


def test_integers_accessor_rejects_string_tensor():
    t = ow.f_packed_varints(1, [4]) + ow.f_varint(2, ow.STRING) + ow.f_bytes(8, "scales")
    with pytest.raises(O.OracleError) as e:
        O.Graph(_upsample_model(t)).run(np.zeros((1, 1, 2, 2), np.float32))
    assert e.value.kind == "unsupportedDtype"                          # the reference calls fatalError (:32)


def test_floats_accessor_fp16_raw_weights_equal_fp32_weights_rounded():
    rng = np.random.default_rng(0)
    w32 = rng.standard_normal((4, 2, 3, 3)).astype(np.float32)

    def model(w):
        b = ow.GraphBuilder("g")
        b.add_input("input", [1, 2, 5, 5])
        wn = b.add_init("w", w)
        y = b.add_node("Conv", ["input", wn], dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1])
        b.add_output(y, [1, 4, 5, 5])
        return b.to_model()
    x = rng.standard_normal((1, 2, 5, 5)).astype(np.float32)
    a = O.Graph(model(w32.astype(np.float16))).run(x)
    b_ = O.Graph(model(w32.astype(np.float16).astype(np.float32))).run(x)
    assert np.array_equal(a, b_)


# ---- round 2: converters no BASELINE model instantiates, pinned by independent numpy / torch restatements ----------------
def _one_op_model(op, in_dims, out_dims, inits=(), **attrs):
    b = ow.GraphBuilder("g")
    b.add_input("input", list(in_dims))
    names = ["input"] + [b.add_init(n, a) for n, a in inits]
    y = b.add_node(op, names, **attrs)
    b.add_output(y, list(out_dims))
    return b.to_model()


@pytest.mark.parametrize("shape,expect_chw", [([6, 2, -1], (4, 2, 6)), ([0, -1, 4], (4, 12, 1)), ([4, 0, 0], (3, 4, 4)), ([48], (1, 1, 48))])
def test_reshape_whc_order_zero_and_minus_one_rules(shape, expect_chw):
    """Converters.swift:844-864: shape is read as (W, H, C), missing entries = 1, `0` copies (channels, width, height)[i] of the
    SOURCE Shape — so W := channels (= 1), H := the source WIDTH, C := the source HEIGHT, a quirk kept —, one -1 is inferred from the total;
    the data keep their linear (c, y, x) order"""
    x = np.arange(2 * 48, dtype=np.float32).reshape(2, 4, 3, 4)         # C=4, H=3, W=4
    m = _one_op_model("Reshape", [1, 4, 3, 4], [1] + list(expect_chw), inits=[("shape", np.array(shape, np.int64))])
    y = O.Graph(m).run(x)
    assert y.shape == (2,) + expect_chw
    assert np.array_equal(y.reshape(2, -1), x.reshape(2, -1))


def test_dropout_is_identity_under_every_output_name_and_needs_ratio():
    x = np.random.default_rng(0).standard_normal((1, 3, 4, 4)).astype(np.float32)
    b = ow.GraphBuilder("g"); b.add_input("input", [1, 3, 4, 4])
    b.add_node_named("Dropout", ["input"], ["d_out", "d_mask"], ratio=0.3)
    y = b.add_node("Relu", ["d_mask"])                                 # the reference publishes the node under every name (:263-273)
    b.add_output(y, [1, 3, 4, 4])
    assert np.array_equal(O.Graph(b.to_model()).run(x), np.maximum(x, 0))
    b2 = ow.GraphBuilder("g"); b2.add_input("input", [1, 3, 4, 4]); y2 = b2.add_node("Dropout", ["input"]); b2.add_output(y2, [1, 3, 4, 4])
    with pytest.raises(O.OracleError) as e:
        O.Graph(b2.to_model()).run(x)
    assert e.value.kind == "noSuchOutput"                              # guard at Converters.swift:920-925


def test_leaky_relu_and_matmul_supersets_against_numpy():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 3, 2)).astype(np.float32)
    for kw, a in (({}, 0.01), ({"alpha": 0.3}, 0.3)):
        y = O.Graph(_one_op_model("LeakyRelu", [1, 5, 3, 2], [1, 5, 3, 2], **kw)).run(x)
        assert np.array_equal(y, np.where(x >= 0, x, x * np.float32(a)).astype(np.float32))
    xf = rng.standard_normal((3, 7, 1, 1)).astype(np.float32)
    w = rng.standard_normal((7, 4)).astype(np.float32)
    y = O.Graph(_one_op_model("MatMul", [1, 7, 1, 1], [1, 4], inits=[("w", w)])).run(xf)
    assert y.shape == (3, 4, 1, 1)
    assert np.abs(y.reshape(3, 4) - xf.reshape(3, 7).astype(np.float64) @ w.astype(np.float64)).max() < 1e-5


def test_clip_superset_against_numpy():
    """ONNX Clip with lower bound 0 (ReLU6 of a torchvision MobileNet-v2 export) — not in the reference registry; both opset forms, the unbounded
    case = Relu, everything else refused"""
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((2, 5, 3, 2)) * 5).astype(np.float32)
    y = O.Graph(_one_op_model("Clip", [1, 5, 3, 2], [1, 5, 3, 2], min=0.0, max=6.0)).run(x)                     # opset 6: attributes
    assert np.array_equal(y, np.clip(x, 0, 6))
    y = O.Graph(_one_op_model("Clip", [1, 5, 3, 2], [1, 5, 3, 2], inits=[("lo", np.zeros((), np.float32)), ("hi", np.full((), 2.5, np.float32))])).run(x)   # opset 11: inputs
    assert np.array_equal(y, np.clip(x, 0, 2.5))
    y = O.Graph(_one_op_model("Clip", [1, 5, 3, 2], [1, 5, 3, 2], min=0.0)).run(x)                              # no upper bound: Relu
    assert np.array_equal(y, np.maximum(x, 0))
    assert np.array_equal(O.neuron("clip", x, 6.0), np.clip(x, 0, 6))
    for kw, kind in ((dict(min=-1.0, max=1.0), "unknownNodeOpType"), (dict(max=6.0), "unknownNodeOpType"), (dict(min=0.0, max=-2.0), "inconsistentState")):
        with pytest.raises(O.OracleError) as e:
            O.Graph(_one_op_model("Clip", [1, 5, 3, 2], [1, 5, 3, 2], **kw)).run(x)
        assert e.value.kind == kind


def test_average_pool_converter_against_torch():
    torch = pytest.importorskip("torch")
    x = np.random.default_rng(1).standard_normal((2, 3, 9, 8)).astype(np.float32)
    m = _one_op_model("AveragePool", [1, 3, 9, 8], [1, 3, 5, 4], kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2])
    y = O.Graph(m).run(x)
    ref = torch.nn.functional.avg_pool2d(torch.from_numpy(x).double(), 3, 2, 1, count_include_pad=True).numpy()
    # 9 -> 5 and 8 -> 4: the reference's window offset -floor(((out-1)s+k-in)/2) equals -p for H (odd total) and 0 for W
    # (PyTorchPoolPadding.swift:60-75), so only the H axis compares 1:1 with torch's padding = 1; W windows start at 0
    refw = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(torch.from_numpy(x).double(), (0, 1, 1, 1)), 3, 2, 0).numpy()
    assert y.shape == (2, 3, 5, 4)
    assert np.abs(y - refw).max() < 1e-5 or np.abs(y - ref).max() < 1e-5


@pytest.mark.parametrize("k,d,pad", [(3, 2, 2), (3, 3, 1), (5, 2, 0)])
def test_dilated_conv_taps_are_centred_the_mps_way(k, d, pad):
    """offset = k/2 - pad (ONNXConvolutionPadding.swift:73-79), taps at centre + (kh - k/2)*d, output size from the dilation-free
    rule (:105-110) == torch conv2d with padding pad + (k/2)(d-1), cropped / zero-extended to the reference's output grid"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(k * 10 + d)
    x = rng.standard_normal((1, 2, 11, 10)).astype(np.float32)
    w = rng.standard_normal((3, 2, k, k)).astype(np.float32)
    ho, wo = 11 + 2 * pad - k + 1, 10 + 2 * pad - k + 1
    m = _one_op_model("Conv", [1, 2, 11, 10], [1, 3, ho, wo], inits=[("w", w)], dilations=[d, d], group=1, kernel_shape=[k, k],
                      pads=[pad] * 4, strides=[1, 1])
    y = O.Graph(m).run(x)
    assert y.shape == (1, 3, ho, wo)
    big = 3 * k * d                                                     # generous zero border, then cut the reference's window out of it
    xp = torch.nn.functional.pad(torch.from_numpy(x).double(), (big, big, big, big))
    full = torch.nn.functional.conv2d(xp, torch.from_numpy(w).double(), dilation=d).numpy()
    off = big - pad - (k // 2) * (d - 1)                                # output o reads input rows o - pad' + kh*d
    ref = full[:, :, off:off + ho, off:off + wo]
    assert np.abs(y - ref).max() < 1e-4


def test_softplus_is_finite_for_large_inputs():
    x = np.array([-100.0, -1.0, 0.0, 1.0, 30.0, 100.0, 65504.0], np.float32).reshape(1, 7, 1, 1)
    y = O.neuron("softplus", x).ravel()
    assert np.all(np.isfinite(y))
    assert np.allclose(y, np.logaddexp(0.0, x.ravel().astype(np.float64)), rtol=1e-6, atol=1e-30)


def test_reshape_net_against_an_independent_numpy_walk():
    """zoo.reshape_net end to end: conv / pools by torch, every reshape as numpy's C-order reshape of the NCHW array"""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    from smelter_amd import onnx_reader as orr, zoo
    model = zoo.reshape_net()
    m = orr.parse_model(model)
    A = {t.name: t.to_array() for t in m.initializer}
    convs = [n for n in m.node if n.op_type == "Conv"]
    x = zoo.synthetic_images(2, 3, (16, 12))
    t = torch.from_numpy(x).double()
    t = F.conv2d(t, torch.from_numpy(A[convs[0].input[1]]).double(), torch.from_numpy(A[convs[0].input[2]]).double(), padding=1)
    t = F.leaky_relu(t, 0.1)
    t = F.avg_pool2d(F.pad(t, (0, 1, 0, 1)), 3, 2, 0)                   # 16x12 -> 8x6, windows start at 0 (even total padding)
    t = F.interpolate(t, scale_factor=2, mode="nearest")
    t = F.avg_pool2d(t, 2, 2)
    a = t.numpy()                                                       # (2, 8, 8, 6)
    a = a.reshape(2, 8, 4, 12).reshape(2, 8, 48, 1)
    t = F.conv2d(torch.from_numpy(a), torch.from_numpy(A[convs[1].input[1]]).double(), torch.from_numpy(A[convs[1].input[2]]).double())
    a = t.mean(dim=(2, 3)).numpy().reshape(2, 2, 2, 2)
    a = 1.0 / (1.0 + np.exp(-a))
    mm = [n for n in m.node if n.op_type == "MatMul"][0]
    ref = a.reshape(2, 8) @ A[mm.input[1]].astype(np.float64)
    got = O.Graph(model).run(x)
    assert got.shape == (2, 5, 1, 1)
    assert np.abs(got.reshape(2, 5) - ref).max() < 1e-5
