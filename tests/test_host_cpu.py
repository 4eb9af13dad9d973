"""CPU tier for the product's host side: the C-ABI library loads and exports every symbol the header declares, the
graph builder mirrors ONNXGraph's behaviour (flavour detection, shapes, every error case, same error KIND as the
oracle), and the weight preparation (BN fold + fp16 cast + OHWI) matches an independent numpy restatement.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
from smelter_amd import _native as N
from smelter_amd import onnx2mps, onnx_reader as orr, onnx_writer as ow, zoo
from smelter_amd.graph import Configuration, Format, ONNXGraph, SmelterError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "smelter_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:sm|smelter)_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 60
    L = C.CDLL(N.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, "declared in include/smelter_hip.h but not exported: %s" % missing
    assert declared == set(N.SYMBOLS), "python binding table drifted from the header: %s" % (declared ^ set(N.SYMBOLS))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        N.lib()


def test_product_tree_never_references_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "smelter_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                code = re.sub(r'""".*?"""', "", src, flags=re.S)                 # prose may mention it; code may not
                code = "\n".join(l for l in code.splitlines() if not l.strip().startswith(("#", "//", "*", "/*")))
                assert not re.search(r"import\s+oracle|from\s+oracle|liborc|orc_[a-z]+\s*\(|oracle/", code), f


def test_flavour_detection_and_output_shapes():
    g = ONNXGraph(zoo.toy())
    assert g.modelFormat == Format.onnx and g.numNodes == 9
    s = g.outputShapes[0]                              # ONNXGraph.swift:74-80: channels=dim1, width=dim2, height=dim3
    assert (s.channels, s.width, s.height, s.depth) == (16, 64, 64, 1)
    mps = onnx2mps.optimize_model(zoo.toy(), half=True)
    g2 = ONNXGraph(mps)
    assert g2.modelFormat == Format.mpsFlavor and g2.numNodes == 6      # 3 BN folded away
    assert O.Graph(mps).is_mps_flavor
    g2.modelFormat = Format.onnx                       # public var in the reference (ONNXGraph.swift:58)
    assert g2.modelFormat == Format.onnx
    r50 = ONNXGraph(zoo.resnet50(hw=(32, 32)))
    assert r50.numNodes == 175
    assert r50.outputShapes == []                      # rank-2 graph output is dropped (ONNXGraph.swift:88)


def _conv_model(attrs, wdtype=np.float32, op="Conv", extra_nodes=(), out="y", n_out=1, weight=True):
    b = ow.GraphBuilder("g")
    b.add_input("input", [1, 8, 6, 6])
    if weight:
        b.add_init("w", np.ones((8, 8, 3, 3), wdtype))
    b.add_node_named(op, ["input", "w"], ["y"], **attrs)
    for n in extra_nodes:
        b.nodes.append(n)
    for i in range(n_out):
        b.add_output(out if i == 0 else "input", [1, 8, 6, 6])
    return b.to_model()


FULL = dict(dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1])
ERRORS = [
    ("notEnoughAttributes", _conv_model({k: v for k, v in FULL.items() if k != "dilations"})),   # Converters.swift:234-237
    ("notEnoughAttributes", _conv_model({k: v for k, v in FULL.items() if k != "kernel_shape"})),
    ("insufficientInputs", _conv_model(FULL, weight=False)),                                      # :193-194
    ("unknownNodeOpType", _conv_model(FULL, op="Gelu")),                                     # not registered: ONNXGraph.swift:173-174
    ("noSuchOutput", _conv_model(FULL, out="nope")),                                              # :182-183
    ("unsupportedOutput", _conv_model(FULL, n_out=2)),                                            # :178-180
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("Relu", ["ghost"], ["z"])])),         # :344-348
    ("unsupportedDtype", _conv_model(FULL, wdtype=np.float64)),                                   # ConvWeightArray.invalid :82-89
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("MaxPool", ["y"], ["z"], kernel_shape=[2, 2], strides=[2, 2])])),  # pads missing :654-661
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("Add", ["y", "w"], ["z"])])),         # Add with an initializer operand :436-437
    ("notEnoughAttributes", _conv_model(FULL, extra_nodes=[ow.node("Upsample", ["y"], ["z"], mode="nearest")])),  # scales missing
    ("unknownNodeOpType", _conv_model(FULL, extra_nodes=[ow.node("Upsample", ["y"], ["z"], mode="cubic", scales=[1.0, 1.0, 2.0, 2.0])])),
    ("insufficientInputs", _conv_model(FULL, extra_nodes=[ow.node("BatchNormalization", ["y", "w"], ["z"])])),    # :803-809
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("Softmax", ["y"], ["z"], axis=2)])),               # axis must be 1 (:703-704)
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("LogSoftmax", ["y"], ["z"])])),                   # axis attribute missing
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("Elu", ["y"], ["z"])])),                          # alpha attribute missing (:392-396)
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("Mul", ["y", "w"], ["z"])])),                     # initializer operand (:455-459)
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("Pad", ["y"], ["z"], mode=b"reflect")])),         # pads missing (:948-949)
    ("inconsistentState", _conv_model(FULL, extra_nodes=[ow.node("Pad", ["y"], ["z"], pads=[0, 0, 1, 1, 0, 0, 1, 1], mode=b"wrap")])),   # :966-967
    ("noSuchOutput", _conv_model(FULL, extra_nodes=[ow.node("InstanceNormalization", ["y", "w"], ["z"])])),   # 3 inputs required (:994-999)
    ("unknownNodeOpType", _conv_model(FULL, extra_nodes=[ow.node("Clip", ["y"], ["z"], min=-1.0, max=1.0)])),     # Clip superset: only the lower bound 0 (ReLU-N family)
    ("insufficientInputs", _conv_model(FULL, extra_nodes=[ow.node("Clip", ["y", "input", "w"], ["z"])])),        # opset-11 form: the bounds must be constant tensors
    ("inconsistentState", _conv_model(FULL, extra_nodes=[ow.node("Clip", ["y"], ["z"], min=0.0, max=0.0)])),     # empty range
    ("inconsistentState", _conv_model(dict(FULL, group=2), op="ConvTranspose")),                  # outside the restated subset (group 1)
    ("notEnoughAttributes", _conv_model({k: v for k, v in FULL.items() if k != "strides"}, op="ConvTranspose")),
]


@pytest.mark.parametrize("kind,model", ERRORS)
def test_error_cases_match_the_reference_and_the_oracle(kind, model):
    with pytest.raises(SmelterError) as e:
        ONNXGraph(model).packWeights(1)
    assert e.value.kind == kind, e.value
    with pytest.raises(O.OracleError) as e2:
        O.Graph(model).run(np.zeros((1, 8, 6, 6), np.float32))
    assert e2.value.kind == kind, e2.value


def test_conv_transpose_shape_rule_and_weight_layout():
    # out = (in-1)*s - pt - pb + k + output_padding (ONNXConvolutionPadding.swift:97-103); weights [I,O,kH,kW] -> O, flipH, flipW, I
    b = ow.GraphBuilder("g")
    b.add_input("input", [1, 8, 5, 7])
    w = np.arange(8 * 4 * 3 * 3, dtype=np.float32).reshape(8, 4, 3, 3) / 64
    b.add_init("w", w)
    b.add_node_named("ConvTranspose", ["input", "w"], ["y"], dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2], output_padding=[1, 0])
    b.add_output("y", [1, 4, 10, 13])
    m = b.to_model()
    x = np.random.default_rng(3).standard_normal((2, 8, 5, 7)).astype(np.float32)
    y = O.Graph(m).run(x)
    assert y.shape == (2, 4, (5 - 1) * 2 - 2 + 3 + 1, (7 - 1) * 2 - 2 + 3 + 0)
    g = ONNXGraph(m)
    # one stride-2 3x3 ConvTranspose = 4 per-phase convolutions with 2x2, 2x1, 1x2 and 1x1 taps (9 in total) of 4x8 fp16 weights,
    # each block padded to the arena's 256-byte alignment: 256 + 128 + 128 + 64 -> 4 x 256
    assert g.weightBytes(1) == 1024
    arena = g.packWeights(1).view(np.uint16)
    flipped = O.reformat_conv_weight(w, 4, 8, 3, 3, is_transpose=True).reshape(4, 3, 3, 8)
    # phase (py, px) = (0, 0): taps kh in {0, 2}, kw in {0, 2}, stored correlation-order (a' = A-1-a) from the flipped master copy
    sub = np.zeros((4, 2, 2, 8), np.float32)
    for a in range(2):
        for bb in range(2):
            kh, kw = 2 * (1 - a), 2 * (1 - bb)
            sub[:, a, bb, :] = flipped[:, 2 - kh, 2 - kw, :]
    assert np.array_equal(arena[:sub.size], sub.astype(np.float16).view(np.uint16).ravel())


def test_unknown_op_error_carries_the_op_type():
    with pytest.raises(SmelterError) as e:
        ONNXGraph(_conv_model(FULL, op="Gelu")).packWeights(1)
    assert e.value.detail == "Gelu"                    # Errors.unknownNodeOpType(opType:)


def test_unsupported_input_rank_and_dims_override():
    b = ow.GraphBuilder("g")
    b.add_input("input", [8, 6])
    b.add_output(b.add_node("Relu", ["input"]), [8, 6])
    with pytest.raises(SmelterError) as e:
        ONNXGraph(b.to_model()).packWeights(1)
    assert e.value.kind == "unsupportedInput"          # ONNXGraph.swift:213-214
    # Configuration.dims overrides a per-axis input dim (:200-202): 64x64 toy planned at 32x48
    g = ONNXGraph(zoo.toy(), Configuration(dims={2: 32, 3: 48}))
    assert g.weightBytes(1) == ONNXGraph(zoo.toy()).weightBytes(1)
    assert O.Graph(zoo.toy(), dims={2: 32, 3: 48}).run(np.zeros((1, 3, 32, 48), np.float32)).shape == (1, 16, 32, 48)


def _lanczos_matrix(n_in, n_out):
    """independent numpy restatement of the separable Lanczos-3 resampler (support widened when shrinking, normalised, edge clamp)"""
    scale = n_in / n_out
    fs = max(scale, 1.0)
    m = np.zeros((n_out, n_in))
    for o in range(n_out):
        c = (o + 0.5) * scale - 0.5
        b = int(np.floor(c)); r = int(np.ceil(3 * fs))
        for i in range(b - r + 1, b + r + 1):
            t = abs((i - c) / fs)
            wgt = 1.0 if t < 1e-6 else (0.0 if t >= 3 else 3 * np.sin(np.pi * t) * np.sin(np.pi * t / 3) / (np.pi * t) ** 2)
            m[o, min(max(i, 0), n_in - 1)] += wgt
        m[o] /= m[o].sum()
    return m


def test_force_input_scale_resamplers():
    # Configuration.inputConstraint .forceInputScale (ONNXGraph.swift:219-241): the graph builds, and the oracle's two resamplers
    # match independent restatements (bilinear is also pinned to a torch fixture in test_golden.py)
    ONNXGraph(zoo.toy(), Configuration(inputConstraint="bilinear")).packWeights(1)
    ONNXGraph(zoo.toy(), Configuration(inputConstraint="lanczos")).packWeights(1)
    x = np.random.default_rng(5).standard_normal((2, 3, 19, 13)).astype(np.float32)
    for size in ((8, 20), (38, 26), (19, 13), (7, 5)):
        my, mx = _lanczos_matrix(19, size[0]), _lanczos_matrix(13, size[1])
        ref = np.einsum("oy,ncyx,px->ncop", my, x.astype(np.float64), mx)
        assert np.abs(O.resize(x, size, "lanczos") - ref).max() <= 2e-5
    assert np.array_equal(O.resize(x, (19, 13), "bilinear"), x)
    # the oracle graph applies the scale node only when configured, like the reference
    big = np.random.default_rng(6).random((1, 3, 100, 90), dtype=np.float32)
    y = O.Graph(zoo.toy(), input_constraint="bilinear").run(big)
    assert np.array_equal(y, O.Graph(zoo.toy()).run(O.resize(big, (64, 64), "bilinear")))
    with pytest.raises(O.OracleError):
        O.Graph(zoo.toy()).run(big)


def _arena_weights(model_bytes, n_convs):
    """first conv's fp16 OHWI weights + fp32 bias from the packed arena (layout: 256-byte aligned blocks)"""
    return ONNXGraph(model_bytes).packWeights(1)


def test_bn_fold_fp16_cast_and_ohwi_match_numpy_restatement():
    # toy: conv(3->8) + BN.  Independent restatement of ONNX2MPS.py's pipeline in numpy:
    m = orr.parse_model(zoo.toy())
    arr = {t.name: t.to_array() for t in m.initializer}
    conv, bn = m.node[0], m.node[1]
    w = arr[conv.input[1]]
    gamma, beta, mean, var = (arr[i] for i in bn.input[1:5])
    s = gamma / np.sqrt(var + np.float32(1e-5))
    w_f = (w * s[:, None, None, None]).astype(np.float32)
    b_f = ((0 - mean) * s + beta).astype(np.float32)
    # first layer ("image stem"): OHWI regrouped as [Cout][kernel-row pair][8 pixels][row 2j | row 2j+1][4 channels],
    # zero where kh >= kH, kw >= kW or c >= Cin
    stem = np.zeros((8, 2, 8, 2, 4), np.float32)
    for kh in range(3):
        for kw in range(3):
            stem[:, kh // 2, kw, kh & 1, :3] = w_f[:, :, kh, kw]
    expect_w = stem.astype(np.float16).view(np.uint16).ravel()
    arena = ONNXGraph(zoo.toy()).packWeights(1)
    got_w = arena[:expect_w.size * 2].view(np.uint16)
    assert np.array_equal(got_w, expect_w)
    off = (expect_w.size * 2 + 255) // 256 * 256
    assert np.array_equal(arena[off:off + 32].view(np.float32), b_f)
    # the offline tool produces the same folded fp16 weights, and loading ITS output packs the identical arena
    arena2 = ONNXGraph(onnx2mps.optimize_model(zoo.toy(), half=True)).packWeights(1)
    assert np.array_equal(arena2[:expect_w.size * 2].view(np.uint16), expect_w)


def test_mps_flavor_round_trip_same_function_as_unfused():
    # ONNX2MPS-flavoured fp32 file (BN folded, weights pre-transposed) == original unfused graph, in the oracle
    x = zoo.synthetic_images(2, 3, (32, 32))
    model = zoo.resnet50(hw=(32, 32))
    a = O.Graph(model).run(x)
    b = O.Graph(onnx2mps.optimize_model(model, half=False)).run(x)
    assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()


def test_weight_bytes_resnet50():
    g = ONNXGraph(zoo.resnet50(hw=(32, 32)))
    nbytes = g.weightBytes(4)
    # 25.5 M fp16 weights + fp32 biases = 51.1 MB, + the fragment-order copies of the layers that sit in persistent runs (executor pass 3k; at 32 x 32 input every stage's
    # images are <= 64 pixels, so nearly every 1x1 / 3x3 stride-1 layer is in one): less than as much again
    assert 51.0e6 < nbytes < 2 * 51.5e6
    assert nbytes == g.weightBytes(64)                 # independent of batch


# ---- round 2: converters no BASELINE model instantiates + the declared supersets (LeakyRelu, MatMul) ------------------------
def _same_error_kind(model, kind):
    with pytest.raises(SmelterError) as e:
        ONNXGraph(model).packWeights(1)
    assert e.value.kind == kind, (e.value.kind, e.value.detail)
    with pytest.raises(O.OracleError) as e2:
        O.Graph(model).run(np.zeros((1,) + O.Graph(model).input_chw, np.float32))
    assert e2.value.kind == kind


def test_reshape_net_plans_and_error_kinds_match_the_oracle():
    for final in ("matmul", "reshape"):
        g = ONNXGraph(zoo.reshape_net(final=final))
        assert g.weightBytes(3) > 0                                          # the whole plan (aliases, materialised reshapes, FC permutation) builds
        ONNXGraph(onnx2mps.optimize_model(zoo.reshape_net(final=final), half=True)).packWeights(2)   # fp16-cast shape tensors still parse (.integers)
    def one(op, inits=(), n_in=1, **attrs):
        b = ow.GraphBuilder("g"); b.add_input("input", [1, 4, 2, 2])
        names = ["input"][:n_in] + [b.add_init(n, a) for n, a in inits]
        b.add_output(b.add_node(op, names, **attrs), [1, 4, 2, 2])
        return b.to_model()
    _same_error_kind(one("Dropout"), "noSuchOutput")                         # ratio attribute required (Converters.swift:920-925)
    _same_error_kind(one("Reshape"), "noSuchOutput")                         # shape tensor required (:832-836)
    _same_error_kind(one("MatMul", inits=[("w", np.zeros((16, 3), np.int64))]), "unsupportedDtype")
    _same_error_kind(one("MatMul", inits=[("w", np.zeros((15, 3), np.float32))]), "inconsistentState")
    _same_error_kind(one("Reshape", inits=[("s", np.array([5, 1, 1], np.int64))]), "inconsistentState")
    # a node without outputs, or a Conv whose int64 weight dims are hostile, is an error — never undefined behaviour
    b = ow.GraphBuilder("g"); b.add_input("input", [1, 4, 2, 2]); b.add_node_named("Dropout", ["input"], [], ratio=0.5); b.add_output("input", [1, 4, 2, 2])
    with pytest.raises(SmelterError) as e:
        ONNXGraph(b.to_model()).packWeights(1)
    assert e.value.kind == "noSuchOutput"
    for dims in ([-1, 4, 3, 3], [1 << 40, 4, 3, 3], [8, 4, 0, 3]):
        b = ow.GraphBuilder("g"); b.add_input("input", [1, 4, 8, 8])
        b.inits.append(ow.tensor("w", np.zeros(4, np.float32), dims=dims))
        b.add_output(b.add_node("Conv", ["input", "w"], dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1]), [1, 8, 8, 8])
        with pytest.raises(SmelterError) as e:
            ONNXGraph(b.to_model()).packWeights(1)
        assert e.value.kind == "inconsistentState"


def test_matmul_weight_is_packed_transposed_to_out_in():
    w = np.arange(6 * 4, dtype=np.float32).reshape(6, 4)                     # ONNX MatMul: [K, N]
    b = ow.GraphBuilder("g"); b.add_input("input", [1, 6, 1, 1])
    b.add_output(b.add_node("MatMul", ["input", b.add_init("w", w)]), [1, 4])
    arena = ONNXGraph(b.to_model()).packWeights(1)
    assert np.array_equal(arena[:48].view(np.float16).astype(np.float32).reshape(4, 6), w.T)   # MPSCNNFullyConnected weights are [out, in]


def test_fused_pool_and_head_predicates_reject_grouped_convolutions():
    """sm_conv2d_pool_supported / sm_conv2d_head_supported are host-only predicates the planner (executor passes 3a / 3d) relies on: a depthwise
    3x3 with 32 channels has the weight-stationary kernel's Cin / Cout but must not be taken (it runs on the depthwise kernel, which has no second output)"""
    L = N.lib()
    L.sm_conv2d_head_supported.argtypes = [C.c_void_p, C.c_int32]
    L.sm_conv2d_pool_supported.argtypes = [C.c_void_p]
    for groups, want in ((1, 1), (32, 0)):
        d = N.ConvDesc(2, 64, 64, 32, 32, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, groups, 64, 64, 1, 0.0, 32, 32, 0)
        assert L.sm_conv2d_head_supported(C.byref(d), 16) == want
        assert L.sm_conv2d_pool_supported(C.byref(d)) == want


def test_docs_are_stamped_from_the_profile_files():
    """every number block of DESIGN.md / README.md is generated from profiles/r05_* by tools/doc_numbers.py; a block edited by hand, or a profile replaced
    without re-stamping, fails here"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "doc_numbers.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for doc in ("DESIGN.md", "README.md"):
        text = open(os.path.join(root, doc)).read()
        assert "<!-- numbers:state -->" in text and "<!-- numbers:parity -->" in text and "<!-- numbers:stage -->" in text
