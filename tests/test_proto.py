"""Wire-format tests: our python writer, the C++ product decoder and the C oracle decoder against an independent
implementation (google.protobuf dynamic descriptors), cf. SURVEY.md §8c(4)."""
import struct

import numpy as np
import pytest

import oracle as O
from smelter_amd import onnx_reader as orr
from smelter_amd import onnx_writer as ow
from smelter_amd import zoo
from smelter_amd.graph import ONNXGraph, SmelterError

import pbdyn

PB = pbdyn.build()


def test_writer_output_parses_with_google_protobuf():
    data = zoo.toy()
    m = PB["ModelProto"]()
    m.ParseFromString(data)
    assert m.ir_version == 5 and m.opset_import[0].version == 9
    assert [n.op_type for n in m.graph.node] == ["Conv", "BatchNormalization", "Relu"] * 3
    conv = m.graph.node[0]
    attrs = {a.name: a for a in conv.attribute}
    assert list(attrs["kernel_shape"].ints) == [3, 3] and list(attrs["pads"].ints) == [1, 1, 1, 1]
    assert attrs["group"].i == 1
    w = [t for t in m.graph.initializer if t.name == conv.input[1]][0]
    assert list(w.dims) == [8, 3, 3, 3] and w.data_type == 1 and len(w.raw_data) == 8 * 27 * 4
    assert [d.dim_value for d in m.graph.input[0].type.tensor_type.shape.dim] == [1, 3, 64, 64]
    # and our own reader agrees with google's view
    mine = orr.parse_model(data)
    assert [n.op_type for n in mine.node] == [n.op_type for n in m.graph.node]
    assert mine.input[0].dims == [1, 3, 64, 64]
    assert np.array_equal(mine.initializer[0].to_array().ravel(), np.frombuffer(m.graph.initializer[0].raw_data, "<f4"))


def _toy_via_google(packed_ints=True, typed_floats=False, unknown_fields=False):
    """re-encode the toy model with google.protobuf, optionally changing the encoding without changing meaning"""
    m = PB["ModelProto"]()
    m.ParseFromString(zoo.toy())
    if typed_floats:
        for t in m.graph.initializer:
            vals = np.frombuffer(t.raw_data, "<f4")
            t.ClearField("raw_data")
            t.float_data.extend(vals.tolist())
    data = m.SerializeToString()
    if unknown_fields:  # ModelProto.doc_string(6), model_version(5) and a made-up field 99 must be skipped
        data = ow.f_bytes(6, "doc") + ow.f_varint(5, 7) + ow.f_bytes(99, b"\x01\x02") + data + ow.f_varint(98, 1)
    return data


@pytest.mark.parametrize("kw", [dict(), dict(typed_floats=True), dict(unknown_fields=True)])
def test_decoders_are_encoding_independent(kw):
    ref_arena = ONNXGraph(zoo.toy()).packWeights(1)
    data = _toy_via_google(**kw)
    assert np.array_equal(ONNXGraph(data).packWeights(1), ref_arena)          # C++ product decoder
    x = zoo.synthetic_images(1, 3, (64, 64))
    assert np.array_equal(O.Graph(data).run(x), O.Graph(zoo.toy()).run(x))     # C oracle decoder


def test_unpacked_repeated_scalars_are_accepted():
    # proto2-style unpacked ints for `pads`/`strides`... (SwiftProtobuf accepts both; onnx.pb.swift:1146-1159)
    w = np.random.default_rng(0).standard_normal((4, 3, 3, 3)).astype(np.float32)

    def model(packed):
        attrs = b"".join(ow.f_bytes(5, a) for a in (
            ow.attr_ints("dilations", [1, 1], packed), ow.attr_i("group", 1), ow.attr_ints("kernel_shape", [3, 3], packed),
            ow.attr_ints("pads", [1, 1, 1, 1], packed), ow.attr_ints("strides", [2, 2], packed)))
        node = ow.f_bytes(1, "input") + ow.f_bytes(1, "w") + ow.f_bytes(2, "y") + ow.f_bytes(4, "Conv") + attrs
        dims = ow.f_packed_varints(1, w.shape) if packed else ow.f_unpacked_varints(1, w.shape)
        t = dims + ow.f_varint(2, ow.FLOAT) + ow.f_bytes(9, w.tobytes()) + ow.f_bytes(8, "w")
        g = ow.graph([node], "g", [ow.value_info("input", ow.FLOAT, [1, 3, 8, 8])], [ow.value_info("y", ow.FLOAT, [1, 4, 4, 4])], [t])
        return ow.model(g)

    x = zoo.synthetic_images(1, 3, (8, 8))
    a, b = O.Graph(model(True)).run(x), O.Graph(model(False)).run(x)
    assert a.shape == (1, 4, 4, 4) and np.array_equal(a, b)
    assert np.array_equal(ONNXGraph(model(True)).packWeights(1), ONNXGraph(model(False)).packWeights(1))


@pytest.mark.parametrize("cut", [1, 7, 100, 5000])
def test_truncated_model_is_a_decode_error_not_a_crash(cut):
    data = zoo.toy()[:-cut]
    with pytest.raises(SmelterError) as e:
        ONNXGraph(data).packWeights(1)
    assert e.value.kind in ("protoDecode", "inconsistentState", "noSuchOutput", "insufficientInputs", "unsupportedOutput", "unsupportedInput")
    with pytest.raises(O.OracleError):
        O.Graph(data).run(zoo.synthetic_images(1, 3, (64, 64)))


def test_garbage_is_a_decode_error():
    junk = bytes([0xFF] * 64)
    with pytest.raises(SmelterError) as e:
        ONNXGraph(junk)
    assert e.value.kind == "protoDecode"
    with pytest.raises(O.OracleError) as e2:
        O.Graph(junk)
    assert e2.value.kind == "protoDecode"


def test_negative_int64_attribute_roundtrip():
    # Flatten axis = -1 travels as a 10-byte varint; must arrive as -1 (and be rejected: axis must be 1)
    b = ow.GraphBuilder("g")
    b.add_input("input", [1, 4, 2, 2])
    y = b.add_node("Flatten", ["input"], axis=-1)
    b.add_output(y, [1, 16])
    data = b.to_model()
    m = PB["ModelProto"]()
    m.ParseFromString(data)
    assert m.graph.node[0].attribute[0].i == -1
    with pytest.raises(SmelterError) as e:
        ONNXGraph(data).packWeights(1)
    assert e.value.kind == "flattenAxis" and "-1" in e.value.detail
    with pytest.raises(O.OracleError) as e2:
        O.Graph(data).run(np.zeros((1, 4, 2, 2), np.float32))
    assert e2.value.kind == "flattenAxis"


def test_float_attribute_bits():
    raw = ow.attr_f("epsilon", 1e-5)
    a = PB["AttributeProto"]()
    a.ParseFromString(raw)
    assert struct.pack("<f", a.f) == struct.pack("<f", 1e-5) and a.type == 1
