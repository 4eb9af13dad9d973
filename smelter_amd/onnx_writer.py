"""Minimal ONNX (proto3 wire format) writer — no `onnx` package needed.

Smelter consumes ONNX IR<=5 files (field numbers: /root/reference/Sources/Smelter/onnx.pb.swift
:1079-1093 Attribute, :1337-1344 Node, :1402-1411 Model, :1597-1605 Graph, :1668-1682 Tensor).
This module emits exactly that subset so that models can be synthesised offline (no network, no
`onnx` wheel in the image).  It is the *encoder*; the product decoder is C++
(smelter_amd/csrc/host/onnx_pb.cpp) and the oracle decoder is C (oracle/orc_proto.c) — three
independent implementations of one wire contract, cross-checked in tests/test_proto.py against
google.protobuf dynamic descriptors.
"""
from __future__ import annotations

import struct
from typing import Iterable, Sequence

import numpy as np

# TensorProto.DataType (onnx.pb.swift:704-749)
FLOAT, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, FLOAT16, DOUBLE, UINT32, UINT64 = range(1, 14)
# AttributeProto.AttributeType (onnx.pb.swift:233-243)
AT_FLOAT, AT_INT, AT_STRING, AT_TENSOR, AT_GRAPH, AT_FLOATS, AT_INTS, AT_STRINGS = range(1, 9)

_NP2ONNX = {
    np.dtype(np.float32): FLOAT,
    np.dtype(np.float16): FLOAT16,
    np.dtype(np.float64): DOUBLE,
    np.dtype(np.int64): INT64,
    np.dtype(np.int32): INT32,
    np.dtype(np.uint8): UINT8,
    np.dtype(np.int8): INT8,
}


def varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64  # proto int64 two's complement, 10 bytes
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return varint((field << 3) | wire)


def f_varint(field: int, v: int) -> bytes:
    return _key(field, 0) + varint(int(v))


def f_bytes(field: int, b: bytes | str) -> bytes:
    if isinstance(b, str):
        b = b.encode("utf-8")
    return _key(field, 2) + varint(len(b)) + bytes(b)


def f_float(field: int, v: float) -> bytes:
    return _key(field, 5) + struct.pack("<f", v)


def f_packed_varints(field: int, vals: Iterable[int]) -> bytes:
    body = b"".join(varint(int(v)) for v in vals)
    return f_bytes(field, body)


def f_packed_floats(field: int, vals: Iterable[float]) -> bytes:
    vals = list(vals)
    return f_bytes(field, struct.pack("<%df" % len(vals), *vals))


def f_unpacked_varints(field: int, vals: Iterable[int]) -> bytes:
    """Legacy (proto2-style) unpacked repeated scalars; decoders must accept both."""
    return b"".join(f_varint(field, v) for v in vals)


# ----------------------------------------------------------------------------------------------
# messages
# ----------------------------------------------------------------------------------------------

def tensor(name: str, array: np.ndarray, *, raw: bool = True, dims: Sequence[int] | None = None,
           data_type: int | None = None) -> bytes:
    """TensorProto.  raw=True -> raw_data (field 9, little endian); raw=False -> typed repeated
    field (float_data 4 / int32_data 5 / int64_data 7 / double_data 10)."""
    array = np.asarray(array)
    dt = data_type if data_type is not None else _NP2ONNX[array.dtype]
    dims = list(array.shape) if dims is None else list(dims)
    out = f_packed_varints(1, dims) if dims else b""
    out += f_varint(2, dt)
    if raw:
        out += f_bytes(9, np.ascontiguousarray(array).astype(array.dtype.newbyteorder("<")).tobytes())
    else:
        flat = array.reshape(-1)
        if dt == FLOAT:
            out += f_bytes(4, flat.astype("<f4").tobytes())
        elif dt == DOUBLE:
            out += f_bytes(10, flat.astype("<f8").tobytes())
        elif dt == INT64:
            out += f_packed_varints(7, flat.tolist())
        elif dt in (INT32, INT16, INT8, UINT16, UINT8, BOOL):
            out += f_packed_varints(5, flat.tolist())
        elif dt in (UINT32, UINT64):
            out += f_packed_varints(11, flat.tolist())
        else:
            raise ValueError("typed storage unsupported for data_type %d" % dt)
    out += f_bytes(8, name)
    return out


def attr_i(name: str, v: int) -> bytes:
    return f_bytes(1, name) + f_varint(3, v) + f_varint(20, AT_INT)


def attr_f(name: str, v: float) -> bytes:
    return f_bytes(1, name) + f_float(2, v) + f_varint(20, AT_FLOAT)


def attr_s(name: str, v: str | bytes) -> bytes:
    return f_bytes(1, name) + f_bytes(4, v) + f_varint(20, AT_STRING)


def attr_ints(name: str, vals: Sequence[int], packed: bool = True) -> bytes:
    body = f_packed_varints(8, vals) if packed else f_unpacked_varints(8, vals)
    return f_bytes(1, name) + body + f_varint(20, AT_INTS)


def attr_floats(name: str, vals: Sequence[float]) -> bytes:
    return f_bytes(1, name) + f_packed_floats(7, vals) + f_varint(20, AT_FLOATS)


def attr_t(name: str, tensor_bytes: bytes) -> bytes:
    return f_bytes(1, name) + f_bytes(5, tensor_bytes) + f_varint(20, AT_TENSOR)


def _attr(name: str, v) -> bytes:
    if isinstance(v, bool):
        return attr_i(name, int(v))
    if isinstance(v, (int, np.integer)):
        return attr_i(name, int(v))
    if isinstance(v, (float, np.floating)):
        return attr_f(name, float(v))
    if isinstance(v, (str, bytes)):
        return attr_s(name, v)
    if isinstance(v, np.ndarray):
        return attr_t(name, tensor("", v))
    if isinstance(v, (list, tuple)):
        if len(v) and isinstance(v[0], (float, np.floating)):
            return attr_floats(name, v)
        return attr_ints(name, v)
    raise TypeError("attribute %s: unsupported value %r" % (name, type(v)))


def node(op_type: str, inputs: Sequence[str], outputs: Sequence[str], name: str = "", **attrs) -> bytes:
    out = b"".join(f_bytes(1, i) for i in inputs)
    out += b"".join(f_bytes(2, o) for o in outputs)
    if name:
        out += f_bytes(3, name)
    out += f_bytes(4, op_type)
    for k, v in attrs.items():
        out += f_bytes(5, _attr(k, v))
    return out


def value_info(name: str, elem_type: int, dims: Sequence[int]) -> bytes:
    dim_msgs = b"".join(f_bytes(1, f_varint(1, d)) for d in dims)        # TensorShapeProto.dim{dim_value}
    ttype = f_varint(1, elem_type) + f_bytes(2, dim_msgs)                # TypeProto.Tensor{elem_type, shape}
    return f_bytes(1, name) + f_bytes(2, f_bytes(1, ttype))             # ValueInfoProto{name, type{tensor_type}}


def graph(nodes: Sequence[bytes], name: str, inputs: Sequence[bytes], outputs: Sequence[bytes],
          initializers: Sequence[bytes], value_infos: Sequence[bytes] = ()) -> bytes:
    out = b"".join(f_bytes(1, n) for n in nodes)
    out += f_bytes(2, name)
    out += b"".join(f_bytes(5, t) for t in initializers)
    out += b"".join(f_bytes(11, i) for i in inputs)
    out += b"".join(f_bytes(12, o) for o in outputs)
    out += b"".join(f_bytes(13, v) for v in value_infos)
    return out


def model(graph_bytes: bytes, *, producer_name: str = "smelter_amd.zoo", producer_version: str = "1",
          ir_version: int = 5, opset: int = 9) -> bytes:
    out = f_varint(1, ir_version)
    out += f_bytes(2, producer_name)
    out += f_bytes(3, producer_version)
    out += f_bytes(7, graph_bytes)
    out += f_bytes(8, f_bytes(1, "") + f_varint(2, opset))
    return out


class GraphBuilder:
    """Tiny convenience layer: collects nodes/initializers, hands out unique value names."""

    def __init__(self, name: str = "g"):
        self.name = name
        self.nodes: list[bytes] = []
        self.inits: list[bytes] = []
        self.inputs: list[bytes] = []
        self.outputs: list[bytes] = []
        self.init_arrays: dict[str, np.ndarray] = {}
        self._n = 0
        self.list_initializers_as_inputs = False  # old exporters list weights in graph.input too

    def fresh(self, hint: str = "v") -> str:
        self._n += 1
        return "%s_%d" % (hint, self._n)

    def add_input(self, name: str, dims: Sequence[int], elem_type: int = FLOAT) -> str:
        self.inputs.append(value_info(name, elem_type, dims))
        return name

    def add_output(self, name: str, dims: Sequence[int], elem_type: int = FLOAT) -> str:
        self.outputs.append(value_info(name, elem_type, dims))
        return name

    def add_init(self, name: str, array: np.ndarray, raw: bool = True) -> str:
        self.inits.append(tensor(name, array, raw=raw))
        self.init_arrays[name] = np.asarray(array)
        if self.list_initializers_as_inputs:
            self.inputs.append(value_info(name, _NP2ONNX[np.asarray(array).dtype], list(np.asarray(array).shape)))
        return name

    def add_node(self, op_type: str, inputs: Sequence[str], n_out: int = 1, name: str = "", **attrs):
        outs = [self.fresh(op_type.lower()) for _ in range(n_out)]
        self.nodes.append(node(op_type, inputs, outs, name=name or outs[0], **attrs))
        return outs[0] if n_out == 1 else outs

    def add_node_named(self, op_type: str, inputs: Sequence[str], outputs: Sequence[str], **attrs):
        self.nodes.append(node(op_type, inputs, outputs, name=outputs[0] if outputs else "", **attrs))

    def to_model(self, **kw) -> bytes:
        return model(graph(self.nodes, self.name, self.inputs, self.outputs, self.inits), **kw)
