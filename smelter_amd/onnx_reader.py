"""Pure-Python reader for the ONNX subset (field numbers: /root/reference/Sources/Smelter/onnx.pb.swift).
Used by the offline tool smelter_amd/onnx2mps.py and by tests; the product decoder is C++ (csrc/host/onnx_pb.cpp)."""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import onnx_writer as ow


def _varint(b: bytes, i: int):
    v = shift = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << shift
        if not c & 0x80:
            return v, i
        shift += 7


def _fields(b: bytes):
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v, i = b[i:i + ln], i + ln
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError("bad wire type %d" % wt)
        yield f, wt, v


def _s64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _rep_varint(wt, v) -> List[int]:
    if wt == 0:
        return [_s64(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_s64(x))
    return out


@dataclass
class Tensor:
    name: str = ""
    dims: List[int] = field(default_factory=list)
    data_type: int = 0
    raw_data: bytes = b""
    float_data: List[float] = field(default_factory=list)
    int32_data: List[int] = field(default_factory=list)
    int64_data: List[int] = field(default_factory=list)
    double_data: List[float] = field(default_factory=list)

    def to_array(self) -> np.ndarray:
        dt = {ow.FLOAT: np.float32, ow.FLOAT16: np.float16, ow.DOUBLE: np.float64, ow.INT64: np.int64, ow.INT32: np.int32}[self.data_type]
        if self.raw_data:
            a = np.frombuffer(self.raw_data, dtype=np.dtype(dt).newbyteorder("<"))
        elif self.data_type == ow.FLOAT:
            a = np.asarray(self.float_data, np.float32)
        elif self.data_type == ow.DOUBLE:
            a = np.asarray(self.double_data, np.float64)
        elif self.data_type == ow.INT64:
            a = np.asarray(self.int64_data, np.int64)
        else:
            a = np.asarray(self.int32_data, dt)
        return a.reshape(self.dims) if self.dims else a


@dataclass
class Attribute:
    name: str = ""
    type: int = 0
    f: float = 0.0
    i: int = 0
    s: bytes = b""
    t: Tensor = None
    floats: List[float] = field(default_factory=list)
    ints: List[int] = field(default_factory=list)
    raw: bytes = b""


@dataclass
class Node:
    op_type: str = ""
    name: str = ""
    input: List[str] = field(default_factory=list)
    output: List[str] = field(default_factory=list)
    attribute: List[Attribute] = field(default_factory=list)
    raw: bytes = b""

    def attr(self, name):
        for a in self.attribute:
            if a.name == name:
                return a
        return None


@dataclass
class ValueInfo:
    name: str = ""
    elem_type: int = 0
    dims: List[int] = field(default_factory=list)


@dataclass
class Model:
    ir_version: int = 0
    producer_name: str = ""
    producer_version: str = ""
    opset: int = 0
    graph_name: str = ""
    node: List[Node] = field(default_factory=list)
    initializer: List[Tensor] = field(default_factory=list)
    input: List[ValueInfo] = field(default_factory=list)
    output: List[ValueInfo] = field(default_factory=list)


def parse_tensor(b: bytes) -> Tensor:
    t = Tensor()
    for f, wt, v in _fields(b):
        if f == 1:
            t.dims += _rep_varint(wt, v)
        elif f == 2:
            t.data_type = v
        elif f == 4:
            t.float_data += list(struct.unpack("<%df" % (len(v) // 4), v))
        elif f == 5:
            t.int32_data += _rep_varint(wt, v)
        elif f == 7:
            t.int64_data += _rep_varint(wt, v)
        elif f == 8:
            t.name = v.decode()
        elif f == 9:
            t.raw_data = bytes(v)
        elif f == 10:
            t.double_data += list(struct.unpack("<%dd" % (len(v) // 8), v))
    return t


def parse_attr(b: bytes) -> Attribute:
    a = Attribute(raw=bytes(b))
    for f, wt, v in _fields(b):
        if f == 1:
            a.name = v.decode()
        elif f == 2:
            a.f = struct.unpack("<f", v)[0]
        elif f == 3:
            a.i = _s64(v)
        elif f == 4:
            a.s = bytes(v)
        elif f == 5:
            a.t = parse_tensor(v)
        elif f == 7:
            a.floats += list(struct.unpack("<%df" % (len(v) // 4), v))
        elif f == 8:
            a.ints += _rep_varint(wt, v)
        elif f == 20:
            a.type = v
    return a


def parse_node(b: bytes) -> Node:
    n = Node(raw=bytes(b))
    for f, wt, v in _fields(b):
        if f == 1:
            n.input.append(v.decode())
        elif f == 2:
            n.output.append(v.decode())
        elif f == 3:
            n.name = v.decode()
        elif f == 4:
            n.op_type = v.decode()
        elif f == 5:
            n.attribute.append(parse_attr(v))
    return n


def parse_value_info(b: bytes) -> ValueInfo:
    vi = ValueInfo()
    for f, wt, v in _fields(b):
        if f == 1:
            vi.name = v.decode()
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 != 1:
                    continue
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        vi.elem_type = v3
                    elif f3 == 2:
                        for f4, _, v4 in _fields(v3):
                            if f4 == 1:
                                dv = 0
                                for f5, _, v5 in _fields(v4):
                                    if f5 == 1:
                                        dv = _s64(v5)
                                vi.dims.append(dv)
    return vi


def parse_model(b: bytes) -> Model:
    m = Model()
    for f, wt, v in _fields(b):
        if f == 1:
            m.ir_version = v
        elif f == 2:
            m.producer_name = v.decode()
        elif f == 3:
            m.producer_version = v.decode()
        elif f == 8:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    m.opset = v2
        elif f == 7:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    m.node.append(parse_node(v2))
                elif f2 == 2:
                    m.graph_name = v2.decode()
                elif f2 == 5:
                    m.initializer.append(parse_tensor(v2))
                elif f2 == 11:
                    m.input.append(parse_value_info(v2))
                elif f2 == 12:
                    m.output.append(parse_value_info(v2))
    return m
