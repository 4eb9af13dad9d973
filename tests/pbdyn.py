"""google.protobuf dynamic descriptors for the ONNX subset — an INDEPENDENT encoder/decoder used only by the
tests to pin our three hand-written wire implementations (python writer, C++ product decoder, C oracle decoder).
Field numbers transcribed from /root/reference/Sources/Smelter/onnx.pb.swift name maps
(:1079-1093, :1260-1263, :1337-1344, :1402-1411, :1597-1605, :1668-1682, :1897-1898, :1926-1929, :1975-1977)."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_T = {"int64": _F.TYPE_INT64, "int32": _F.TYPE_INT32, "string": _F.TYPE_STRING, "bytes": _F.TYPE_BYTES,
      "float": _F.TYPE_FLOAT, "double": _F.TYPE_DOUBLE, "uint64": _F.TYPE_UINT64}


def _msg(fd, name, fields):
    m = fd.message_type.add()
    m.name = name
    for fname, num, typ, rep in fields:
        f = m.field.add()
        f.name, f.number = fname, num
        f.label = _F.LABEL_REPEATED if rep else _F.LABEL_OPTIONAL
        if typ in _T:
            f.type = _T[typ]
        else:
            f.type = _F.TYPE_MESSAGE
            f.type_name = ".onnxdyn." + typ
    return m


def build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "onnxdyn.proto", "onnxdyn", "proto3"
    _msg(fd, "TensorProto", [("dims", 1, "int64", 1), ("data_type", 2, "int32", 0), ("float_data", 4, "float", 1),
                             ("int32_data", 5, "int32", 1), ("int64_data", 7, "int64", 1), ("name", 8, "string", 0),
                             ("raw_data", 9, "bytes", 0), ("double_data", 10, "double", 1), ("uint64_data", 11, "uint64", 1)])
    _msg(fd, "AttributeProto", [("name", 1, "string", 0), ("f", 2, "float", 0), ("i", 3, "int64", 0), ("s", 4, "bytes", 0),
                                ("t", 5, "TensorProto", 0), ("floats", 7, "float", 1), ("ints", 8, "int64", 1), ("type", 20, "int32", 0)])
    _msg(fd, "NodeProto", [("input", 1, "string", 1), ("output", 2, "string", 1), ("name", 3, "string", 0), ("op_type", 4, "string", 0),
                           ("attribute", 5, "AttributeProto", 1), ("doc_string", 6, "string", 0), ("domain", 7, "string", 0)])
    _msg(fd, "Dim", [("dim_value", 1, "int64", 0), ("dim_param", 2, "string", 0)])
    _msg(fd, "TensorShapeProto", [("dim", 1, "Dim", 1)])
    _msg(fd, "TypeTensor", [("elem_type", 1, "int32", 0), ("shape", 2, "TensorShapeProto", 0)])
    _msg(fd, "TypeProto", [("tensor_type", 1, "TypeTensor", 0)])
    _msg(fd, "ValueInfoProto", [("name", 1, "string", 0), ("type", 2, "TypeProto", 0), ("doc_string", 3, "string", 0)])
    _msg(fd, "GraphProto", [("node", 1, "NodeProto", 1), ("name", 2, "string", 0), ("initializer", 5, "TensorProto", 1),
                            ("doc_string", 10, "string", 0), ("input", 11, "ValueInfoProto", 1), ("output", 12, "ValueInfoProto", 1),
                            ("value_info", 13, "ValueInfoProto", 1)])
    _msg(fd, "OperatorSetIdProto", [("domain", 1, "string", 0), ("version", 2, "int64", 0)])
    _msg(fd, "ModelProto", [("ir_version", 1, "int64", 0), ("producer_name", 2, "string", 0), ("producer_version", 3, "string", 0),
                            ("domain", 4, "string", 0), ("model_version", 5, "int64", 0), ("doc_string", 6, "string", 0),
                            ("graph", 7, "GraphProto", 0), ("opset_import", 8, "OperatorSetIdProto", 1)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:  # older protobuf
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName("onnxdyn." + n)) for n in
            ("TensorProto", "AttributeProto", "NodeProto", "ValueInfoProto", "GraphProto", "ModelProto")}
