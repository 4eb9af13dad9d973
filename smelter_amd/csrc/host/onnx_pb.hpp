// onnx_pb.hpp — the ONNX proto3 messages Smelter reads, as plain C++ structs, plus the typed tensor views.
//
// Stands in for the generated /root/reference/Sources/Smelter/onnx.pb.swift (field numbers = the wire
// contract; decoders at :1146-1159 Attribute, :1350-1356 Node, :1454-1462 Model, :1611-1618 Graph,
// :1735-1748 Tensor, :1294-1296 ValueInfo, :1870-2014 Type/Shape/Dim, :2082-2083 OperatorSetId) and for
// Onnx_TensorProto+Extensions.swift (.floats / .integers / .length).
// String and bytes fields are views into the model buffer the graph keeps alive (no per-tensor copies).
#pragma once
#include <stdint.h>

#include <string>
#include <string_view>
#include <vector>

namespace smelter {
namespace pb {

enum DataType : int32_t { UNDEFINED = 0, FLOAT = 1, UINT8 = 2, INT8 = 3, UINT16 = 4, INT16 = 5, INT32 = 6, INT64 = 7, STRING = 8,
                          BOOL = 9, FLOAT16 = 10, DOUBLE = 11, UINT32 = 12, UINT64 = 13 };

struct Tensor {
    std::vector<int64_t> dims;
    int32_t data_type = 0;
    std::string_view name, raw_data;
    std::vector<float> float_data;
    std::vector<int32_t> int32_data;
    std::vector<int64_t> int64_data;
    std::vector<double> double_data;
    std::vector<uint64_t> uint64_data;
    int64_t length() const { int64_t n = 1; for (auto d : dims) n *= d; return n; }  // Onnx_TensorProto+Extensions.swift:64-66
};

struct Attribute {
    std::string_view name, s;
    float f = 0.f; int64_t i = 0; int32_t type = 0;
    bool has_t = false; Tensor t;
    std::vector<float> floats; std::vector<int64_t> ints;
};

struct Node {
    std::vector<std::string_view> input, output;
    std::string_view name, op_type, domain;
    std::vector<Attribute> attribute;
    const Attribute* attr(std::string_view n) const { for (auto& a : attribute) if (a.name == n) return &a; return nullptr; }
};

struct ValueInfo { std::string_view name; int32_t elem_type = 0; std::vector<int64_t> dims; };

struct Graph {
    std::vector<Node> node; std::string_view name;
    std::vector<Tensor> initializer; std::vector<ValueInfo> input, output, value_info;
};

struct Model {
    int64_t ir_version = 0; std::string_view producer_name, producer_version; int64_t opset_version = 0;
    Graph graph;
};

// false on malformed input (truncated varint / length, bad wire type)
bool decode_model(const uint8_t* p, size_t n, Model* out);

// Typed views.  Return 0 or SM_ERR_UNSUPPORTED_DTYPE (the reference's fatalError).
int tensor_floats(const Tensor& t, std::vector<float>* out);      // :36-62
int tensor_integers(const Tensor& t, std::vector<int64_t>* out);  // :2-34

}  // namespace pb
}  // namespace smelter
