// graph.cpp — ONNXGraph: decode, flavour detection, tensor index, converter registry and the node loop
// (/root/reference/Sources/Smelter/ONNXGraph.swift:95-285).
#include "graph.hpp"

#include "../kernels/common.hpp"

namespace smelter {

ONNXGraph::ONNXGraph(const void* data, size_t n, const Configuration& cfg, int* err) : configuration(cfg) {
    bytes_.assign((const char*)data, n);                                 // caller may free its buffer right away
    if (!pb::decode_model((const uint8_t*)bytes_.data(), bytes_.size(), &model_)) {
        sm_set_error("protobuf decode failed (%zu bytes)", n);
        *err = SM_ERR_PROTO_DECODE; return;
    }
    modelFormat = model_.producer_name == "ONNX2MPS" ? Format::mpsFlavor : Format::onnx;   // :98-103
    for (auto& t : model_.graph.initializer) tensors_[t.name] = &t;                        // :106-108
    registerDefaultConverters(*this);                                                      // :110-155
    *err = SM_OK;
}

std::vector<Shape> ONNXGraph::outputShapes() const {                     // :69-91 (note W := dim2, H := dim3 for rank 4)
    std::vector<Shape> r;
    for (auto& o : model_.graph.output) {
        Shape s;
        if (o.dims.size() == 3) { s.channels = (int)o.dims[0]; s.width = (int)o.dims[1]; s.height = (int)o.dims[2]; s.depth = 1; }
        else if (o.dims.size() == 4) { s.channels = (int)o.dims[1]; s.width = (int)o.dims[2]; s.height = (int)o.dims[3]; s.depth = 1; }
        else continue;
        r.push_back(s);
    }
    return r;
}

ONNXGraph& ONNXGraph::registerConverter(const std::string& name, std::shared_ptr<NodeConverter> c) {
    converters_[name] = std::move(c);                                     // last registration wins (:254-257)
    return *this;
}
void ONNXGraph::initTensor(std::string_view name, const pb::Tensor* t) { tensors_[name] = t; }

int ONNXGraph::addFilter(Op&& op, int c, int h, int w, const Shape& outputShape, const std::vector<std::string_view>& outputs) {
    if (outputs.empty()) { sm_set_error("node without outputs"); return SM_ERR_NO_SUCH_OUTPUT; }
    if (c <= 0 || h <= 0 || w <= 0 || (long)c * h > (1L << 31) - 1 || (long)c * h * w > (1L << 31) - 1) {       // per-image element count stays an int everywhere downstream
        sm_set_error("value '%.*s' has shape C=%d H=%d W=%d (empty, or more than 2^31 elements per image)", (int)outputs[0].size(), outputs[0].data(), c, h, w);
        return SM_ERR_INCONSISTENT_STATE;
    }
    Value v; v.name = std::string(outputs[0]); v.c = c; v.h = h; v.w = w; v.ref = outputShape; v.producer = (int)ops_.size();
    int id = (int)values_.size();
    values_.push_back(v);
    op.output = id; if (op.name.empty()) op.name = v.name;
    ops_.push_back(std::move(op));
    for (auto& o : outputs) outputs_[std::string(o)] = id;              // every output name maps to the same image (:269-272)
    return SM_OK;
}
int ONNXGraph::output(std::string_view name) const { auto it = outputs_.find(std::string(name)); return it == outputs_.end() ? -1 : it->second; }
const pb::Tensor* ONNXGraph::tensor(std::string_view name) const { auto it = tensors_.find(name); return it == tensors_.end() ? nullptr : it->second; }
const Shape* ONNXGraph::shape(std::string_view name) const { int id = output(name); return id < 0 ? nullptr : &values_[id].ref; }

int ONNXGraph::initOutputs() {                                           // :197-251
    for (auto& vi : model_.graph.input) {
        if (tensor(vi.name)) continue;                                   // initializers listed as inputs (:199)
        std::vector<int> shape;
        for (size_t i = 0; i < vi.dims.size(); i++) {
            auto it = configuration.dims.find((int)i);
            const int64_t dv = vi.dims[i];                                // int64 in the file: saturate, the range check below rejects it
            shape.push_back(it != configuration.dims.end() ? it->second : (int)(dv > (1 << 24) ? (1 << 24) : dv < -1 ? -1 : dv));   // :200-202
        }
        int channels, height, width;
        if (shape.size() == 3) { channels = shape[0]; height = shape[1]; width = shape[2]; }
        else if (shape.size() == 4) { channels = shape[1]; height = shape[2]; width = shape[3]; }   // N dropped (:209-212)
        else { sm_set_error("graph input '%.*s' has rank %zu", (int)vi.name.size(), vi.name.data(), shape.size()); return SM_ERR_UNSUPPORTED_INPUT; }
        // Configuration.inputConstraint .forceInputScale(:219-241) puts a scale node in front of the graph; the model-side shape
        // recorded here is the same either way, the executor resamples the caller's image first (smelter_exec_encode_sized)
        if (input_value_ >= 0) { sm_set_error("more than one image input"); return SM_ERR_UNSUPPORTED_INPUT; }
        if (channels <= 0 || height <= 0 || width <= 0 || (long)channels * height * width > (1L << 31) - 1) { sm_set_error("graph input has a non-positive dimension"); return SM_ERR_UNSUPPORTED_INPUT; }
        Value v; v.name = std::string(vi.name); v.c = channels; v.h = height; v.w = width;
        v.ref = Shape{1, width, height, channels};                        // :243-248
        input_value_ = (int)values_.size();
        values_.push_back(v);
        outputs_[v.name] = input_value_;
    }
    if (input_value_ < 0) { sm_set_error("graph has no image input"); return SM_ERR_UNSUPPORTED_INPUT; }
    return SM_OK;
}

int ONNXGraph::build() {
    values_.clear(); ops_.clear(); outputs_.clear(); input_value_ = output_value_ = -1;
    tensors_.clear();
    for (auto& t : model_.graph.initializer) tensors_[t.name] = &t;
    int rc = initOutputs(); if (rc) return rc;
    for (auto& node : model_.graph.node) {                                // file order == topological order (:172-176)
        auto it = converters_.find(std::string(node.op_type));
        if (it == converters_.end()) { sm_set_error("%.*s", (int)node.op_type.size(), node.op_type.data()); return SM_ERR_UNKNOWN_NODE_OP_TYPE; }
        rc = it->second->convert(*this, node);
        if (rc) return rc;
    }
    if (model_.graph.output.size() != 1) { sm_set_error("graph has %zu outputs, exactly one is supported", model_.graph.output.size()); return SM_ERR_UNSUPPORTED_OUTPUT; }  // :178-180
    output_value_ = output(model_.graph.output[0].name);
    if (output_value_ < 0) { sm_set_error("graph output '%.*s' is produced by no node", (int)model_.graph.output[0].name.size(), model_.graph.output[0].name.data()); return SM_ERR_NO_SUCH_OUTPUT; }  // :182-183
    return SM_OK;
}

}  // namespace smelter
