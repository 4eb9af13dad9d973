// graph.hpp — ONNXGraph, Configuration, Shape, NodeConverter: the load-and-build surface of
// /root/reference/Sources/Smelter/ONNXGraph.swift, NodeConverter.swift and TypeDefinitions.swift, re-stated in
// C++.  Where the reference's converters create an MPSNNFilterNode, ours append one `Op` record; the executor
// (executor.cpp) turns the records into gfx950 kernel launches through the sm_* C ABI.
#pragma once
#include "../../../include/smelter_hip.h"
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "onnx_pb.hpp"

namespace smelter {

// TypeDefinitions.swift:1-33
struct Shape {
    int channels = 0, width = 0, height = 0, depth = 0;
    std::vector<int> toArray() const { return {channels, width, height, depth}; }
};

// ONNXGraph.swift:6-36
struct Configuration {
    enum class InputConstraint { none, forceInputScaleLanczos, forceInputScaleBilinear };
    InputConstraint inputConstraint = InputConstraint::none;
    bool alignCorners = true;            // billinearUpsamplingConfiguration.alignCorners (default true, :20)
    std::map<int, int> dims;             // per-axis input-dim override (:25, :200-202)
    // executor options (no reference counterpart)
    bool fuse = true, useHipGraph = true, autotune = false, keepIntermediates = false, branchConcurrency = false;
    int fusions = SM_FUSE_DEFAULT;       // (smelter_hip.h): which fusion passes may run
    int autotuneMode = 0, autotuneMaxCfg = 0, mbconvDomain[3] = {0, 0, 0};
    bool allows(int bit) const { return (fusions & bit) != 0; }
};

enum class Format { onnx, mpsFlavor };   // ONNXGraph.swift:49-52

enum class OpKind { Conv, FullyConnected, BatchNorm, Neuron, Add, MaxPool, AvgPool, GlobalAvgPool, Upsample, Concat, Alias, ConvTranspose, Softmax, GroupNorm, Pad, Reshape };

// One record per converted node (the analogue of an MPSNNFilterNode + its data source / padding policy).
struct Op {
    OpKind kind;
    std::string name;                    // first ONNX output name
    std::vector<int> inputs;             // value ids
    int output = -1;                     // value id
    // convolution / fully connected (ConvDataSource, Converters.swift:12-185)
    int cout = 0, cin = 0, kh = 1, kw = 1, sh = 1, sw = 1, dh = 1, dw = 1, pt = 0, pl = 0, pb = 0, pr = 0, groups = 1;
    std::vector<float> weight;           // OHWI fp32 master copy (cast to fp16 when the arena is packed)
    std::vector<float> bias;             // empty = none
    bool weight_was_fp16 = false;        // ConvDataSource.dataType() (:140-149)
    // output placement of a ConvTranspose phase (executor.cpp): sm_conv_desc.y_sH ... ; conv grid of the phase
    int y_sh = 0, y_sw = 0, y_oh = 0, y_ow = 0, y_h = 0, y_w = 0, grid_h = 0, grid_w = 0;
    bool x_ups = false;                  // executor: a nearest 2x Upsample in front of this conv is fused into its patch gather (sm_conv_desc.x_ups)
    // fused epilogue (set by the executor's fusion passes, or by Neuron ops)
    int act = 0; float slope = 0.f;      // SM_ACT_*
    int residual = -1;                   // value id added before the activation
    // batch norm (BNDataSource :730-794) folded to y = x*scale + shift
    std::vector<float> scale, shift;
    // pooling (PyTorchPoolPadding) : window start of output (0,0)
    int y0 = 0, x0 = 0;
    // upsample
    int sy = 1, sx = 1; bool bilinear = false, alignCorners = true;
    // Add family (SM_BIN_*), Softmax (log flag), GroupNorm / InstanceNorm (groups, eps; gamma/beta in scale/shift), Pad (SM_PAD_*, pads in pt..pr)
    int binop = 0; bool logSoftmax = false; int normGroups = 0; float eps = 1e-5f; int padMode = 0;
};

// A value = one named tensor flowing through the graph (MPSNNImageNode in the reference).
struct Value {
    std::string name;
    int c = 0, h = 0, w = 0;             // per-image dims actually computed
    Shape ref;                           // the reference's bookkeeping (nodeShapes, ONNXGraph.swift:64)
    int producer = -1;                   // op index, -1 = graph input
};

class ONNXGraph;

// NodeConverter.swift:3-5 — the operator-plugin protocol.  Returns SM_OK or an SM_ERR_* (Swift `throws`).
struct NodeConverter {
    virtual ~NodeConverter() = default;
    virtual int convert(ONNXGraph& graph, const pb::Node& node) = 0;
};

class ONNXGraph {
   public:
    Format modelFormat = Format::onnx;   // public var, ONNXGraph.swift:58
    Configuration configuration;         // public var, :59

    // init(data:configuration:) (:95-156).  `err` receives SM_OK or SM_ERR_PROTO_DECODE.
    ONNXGraph(const void* data, size_t n, const Configuration& cfg, int* err);

    std::vector<Shape> outputShapes() const;                                       // :69-91
    // The node loop of metalGraph(device:) (:169-183) up to, not including, MPSNNGraph creation.
    // Idempotent: the op list is rebuilt from scratch on every call.
    int build();

    // ---- what a converter may call (ONNXGraph.swift:253-285) ----
    ONNXGraph& registerConverter(const std::string& name, std::shared_ptr<NodeConverter> c);   // register(name:converter:) :254
    void initTensor(std::string_view name, const pb::Tensor* t);                               // :259
    int addFilter(Op&& op, int c, int h, int w, const Shape& outputShape, const std::vector<std::string_view>& outputs);  // :263-273
    int output(std::string_view name) const;                                                   // :275 (value id or -1)
    const pb::Tensor* tensor(std::string_view name) const;                                     // :279
    const Shape* shape(std::string_view name) const;                                           // :283
    const Value& value(int id) const { return values_[id]; }

    // ---- results of build() ----
    const std::vector<Op>& ops() const { return ops_; }
    const std::vector<Value>& values() const { return values_; }
    int inputValue() const { return input_value_; }
    int outputValue() const { return output_value_; }
    size_t numNodes() const { return model_.graph.node.size(); }
    const pb::Model& model() const { return model_; }

   private:
    int initOutputs();                                                                          // :197-251
    std::string bytes_;                   // private copy of the file; every string_view points in here
    pb::Model model_;
    std::map<std::string, std::shared_ptr<NodeConverter>> converters_;
    std::map<std::string_view, const pb::Tensor*> tensors_;
    std::map<std::string, int> outputs_;  // name -> value id (outputs + nodeShapes tables)
    std::vector<Value> values_;
    std::vector<Op> ops_;
    int input_value_ = -1, output_value_ = -1;
};

void registerDefaultConverters(ONNXGraph& g);   // the register chain of ONNXGraph.swift:110-155 (hot-path subset)

}  // namespace smelter
