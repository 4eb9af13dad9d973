// converters.cpp — one NodeConverter class per ONNX op_type on the hot path, same class names and the same
// attribute handling, defaults, error cases and shape rules as /root/reference/Sources/Smelter/Converters.swift
// (line ranges cited per class).  A converter reads producers / shapes / tensors by value NAME through the
// graph tables and publishes exactly one Op record under every node.output name.
//
// Reference quirks — kept / dropped (full table in DESIGN.md):
//   kept:    conv out-size rule ignores dilation; kernel_shape+dilations+strides all required; Gemm attributes
//            ignored, weight read as [out,in]; PRelu slope = first fp32 of raw_data; pools use pads[0],pads[1]
//            and the Float-division size rule; Upsample scales truncated to Int; Flatten axis != 1 is an error
//            (fatalError -> SM_ERR_FLATTEN_AXIS); unsupported tensor dtype is an error (fatalError ->
//            SM_ERR_UNSUPPORTED_DTYPE); the reference Shape bookkeeping incl. Concat depth*=2.
//   dropped: BN epsilon attribute IS honoured (the reference never reads it); fp32 bias is also accepted from
//            float_data; Concat's real channel count is the sum of its operands.
#include <cmath>
#include <string.h>

#include "../kernels/common.hpp"
#include "graph.hpp"

namespace smelter {
namespace {

#define SM_FAIL(code, ...) do { sm_set_error(__VA_ARGS__); return (code); } while (0)
#define SV(s) (int)(s).size(), (s).data()

// Every integer / float the converters take out of the file is untrusted: saturate to +-2^20 (no legitimate kernel size, stride,
// pad, scale, group count or channel count is larger; NaN -> 0) so that a hostile value ends in an error code, never in
// implementation-defined narrowing or signed overflow.  Products of two such values go through mulSat.
constexpr int kIntCap = 1 << 20;
int I(int64_t v) { return v > kIntCap ? kIntCap : v < -kIntCap ? -kIntCap : (int)v; }
int I(double v) { return v != v ? 0 : v > kIntCap ? kIntCap : v < -kIntCap ? -kIntCap : (int)v; }
int I(float v) { return I((double)v); }
int mulSat(int a, int b) { const long r = (long)a * b; return r > INT32_MAX ? INT32_MAX : r < INT32_MIN ? INT32_MIN : (int)r; }

Shape mkShape(int channels, int width, int height, int depth) { Shape s; s.channels = channels; s.width = width; s.height = height; s.depth = depth; return s; }

// ---- ConvolutionConverter (Converters.swift:187-338) + ConvDataSource (:12-185) ----------------------------
struct ConvolutionConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        if (node.input.empty()) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%.*s: no inputs", SV(node.op_type));
        const int in = graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%.*s: input '%.*s' is not a graph value", SV(node.op_type), SV(node.input[0]));   // :189-191
        const pb::Tensor* weight = node.input.size() > 1 ? graph.tensor(node.input[1]) : nullptr;
        if (!weight) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "%.*s: weight tensor missing", SV(node.op_type));                             // :193-194
        const pb::Tensor* bias = node.input.size() > 2 ? graph.tensor(node.input[2]) : nullptr;                                        // :196-199
        const Value iv = graph.value(in);

        bool hasK = false, hasD = false, hasS = false;
        int kh = 0, kw = 0, dh = 1, dw = 1, sh = 1, sw = 1, groups = 1, pt = 0, pl = 0, pb_ = 0, pr = 0, oph = 0, opw = 0;
        for (auto& a : node.attribute) {                                                                                                // :208-225
            if (a.name == "dilations" && a.ints.size() >= 2) { dh = I(a.ints[0]); dw = I(a.ints[1]); hasD = true; }
            else if (a.name == "strides" && a.ints.size() >= 2) { sh = I(a.ints[0]); sw = I(a.ints[1]); hasS = true; }
            else if (a.name == "group") groups = I(a.i);
            else if (a.name == "pads" && a.ints.size() >= 4) { pt = I(a.ints[0]); pl = I(a.ints[1]); pb_ = I(a.ints[2]); pr = I(a.ints[3]); }
            else if (a.name == "kernel_shape" && a.ints.size() >= 2) { kh = I(a.ints[0]); kw = I(a.ints[1]); hasK = true; }
            else if (a.name == "output_padding" && a.ints.size() >= 2) { oph = I(a.ints[0]); opw = I(a.ints[1]); }
        }
        const bool isGemm = node.op_type == "Gemm", isTranspose = node.op_type == "ConvTranspose";
        if (isGemm) { kh = kw = dh = dw = sh = sw = 1; hasK = hasD = hasS = true; }                                                     // :228-232
        if (!hasK || !hasD || !hasS) SM_FAIL(SM_ERR_NOT_ENOUGH_ATTRIBUTES, "%.*s: kernel_shape, dilations and strides are all required", SV(node.op_type));  // :234-237
        if (isTranspose && groups != 1) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "ConvTranspose: group=%d is not supported (group=1 only)", groups);
        if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || groups <= 0 || kh <= 0 || kw <= 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Conv: non-positive attribute");

        // ConvDataSource.init: channel/kernel extents from the weight dims (:35-54)
        const bool mps = graph.modelFormat == Format::mpsFlavor;
        int OC, IC, wkh, wkw;
        if (isGemm) { if (weight->dims.size() < 2) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "Gemm: weight rank < 2"); OC = I(weight->dims[0]); IC = I(weight->dims[1]); wkh = wkw = 1; }
        else {
            if (weight->dims.size() < 4) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "Conv: weight rank < 4");
            if (mps) { OC = I(weight->dims[0]); wkh = I(weight->dims[1]); wkw = I(weight->dims[2]); IC = I(weight->dims[3]); }     // :40-44
            else if (isTranspose) { OC = I(weight->dims[1]); IC = I(weight->dims[0]); wkh = I(weight->dims[2]); wkw = I(weight->dims[3]); }  // :48-50 [I,O,kH,kW]
            else { OC = I(weight->dims[0]); IC = I(weight->dims[1]); wkh = I(weight->dims[2]); wkw = I(weight->dims[3]); }        // :46-54
        }
        if (OC <= 0 || IC <= 0 || wkh <= 0 || wkw <= 0 || (long)OC * IC > (1L << 30) / ((long)wkh * wkw))                                 // dims are int64 in the file: untrusted
            SM_FAIL(SM_ERR_INCONSISTENT_STATE, "%.*s: weight dims out of range", SV(node.op_type));
        if (weight->data_type != pb::FLOAT && weight->data_type != pb::FLOAT16)                                                          // :82-89 (.invalid)
            SM_FAIL(SM_ERR_UNSUPPORTED_DTYPE, "%.*s: weight data_type %d (FLOAT or FLOAT16 expected)", SV(node.op_type), weight->data_type);
        if (!isGemm && (wkh != kh || wkw != kw)) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Conv: kernel_shape %dx%d disagrees with weight dims %dx%d", kh, kw, wkh, wkw);

        Op op; op.kind = isGemm ? OpKind::FullyConnected : isTranspose ? OpKind::ConvTranspose : OpKind::Conv;
        op.inputs = {in}; op.cout = OC; op.kh = kh; op.kw = kw; op.sh = sh; op.sw = sw; op.dh = dh; op.dw = dw;
        op.pt = pt; op.pl = pl; op.pb = pb_; op.pr = pr; op.groups = groups;
        op.weight_was_fp16 = weight->data_type == pb::FLOAT16;
        std::vector<float> w;
        int rc = pb::tensor_floats(*weight, &w); if (rc) SM_FAIL(rc, "Conv: weight accessor");
        const size_t expect = (size_t)OC * IC * wkh * wkw;
        if (w.size() != expect) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "%.*s: weight holds %zu elements, dims say %zu", SV(node.op_type), w.size(), expect);
        if (!mps && !isGemm) {                                            // re-layout unless ONNX2MPS already did it (:91-124)
            op.weight.resize(expect);
            sm_reformat_conv_weight(w.data(), op.weight.data(), 4, OC, IC, wkh, wkw, isTranspose ? 1 : 0);   // transpose: + flip (:71-76)
        } else op.weight = std::move(w);
        if (bias) {                                                       // :126-135 (+ typed float_data accepted)
            rc = pb::tensor_floats(*bias, &op.bias); if (rc) SM_FAIL(rc, "Conv: bias accessor");
            if ((int)op.bias.size() != OC) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "%.*s: bias holds %zu elements for %d channels", SV(node.op_type), op.bias.size(), OC);
        }

        int Ho = 1, Wo = 1;
        if (isGemm) {
            const long feat = (long)iv.c * iv.h * iv.w;
            if (feat != IC) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Gemm: input has %ld features, weight expects %d", feat, IC);
            op.cin = IC;
        } else {
            const bool depthwise = groups != 1 && groups == OC;                                                                         // :57
            const int cinTotal = depthwise ? OC : IC * groups;                                                                          // :59-74
            if (cinTotal != iv.c) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Conv: input has %d channels, weight expects %d", iv.c, cinTotal);
            if (OC % groups) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Conv: %d output channels not divisible by group %d", OC, groups);
            op.cin = cinTotal;
            sm_conv_padded_size(iv.w, iv.h, kh, kw, sh, sw, pt, pl, pb_, pr, oph, opw, isTranspose ? 1 : 0, &Wo, &Ho);                  // :311-322
            if (Ho <= 0 || Wo <= 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Conv: empty output %dx%d", Ho, Wo);
            // Dilated taps, the reference's way (quirk KEPT): the padding policy centres the kernel at offset = k/2 - pad
            // (ONNXConvolutionPadding.swift:73-79) and MPSCNNConvolution places tap kh at centre + (kh - k/2)*d, i.e. input row
            // o*s - pad + kh*d - (k/2)(d-1): the ONNX tap rule with pad' = pad + (k/2)(d-1), on an output grid whose size
            // ignores the dilation (:105-110).  d = 1 (every zoo model): nothing changes.
            if (!isTranspose) { op.pt += (kh / 2) * (dh - 1); op.pl += (kw / 2) * (dw - 1); }
        }
        return graph.addFilter(std::move(op), OC, Ho, Wo, mkShape(1, Wo, Ho, OC), node.output);                                         // :311-336
    }
};

// ---- ReluConverter (:342-359), PReluConverter (:361-384) ----------------------------------------------------
struct ReluConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Relu: input is not a graph value");
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Neuron; op.inputs = {in}; op.act = SM_ACT_RELU; op.slope = 0.f;
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};
struct PReluConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.size() >= 2 ? graph.output(node.input[0]) : -1;
        const pb::Tensor* a = node.input.size() >= 2 ? graph.tensor(node.input[1]) : nullptr;
        if (in < 0 || !a) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "PRelu: input value or slope tensor missing");
        float alpha = 0.f;
        if (a->raw_data.size() >= 4) memcpy(&alpha, a->raw_data.data(), 4);   // first fp32 of raw_data (:372-374)
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Neuron; op.inputs = {in}; op.act = SM_ACT_PRELU; op.slope = alpha;
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};

// ---- LeakyReluConverter — NOT in the reference registry (ONNXGraph.swift:110-155 has no "LeakyRelu": the reference throws
// unknownNodeOpType).  Declared superset named by BASELINE.json's north_star: ONNX LeakyRelu, y = x >= 0 ? x : alpha*x with the
// `alpha` attribute (ONNX default 0.01), lowered to the node PReluConverter builds (MPSCNNNeuronPReLUNode, :376) with a = alpha.
struct LeakyReluConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "LeakyRelu: input is not a graph value");
        const pb::Attribute* al = node.attr("alpha");
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Neuron; op.inputs = {in}; op.act = SM_ACT_PRELU; op.slope = al ? al->f : 0.01f;
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};

// ---- ClipConverter — NOT in the reference registry (a torchvision MobileNet-v2 export stops Smelter with unknownNodeOpType("Clip"), ONNXGraph.swift:173-174).
// Declared superset: ONNX Clip with lower bound 0, i.e. the ReLU-N family (ReLU6 = Clip(0, 6)); bounds from the `min` / `max` attributes (opset 6) or from
// constant inputs 1 / 2 (opset 11: initializer or Constant node; an empty name = unbounded).  max = +inf lowers to the node ReluConverter builds; a non-zero
// lower bound is refused (no single MPSCNNNeuron node expresses it either).
struct ClipConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Clip: input is not a graph value");
        float lo = -INFINITY, hi = INFINITY;
        if (const pb::Attribute* a = node.attr("min")) lo = a->f;
        if (const pb::Attribute* a = node.attr("max")) hi = a->f;
        for (int k = 1; k <= 2 && k < (int)node.input.size(); k++) {
            if (node.input[k].empty()) continue;
            const pb::Tensor* t = graph.tensor(node.input[k]);
            if (!t) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "Clip: bound '%.*s' must be a constant tensor", (int)node.input[k].size(), node.input[k].data());
            std::vector<float> v; int rc = pb::tensor_floats(*t, &v); if (rc) return rc;
            if (v.size() != 1) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Clip: bound '%.*s' has %zu elements (a scalar expected)", (int)node.input[k].size(), node.input[k].data(), v.size());
            (k == 1 ? lo : hi) = v[0];
        }
        if (lo != 0.f) SM_FAIL(SM_ERR_UNKNOWN_NODE_OP_TYPE, "Clip with lower bound %g (only 0: the ReLU-N family is implemented)", (double)lo);
        if (!(hi > 0.f)) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Clip: upper bound %g is not above the lower bound 0", (double)hi);
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Neuron; op.inputs = {in};
        if (std::isinf(hi)) { op.act = SM_ACT_RELU; op.slope = 0.f; } else { op.act = SM_ACT_CLIP; op.slope = hi; }
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};

// ---- MatMulConverter — NOT in the reference registry either (declared superset, north_star "MatMul/Gemm").  2-D case only:
// Y[batch, N] = A[batch, K] x B[K, N] with B a constant tensor (initializer or Constant node) and A a graph value of K features.
// Lowered to the node the Gemm branch of ConvolutionConverter builds (MPSCNNFullyConnectedNode, :288-302): weight [out,in] = B^T,
// no bias.  ONNX stores B as [K, N] (in x out) where Gemm(transB=1) stores [out, in].
struct MatMulConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.size() >= 2 ? graph.output(node.input[0]) : -1;
        const pb::Tensor* b = node.input.size() >= 2 ? graph.tensor(node.input[1]) : nullptr;
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "MatMul: left operand is not a graph value");
        if (!b) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "MatMul: right operand must be a constant tensor");
        if (b->dims.size() != 2) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "MatMul: right operand has rank %zu (2 expected)", b->dims.size());
        if (b->data_type != pb::FLOAT && b->data_type != pb::FLOAT16) SM_FAIL(SM_ERR_UNSUPPORTED_DTYPE, "MatMul: weight data_type %d", b->data_type);
        const long K = b->dims[0], N = b->dims[1];
        if (K <= 0 || N <= 0 || K * N > (1L << 30)) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "MatMul: weight dims out of range");
        const Value iv = graph.value(in);
        if ((long)iv.c * iv.h * iv.w != K) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "MatMul: input has %ld features, weight expects %ld", (long)iv.c * iv.h * iv.w, K);
        std::vector<float> w; int rc = pb::tensor_floats(*b, &w); if (rc) SM_FAIL(rc, "MatMul: weight accessor");
        if ((long)w.size() != K * N) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "MatMul: weight holds %zu elements, dims say %ld", w.size(), K * N);
        Op op; op.kind = OpKind::FullyConnected; op.inputs = {in}; op.cout = (int)N; op.cin = (int)K;
        op.weight_was_fp16 = b->data_type == pb::FLOAT16;
        op.weight.resize(w.size());
        for (long k = 0; k < K; k++) for (long n = 0; n < N; n++) op.weight[(size_t)n * K + k] = w[(size_t)k * N + n];
        return graph.addFilter(std::move(op), (int)N, 1, 1, mkShape(1, 1, 1, (int)N), node.output);
    }
};

// ---- AddConverter (:430-451): both operands must be graph values, same shape -------------------------------
struct AddConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int a = node.input.size() >= 2 ? graph.output(node.input[0]) : -1;
        const int b = node.input.size() >= 2 ? graph.output(node.input[1]) : -1;
        if (a < 0 || b < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Add: an operand is not a graph value");
        const Value va = graph.value(a), vb = graph.value(b);
        if (va.c != vb.c || va.h != vb.h || va.w != vb.w) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Add: operand shapes differ (the reference has no broadcasting)");
        Op op; op.kind = OpKind::Add; op.inputs = {a, b};
        return graph.addFilter(std::move(op), va.c, va.h, va.w, va.ref, node.output);
    }
};

// ---- UpsampleConverter (:478-552) ---------------------------------------------------------------------------
struct UpsampleConverter : NodeConverter {
    bool alignCorners;
    explicit UpsampleConverter(bool ac) : alignCorners(ac) {}
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Upsample: input is not a graph value");
        const pb::Attribute* mode = node.attr("mode"); const pb::Attribute* sc = node.attr("scales");
        bool have = false; int sy = 1, sx = 1;
        if (sc && sc->floats.size() >= 4) { sy = I(sc->floats[2]); sx = I(sc->floats[3]); have = true; }                           // :499
        if (!have) {                                                                                                                    // :505-514
            const pb::Tensor* st = node.input.size() >= 2 ? graph.tensor(node.input[1]) : nullptr;
            if (!st || st->dims.empty() || st->dims[0] != 4) SM_FAIL(SM_ERR_NOT_ENOUGH_ATTRIBUTES, "Upsample: scales missing");
            std::vector<int64_t> v; int rc = pb::tensor_integers(*st, &v); if (rc) SM_FAIL(rc, "Upsample: scales dtype");
            if (v.size() < 4) SM_FAIL(SM_ERR_NOT_ENOUGH_ATTRIBUTES, "Upsample: scales tensor too short");
            sy = I(v[2]); sx = I(v[3]);
        }
        if (!mode) SM_FAIL(SM_ERR_NOT_ENOUGH_ATTRIBUTES, "Upsample: mode missing");                                                     // :517-519
        bool bilinear;
        if (mode->s == "nearest") bilinear = false; else if (mode->s == "bilinear" || mode->s == "linear") bilinear = true;
        else SM_FAIL(SM_ERR_UNKNOWN_NODE_OP_TYPE, "%.*s", SV(node.op_type));                                                            // :539-540
        if (sy <= 0 || sx <= 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Upsample: scale %dx%d", sy, sx);
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Upsample; op.inputs = {in}; op.sy = sy; op.sx = sx; op.bilinear = bilinear; op.alignCorners = alignCorners;
        return graph.addFilter(std::move(op), iv.c, mulSat(iv.h, sy), mulSat(iv.w, sx), mkShape(iv.ref.channels, mulSat(sx, iv.ref.width), mulSat(sy, iv.ref.height), iv.ref.depth), node.output);
    }
};

// ---- ConcatConverter (:554-574): feature-channel concatenation, `axis` ignored ------------------------------
struct ConcatConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        if (node.input.empty()) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Concat: no inputs");
        Op op; op.kind = OpKind::Concat; int c = 0;
        for (auto& name : node.input) { int id = graph.output(name); if (id < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Concat: operand '%.*s' is not a graph value", SV(name)); op.inputs.push_back(id); }
        const Value v0 = graph.value(op.inputs[0]);
        for (int id : op.inputs) { const Value& v = graph.value(id); if (v.h != v0.h || v.w != v0.w) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Concat: operands differ in H/W"); c += v.c; }
        Shape ref = v0.ref; ref.depth *= 2;                                                                                             // :566-567
        return graph.addFilter(std::move(op), c, v0.h, v0.w, ref, node.output);
    }
};

// ---- GlobalAveragePoolConverter (:578-605) -----------------------------------------------------------------
struct GlobalAveragePoolConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "GlobalAveragePool: input is not a graph value");
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::GlobalAvgPool; op.inputs = {in};
        return graph.addFilter(std::move(op), iv.c, 1, 1, mkShape(iv.ref.channels, 1, 1, iv.ref.depth), node.output);
    }
};

// ---- AveragePoolConverter (:607-650), MaxPoolConverter (:652-695) + PyTorchPoolPadding ----------------------
struct PoolConverter : NodeConverter {
    bool isMax;
    explicit PoolConverter(bool m) : isMax(m) {}
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        const pb::Attribute *ks = node.attr("kernel_shape"), *pads = node.attr("pads"), *st = node.attr("strides");
        if (in < 0 || node.attribute.size() < 3 || !ks || !pads || !st || ks->ints.size() < 2 || pads->ints.size() < 2 || st->ints.size() < 2)
            SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%.*s: input, kernel_shape, pads and strides are all required", SV(node.op_type));           // :654-661
        const Value iv = graph.value(in);
        Op op; op.kind = isMax ? OpKind::MaxPool : OpKind::AvgPool; op.inputs = {in};
        op.kh = I(ks->ints[0]); op.kw = I(ks->ints[1]); op.sh = I(st->ints[0]); op.sw = I(st->ints[1]);
        op.pt = I(pads->ints[0]); op.pl = I(pads->ints[1]);                                                                         // :673-674 (symmetric)
        if (op.kh <= 0 || op.kw <= 0 || op.sh <= 0 || op.sw <= 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Pool: non-positive attribute");
        int Ho, Wo; sm_pool_padded_size(iv.w, iv.h, op.kh, op.kw, op.sh, op.sw, op.pt, op.pl, &Wo, &Ho);
        if (Ho <= 0 || Wo <= 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Pool: empty output");
        op.y0 = sm_pool_window_start(iv.h, Ho, op.kh, op.sh); op.x0 = sm_pool_window_start(iv.w, Wo, op.kw, op.sw);
        return graph.addFilter(std::move(op), iv.c, Ho, Wo, mkShape(iv.ref.channels, Wo, Ho, iv.ref.depth), node.output);
    }
};

// ---- ConstantConverter (:716-727) --------------------------------------------------------------------------
struct ConstantConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const pb::Attribute* v = node.attr("value");
        if (node.output.empty() || !v) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Constant: output or value attribute missing");
        graph.initTensor(node.output[0], &v->t);
        return SM_OK;
    }
};

// ---- BatchNormalizationConverter (:797-827) + BNDataSource (:730-794) --------------------------------------
struct BatchNormalizationConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "BatchNormalization: input is not a graph value");
        if (node.input.size() < 5) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "BatchNormalization: 5 inputs required");
        const Value iv = graph.value(in);
        std::vector<float> p[4];                                         // gamma, beta, mean, variance (:805-808)
        for (int i = 0; i < 4; i++) {
            const pb::Tensor* t = graph.tensor(node.input[1 + i]);
            if (!t) SM_FAIL(SM_ERR_INSUFFICIENT_INPUTS, "BatchNormalization: tensor '%.*s' missing", SV(node.input[1 + i]));
            int rc = pb::tensor_floats(*t, &p[i]); if (rc) SM_FAIL(rc, "BatchNormalization: parameter dtype");
            if ((int)p[i].size() != iv.c) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "BatchNormalization: parameter %d holds %zu values for %d channels", i + 1, p[i].size(), iv.c);
        }
        const pb::Attribute* e = node.attr("epsilon"); const float eps = e ? e->f : 1e-5f;
        Op op; op.kind = OpKind::BatchNorm; op.inputs = {in}; op.scale.resize(iv.c); op.shift.resize(iv.c);
        for (int c = 0; c < iv.c; c++) { float s = p[0][c] / sqrtf(p[3][c] + eps); op.scale[c] = s; op.shift[c] = p[1][c] - p[2][c] * s; }
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};

// ---- ReshapeConverter (:830-876), FlattenConverter (:879-915), DropoutConverter (:918-939) ------------------
// All three are data-order-preserving in our layout only when H == W == 1 (or when the sole consumers are
// fully-connected layers, whose weights the executor permutes); the executor checks that and aliases.
struct ReshapeConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.size() >= 2 ? graph.output(node.input[0]) : -1;
        const pb::Tensor* st = node.input.size() >= 2 ? graph.tensor(node.input[1]) : nullptr;
        if (in < 0 || !st) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Reshape: input value or shape tensor missing");
        const Value iv = graph.value(in);
        std::vector<int64_t> shape; int rc = pb::tensor_integers(*st, &shape); if (rc) SM_FAIL(rc, "Reshape: shape dtype");
        const int total = iv.ref.channels * iv.ref.depth * iv.ref.height * iv.ref.width;                                               // :840-843
        const std::vector<int> src = iv.ref.toArray(); int nd[3];
        for (int i = 0; i < 3; i++) { int d = i < (int)shape.size() ? I(shape[i]) : 1; nd[i] = d == 0 ? src[i] : d; }                 // :844-849
        int inferred = -1; for (int i = 0; i < 3; i++) if (nd[i] == -1) { inferred = i; break; }
        if (inferred >= 0) { int v = total; for (int i = 0; i < 3; i++) if (i != inferred) { if (nd[i] == 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Reshape: zero dim"); v /= nd[i]; } nd[inferred] = v; }   // :851-857
        if ((long)nd[0] * nd[1] * nd[2] != (long)iv.c * iv.h * iv.w) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Reshape: element count changes");
        Op op; op.kind = OpKind::Alias; op.inputs = {in};
        return graph.addFilter(std::move(op), nd[2], nd[1], nd[0], mkShape(iv.ref.channels, nd[0], nd[1], nd[2]), node.output);       // (W,H,C) = nd (:859-872)
    }
};
struct FlattenConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        const pb::Attribute* ax = node.attr("axis");
        if (in < 0 || node.attribute.empty() || !ax) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Flatten: input value or axis attribute missing");  // :881-886
        if (ax->i != 1) SM_FAIL(SM_ERR_FLATTEN_AXIS, "Flatten: axis %lld (only 1 is supported)", (long long)ax->i);                     // :893-896
        const Value iv = graph.value(in);
        const int total = iv.ref.channels * iv.ref.width * iv.ref.height * iv.ref.depth;                                               // :888-889
        Op op; op.kind = OpKind::Alias; op.inputs = {in};
        return graph.addFilter(std::move(op), iv.c * iv.h * iv.w, 1, 1, mkShape(total, 1, 1, 1), node.output);                        // :906-911
    }
};
struct DropoutConverter : NodeConverter {   // inference identity (the reference builds an MPSCNNDropoutNode, :926-929)
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0 || node.attribute.empty() || !node.attr("ratio")) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Dropout: input value or ratio attribute missing");
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Alias; op.inputs = {in};
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);   // (addFilter rejects an empty output list)
    }
};

// ---- SubConverter (:1177-1193), MulConverter (:453-464), DivConverter (:1195-1211): MPSNN{Subtraction,Multiplication,Division}Node
struct BinaryConverter : NodeConverter {
    int binop; const char* name;
    BinaryConverter(int op, const char* n) : binop(op), name(n) {}
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int a = node.input.size() >= 2 ? graph.output(node.input[0]) : -1;
        const int b = node.input.size() >= 2 ? graph.output(node.input[1]) : -1;
        if (a < 0 || b < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%s: an operand is not a graph value", name);
        const Value va = graph.value(a), vb = graph.value(b);
        if (va.c != vb.c || va.h != vb.h || va.w != vb.w) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "%s: operand shapes differ (no broadcasting)", name);
        Op op; op.kind = OpKind::Add; op.binop = binop; op.inputs = {a, b};
        return graph.addFilter(std::move(op), va.c, va.h, va.w, va.ref, node.output);
    }
};
struct SubConverter : BinaryConverter { SubConverter() : BinaryConverter(SM_BIN_SUB, "Sub") {} };
struct MulConverter : BinaryConverter { MulConverter() : BinaryConverter(SM_BIN_MUL, "Mul") {} };
struct DivConverter : BinaryConverter { DivConverter() : BinaryConverter(SM_BIN_DIV, "Div") {} };

// ---- the MPSCNNNeuron*Node converters, all with the node's DEFAULT parameters: Elu (:386-409, alpha attribute required),
// Exp (:411-428), Sigmoid (:466-476), Abs (:1056), HardSigmoid (:1073, attributes not read), Softplus (:1090), Softsign (:1107),
// Tanh (:1124), Log (:1142), Pow (:1160: MPSCNNNeuronPowerNode(source:) has exponent 1 -> identity, aliased like Dropout) ----
struct NeuronConverter : NodeConverter {
    int act; const char* name;
    NeuronConverter(int a, const char* n) : act(a), name(n) {}
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        if (in < 0) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%s: input is not a graph value", name);
        float a = 0.f;
        if (act == SM_ACT_ELU) { const pb::Attribute* al = node.attr("alpha"); if (node.attribute.empty() || !al) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Elu: alpha attribute required"); a = al->f; }
        const Value iv = graph.value(in);
        Op op; op.inputs = {in};
        if (act == SM_ACT_NONE) op.kind = OpKind::Alias;
        else { op.kind = OpKind::Neuron; op.act = act; op.slope = a; }
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};
struct EluConverter : NeuronConverter { EluConverter() : NeuronConverter(SM_ACT_ELU, "Elu") {} };
struct ExpConverter : NeuronConverter { ExpConverter() : NeuronConverter(SM_ACT_EXP, "Exp") {} };
struct SigmoidConverter : NeuronConverter { SigmoidConverter() : NeuronConverter(SM_ACT_SIGMOID, "Sigmoid") {} };
struct AbsConverter : NeuronConverter { AbsConverter() : NeuronConverter(SM_ACT_ABS, "Abs") {} };
struct HardSigmoidConverter : NeuronConverter { HardSigmoidConverter() : NeuronConverter(SM_ACT_HARD_SIGMOID, "HardSigmoid") {} };
struct SoftplusConverter : NeuronConverter { SoftplusConverter() : NeuronConverter(SM_ACT_SOFTPLUS, "Softplus") {} };
struct SoftsignConverter : NeuronConverter { SoftsignConverter() : NeuronConverter(SM_ACT_SOFTSIGN, "Softsign") {} };
struct TanhConverter : NeuronConverter { TanhConverter() : NeuronConverter(SM_ACT_TANH, "Tanh") {} };
struct LogConverter : NeuronConverter { LogConverter() : NeuronConverter(SM_ACT_LOG, "Log") {} };
struct PowConverter : NeuronConverter { PowConverter() : NeuronConverter(SM_ACT_NONE, "Pow") {} };

// ---- SoftmaxConverter (:697-714), LogSoftmaxConverter (:1213-1232): an `axis` attribute must be present and equal 1 -------
struct SoftmaxConverter : NodeConverter {
    bool logsm;
    explicit SoftmaxConverter(bool l = false) : logsm(l) {}
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        const pb::Attribute* ax = node.attr("axis");
        if (in < 0 || node.attribute.empty() || !ax || ax->i != 1) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%s: input missing or axis != 1", logsm ? "LogSoftmax" : "Softmax");
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Softmax; op.inputs = {in}; op.logSoftmax = logsm;
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};
struct LogSoftmaxConverter : SoftmaxConverter { LogSoftmaxConverter() : SoftmaxConverter(true) {} };

// ---- InstanceNormConverter (:992-1011: x, gamma, beta), GroupNormConverter (:1273-1300: x, groups, gamma, beta) ------------
// epsilon: the ONNX attribute is honoured (default 1e-5), as for BatchNormalization.
struct InstanceNormConverter : NodeConverter {
    bool isGroup;
    explicit InstanceNormConverter(bool g = false) : isGroup(g) {}
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const size_t need = isGroup ? 4 : 3;
        const char* nm = isGroup ? "custom_group_norm" : "InstanceNormalization";
        const int in = node.input.size() >= need ? graph.output(node.input[0]) : -1;
        const pb::Tensor* gt = node.input.size() >= need ? graph.tensor(node.input[need - 2]) : nullptr;
        const pb::Tensor* bt = node.input.size() >= need ? graph.tensor(node.input[need - 1]) : nullptr;
        const pb::Tensor* nt = isGroup && node.input.size() >= need ? graph.tensor(node.input[1]) : nullptr;
        if (in < 0 || !gt || !bt || (isGroup && !nt)) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "%s: input value or parameter tensor missing", nm);
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::GroupNorm; op.inputs = {in};
        int rc = pb::tensor_floats(*gt, &op.scale); if (rc) SM_FAIL(rc, "%s: gamma dtype", nm);
        rc = pb::tensor_floats(*bt, &op.shift); if (rc) SM_FAIL(rc, "%s: beta dtype", nm);
        op.normGroups = iv.c;
        if (isGroup) { std::vector<int64_t> gi; rc = pb::tensor_integers(*nt, &gi); if (rc) SM_FAIL(rc, "custom_group_norm: groups dtype");
            if (gi.empty()) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "custom_group_norm: groups tensor empty"); op.normGroups = I(gi[0]); }
        if ((int)op.scale.size() != iv.c || (int)op.shift.size() != iv.c || op.normGroups <= 0 || iv.c % op.normGroups)
            SM_FAIL(SM_ERR_INCONSISTENT_STATE, "%s: parameter sizes / groups do not fit %d channels", nm, iv.c);
        const pb::Attribute* e = node.attr("epsilon"); op.eps = e ? e->f : 1e-5f;
        return graph.addFilter(std::move(op), iv.c, iv.h, iv.w, iv.ref, node.output);
    }
};
struct GroupNormConverter : InstanceNormConverter { GroupNormConverter() : InstanceNormConverter(true) {} };

// ---- PaddingConverter (:944-990): pads = [n,c,h,w begins ; n,c,h,w ends], mode constant(0) / reflect / edge ------------------
struct PaddingConverter : NodeConverter {
    int convert(ONNXGraph& graph, const pb::Node& node) override {
        const int in = node.input.empty() ? -1 : graph.output(node.input[0]);
        const pb::Attribute* pd = node.attr("pads");
        if (in < 0 || !pd) SM_FAIL(SM_ERR_NO_SUCH_OUTPUT, "Pad: input value or pads attribute missing");
        if (pd->ints.size() < 8) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Pad: 8 pads expected");
        int mode = SM_PAD_CONSTANT; const pb::Attribute* ma = node.attr("mode");
        if (ma && !ma->s.empty()) {
            if (ma->s == "constant") mode = SM_PAD_CONSTANT; else if (ma->s == "reflect") mode = SM_PAD_REFLECT; else if (ma->s == "edge") mode = SM_PAD_EDGE;
            else SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Pad: unknown mode '%.*s'", SV(ma->s));                                           // :966-967
        }
        if (pd->ints[0] || pd->ints[1] || pd->ints[4] || pd->ints[5]) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Pad: batch / channel padding is not supported");
        const Value iv = graph.value(in);
        Op op; op.kind = OpKind::Pad; op.inputs = {in}; op.padMode = mode;
        op.pt = I(pd->ints[2]); op.pl = I(pd->ints[3]); op.pb = I(pd->ints[6]); op.pr = I(pd->ints[7]);
        if (op.pt < 0 || op.pl < 0 || op.pb < 0 || op.pr < 0) SM_FAIL(SM_ERR_INCONSISTENT_STATE, "Pad: negative pads");
        const int ho = iv.h + op.pt + op.pb, wo = iv.w + op.pl + op.pr;
        return graph.addFilter(std::move(op), iv.c, ho, wo, mkShape(iv.ref.channels, wo, ho, iv.ref.depth), node.output);              // :980-985
    }
};

}  // namespace

void registerDefaultConverters(ONNXGraph& g) {    // ONNXGraph.swift:110-155, hot-path subset (SURVEY.md §2a ★ rows)
    auto conv = std::make_shared<ConvolutionConverter>();
    g.registerConverter("Conv", conv)
        .registerConverter("Gemm", conv)
        .registerConverter("ConvTranspose", conv)
        .registerConverter("Relu", std::make_shared<ReluConverter>())
        .registerConverter("PRelu", std::make_shared<PReluConverter>())
        .registerConverter("Add", std::make_shared<AddConverter>())
        .registerConverter("Upsample", std::make_shared<UpsampleConverter>(g.configuration.alignCorners))
        .registerConverter("Concat", std::make_shared<ConcatConverter>())
        .registerConverter("AveragePool", std::make_shared<PoolConverter>(false))
        .registerConverter("MaxPool", std::make_shared<PoolConverter>(true))
        .registerConverter("Constant", std::make_shared<ConstantConverter>())
        .registerConverter("GlobalAveragePool", std::make_shared<GlobalAveragePoolConverter>())
        .registerConverter("BatchNormalization", std::make_shared<BatchNormalizationConverter>())
        .registerConverter("Dropout", std::make_shared<DropoutConverter>())
        .registerConverter("Reshape", std::make_shared<ReshapeConverter>())
        .registerConverter("Flatten", std::make_shared<FlattenConverter>())
        // the rest of the registry of ONNXGraph.swift:110-155
        .registerConverter("Sub", std::make_shared<SubConverter>())
        .registerConverter("Mul", std::make_shared<MulConverter>())
        .registerConverter("Div", std::make_shared<DivConverter>())
        .registerConverter("Elu", std::make_shared<EluConverter>())
        .registerConverter("Exp", std::make_shared<ExpConverter>())
        .registerConverter("Sigmoid", std::make_shared<SigmoidConverter>())
        .registerConverter("Abs", std::make_shared<AbsConverter>())
        .registerConverter("HardSigmoid", std::make_shared<HardSigmoidConverter>())
        .registerConverter("Softplus", std::make_shared<SoftplusConverter>())
        .registerConverter("Softsign", std::make_shared<SoftsignConverter>())
        .registerConverter("Tanh", std::make_shared<TanhConverter>())
        .registerConverter("Log", std::make_shared<LogConverter>())
        .registerConverter("Pow", std::make_shared<PowConverter>())
        .registerConverter("Softmax", std::make_shared<SoftmaxConverter>())
        .registerConverter("LogSoftmax", std::make_shared<LogSoftmaxConverter>())
        .registerConverter("InstanceNormalization", std::make_shared<InstanceNormConverter>())
        .registerConverter("custom_group_norm", std::make_shared<GroupNormConverter>())
        .registerConverter("Pad", std::make_shared<PaddingConverter>())
        // declared supersets (not in the reference registry; BASELINE.json north_star names them)
        .registerConverter("LeakyRelu", std::make_shared<LeakyReluConverter>())
        .registerConverter("MatMul", std::make_shared<MatMulConverter>())
        .registerConverter("Clip", std::make_shared<ClipConverter>());
}

}  // namespace smelter
