// capi.cpp — graph-level C ABI (include/smelter_hip.h §2): the ONNXGraph load-and-encode surface
// (/root/reference/Sources/Smelter/ONNXGraph.swift:95,158,169; README.md:43-44;
// Extensions/Foundation/MPSImage+Extensions.swift:9) over opaque handles.
#include <memory>

#include "../kernels/common.hpp"
#include "executor.hpp"
#include "graph.hpp"

using namespace smelter;

struct smelter_graph { std::unique_ptr<ONNXGraph> g; };
struct smelter_exec { std::unique_ptr<Executor> e; };

extern "C" {

void smelter_config_default(smelter_config* c) {
    if (!c) return;
    c->input_constraint = SM_INPUT_NONE; c->align_corners = 1;
    for (int i = 0; i < 4; i++) c->dims[i] = -1;
    c->fuse = 1; c->use_hip_graph = 1; c->autotune = 0; c->keep_intermediates = 0; c->branch_concurrency = 0;
}

int smelter_graph_create(const void* onnx, size_t n, const smelter_config* cfg, smelter_graph** out) {
    SM_REQUIRE(onnx && out, "smelter_graph_create: null argument");
    smelter_config dc; if (!cfg) { smelter_config_default(&dc); cfg = &dc; }
    Configuration c;
    c.inputConstraint = cfg->input_constraint == SM_INPUT_FORCE_SCALE_LANCZOS ? Configuration::InputConstraint::forceInputScaleLanczos
                      : cfg->input_constraint == SM_INPUT_FORCE_SCALE_BILINEAR ? Configuration::InputConstraint::forceInputScaleBilinear
                      : Configuration::InputConstraint::none;
    c.alignCorners = cfg->align_corners != 0;
    for (int i = 0; i < 4; i++) if (cfg->dims[i] >= 0) c.dims[i] = cfg->dims[i];
    c.fuse = cfg->fuse != 0; c.useHipGraph = cfg->use_hip_graph != 0; c.autotune = cfg->autotune != 0; c.keepIntermediates = cfg->keep_intermediates != 0;
    c.branchConcurrency = cfg->branch_concurrency != 0;
    int err = SM_OK;
    std::unique_ptr<ONNXGraph> g(new ONNXGraph(onnx, n, c, &err));
    if (err) return err;
    *out = new smelter_graph{std::move(g)};
    return SM_OK;
}
void smelter_graph_destroy(smelter_graph* g) { delete g; }
int smelter_graph_format(const smelter_graph* g) { return g && g->g->modelFormat == Format::mpsFlavor ? SM_FORMAT_MPS_FLAVOR : SM_FORMAT_ONNX; }
int smelter_graph_set_format(smelter_graph* g, int32_t f) {
    SM_REQUIRE(g && (f == SM_FORMAT_ONNX || f == SM_FORMAT_MPS_FLAVOR), "smelter_graph_set_format: bad argument");
    g->g->modelFormat = f == SM_FORMAT_MPS_FLAVOR ? Format::mpsFlavor : Format::onnx; return SM_OK;
}
int smelter_graph_output_shapes(const smelter_graph* g, smelter_shape* dst, int32_t cap) {
    if (!g) return 0;
    auto v = g->g->outputShapes();
    for (int i = 0; i < (int)v.size() && i < cap; i++) dst[i] = smelter_shape{v[i].channels, v[i].width, v[i].height, v[i].depth};
    return (int)v.size();
}
int smelter_graph_num_nodes(const smelter_graph* g) { return g ? (int)g->g->numNodes() : 0; }
int smelter_graph_weight_bytes(smelter_graph* g, int32_t batch, size_t* bytes) {
    SM_REQUIRE(g && bytes && batch > 0, "smelter_graph_weight_bytes: bad argument");
    return Executor::weightBytes(*g->g, batch, bytes);
}
int smelter_graph_pack_weights(smelter_graph* g, int32_t batch, void* dst, size_t cap) {
    SM_REQUIRE(g && dst && batch > 0, "smelter_graph_pack_weights: bad argument");
    return Executor::packWeightsHost(*g->g, batch, dst, cap);
}
int smelter_graph_compile(smelter_graph* g, int32_t dev, int32_t batch, void* arena, int32_t upload, smelter_exec** out) {
    SM_REQUIRE(g && out, "smelter_graph_compile: null argument");
    Executor* e = nullptr;
    int rc = Executor::create(*g->g, dev, batch, arena, upload != 0, &e);
    if (rc) return rc;
    *out = new smelter_exec{std::unique_ptr<Executor>(e)};
    return SM_OK;
}
void smelter_exec_destroy(smelter_exec* e) { delete e; }

int smelter_exec_encode(smelter_exec* e, sm_stream_t s, const float* in) {
    SM_REQUIRE(e, "smelter_exec_encode: null executor");
    return e->e->encode((hipStream_t)s, in);
}
int smelter_exec_encode_sized(smelter_exec* e, sm_stream_t s, const float* in, int32_t in_h, int32_t in_w) {
    SM_REQUIRE(e, "smelter_exec_encode_sized: null executor");
    return e->e->encodeSized((hipStream_t)s, in, in_h, in_w);
}
int smelter_exec_output_dims(const smelter_exec* e, int32_t* n, int32_t* c, int32_t* h, int32_t* w) {
    SM_REQUIRE(e && n && c && h && w, "smelter_exec_output_dims: null argument");
    e->e->outputDims(n, c, h, w); return SM_OK;
}
int smelter_exec_read_float(smelter_exec* e, sm_stream_t s, float* dst, size_t cap, int32_t layout) {
    SM_REQUIRE(e && dst, "smelter_exec_read_float: null argument");
    return e->e->readFloat((hipStream_t)s, dst, cap, layout);
}
const void* smelter_exec_output_device(const smelter_exec* e) { return e ? e->e->outputDevice() : nullptr; }
int smelter_exec_num_launches(const smelter_exec* e) { return e ? e->e->numLaunches() : 0; }
int smelter_exec_num_ops(const smelter_exec* e) { return e ? e->e->numOps() : 0; }
size_t smelter_exec_weight_bytes(const smelter_exec* e) { return e ? e->e->weightBytesUsed() : 0; }
void* smelter_exec_weight_arena(const smelter_exec* e) { return e ? e->e->weightArena() : nullptr; }
size_t smelter_exec_activation_bytes(const smelter_exec* e) { return e ? e->e->activationBytes() : 0; }
double smelter_exec_flops(const smelter_exec* e) { return e ? e->e->flops() : 0; }
double smelter_exec_algorithmic_bytes(const smelter_exec* e) { return e ? e->e->algorithmicBytes() : 0; }
double smelter_exec_launch_bytes(const smelter_exec* e) { return e ? e->e->launchBytes() : 0; }
const char* smelter_exec_op_desc(const smelter_exec* e, int32_t idx) { return e ? e->e->opDesc(idx) : ""; }
int smelter_exec_num_values(const smelter_exec* e) { return e ? e->e->numMaterialized() : 0; }
const char* smelter_exec_value_name(const smelter_exec* e, int32_t idx) { return e ? e->e->materializedName(idx) : ""; }
int smelter_exec_time_op(smelter_exec* e, sm_stream_t s, int32_t idx, int32_t iters, float* ms) {
    SM_REQUIRE(e && ms, "smelter_exec_time_op: null argument");
    return e->e->timeOp((hipStream_t)s, idx, iters, ms);
}
int smelter_exec_profile(smelter_exec* e, sm_stream_t s, const float* in, int32_t iters, float* ms, int32_t cap) {
    SM_REQUIRE(e, "smelter_exec_profile: null executor");
    return e->e->profile((hipStream_t)s, in, iters, ms, cap);
}
int smelter_exec_read_value(smelter_exec* e, sm_stream_t s, const char* name, float* dst, size_t cap, int32_t* c, int32_t* h, int32_t* w) {
    SM_REQUIRE(e && name && dst && c && h && w, "smelter_exec_read_value: null argument");
    return e->e->readValue((hipStream_t)s, name, dst, cap, c, h, w);
}

}  // extern "C"
