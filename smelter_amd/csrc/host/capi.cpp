// capi.cpp — graph-level C ABI (include/smelter_hip.h §2): the ONNXGraph load-and-encode surface
// (/root/reference/Sources/Smelter/ONNXGraph.swift:95,158,169; README.md:43-44;
// Extensions/Foundation/MPSImage+Extensions.swift:9) over opaque handles.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
    c->fusions = SM_FUSE_DEFAULT; c->autotune_mode = SM_TUNE_IN_SITU; c->autotune_max_cfg = 0; c->mbconv_domain[0] = c->mbconv_domain[1] = c->mbconv_domain[2] = 0;
    // The ONLY place the executor's environment switches are read (bench / A-B tools; documented in smelter_hip.h): they edit the defaults, once.
    auto off = [&](const char* name, int bit) { const char* e = getenv(name); if (e && e[0] == '0') c->fusions &= ~bit; };
    off("SMELTER_MBCONV", SM_FUSE_MBCONV); off("SMELTER_XHEAD", SM_FUSE_XHEAD); off("SMELTER_XHEAD_PROJ", SM_FUSE_XHEAD_PROJ); off("SMELTER_KCAT", SM_FUSE_KCAT);
    off("SMELTER_BNECK", SM_FUSE_BNECK); off("SMELTER_MBCONV_STEM", SM_FUSE_MBCONV_STEM); off("SMELTER_STEM_F32", SM_FUSE_STEM_F32); off("SMELTER_STEM_POOL", SM_FUSE_STEM_POOL); off("SMELTER_STEM3", SM_FUSE_STEM3); off("SMELTER_STAGE", SM_FUSE_STAGE);
    if (const char* e = getenv("SMELTER_STAGE")) if (!strcmp(e, "wide")) c->fusions |= SM_FUSE_STAGE_WIDE;
    if (const char* e = getenv("SMELTER_KCAT")) if (!strcmp(e, "copy")) c->fusions |= SM_FUSE_KCAT_COPY;
    if (const char* e = getenv("SMELTER_AUTOTUNE")) if (!strcmp(e, "cold")) c->autotune_mode = SM_TUNE_COLD;
    if (const char* e = getenv("SMELTER_AUTOTUNE_MAXCFG")) c->autotune_max_cfg = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("SMELTER_MBCONV_DOMAIN")) { int a = 0, b = 0, d = 0; if (sscanf(e, "%d,%d,%d", &a, &b, &d) == 3) { c->mbconv_domain[0] = a; c->mbconv_domain[1] = b; c->mbconv_domain[2] = d; } }
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
    c.fusions = cfg->fusions; c.autotuneMode = cfg->autotune_mode; c.autotuneMaxCfg = cfg->autotune_max_cfg;
    for (int i = 0; i < 3; i++) c.mbconvDomain[i] = cfg->mbconv_domain[i];
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
int smelter_graph_compile_with_plan(smelter_graph* g, int32_t dev, int32_t batch, void* arena, int32_t upload, const int32_t* plan, size_t plan_words, smelter_exec** out) {
    SM_REQUIRE(g && out && plan, "smelter_graph_compile_with_plan: null argument");
    Executor* e = nullptr;
    int rc = Executor::create(*g->g, dev, batch, arena, upload != 0, &e, plan, plan_words);
    if (rc) return rc;
    *out = new smelter_exec{std::unique_ptr<Executor>(e)};
    return SM_OK;
}
int smelter_exec_get_plan(const smelter_exec* e, int32_t* dst, size_t cap_words, size_t* words) {
    SM_REQUIRE(e && words, "smelter_exec_get_plan: null argument");
    const std::vector<int32_t> p = e->e->exportPlan();
    *words = p.size();
    if (dst) { SM_REQUIRE(cap_words >= p.size(), "smelter_exec_get_plan: buffer holds %zu words, the plan has %zu", cap_words, p.size()); memcpy(dst, p.data(), p.size() * sizeof(int32_t)); }
    return SM_OK;
}
int smelter_graph_default_plan(smelter_graph* g, int32_t batch, int32_t* dst, size_t cap_words, size_t* words) {
    SM_REQUIRE(g && words, "smelter_graph_default_plan: null argument");
    std::vector<int32_t> p; int rc = Executor::defaultPlanHost(*g->g, batch, &p); if (rc) return rc;
    *words = p.size();
    if (dst) { SM_REQUIRE(cap_words >= p.size(), "smelter_graph_default_plan: buffer holds %zu words, the plan has %zu", cap_words, p.size()); memcpy(dst, p.data(), p.size() * sizeof(int32_t)); }
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
int smelter_exec_num_runs(const smelter_exec* e) { return e ? e->e->numRuns() : 0; }
int smelter_exec_run_info(const smelter_exec* e, int32_t idx, smelter_run_info* out) {
    SM_REQUIRE(e && out, "smelter_exec_run_info: null argument");
    const StageRun* r = e->e->run(idx);
    SM_REQUIRE(r, "smelter_exec_run_info: this executor has %d persistent runs, index %d", e->e->numRuns(), idx);
    *out = smelter_run_info{r->first, r->count, r->on ? 1 : 0, r->timed ? 1 : 0, r->samples, r->mode_bits, r->err_bits, r->why_off, r->t_run_us, r->t_layers_us};
    return SM_OK;
}
size_t smelter_exec_weight_bytes(const smelter_exec* e) { return e ? e->e->weightBytesUsed() : 0; }
void* smelter_exec_weight_arena(const smelter_exec* e) { return e ? e->e->weightArena() : nullptr; }
size_t smelter_exec_activation_bytes(const smelter_exec* e) { return e ? e->e->activationBytes() : 0; }
double smelter_exec_flops(const smelter_exec* e) { return e ? e->e->flops() : 0; }
double smelter_exec_algorithmic_bytes(const smelter_exec* e) { return e ? e->e->algorithmicBytes() : 0; }
double smelter_exec_launch_bytes(const smelter_exec* e) { return e ? e->e->launchBytes() : 0; }
const char* smelter_exec_op_desc(const smelter_exec* e, int32_t idx) { return e ? e->e->opDesc(idx) : ""; }
int smelter_exec_num_values(const smelter_exec* e) { return e ? e->e->numMaterialized() : 0; }
const char* smelter_exec_value_name(const smelter_exec* e, int32_t idx) { return e ? e->e->materializedName(idx) : ""; }
int smelter_exec_value_stored(const smelter_exec* e, int32_t idx) { return e ? e->e->materializedStored(idx) : 0; }
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
