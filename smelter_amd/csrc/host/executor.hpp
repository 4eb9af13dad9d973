// executor.hpp — what `metalGraph(device:)` returns in the reference is an MPSNNGraph
// (/root/reference/Sources/Smelter/ONNXGraph.swift:185-190); here it is an Executor: the converted op list
// after epilogue fusion, a static memory plan, a packed weight arena on the GPU and a launch sequence of sm_*
// kernel calls, optionally replayed as one hipGraph.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "graph.hpp"
#include "../../../include/smelter_hip.h"

namespace smelter {

struct Placement {           // where a value lives
    int root = -1;           // value id owning the physical buffer
    int c_off = 0;           // channel offset inside the root's pixel
    int ld = 0;              // pixel stride in elements
};

struct Step {                // one op after fusion, with everything resolved
    Op op;
    int M = 0;               // output pixels (batch*Ho*Wo)
    size_t w_off = 0, b_off = 0, s_off = 0, t_off = 0;   // arena offsets: weight fp16, bias f32, bn scale/shift f32
    bool has_bias = false;
    int conv_cfg = -1;       // tile configuration (-1 = heuristic)
    int launches = 1;
    int real_cin = 0;        // before first-layer channel padding (FLOP accounting)
    int fc_h = 1, fc_w = 1;  // FullyConnected over a flattened HxW map: weight columns permuted NCHW->NHWC order
    bool stem = false;       // first-layer conv running on the row-pair image (executor.cpp "image stem")
    bool stem3 = false; int stem3_p = -1;   // ... launched as sm_stem3x3_f32_fwd instead (pass 3j: 3x3, C <= 3, symmetric padding stem3_p in {0, 1}); weights = stem_w at ws_off
    int real_k = 0;          // kH*kW*Cin/groups of the model (FLOP / byte accounting, description)
    int real_kh = 0, real_kw = 0;
    // branch concurrency (executor.cpp "side branch"): a conv whose result is only read as the residual operand of a later
    // step `join` runs on the side stream, concurrently with the steps in between
    int join = -1;           // >= 0: this step runs on the side stream and step `join` waits for it
    bool waits_side = false; // this step is the join of an earlier side step
    int pool3_out = -1;      // >= 0 (image stem only): value id of the 3x3 / stride-2 max-pool of the stem's output, the ONLY tensor the launch writes (sm_conv2d_stem_f32_pool_fwd)
    int pool_out = -1;       // >= 0: value id of the 2x2 / stride-2 max-pool of this conv's output, written by the same launch (sm_conv2d_pool_fwd_cfg)
    size_t w2_off = 0, b2_off = 0; bool has_bias2 = false;                // second weight block of a fused step (head / projection of an inverted-residual block)
    // inverted-residual block (executor.cpp pass 3c): this step is the 1x1 EXPAND; `mb_dw` (depthwise 3x3) and `mb_proj` (1x1 project, optional
    // residual = this step's input) run in the same launch (sm_mbconv_fwd) and the two hidden tensors are never materialised
    bool has_mb = false; Op mb_dw, mb_proj; size_t wd_off = 0, bd_off = 0; bool has_biasd = false;
    // 3x3 convolution -> 1x1 "head" (executor.cpp pass 3d): `head` is the pointwise convolution that is the only reader of this step's output; it
    // runs in the same launch (sm_conv2d_head_fwd_cfg; weights at w2_off / b2_off) and the 3x3 output is never stored
    bool has_head = false; Op head;
    // ... and the projection shortcut of a first bottleneck block: `side` is the 1x1 convolution of the 64-channel block input whose fused residual operand
    // is the head's output; it becomes the head's second GEMM (sm_conv2d_xhead_proj_fwd_cfg; weights at wd_off, the two biases summed at b2_off) and the
    // head's own output is never materialised
    bool has_side = false; Op side;
    // ... and the 1x1 REDUCE layer in front of the 3x3 (executor.cpp pass 3l): `reduce` is the pointwise convolution whose only reader is this step; the whole bottleneck block —
    // reduce, 3x3, expand head (+ residual) — is one launch (sm_bottleneck_fwd; reduce weights at w0_off / b0_off) and neither hidden tensor is stored (both stay in the rounding list)
    bool has_reduce = false; Op reduce; size_t w0_off = 0, b0_off = 0; bool has_bias0 = false;
    // ... behind an image stem (executor.cpp pass 3g): the stem convolution is the block kernel's "expand" layer, gathered from the fp32 batch
    // (sm_mbconv_stem_fwd): `stem_w` = the stem's weights [Ch][3][3][C] before the row-pair packing (kept for every stem step), at ws_off / bs_off
    bool mb_stem = false; std::vector<float> stem_w, stem_b; size_t ws_off = 0, bs_off = 0; int stem_oh = 0, stem_ow = 0, stem_s = 1, stem_p = 0, stem_act = 0; float stem_slope = 0.f;
    int stem_out = -1; std::string stem_name;
    // pass 3h: `kcat` = this 1x1 convolution is an expand layer and the projection shortcut that carried its Add, concatenated along K (its extra
    // algorithmic bytes = the expand output the graph writes and reads back); `gather` = the strided copy of the block input placed beside the expand
    // layer's input (a MaxPool 1x1 step the graph does not hold: not accounted)
    int dual_src = -1, dual_stride = 1;   // >= 0: value id of the SECOND source of a K-concatenated pointwise convolution, read in place at this pixel stride (sm_conv2d_dual_fwd_cfg)
    bool kcat = false, gather = false; double kcat_extra_elems = 0;   // elements the two graph layers move beyond the merged layer's own operands
    size_t wp_off = 0; bool has_wp = false;   // member of a persistent run (executor.cpp pass 3k): the weights once more, in the fragment order of the run's tile form (sm_stage_pack_weights)
    bool mb_noexp = false;   // the block has no expand layer: this step IS the depthwise convolution (mb_dw is a copy of it), mb_proj follows
    std::string desc;
};

// pass 3k: steps [first, first + count) — plain 1x1 / 3x3 stride-1 convolutions at one resolution — run as ONE persistent launch (sm_stage_fwd) when `on`
struct StageRun {
    int first = 0, count = 0; bool on = true, wide = false;
    // what the compile's in-situ A/B measured (autotune): medians over `samples` interleaved launches of each form, the hand-off modes the timed
    // launches ran in, whether a hand-off wait ever ran out, and why the run was turned down (SM_RUN_* of smelter_hip.h; 0 = not turned down)
    bool timed = false; float t_run_us = 0.f, t_layers_us = 0.f; int samples = 0, mode_bits = 0, err_bits = 0, why_off = 0;
};

class Executor {
   public:
    // plan = the tile-configuration picks of another executor of the same graph / batch / library build (exportPlan): applied instead of autotuning
    static int create(ONNXGraph& g, int device, int batch, void* ext_arena, bool upload, Executor** out, const int32_t* plan = nullptr, size_t plan_words = 0);
    // [0] = 'PLAN', [1] = tile configurations in this build, [2] = steps, [3 ..] = conv_cfg per step (-1: heuristic / no tile configuration)
    std::vector<int32_t> exportPlan() const;
    int importPlan(const int32_t* plan, size_t words);
    static int defaultPlanHost(ONNXGraph& g, int batch, std::vector<int32_t>* out);   // host only: the step list's plan with no picks (what a non-autotuned compile runs)
    static int weightBytes(ONNXGraph& g, int batch, size_t* bytes);
    // host-side arena image (fold + cast + pad + pack) without touching a GPU
    static int packWeightsHost(ONNXGraph& g, int batch, void* dst, size_t cap);
    ~Executor();

    int encode(hipStream_t s, const float* in_nchw);
    // Configuration.inputConstraint .forceInputScale: source images of any size, resampled to the model's input size first
    int encodeSized(hipStream_t s, const float* in_nchw, int in_h, int in_w);
    int readFloat(hipStream_t s, float* dst, size_t cap, int layout);
    int readValue(hipStream_t s, const char* name, float* dst, size_t cap, int* c, int* h, int* w);
    int timeOp(hipStream_t s, int idx, int iters, float* ms);
    // in-situ timing: `iters` full eager passes with HIP events between consecutive launches on `s`;
    // ms[0] = input conversion, ms[1 + i] = op i (averages).  cap >= numOps() + 1.
    int profile(hipStream_t s, const float* in, int iters, float* ms, int cap);

    int batch() const { return batch_; }
    void outputDims(int* n, int* c, int* h, int* w) const;
    const void* outputDevice() const;
    int numLaunches() const { return numLaunchesNow(); }
    int numOps() const { return (int)steps_.size(); }
    int numRuns() const { return (int)runs_.size(); }
    const StageRun* run(int i) const { return i >= 0 && i < (int)runs_.size() ? &runs_[i] : nullptr; }
    size_t weightBytesUsed() const { return arena_bytes_; }
    void* weightArena() const { return arena_; }
    size_t activationBytes() const { return act_bytes_; }
    double flops() const { return flops_; }
    double algorithmicBytes() const { return alg_bytes_; }
    double launchBytes() const { return launch_bytes_; }       // post-fusion compulsory bytes: what the launches of one encode must read and write
    const char* opDesc(int idx) const { return idx >= 0 && idx < (int)steps_.size() ? steps_[idx].desc.c_str() : ""; }
    // names of the tensors that exist in fp16 storage after fusion (graph input + every step output)
    int numMaterialized() const { return (int)materialized_.size(); }
    const char* materializedName(int i) const { return i >= 0 && i < (int)materialized_.size() ? materialized_[i].c_str() : ""; }
    int materializedStored(int i) const;   // 1 / 0: does the i-th listed tensor have storage (readValue works)

   private:
    Executor() = default;
    int plan(ONNXGraph& g, int batch);                   // fusion + placement + memory plan + arena layout
    int packWeights(std::vector<uint8_t>* host) const;   // fold/cast/pad -> arena image
    int launchStep(const Step& st, hipStream_t s) const;
    int launchRun(const StageRun& r, hipStream_t s) const;
    // the phase list of a run over the given arena / activation base addresses (real ones, or any two disjoint fake ranges for host-side planning and packing)
    std::vector<sm_stage_phase> runPhases(const StageRun& r, const char* arena, const char* act, bool packed) const;
    int numLaunchesNow() const;
    void annotateRuns();
    int checkRuns(hipStream_t s) const;
    sm_conv_desc convDesc(const Step& st) const;
    int convertInput(hipStream_t s, const float* in) const;
    int launchAll(hipStream_t s, const float* in) const;
    int autotune(hipStream_t s);
    void* valuePtr(int v) const;
    int valueLd(int v) const { return place_[v].ld; }

    int device_ = 0, batch_ = 1;
    Configuration cfg_;
    std::vector<Value> values_;
    std::vector<Placement> place_;
    std::vector<size_t> root_offset_;    // per value id (roots only), byte offset in the activation arena
    std::vector<Step> steps_;
    std::vector<StageRun> runs_; std::vector<int> step_run_;   // per step: index into runs_ or -1
    void* stage_ctl_ = nullptr;
    int input_value_ = -1, output_value_ = -1, input_ld_ = 0;
    float* resized_ = nullptr;       // forceInputScale: the resampled batch (NCHW fp32), allocated on first use
    bool stem_ = false; int stem_h_ = 0, stem_w_ = 0, stem_pt_ = 0, stem_pl_ = 0;   // row-pair image geometry
    // the stem kernel's loader waves build its patches from the fp32 batch (sm_conv2d_stem_f32_fwd_cfg): no conversion launch, no row-pair image
    bool stem_f32_ = false; mutable const float* cur_in_ = nullptr;
    bool stem_in_block_ = false;     // pass 3g: the stem runs inside the first block's launch
    std::vector<std::string> value_names_, materialized_;
    std::vector<char> stored_;           // per value id: 1 = the tensor has storage in the activation arena (readValue can fetch it); 0 = it exists in registers / LDS only

    char* arena_ = nullptr; size_t arena_bytes_ = 0; bool own_arena_ = false;
    char* act_ = nullptr; size_t act_bytes_ = 0;
    char* ws_ = nullptr; size_t ws_bytes_ = 0;           // split-K scratch
    float* out_f32_ = nullptr; size_t out_f32_count_ = 0; // readback staging
    int launches_ = 0;
    double flops_ = 0, alg_bytes_ = 0, launch_bytes_ = 0;

    // hipGraph replay
    hipGraph_t graph_ = nullptr; hipGraphExec_t graph_exec_ = nullptr; const float* captured_in_ = nullptr;
    // side branch: one extra stream, fork/join events (timing disabled)
    hipStream_t side_ = nullptr; hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
};

}  // namespace smelter
