// executor.cpp — compile the converted op list into a GPU launch sequence (see executor.hpp).
//
// Passes, in order:
//   1. epilogue fusion   BN folded into the producing Conv/FC (the arithmetic ONNX2MPS.py:107 delegates to
//                        onnx.optimizer `fuse_bn_into_conv`: s = gamma/sqrt(var+eps); W' = W*s[oc];
//                        b' = (b-mean)*s+beta), residual Add and Relu/PRelu fused into conv epilogues.
//   2. aliasing          Flatten / Reshape / Dropout never move data.
//   3. concat placement  producers write straight into channel slices of the concat buffer (pixel stride =
//                        total C), so MPSNNConcatenationNode (Converters.swift:564) costs no kernel.
//   4. memory plan       liveness-based first-fit packing of activation buffers into one arena: buffers are
//                        reused as soon as their last consumer ran, which keeps the working set inside the
//                        256 MiB Infinity Cache for ResNet-50 bs=64.
//   5. weight arena      fp16 OHWI weights + fp32 bias/BN vectors packed into ONE device buffer, so that a
//                        batch-shard replica can receive it with a single RCCL broadcast.
//   6. launch sequence   sm_* C-ABI calls only; replayed as a hipGraph when configured.
#include "executor.hpp"
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <numeric>

#include "../kernels/common.hpp"
#include "half.hpp"

namespace smelter {
namespace {

constexpr size_t kAlign = 256;
size_t alignUp(size_t v, size_t a = kAlign) { return (v + a - 1) / a * a; }
// activations the fused block kernels (mbconv.hip, stem3.hip) apply: the ReLU family and ONNX Clip(0, hi) (ReLU6) — ADVICE r4: a torchvision MobileNet-v2 export holds Clip
bool blockAct(int act) { return act <= SM_ACT_PRELU || act == SM_ACT_CLIP; }
bool isConvLike(OpKind k) { return k == OpKind::Conv || k == OpKind::FullyConnected || k == OpKind::ConvTranspose; }
const char* kindName(OpKind k) {
    switch (k) {
    case OpKind::Conv: return "Conv"; case OpKind::FullyConnected: return "FC"; case OpKind::BatchNorm: return "BatchNorm";
    case OpKind::Neuron: return "Neuron"; case OpKind::Add: return "Add"; case OpKind::MaxPool: return "MaxPool";
    case OpKind::AvgPool: return "AvgPool"; case OpKind::GlobalAvgPool: return "GlobalAvgPool"; case OpKind::Upsample: return "Upsample";
    case OpKind::Concat: return "Concat"; case OpKind::Alias: return "Alias"; case OpKind::ConvTranspose: return "ConvTranspose";
    case OpKind::Softmax: return "Softmax"; case OpKind::GroupNorm: return "GroupNorm"; case OpKind::Pad: return "Pad";
    case OpKind::Reshape: return "Reshape";
    }
    return "?";
}

// Optional trace ranges (SMELTER_ROCTX=1): every launch of a step is bracketed by roctxRangePush(<smelter_exec_op_desc>) / roctxRangePop, so a
// `rocprofv3 --kernel-trace --marker-trace` run names its kernel rows by ONNX node — what the reference's node labels do for a Metal GPU
// capture (Converters.swift:178, :787, :931, :1064).  libroctx64 is looked up at run time: the product library does not link against it.
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr; bool on = false;
    Roctx() {
        const char* e = getenv("SMELTER_ROCTX");
        if (!e || e[0] == '0') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);       // the library rocprofv3 --marker-trace intercepts
        if (!h) h = dlopen("/opt/rocm/lib/librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "SMELTER_ROCTX: no roctx library found, ranges disabled\n"); return; }
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop");
        on = push && pop;
    }
};
const Roctx& roctx() { static const Roctx r; return r; }

}  // namespace

Executor::~Executor() {
    if (stage_ctl_) (void)hipFree(stage_ctl_);
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    if (own_arena_ && arena_) (void)hipFree(arena_);
    if (act_) (void)hipFree(act_);
    if (resized_) (void)hipFree(resized_);
    if (ws_) (void)hipFree(ws_);
    if (out_f32_) (void)hipFree(out_f32_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (side_) (void)hipStreamDestroy(side_);
}

void* Executor::valuePtr(int v) const {
    const Placement& p = place_[v];
    return act_ + root_offset_[p.root] + (size_t)p.c_off * 2;
}
void Executor::outputDims(int* n, int* c, int* h, int* w) const {
    const Value& v = values_[output_value_]; *n = batch_; *c = v.c; *h = v.h; *w = v.w;
}
const void* Executor::outputDevice() const { return valuePtr(output_value_); }

// ------------------------------------------------------------------------------------------------------------
int Executor::plan(ONNXGraph& g, int batch) {
    batch_ = batch; cfg_ = g.configuration;
    values_ = g.values(); input_value_ = g.inputValue(); output_value_ = g.outputValue();
    std::vector<Op> ops = g.ops();
    int nv = (int)values_.size(); const int no = (int)ops.size();
    std::vector<char> dead(no, 0);

    auto consumers = [&](int v) { int c = v == output_value_ ? 1 : 0; for (int i = 0; i < no; i++) if (!dead[i]) { for (int u : ops[i].inputs) c += u == v; c += ops[i].residual == v; } return c; };
    auto producerOf = [&](int v) { for (int i = 0; i < no; i++) if (!dead[i] && ops[i].output == v) return i; return -1; };

    // ---- 1. epilogue fusion --------------------------------------------------------------------------------
    if (cfg_.fuse) {
        for (int i = 0; i < no; i++) {
            Op& op = ops[i];
            if (op.kind == OpKind::BatchNorm) {
                int v = op.inputs[0], p = producerOf(v);
                if (p >= 0 && isConvLike(ops[p].kind) && ops[p].act == SM_ACT_NONE && ops[p].residual < 0 && consumers(v) == 1) {
                    Op& c = ops[p];
                    const size_t per = c.weight.size() / c.cout;
                    if (c.bias.empty()) c.bias.assign(c.cout, 0.f);
                    for (int o = 0; o < c.cout; o++) {
                        const float s = op.scale[o];
                        float* w = c.weight.data() + (size_t)o * per;
                        for (size_t k = 0; k < per; k++) w[k] *= s;
                        c.bias[o] = c.bias[o] * s + op.shift[o];          // (b - mean)*s + beta
                    }
                    c.output = op.output; dead[i] = 1;
                }
            } else if (op.kind == OpKind::Add && op.binop == SM_BIN_ADD) {
                int pa = producerOf(op.inputs[0]), pb = producerOf(op.inputs[1]);
                auto ok = [&](int p, int v) { return p >= 0 && isConvLike(ops[p].kind) && ops[p].kind != OpKind::ConvTranspose && ops[p].act == SM_ACT_NONE && ops[p].residual < 0 && consumers(v) == 1; };
                int cand = -1, other = -1;
                bool oka = ok(pa, op.inputs[0]), okb = ok(pb, op.inputs[1]) && op.inputs[0] != op.inputs[1];
                if (oka && okb && cfg_.branchConcurrency) {
                    // Projection shortcut: of the two convolutions feeding the Add, the LATER one in file order may depend only on
                    // values that existed before the other branch started (ResNet: the 1x1 on the block input, exported after
                    // the three main-branch convs).  Hoisted to right behind the producer of its input it becomes the plain
                    // conv, the end of the main branch fuses the Add, and the executor can run it on the side stream.
                    const int hi = std::max(pa, pb), lo = std::min(pa, pb);
                    int ready = -1; for (int u : ops[hi].inputs) ready = std::max(ready, producerOf(u));
                    if (ready + 1 < lo) {
                        std::rotate(ops.begin() + ready + 1, ops.begin() + hi, ops.begin() + hi + 1);
                        std::rotate(dead.begin() + ready + 1, dead.begin() + hi, dead.begin() + hi + 1);
                        pa = producerOf(ops[i].inputs[0]); pb = producerOf(ops[i].inputs[1]);
                    }
                }
                if (oka && okb) { if (pa > pb) { cand = pa; other = op.inputs[1]; } else { cand = pb; other = op.inputs[0]; } }
                else if (oka) { cand = pa; other = op.inputs[1]; }
                else if (okb) { cand = pb; other = op.inputs[0]; }
                if (cand >= 0 && producerOf(other) < cand) {               // the other operand is ready when `cand` runs
                    ops[cand].residual = other; ops[cand].output = op.output; dead[i] = 1;
                }
            } else if (op.kind == OpKind::Neuron) {
                int v = op.inputs[0], p = producerOf(v);
                if (p >= 0 && ops[p].act == SM_ACT_NONE && consumers(v) == 1 &&
                    (isConvLike(ops[p].kind) || ops[p].kind == OpKind::Add || ops[p].kind == OpKind::BatchNorm)) {
                    ops[p].act = op.act; ops[p].slope = op.slope; ops[p].output = op.output; dead[i] = 1;
                }
            }
        }
    }

    // ---- 1b. nearest 2x Upsample -> 3x3 convolution: the convolution gathers its patches straight from the half-resolution tensor
    // (sm_conv_desc.x_ups, conv3x3_p9.hpp); the upsampled image — 4x the bytes, written once and read once — never exists.
    if (cfg_.fuse && !cfg_.keepIntermediates) {
        for (int i = 0; i < no; i++) {
            Op& up = ops[i];
            if (dead[i] || up.kind != OpKind::Upsample || up.bilinear || up.sy != 2 || up.sx != 2 || up.output == output_value_ || consumers(up.output) != 1) continue;
            int c = -1; for (int j = 0; j < no; j++) if (!dead[j] && !ops[j].inputs.empty() && ops[j].inputs[0] == up.output) c = j;
            if (c < 0) continue;
            Op& cv = ops[c];
            if (cv.kind != OpKind::Conv || cv.groups != 1 || cv.kh != 3 || cv.kw != 3 || cv.sh != 1 || cv.sw != 1 || cv.dh != 1 || cv.dw != 1 || cv.pt != 1 || cv.pl != 1 ||
                cv.pb != 1 || cv.pr != 1 || ((cv.cin % 64) && !(cv.cin == 32 && (cv.cout == 32 || cv.cout == 64))) || (cv.cout % 8) || cv.residual >= 0 || cv.act > SM_ACT_PRELU || up.inputs[0] == input_value_) continue;
            // the fused read needs 16-byte channel vectors on both sides; placement happens later (pass 3), so be conservative here: neither tensor
            // may sit in a Concat that has an operand whose channel count is not a multiple of 8 (it could then land on an odd channel offset)
            bool odd_concat = (values_[up.inputs[0]].c & 7) != 0 || (cv.cout & 7) != 0;
            for (int j = 0; j < no && !odd_concat; j++) if (!dead[j] && ops[j].kind == OpKind::Concat) {
                bool member = false, odd = false;
                for (int u : ops[j].inputs) { member |= u == up.inputs[0] || u == cv.output; odd |= (values_[u].c & 7) != 0; }
                odd_concat = member && odd;
            }
            if (odd_concat) continue;
            cv.inputs[0] = up.inputs[0]; cv.x_ups = true; dead[i] = 1;
        }
    }

    // ---- 2. aliasing + 3. concat placement -----------------------------------------------------------------
    place_.assign(nv, Placement());
    for (int v = 0; v < nv; v++) { place_[v].root = v; place_[v].c_off = 0; place_[v].ld = values_[v].c; }
    // first-layer image: pad channels to a multiple of 8 so the MFMA implicit-GEMM path applies (zeros written
    // by sm_nchw_f32_to_nhwc_f16, matching zero weight columns)
    input_ld_ = (values_[input_value_].c + 7) / 8 * 8;
    place_[input_value_].ld = input_ld_;
    // "image stem": an RGB(A) image feeding ONE ordinary convolution.  With C <= 4 the im2col row of a pixel is kH runs of
    // kW*C <= 32 contiguous halfs; stored as a row-pair image (sm_nchw_f32_to_rowpair_f16: pixel = [row y | row y+1] x 4
    // channels, zero halo instead of padding) two kernel rows of 8 pixels are 64 contiguous halfs, so the layer runs on
    // the BK = 64 LDS-DMA kernel as a Cin = 64, kH' = ceil(kH/2), kW' = 1, dilation-2 convolution with K' = 64 kH' instead
    // of the generic BK = 32 kernel with K = kH*kW*8 (ResNet-50 7x7/2: K' = 256 vs 392).
    int stem_op = -1;
    if (!cfg_.keepIntermediates && input_value_ != output_value_ && values_[input_value_].c <= 4) {
        int users = 0;
        for (int j = 0; j < no; j++) if (!dead[j]) {
            for (int u : ops[j].inputs) if (u == input_value_) { users++; stem_op = j; }
            if (ops[j].residual == input_value_) users += 2;
        }
        if (users != 1) stem_op = -1;
        if (stem_op >= 0) {
            const Op& o = ops[stem_op];
            if (o.kind != OpKind::Conv || o.groups != 1 || o.kw > 8 || o.dh != 1 || o.dw != 1 || o.cin != values_[input_value_].c) stem_op = -1;
        }
    }

    std::vector<int> alias_of(nv, -1);       // value -> value it aliases
    std::vector<int> placed_in(nv, -1);      // value -> concat value it was placed into
    std::vector<int> fc_permute(no, 0);
    for (int i = 0; i < no; i++) {
        if (dead[i]) continue;
        Op& op = ops[i];
        if (op.kind == OpKind::Alias) {
            // Flatten / Reshape / Dropout / Pow.  The reference's image keeps the linear (c, y, x) element order across a reshape
            // (MPSNNReshapeNode); ours is NHWC with a pixel stride `ld`.  The two orders coincide — and the output may share the
            // source's buffer AND stride — only when the shape is unchanged, or when both sides are 1x1 maps (one dense row of C
            // values per image).  A spatial map flattened for fully-connected consumers only is also left in place: those layers
            // read one dense row per image and get their weight columns permuted NCHW -> NHWC at pack time.  Every other reshape
            // (1x1xC -> a spatial map, spatial -> spatial with other dims, a reshape that is the graph output ...) is MATERIALISED
            // by sm_reshape_nchw_order into a dense buffer of the output's own shape.
            int in = op.inputs[0]; const Value& iv = values_[in]; const Value& ov = values_[op.output];
            const bool same = ov.c == iv.c && ov.h == iv.h && ov.w == iv.w;
            const bool both_flat = iv.h * iv.w == 1 && ov.h * ov.w == 1;
            if (!same && !both_flat) {
                bool all_fc = op.output != output_value_ && ov.h * ov.w == 1 && in != input_value_; int users = 0;
                for (int j = 0; j < no; j++) if (!dead[j]) {
                    for (int u : ops[j].inputs) if (u == op.output) { users++; if (ops[j].kind != OpKind::FullyConnected) all_fc = false; }
                    if (ops[j].residual == op.output) all_fc = false;
                }
                if (all_fc && users > 0) {
                    for (int j = 0; j < no; j++) if (!dead[j] && ops[j].kind == OpKind::FullyConnected && ops[j].inputs[0] == op.output) fc_permute[j] = in + 1;
                } else { op.kind = OpKind::Reshape; continue; }             // gets its own buffer and a copy kernel
            }
            alias_of[op.output] = in;
        }
    }
    auto rootOf = [&](int v) { while (alias_of[v] >= 0) v = alias_of[v]; return v; };
    std::vector<std::vector<std::pair<int, int>>> concat_copies(no);   // (input value, channel offset) needing a copy kernel
    for (int i = 0; i < no; i++) {
        if (dead[i] || ops[i].kind != OpKind::Concat) continue;
        Op& op = ops[i]; int off = 0; const int ctot = values_[op.output].c;
        std::vector<int> seen;
        for (int u : op.inputs) {
            int r = rootOf(u);
            bool reorder_src = false; for (int j = 0; j < no; j++) if (fc_permute[j] == r + 1) reorder_src = true;
            bool can = r == u && u != input_value_ && u != output_value_ && placed_in[u] < 0 && !reorder_src &&
                       std::find(seen.begin(), seen.end(), u) == seen.end() && producerOf(u) >= 0 && producerOf(u) < i;
            if (can) { placed_in[u] = op.output; place_[u].c_off = off; }
            else concat_copies[i].push_back({u, off});
            seen.push_back(u); off += values_[u].c;
        }
        (void)ctot;
    }
    // resolve placement chains: alias -> source; placed -> concat buffer (+offset), recursively
    std::vector<int> state(nv, 0);
    std::function<void(int)> resolve = [&](int v) {
        if (state[v]) return; state[v] = 1;
        if (alias_of[v] >= 0) { int s = alias_of[v]; resolve(s); place_[v] = place_[s]; return; }
        if (placed_in[v] >= 0) { int b = placed_in[v]; resolve(b); int own = place_[v].c_off; place_[v].root = place_[b].root; place_[v].c_off = place_[b].c_off + own; place_[v].ld = place_[b].ld; return; }
    };
    for (int v = 0; v < nv; v++) resolve(v);
    // every fused kernel moves 16-byte channel vectors: an operand placed inside a concat buffer qualifies only when its channel offset and the
    // buffer's pixel stride are multiples of 8 halfs (arena blocks are 256-byte aligned, so this is the pointer's alignment as well).  The fusion
    // passes ask this BEFORE they erase a step: a fused step has no unfused fallback at launch time.
    auto vecOk = [&](int v) { return v >= 0 && (place_[v].c_off & 7) == 0 && (place_[v].ld & 7) == 0; };

    // ---- steps ---------------------------------------------------------------------------------------------
    steps_.clear(); launches_ = 1 /* input conversion */; flops_ = 0; alg_bytes_ = 0;
    for (int i = 0; i < no; i++) {
        if (dead[i]) continue;
        Op& op = ops[i];
        if (op.kind == OpKind::Alias) continue;
        if (op.kind == OpKind::Concat) {
            for (auto& cp : concat_copies[i]) {           // fallback copies, expressed as single-input Concat steps
                Step st; st.op.kind = OpKind::Concat; st.op.name = op.name; st.op.inputs = {cp.first}; st.op.output = op.output; st.op.pl = cp.second;
                st.M = batch * values_[op.output].h * values_[op.output].w; st.launches = 1;
                steps_.push_back(std::move(st));
            }
            continue;
        }
        if (op.kind == OpKind::ConvTranspose && op.groups == 1 && op.dh == 1 && op.dw == 1 && (op.cin % 8) == 0) {
            // A stride-s transposed convolution = s*s ordinary stride-1 convolutions, one per output phase
            // (py, px) = ((oy + pt) mod sH, (ox + pl) mod sW): output row oy = q*sH + py - pt only sees kernel rows
            // kh = py + sH*a, through input row iy = q - a.  Each phase gets its sub-kernel (taken from the flipped OHWI
            // master copy), its own padding (A - 1 - q_min, may be negative) and writes its pixels strided into the
            // shared output image (sm_conv_desc.y_sH ...).  No zero-inserted input, no wasted MACs.
            const Value& iv = values_[op.inputs[0]]; const Value& ov = values_[op.output];
            struct Ph { int A, qmin, Q, o0, pad; };
            auto phase = [](int k, int s, int p, int out, int ph) {
                Ph r; r.A = ph < k ? (k - ph + s - 1) / s : 0;
                r.qmin = (p - ph + s - 1) / s;                              // ceil((p - ph) / s), numerator + s - 1 >= 0
                const int num = out - 1 + p - ph;
                r.Q = num < 0 ? 0 : num / s - r.qmin + 1; if (r.Q < 0) r.Q = 0;
                r.o0 = r.qmin * s + ph - p; r.pad = (r.A > 0 ? r.A - 1 : 0) - r.qmin;
                return r;
            };
            (void)iv;
            for (int py = 0; py < op.sh; py++)
                for (int px = 0; px < op.sw; px++) {
                    const Ph y = phase(op.kh, op.sh, op.pt, ov.h, py), x = phase(op.kw, op.sw, op.pl, ov.w, px);
                    if (y.Q == 0 || x.Q == 0) continue;
                    Step st; Op po = op;                                    // copy: inputs, output, bias, act, name
                    po.kind = OpKind::Conv; po.sh = po.sw = 1; po.dh = po.dw = 1; po.pb = po.pr = 0;
                    const int Ay = y.A > 0 && x.A > 0 ? y.A : 1, Ax = y.A > 0 && x.A > 0 ? x.A : 1;   // tap-less phase: 1x1 zero kernel
                    po.kh = Ay; po.kw = Ax; po.pt = y.pad; po.pl = x.pad;
                    po.weight.assign((size_t)op.cout * Ay * Ax * op.cin, 0.f);
                    if (y.A > 0 && x.A > 0)
                        for (int oc = 0; oc < op.cout; oc++)
                            for (int a = 0; a < Ay; a++)
                                for (int b = 0; b < Ax; b++) {
                                    const int kh = py + op.sh * (Ay - 1 - a), kw = px + op.sw * (Ax - 1 - b);
                                    const float* src = op.weight.data() + (((size_t)oc * op.kh + (op.kh - 1 - kh)) * op.kw + (op.kw - 1 - kw)) * op.cin;
                                    std::copy(src, src + op.cin, po.weight.begin() + (((size_t)oc * Ay + a) * Ax + b) * op.cin);
                                }
                    po.y_sh = op.sh; po.y_sw = op.sw; po.y_oh = y.o0; po.y_ow = x.o0; po.y_h = ov.h; po.y_w = ov.w; po.grid_h = y.Q; po.grid_w = x.Q;
                    st.M = batch * y.Q * x.Q; st.real_cin = op.cin; st.real_k = (y.A > 0 && x.A > 0) ? Ay * Ax * op.cin : 0; st.real_kh = Ay; st.real_kw = Ax;
                    st.op = std::move(po);
                    steps_.push_back(std::move(st));
                }
            continue;
        }
        Step st; st.M = batch * values_[op.output].h * values_[op.output].w;
        st.real_cin = op.cin;
        if (op.kind == OpKind::FullyConnected && fc_permute[i]) { const Value& src = values_[fc_permute[i] - 1]; st.fc_h = src.h; st.fc_w = src.w; }
        st.real_k = isConvLike(op.kind) ? op.kh * op.kw * (op.cin / op.groups) : 0; st.real_kh = op.kh; st.real_kw = op.kw;
        if (op.kind == OpKind::ConvTranspose) st.real_k = std::max(1, st.real_k / (op.sh * op.sw));   // an output pixel sees ~1/(sH*sW) of the taps
        if (i == stem_op) {
            // weights [Cout][kH][kW][C] -> [Cout][P][8 pixels][row 2j | row 2j+1][4 channels], zero where kh >= kH, kw >= kW, c >= C
            const Value& ov = values_[op.output];
            const int C = op.cin, P = (op.kh + 1) / 2;
            std::vector<float> w((size_t)op.cout * P * 64, 0.f);
            for (int co = 0; co < op.cout; co++)
                for (int kh = 0; kh < op.kh; kh++)
                    for (int kw = 0; kw < op.kw; kw++)
                        for (int c = 0; c < C; c++)
                            w[((size_t)co * P + kh / 2) * 64 + kw * 8 + (kh & 1) * 4 + c] = op.weight[(((size_t)co * op.kh + kh) * op.kw + kw) * C + c];
            stem_ = true; stem_pt_ = op.pt; stem_pl_ = op.pl;
            stem_h_ = (ov.h - 1) * op.sh + 2 * (P - 1) + 1; stem_w_ = (ov.w - 1) * op.sw + 8;
            const bool sym3 = op.pt == op.pl && op.pt == op.pb && op.pt == op.pr && op.sh == op.sw && op.kh == 3 && op.kw == 3;
            st.stem_s = op.sh; st.stem_p = (sym3 && C == 3) ? op.pt : -1;   // (-1: not a form pass 3g takes)
            st.stem3_p = (sym3 && C <= 3 && op.pt <= 1) ? op.pt : -1;      // (-1: not a form pass 3j takes)
            op.weight.swap(w); op.cin = 64; op.kh = P; op.kw = 1; op.dh = 2; op.dw = 1; op.pt = op.pl = op.pb = op.pr = 0;
            st.stem_w = std::move(w);                                  // the original [Cout][kH][kW][C] weights
            st.stem = true; input_ld_ = 8; place_[input_value_].ld = 8;
        } else if (isConvLike(op.kind) && op.groups == 1 && op.kind == OpKind::Conv && rootOf(op.inputs[0]) == input_value_ && input_ld_ != op.cin) {
            // pad Cin (OHWI innermost) with zero columns up to the padded image stride
            const int cin = op.cin, cp = input_ld_; const size_t rows = op.weight.size() / cin;
            std::vector<float> w(rows * cp, 0.f);
            for (size_t r = 0; r < rows; r++) std::copy(op.weight.begin() + r * cin, op.weight.begin() + (r + 1) * cin, w.begin() + r * cp);
            op.weight.swap(w); op.cin = cp;
        }
        st.op = std::move(op);
        steps_.push_back(std::move(st));
    }

    // ---- 3a. Conv 3x3 -> MaxPool 2x2 / 2: the pooled tensor is a second output of the weight-stationary convolution kernel or of the pooling form of the nine-tap patch kernel (the
    // window is the two fragments of a lane and its neighbour lane); the full-resolution tensor is still written (skip connection).
    // (not with branchConcurrency: the side-branch selection below knows nothing of a step's second output)
    if (cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency) {
        for (size_t t = 0; t < steps_.size(); t++) {
            Step& c = steps_[t];
            if (c.op.kind != OpKind::Conv || c.stem || c.op.grid_h || c.op.residual >= 0 || c.op.output == output_value_ || !vecOk(c.op.inputs[0]) || !vecOk(c.op.output)) continue;
            const Value& ov = values_[c.op.output];
            sm_conv_desc d{}; d.N = batch; d.H = d.Ho = ov.h; d.W = d.Wo = ov.w; d.Cin = c.op.cin; d.Cout = c.op.cout; d.kH = c.op.kh; d.kW = c.op.kw; d.sH = c.op.sh; d.sW = c.op.sw;
            d.dH = c.op.dh; d.dW = c.op.dw; d.pt = c.op.pt; d.pl = c.op.pl; d.pb = c.op.pb; d.pr = c.op.pr; d.groups = c.op.groups; d.act = c.op.act; d.x_ups = c.op.x_ups ? 1 : 0;
            if (!sm_conv2d_pool_supported(&d)) continue;
            for (size_t u = t + 1; u < steps_.size(); u++) {
                Step& m = steps_[u];
                if (m.op.kind != OpKind::MaxPool || m.op.inputs[0] != c.op.output || m.op.kh != 2 || m.op.kw != 2 || m.op.sh != 2 || m.op.sw != 2 || m.op.y0 || m.op.x0 ||
                    values_[m.op.output].h * 2 != ov.h || values_[m.op.output].w * 2 != ov.w || !vecOk(m.op.output)) continue;
                c.pool_out = m.op.output;
                steps_.erase(steps_.begin() + u);
                break;
            }
        }
    }

    // ---- 3c. inverted-residual blocks (MobileNet-v2): 1x1 expand (+act) -> depthwise 3x3 / stride 1|2 / pad 1 (+act) -> 1x1 project
    // (+ the block input as residual) run as ONE launch (mbconv.hip) when the two hidden tensors have no other reader: they are 6x the block's
    // input and are otherwise written once and read once.  Domain = the all-in-LDS kernel's (Cin <= 32, Ch <= 192, Cout <= 64), where the fused
    // launch measured faster than the three separate ones; SMELTER_MBCONV=0 keeps the layers apart.
    {
        auto readers = [&](int v) { int n = v == output_value_ ? 2 : 0; for (const Step& q : steps_) { for (int u : q.op.inputs) if (u == v) n++; if (q.op.residual == v) n++; } return n; };
        auto pointwise = [&](const Step& q) { const Op& o = q.op; return o.kind == OpKind::Conv && o.kh == 1 && o.kw == 1 && o.sh == 1 && o.sw == 1 && !o.pt && !o.pl && !o.pb && !o.pr && o.groups == 1 &&
                                                     blockAct(o.act) && !q.stem && !o.grid_h && !o.x_ups && q.pool_out < 0 && !q.has_mb; };
        if (cfg_.allows(SM_FUSE_MBCONV) && cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency)
            for (size_t t = 0; t < steps_.size(); t++) {
                Step& e = steps_[t];
                if (!pointwise(e) || e.op.residual >= 0 || readers(e.op.output) != 1) continue;
                size_t ud = 0, up = 0;
                for (size_t u = t + 1; u < steps_.size() && !ud; u++) if (!steps_[u].op.inputs.empty() && steps_[u].op.inputs[0] == e.op.output) ud = u;
                if (!ud) continue;
                const Op& dw = steps_[ud].op;
                if (dw.kind != OpKind::Conv || dw.groups != dw.cin || dw.cin != dw.cout || dw.cin != e.op.cout || dw.kh != 3 || dw.kw != 3 || dw.dh != 1 || dw.dw != 1 || dw.sh != dw.sw ||
                    (dw.sh != 1 && dw.sh != 2) || dw.pt != 1 || dw.pl != 1 || dw.residual >= 0 || !blockAct(dw.act) || steps_[ud].pool_out >= 0 || readers(dw.output) != 1) continue;
                for (size_t u = ud + 1; u < steps_.size() && !up; u++) if (!steps_[u].op.inputs.empty() && steps_[u].op.inputs[0] == dw.output) up = u;
                if (!up || !pointwise(steps_[up])) continue;
                const Op& pj = steps_[up].op;
                if (pj.cin != dw.cout || (pj.residual >= 0 && pj.residual != e.op.inputs[0])) continue;
                const Value& iv = values_[e.op.inputs[0]]; const Value& ov = values_[pj.output];
                sm_mbconv_desc d{}; d.N = batch; d.H = iv.h; d.W = iv.w; d.Ho = ov.h; d.Wo = ov.w; d.stride = dw.sh; d.Cin = e.op.cin; d.Ch = e.op.cout; d.Cout = pj.cout; d.expand = 1;
                d.residual = pj.residual >= 0 ? 1 : 0; d.act1 = e.op.act; d.act2 = dw.act; d.act3 = pj.act;
                // (smelter_config.mbconv_domain = {cin, hidden, cout} widens / narrows the domain; {32, 192, 64} = the all-in-LDS kernel's only)
                const bool dom_env = cfg_.mbconvDomain[0] > 0;
                int max_cin = 96, max_ch = 576, max_cout = 160;
                if (dom_env) { max_cin = cfg_.mbconvDomain[0]; max_ch = cfg_.mbconvDomain[1]; max_cout = cfg_.mbconvDomain[2]; }
                // (the image-per-block kernel, kind 4, brings its own domain: the 14 x 14 and 7 x 7 blocks up to Ch = 960)
                if (((d.Cin > max_cin || d.Ch > max_ch || d.Cout > max_cout) && (dom_env || sm_mbconv_kernel_kind(&d) != 4)) || !vecOk(e.op.inputs[0]) || !vecOk(pj.output) || !sm_mbconv_supported(&d)) continue;
                e.has_mb = true; e.mb_dw = steps_[ud].op; e.mb_proj = steps_[up].op;
                steps_.erase(steps_.begin() + up); steps_.erase(steps_.begin() + ud);
            }
        // the block without an expand layer (MobileNet-v2's first: depthwise 3x3 on the stem output -> 1x1 project)
        if (cfg_.allows(SM_FUSE_MBCONV) && cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency)
            for (size_t t = 0; t < steps_.size(); t++) {
                Step& e = steps_[t];
                const Op& dw = e.op;
                if (e.has_mb || e.pool_out >= 0 || e.stem || dw.kind != OpKind::Conv || dw.groups != dw.cin || dw.cin != dw.cout || dw.groups < 8 || dw.kh != 3 || dw.kw != 3 || dw.dh != 1 ||
                    dw.dw != 1 || dw.sh != dw.sw || (dw.sh != 1 && dw.sh != 2) || dw.pt != 1 || dw.pl != 1 || dw.residual >= 0 || !blockAct(dw.act) || dw.grid_h || dw.x_ups || readers(dw.output) != 1) continue;
                size_t up = 0;
                for (size_t u = t + 1; u < steps_.size() && !up; u++) if (!steps_[u].op.inputs.empty() && steps_[u].op.inputs[0] == dw.output) up = u;
                if (!up || !pointwise(steps_[up]) || steps_[up].op.residual >= 0 || steps_[up].op.cin != dw.cout) continue;
                const Value& iv = values_[dw.inputs[0]]; const Value& ov = values_[steps_[up].op.output];
                sm_mbconv_desc d{}; d.N = batch; d.H = iv.h; d.W = iv.w; d.Ho = ov.h; d.Wo = ov.w; d.stride = dw.sh; d.Cin = dw.cin; d.Ch = dw.cin; d.Cout = steps_[up].op.cout; d.expand = 0;
                d.act2 = dw.act; d.act3 = steps_[up].op.act;
                if (d.Ch > 32 || d.Cout > 64 || !vecOk(dw.inputs[0]) || !vecOk(steps_[up].op.output) || !sm_mbconv_supported(&d)) continue;      // (one chunk of hidden channels: where the fused launch measured faster)
                e.has_mb = true; e.mb_noexp = true; e.mb_dw = e.op; e.mb_proj = steps_[up].op;
                steps_.erase(steps_.begin() + up);
            }
    }

    // ---- 3d. 3x3 convolution -> 1x1 head (a segmentation net's last two layers): when the 3x3 runs on the weight-stationary kernel and the 1x1 is
    // the only reader of its output, the 1x1 is one more MFMA on the output tile while it is still in registers (conv3x3_ws.hpp HEAD): the
    // full-resolution intermediate tensor is neither written nor read.  U-Net 512^2 bs=16: 145 + 110 us -> one launch.
    if (cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency) {
        auto readers = [&](int v) { int n = v == output_value_ ? 2 : 0; for (const Step& q : steps_) { for (int u : q.op.inputs) if (u == v) n++; if (q.op.residual == v) n++; } return n; };
        for (size_t t = 0; t < steps_.size(); t++) {
            Step& c = steps_[t];
            if (c.op.kind != OpKind::Conv || c.stem || c.op.grid_h || c.op.residual >= 0 || c.pool_out >= 0 || c.has_mb || c.op.kh != 3 || readers(c.op.output) != 1 || !vecOk(c.op.inputs[0])) continue;
            const Value& ov = values_[c.op.output];
            sm_conv_desc d{}; d.N = batch; d.H = d.Ho = ov.h; d.W = d.Wo = ov.w; d.Cin = c.op.cin; d.Cout = c.op.cout; d.kH = c.op.kh; d.kW = c.op.kw; d.sH = c.op.sh; d.sW = c.op.sw;
            d.dH = c.op.dh; d.dW = c.op.dw; d.pt = c.op.pt; d.pl = c.op.pl; d.pb = c.op.pb; d.pr = c.op.pr; d.groups = c.op.groups; d.act = c.op.act; d.x_ups = c.op.x_ups ? 1 : 0;
            for (size_t u = t + 1; u < steps_.size(); u++) {
                Step& h = steps_[u];
                if (h.op.inputs.empty() || h.op.inputs[0] != c.op.output) continue;
                const Op& o = h.op;
                if (o.kind != OpKind::Conv || o.kh != 1 || o.kw != 1 || o.sh != 1 || o.sw != 1 || o.pt || o.pl || o.pb || o.pr || o.groups != 1 || o.act > SM_ACT_PRELU || o.grid_h ||
                    o.x_ups || h.has_mb || h.pool_out >= 0 || o.cin != c.op.cout || !vecOk(o.output) || (o.residual >= 0 && !vecOk(o.residual))) break;
                // narrow head without residual: the HEAD instantiations; a bottleneck's tail (1x1 expand to <= 256 channels + block input + ReLU): the
                // expand head (conv3x3_ws.hpp HX), SMELTER_XHEAD=0 keeps the two launches
                const bool xhead_on = cfg_.allows(SM_FUSE_XHEAD);
                const bool narrow = o.residual < 0 && sm_conv2d_head_supported(&d, o.cout);
                if (!narrow && !(xhead_on && sm_conv2d_xhead_supported(&d, o.cout, o.residual >= 0 ? 1 : 0))) break;
                if (o.residual >= 0) {
                    // the head runs where the 3x3 runs, so its residual operand has to exist by then: a producer between the two steps (the projection
                    // shortcut of a first block) moves in front of the 3x3 when its own operands allow it
                    auto producer = [&](int v) -> long {
                        for (size_t q = 0; q < steps_.size(); q++)
                            if (steps_[q].op.output == v || steps_[q].pool_out == v || (steps_[q].has_mb && steps_[q].mb_proj.output == v) ||
                                (steps_[q].has_head && steps_[q].head.output == v)) return (long)q;
                        return -1L;
                    };
                    long pr = producer(rootOf(o.residual));
                    for (int v = 0; v < nv; v++) if (v != rootOf(o.residual) && place_[v].root == place_[rootOf(o.residual)].root && placed_in[v] >= 0) pr = std::max(pr, producer(v));   // a concat root is ready when its last member is
                    if (pr > (long)u) break;
                    if (pr > (long)t) {
                        const Step& ps = steps_[pr];
                        // (one step must be the WHOLE producer: the phase steps of a ConvTranspose and the fallback copies of a Concat share an output)
                        bool movable = ps.pool_out < 0 && !ps.has_mb && !ps.has_head && ps.op.output == rootOf(o.residual) && !ps.op.grid_h && ps.op.kind != OpKind::Concat;
                        for (size_t q = 0; q < steps_.size(); q++) if ((long)q != pr && steps_[q].op.output == ps.op.output) movable = false;
                        for (int v : ps.op.inputs) { const long q = producer(rootOf(v)); if (q >= (long)t) movable = false; }
                        if (ps.op.residual >= 0 && producer(rootOf(ps.op.residual)) >= (long)t) movable = false;
                        if (!movable) break;
                        Step moved = std::move(steps_[pr]);
                        steps_.erase(steps_.begin() + pr); steps_.insert(steps_.begin() + t, std::move(moved));
                        t++;                                           // (c and h are re-fetched below: the vector moved)
                    }
                }
                Step& c2 = steps_[t]; Step& h2 = steps_[u];
                c2.has_head = true; c2.head = std::move(h2.op);
                steps_.erase(steps_.begin() + u);
                break;
            }
        }
    }

    // ---- 3e. ... + projection shortcut: in a stage's first block the Add sits in the epilogue of the shortcut convolution (1x1 on the 64-channel block
    // input, residual = the expand layer's output).  When that expand layer is an expand head without activation and the shortcut is its only reader, the
    // shortcut becomes the head's second GEMM: y = act(w3 . h + ws . x0 + (b3 + bs)) — three layers, one launch, and the expand layer's output (102.8 MB
    // at 56^2 x 256 x bs 64, written once and read once) never exists.  SMELTER_XHEAD_PROJ=0 keeps the shortcut as its own launch.
    if (cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency && cfg_.allows(SM_FUSE_XHEAD_PROJ)) {
        auto readers = [&](int v) { int n = v == output_value_ ? 2 : 0; for (const Step& q : steps_) { for (int u : q.op.inputs) if (u == v) n++; if (q.op.residual == v) n++; } return n; };
        auto producer = [&](int v) -> long {
            for (size_t q = 0; q < steps_.size(); q++)
                if (steps_[q].op.output == v || steps_[q].pool_out == v || (steps_[q].has_mb && steps_[q].mb_proj.output == v) ||
                    (steps_[q].has_head && steps_[q].head.output == v)) return (long)q;
            return -1L;
        };
        for (size_t t = 0; t < steps_.size(); t++) {
            Step& c = steps_[t];
            if (!c.has_head || c.has_side || c.head.residual >= 0 || c.head.act != SM_ACT_NONE || readers(c.head.output) != 1) continue;
            const Value& ov = values_[c.op.output];
            sm_conv_desc d{}; d.N = batch; d.H = d.Ho = ov.h; d.W = d.Wo = ov.w; d.Cin = c.op.cin; d.Cout = c.op.cout; d.kH = c.op.kh; d.kW = c.op.kw; d.sH = c.op.sh; d.sW = c.op.sw;
            d.dH = c.op.dh; d.dW = c.op.dw; d.pt = c.op.pt; d.pl = c.op.pl; d.pb = c.op.pb; d.pr = c.op.pr; d.groups = c.op.groups; d.act = c.op.act;
            for (size_t u = t + 1; u < steps_.size(); u++) {
                Step& sd = steps_[u];
                if (sd.op.residual != c.head.output) continue;
                const Op& o = sd.op;
                if (o.kind != OpKind::Conv || o.kh != 1 || o.kw != 1 || o.sh != 1 || o.sw != 1 || o.pt || o.pl || o.pb || o.pr || o.groups != 1 || o.act > SM_ACT_PRELU || o.grid_h || o.x_ups ||
                    sd.stem || sd.has_mb || sd.has_head || sd.pool_out >= 0 || o.cin != 64 || o.cout != c.head.cout || o.inputs.empty()) break;
                const Value& sv = values_[o.inputs[0]];
                if (sv.h != ov.h || sv.w != ov.w || producer(rootOf(o.inputs[0])) >= (long)t || !vecOk(o.inputs[0]) || !vecOk(o.output) || !sm_conv2d_xhead_proj_supported(&d, o.cout)) break;
                c.has_side = true; c.side = std::move(sd.op);
                steps_.erase(steps_.begin() + u);
                break;
            }
        }
    }

    // ---- 3l. whole bottleneck block (round 6): an expand-head step (3x3 64 -> 64 + 1x1 expand + residual, pass 3d') whose input comes from a plain 1x1
    // convolution to 64 channels that nobody else reads — the block's REDUCE layer — takes that layer in as well: sm_bottleneck_fwd computes reduce -> 3x3 -> expand (+ block
    // input) per output tile with both hidden tensors on chip (csrc/kernels/bneck.hip).  Layer by layer a 56 x 56 x 256 block at bs 64 moves 360 MB (x twice, T1 out and back,
    // the output); as one pass x once (+ halo) + the residual + the output.  The hidden tensors keep their places in the rounding list.  SMELTER_BNECK=0 keeps the two launches.
    if (cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency && cfg_.allows(SM_FUSE_BNECK)) {
        auto readers = [&](int v) {
            int n = v == output_value_ ? 2 : 0;
            for (const Step& q : steps_) {
                for (int u : q.op.inputs) if (u == v) n++;
                if (q.op.residual == v) n++;
                if (q.has_head && q.head.residual == v) n++;
                if (q.has_side) for (int u : q.side.inputs) if (u == v) n++;
                if (q.dual_src == v) n++;
            }
            return n;
        };
        // (a) the 128-channel middle (ResNet-50's 28 x 28 stage; bneck128.hpp): no expand-head step exists for it — a plain 3x3 128 -> 128 whose only reader is a 1x1 expand
        // (+ residual, ReLU family) first takes that layer as its head; the reduce layer then joins below as it does for the 64-channel blocks
        for (size_t t = 0; t < steps_.size(); t++) {
            Step& c = steps_[t];
            const Op& o3 = c.op;
            if (o3.kind != OpKind::Conv || c.has_head || c.has_side || c.has_mb || c.stem || c.stem3 || c.pool_out >= 0 || c.pool3_out >= 0 || c.dual_src >= 0 || c.gather || c.kcat || c.join >= 0 || c.waits_side ||
                o3.x_ups || o3.grid_h || o3.kh != 3 || o3.kw != 3 || o3.sh != 1 || o3.sw != 1 || o3.dh != 1 || o3.dw != 1 || o3.pt != 1 || o3.pl != 1 || o3.pb != 1 || o3.pr != 1 || o3.groups != 1 ||
                o3.cin != 128 || o3.cout != 128 || o3.residual >= 0 || o3.act > SM_ACT_PRELU || readers(o3.output) != 1 || o3.inputs.empty()) continue;
            const int v = o3.inputs[0];
            long a = -1;
            for (size_t q = 0; q < t; q++) if (steps_[q].op.output == v) a = (long)q;
            if (a < 0 || readers(v) != 1 || place_[v].root != v || place_[v].c_off != 0) continue;
            const Op& o1 = steps_[a].op;
            const Step& r1 = steps_[a];
            if (o1.kind != OpKind::Conv || o1.kh != 1 || o1.kw != 1 || o1.sh != 1 || o1.sw != 1 || o1.pt || o1.pl || o1.pb || o1.pr || o1.groups != 1 || o1.act > SM_ACT_PRELU || o1.grid_h || o1.x_ups ||
                o1.residual >= 0 || r1.stem || r1.stem3 || r1.has_mb || r1.has_head || r1.has_side || r1.pool_out >= 0 || r1.pool3_out >= 0 || r1.dual_src >= 0 || r1.gather || r1.kcat || r1.join >= 0 || r1.waits_side ||
                o1.cout != 128 || o1.inputs.empty() || !vecOk(o1.inputs[0])) continue;      // (everything the second loop asks of a reduce layer: the head is only taken where the whole block follows)
            long u = -1;
            for (size_t q = t + 1; q < steps_.size(); q++) if (!steps_[q].op.inputs.empty() && steps_[q].op.inputs[0] == o3.output) { u = (long)q; break; }
            if (u < 0) continue;
            Step& h = steps_[u];
            const Op& oh = h.op;
            if (oh.kind != OpKind::Conv || oh.kh != 1 || oh.kw != 1 || oh.sh != 1 || oh.sw != 1 || oh.pt || oh.pl || oh.pb || oh.pr || oh.groups != 1 || oh.act > SM_ACT_PRELU || oh.grid_h || oh.x_ups ||
                h.stem || h.stem3 || h.has_mb || h.has_head || h.has_side || h.pool_out >= 0 || h.pool3_out >= 0 || h.dual_src >= 0 || h.gather || h.kcat || h.join >= 0 || h.waits_side ||
                oh.cin != 128 || !vecOk(oh.output) || (oh.residual >= 0 && !vecOk(oh.residual))) continue;
            if (oh.residual >= 0) {                                    // the launch runs where the 3x3 runs: the residual operand must exist by then
                bool late = false;
                for (size_t q = t; q < steps_.size(); q++) if (steps_[q].op.output == oh.residual || (steps_[q].has_head && steps_[q].head.output == oh.residual)) late = true;
                if (late) continue;
            }
            const Value& iv = values_[o1.inputs[0]]; const Value& ov = values_[oh.output];
            sm_bneck_desc bd{}; bd.N = batch; bd.H = ov.h; bd.W = ov.w; bd.C0 = o1.cin; bd.Cm = 128; bd.C3 = oh.cout;
            bd.x_ld = std::max(place_[o1.inputs[0]].ld, o1.cin); bd.y_ld = std::max(place_[oh.output].ld, oh.cout); bd.res_ld = oh.residual >= 0 ? std::max(place_[oh.residual].ld, oh.cout) : 0;
            bd.act1 = o1.act; bd.slope1 = o1.slope; bd.act2 = o3.act; bd.slope2 = o3.slope; bd.act3 = oh.act; bd.slope3 = oh.slope;
            // (images of <= 208 pixels are the persistent runs' domain, pass 3k: a stage of such blocks leaves in ONE launch there; the block kernel is for the larger ones, where a
            //  tile is a band of an image and the layers are memory-bound — ResNet-50's 28 x 28 stage)
            if (iv.h != ov.h || iv.w != ov.w || ov.h * ov.w <= 208 || !sm_bottleneck_supported(&bd)) continue;
            c.has_head = true; c.head = std::move(h.op);
            steps_.erase(steps_.begin() + u);
        }
        for (size_t t = 0; t < steps_.size(); t++) {
            Step& c = steps_[t];
            if (!c.has_head || c.has_side || c.has_reduce || c.pool_out >= 0 || c.op.x_ups || c.op.grid_h || c.op.kind != OpKind::Conv || c.op.kh != 3 || c.op.kw != 3 || c.op.cin != c.op.cout || (c.op.cout != 64 && c.op.cout != 128)) continue;
            if (c.op.act > SM_ACT_PRELU || c.head.act > SM_ACT_PRELU) continue;
            const int v = c.op.inputs[0];
            long a = -1;
            for (size_t q = 0; q < t; q++) if (steps_[q].op.output == v) a = (long)q;
            if (a < 0 || readers(v) != 1) continue;
            Step& r = steps_[a];
            const Op& o = r.op;
            if (o.kind != OpKind::Conv || o.kh != 1 || o.kw != 1 || o.sh != 1 || o.sw != 1 || o.pt || o.pl || o.pb || o.pr || o.groups != 1 || o.act > SM_ACT_PRELU || o.grid_h || o.x_ups ||
                o.residual >= 0 || r.stem || r.stem3 || r.has_mb || r.has_head || r.has_side || r.pool_out >= 0 || r.pool3_out >= 0 || r.dual_src >= 0 || r.gather || r.kcat || r.join >= 0 || r.waits_side ||
                o.cout != c.op.cout || o.inputs.empty() || !vecOk(o.inputs[0]) || place_[v].root != v || place_[v].c_off != 0) continue;
            const Value& iv = values_[o.inputs[0]]; const Value& ov = values_[c.head.output];
            if (iv.h != ov.h || iv.w != ov.w) continue;
            sm_bneck_desc bd{}; bd.N = batch; bd.H = ov.h; bd.W = ov.w; bd.C0 = o.cin; bd.Cm = c.op.cout; bd.C3 = c.head.cout;
            bd.x_ld = std::max(place_[o.inputs[0]].ld, o.cin); bd.y_ld = std::max(place_[c.head.output].ld, c.head.cout); bd.res_ld = c.head.residual >= 0 ? std::max(place_[c.head.residual].ld, c.head.cout) : 0;
            bd.act1 = o.act; bd.slope1 = o.slope; bd.act2 = c.op.act; bd.slope2 = c.op.slope; bd.act3 = c.head.act; bd.slope3 = c.head.slope;
            if (!sm_bottleneck_supported(&bd)) {
                // (a 128-channel 3x3 that took its expand layer as head above cannot run without the reduce layer: no two-layer kernel exists for it)
                if (c.op.cout == 128) { sm_set_error("executor pass 3l: a 128-channel bottleneck tail without its reduce layer (internal)"); return SM_ERR_INCONSISTENT_STATE; }
                continue;
            }
            c.has_reduce = true; c.reduce = std::move(r.op);
            steps_.erase(steps_.begin() + a);
            t--;                                                       // (the step list moved down by one: `c` now sits at t - 1, the next candidate at t)
        }
    }

    // ---- 3h. projection shortcut concatenated along K: in the first block of a later stage (ResNet stages 2-4) the expand layer E (1x1, K1 -> N, no
    // activation) is read only as the residual operand of the shortcut convolution S (1x1 / stride s on the block input, K2 -> N, + Add + ReLU).  The two
    // are ONE GEMM over K1 + K2: y = act([W_E | W_S] . [h ; x_s] + (b_E + b_S)).  The producer of h writes its rows with pixel stride K1 + K2, a
    // MaxPool 1x1 / stride s step (an exact strided copy) puts the block input's pixels beside them, and a single 1x1 convolution replaces both layers:
    // E's output (51 MB at 28^2 x 512, bs 64) is neither written nor read back, and one launch per stage goes away.  SMELTER_KCAT=0 keeps the graph.
    if (cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency && cfg_.allows(SM_FUSE_KCAT)) {
        auto readers = [&](int v) {
            int n = v == output_value_ ? 2 : 0;
            for (const Step& q : steps_) {
                for (int u : q.op.inputs) if (u == v) n++;
                if (q.op.residual == v) n++;
                if (q.has_head && q.head.residual == v) n++;
                if (q.has_side) for (int u : q.side.inputs) if (u == v) n++;
               
            }
            return n;
        };
        auto plainConv = [&](const Step& q) { return q.op.kind == OpKind::Conv && !q.stem && !q.has_mb && !q.has_head && !q.has_side && q.pool_out < 0 && !q.op.grid_h && !q.op.x_ups && q.join < 0; };
        auto pointwise = [&](const Op& o) { return o.kh == 1 && o.kw == 1 && !o.pt && !o.pl && !o.pb && !o.pr && o.groups == 1 && o.dh == 1 && o.dw == 1 && o.sh == o.sw; };
        for (size_t u = 0; u < steps_.size(); u++) {
            if (!plainConv(steps_[u])) continue;
            const Op& so = steps_[u].op;
            if (!pointwise(so) || so.residual < 0 || so.act > SM_ACT_PRELU || (so.sh != 1 && so.sh != 2) || (so.cin & 7)) continue;
            long ei = -1, pi = -1;
            for (size_t q = 0; q < u; q++) if (steps_[q].op.output == so.residual) ei = (long)q;
            if (ei < 0 || !plainConv(steps_[ei])) continue;
            const Op& eo = steps_[ei].op;
            if (!pointwise(eo) || eo.sh != 1 || eo.act != SM_ACT_NONE || eo.residual >= 0 || eo.cout != so.cout || (eo.cin & 7) || readers(eo.output) != 1) continue;
            const int h2 = eo.inputs[0], xs = so.inputs[0];
            if (h2 == input_value_ || h2 == output_value_ || alias_of[h2] >= 0 || placed_in[h2] >= 0 || place_[h2].root != h2 || place_[h2].c_off || readers(h2) != 1) continue;
            bool aliased = rootOf(output_value_) == h2; for (int v = 0; v < nv && !aliased; v++) aliased = v != h2 && (alias_of[v] == h2 || place_[v].root == h2);   // a Dropout / same-shape Reshape of h2 kept a copy of its old placement
            if (aliased) continue;
            for (size_t q = 0; q < (size_t)ei; q++) if (steps_[q].op.output == h2) pi = (long)q;
            if (pi < 0 || !plainConv(steps_[pi]) || steps_[pi].op.residual == h2) continue;
            const Value& hv = values_[h2]; const Value& ov = values_[so.output]; const Value& xv = values_[xs];
            if (hv.h != ov.h || hv.w != ov.w || (xv.h - 1) / so.sh + 1 != ov.h || (xv.w - 1) / so.sw + 1 != ov.w || hv.c != eo.cin || xv.c != so.cin) continue;
            const int K1 = eo.cin, K2 = so.cin, N = so.cout, oh = hv.h, ow = hv.w;        // (values_ grows below: no reference into it survives)
            const double x_full = (double)batch * xv.c * xv.h * xv.w;
            // Form 1 (default when both K parts are whole K tiles): the merged convolution reads its SECOND source in place — K tiles beyond K1 come from the block
            // input at pixel stride s (conv1x1_ps.hpp): no concatenated buffer, no strided copy, no extra launch.  SMELTER_KCAT=copy keeps form 2 below.
            const bool kcat_copy = cfg_.allows(SM_FUSE_KCAT_COPY);
            if (!kcat_copy && (K1 % 64) == 0 && (K2 % 64) == 0 && vecOk(xs) && vecOk(h2) && vecOk(so.output)) {
                Step& S = steps_[u]; Op& mo = S.op;
                std::vector<float> w((size_t)N * (K1 + K2));
                for (int n = 0; n < N; n++) {
                    std::copy(eo.weight.begin() + (size_t)n * K1, eo.weight.begin() + (size_t)(n + 1) * K1, w.begin() + (size_t)n * (K1 + K2));
                    std::copy(mo.weight.begin() + (size_t)n * K2, mo.weight.begin() + (size_t)(n + 1) * K2, w.begin() + (size_t)n * (K1 + K2) + K1);
                }
                if (!eo.bias.empty()) { if (mo.bias.empty()) mo.bias.assign(N, 0.f); for (int n = 0; n < N; n++) mo.bias[n] += eo.bias[n]; }
                S.dual_src = xs; S.dual_stride = mo.sh;
                mo.name = eo.name + "+" + mo.name; mo.weight.swap(w); mo.cin = K1 + K2; mo.inputs = {h2, xs}; mo.residual = -1; mo.sh = mo.sw = 1;
                S.real_k = K1 + K2; S.kcat = true;
                S.kcat_extra_elems = 2.0 * (double)S.M * N + x_full;       // E's output written and read back; S's whole input tensor (the merged step is charged h2 only)
                steps_.erase(steps_.begin() + ei);                        // ei < u
                continue;
            }
            // Form 2: two new values: the concatenated operand, and the strided copy of the block input inside it
            Value cv; cv.name = values_[eo.output].name + "/kcat"; cv.c = K1 + K2; cv.h = oh; cv.w = ow;
            Value gv; gv.name = values_[xs].name + "/strided"; gv.c = K2; gv.h = oh; gv.w = ow;
            const int cat = (int)values_.size(); values_.push_back(cv); const int gq = (int)values_.size(); values_.push_back(gv);
            Placement pc; pc.root = cat; pc.c_off = 0; pc.ld = K1 + K2; place_.push_back(pc);
            Placement pg; pg.root = cat; pg.c_off = K1; pg.ld = K1 + K2; place_.push_back(pg);
            alias_of.push_back(-1); alias_of.push_back(-1); placed_in.push_back(-1); placed_in.push_back(cat); nv += 2;
            place_[h2] = pc; placed_in[h2] = cat;
            // the merged convolution takes the shortcut's place
            Step& S = steps_[u]; Op& mo = S.op;
            std::vector<float> w((size_t)N * (K1 + K2));
            for (int n = 0; n < N; n++) {
                std::copy(eo.weight.begin() + (size_t)n * K1, eo.weight.begin() + (size_t)(n + 1) * K1, w.begin() + (size_t)n * (K1 + K2));
                std::copy(mo.weight.begin() + (size_t)n * K2, mo.weight.begin() + (size_t)(n + 1) * K2, w.begin() + (size_t)n * (K1 + K2) + K1);
            }
            if (!eo.bias.empty()) { if (mo.bias.empty()) mo.bias.assign(N, 0.f); for (int n = 0; n < N; n++) mo.bias[n] += eo.bias[n]; }
            Step G; G.op.kind = OpKind::MaxPool; G.op.name = gv.name; G.op.inputs = {xs}; G.op.output = gq; G.op.kh = G.op.kw = 1; G.op.sh = mo.sh; G.op.sw = mo.sw;
            G.M = batch * oh * ow; G.launches = 1; G.gather = true;
            mo.name = eo.name + "+" + mo.name; mo.weight.swap(w); mo.cin = K1 + K2; mo.inputs[0] = cat; mo.residual = -1; mo.sh = mo.sw = 1;
            S.real_k = K1 + K2; S.kcat = true;
            S.kcat_extra_elems = 2.0 * (double)S.M * N + (x_full - (double)S.M * K2);   // E's output written and read back; S accounted with its whole input tensor
            steps_.insert(steps_.begin() + u, std::move(G));         // (u now names the gather, u + 1 the merged convolution)
            steps_.erase(steps_.begin() + ei);                        // ei < u: the pair moves down by one
        }
    }

    // ---- 3g. stem inside the first block: a 3x3 stem on 3 planes whose only reader is a depthwise-first block of pass 3c (MobileNet-v2's first three
    // layers) becomes that block's "expand" layer — an im2col gather of the fp32 batch inside the block kernel (mbconv.hip IMC, sm_mbconv_stem_fwd): the
    // input conversion, the stem launch and the stem's output tensor (205 MB at 112^2 x 32, bs 256: written once, read once) disappear.
    // SMELTER_MBCONV_STEM=0 keeps the stem launch.
    stem_in_block_ = false;
    if (stem_ && cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency && cfg_.allows(SM_FUSE_MBCONV_STEM)) {
        auto readers = [&](int v) { int n = v == output_value_ ? 2 : 0; for (const Step& q : steps_) { for (int u : q.op.inputs) if (u == v) n++; if (q.op.residual == v) n++; } return n; };
        for (size_t t = 0; t + 1 < steps_.size(); t++) {
            Step& sc = steps_[t];
            if (!sc.stem || sc.stem_p < 0 || !blockAct(sc.op.act) || sc.op.residual >= 0 || readers(sc.op.output) != 1) continue;
            size_t ub = 0;
            for (size_t u = t + 1; u < steps_.size() && !ub; u++) if (!steps_[u].op.inputs.empty() && steps_[u].op.inputs[0] == sc.op.output) ub = u;
            if (!ub || !steps_[ub].has_mb || !steps_[ub].mb_noexp) continue;
            Step& bk = steps_[ub];
            const Value& hv = values_[sc.op.output]; const Value& pv = values_[bk.mb_proj.output]; const Value& iv = values_[input_value_];
            sm_mbconv_desc d{}; d.N = batch; d.H = hv.h; d.W = hv.w; d.Ho = pv.h; d.Wo = pv.w; d.stride = bk.mb_dw.sh; d.Cin = 27; d.Ch = sc.op.cout; d.Cout = bk.mb_proj.cout; d.expand = 1;
            d.act1 = sc.op.act; d.act2 = bk.mb_dw.act; d.act3 = bk.mb_proj.act;
            if (!vecOk(bk.mb_proj.output) || !sm_mbconv_stem_supported(&d, iv.c, iv.h, iv.w, sc.stem_s, sc.stem_p)) continue;
            bk.mb_stem = true; bk.stem_w = std::move(sc.stem_w); bk.stem_b = sc.op.bias; bk.stem_oh = hv.h; bk.stem_ow = hv.w; bk.stem_s = sc.stem_s; bk.stem_p = sc.stem_p;
            bk.stem_act = sc.op.act; bk.stem_slope = sc.op.slope; bk.stem_out = sc.op.output; bk.stem_name = sc.op.name;
            bk.op.inputs[0] = input_value_; bk.mb_dw.inputs[0] = input_value_;
            steps_.erase(steps_.begin() + t);
            stem_in_block_ = true;
            break;
        }
    }

    // ---- 3f. image stem: when a loader-wave instantiation of the stem kernel takes the layer, the input conversion happens inside its launch
    // (conv3x3_ws.hpp NLW): the row-pair image is never written; SMELTER_STEM_F32=0 keeps the conversion launch.
    stem_f32_ = false;
    if (stem_ && cfg_.fuse && !cfg_.keepIntermediates && cfg_.allows(SM_FUSE_STEM_F32))
        for (const Step& st : steps_) if (st.stem) {
            const sm_conv_desc d = convDesc(st); const Value& iv = values_[input_value_];
            stem_f32_ = sm_conv2d_stem_f32_supported(&d, iv.c, iv.h, iv.w) != 0;
        }
    if (stem_in_block_) stem_f32_ = true;                             // (no conversion launch, no row-pair image either)
    // ---- 3i. image stem -> MaxPool 3x3 / stride 2 (ResNet's first two layers): the stem's output — the largest tensor of the network, written once and read
    // once — never exists: the loader waves of the stem kernel pool the previous tile from an LDS hand-over buffer (stem_pool.hpp).  SMELTER_STEM_POOL=0
    // keeps the two launches.
    if (stem_ && !stem_in_block_ && cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency && cfg_.allows(SM_FUSE_STEM_POOL) &&
        cfg_.allows(SM_FUSE_STEM_F32)) {                     // (the fused kernel is a loader-wave stem: it reads the fp32 batch)
        auto readers = [&](int v) { int n = v == output_value_ ? 2 : 0; for (const Step& q : steps_) { for (int u : q.op.inputs) if (u == v) n++; if (q.op.residual == v) n++; } return n; };
        for (size_t t = 0; t < steps_.size(); t++) {
            Step& sc = steps_[t];
            if (!sc.stem || sc.has_mb || sc.op.residual >= 0 || sc.op.act > SM_ACT_PRELU || readers(sc.op.output) != 1) continue;
            for (size_t u = t + 1; u < steps_.size(); u++) {
                const Op& m = steps_[u].op;
                if (m.inputs.empty() || m.inputs[0] != sc.op.output) continue;
                const Value& pv = values_[m.output]; const Value& iv = values_[input_value_];
                const sm_conv_desc d = convDesc(sc);
                if (m.kind != OpKind::MaxPool || m.kh != 3 || m.kw != 3 || m.sh != 2 || m.sw != 2 || m.y0 || m.x0 || m.act != SM_ACT_NONE || !vecOk(m.output) ||
                    !sm_conv2d_stem_f32_pool_supported(&d, iv.c, iv.h, iv.w, pv.h, pv.w)) break;
                sc.pool3_out = m.output;
                steps_.erase(steps_.begin() + u);
                stem_f32_ = true;
                break;
            }
            break;
        }
    }
    // ---- 3j. 3x3 image stem straight from the fp32 batch (stem3.hip): the layer is bound by its output write, which the staged kernel does at ~80 % of the loader-wave
    // form's time (U-Net 512^2: 110 -> 68 us); taken whenever the stem is a 3x3 on <= 3 planes that neither pass 3g nor 3i fused away
    if (stem_ && !stem_in_block_ && cfg_.fuse && !cfg_.keepIntermediates && cfg_.allows(SM_FUSE_STEM_F32) && cfg_.allows(SM_FUSE_STEM3))
        for (Step& st : steps_) {
            if (!st.stem || st.has_mb || st.pool3_out >= 0 || st.pool_out >= 0 || st.has_head || st.stem3_p < 0 || st.op.residual >= 0 || !blockAct(st.op.act) || !vecOk(st.op.output)) continue;
            const Value& iv = values_[input_value_];
            if (!sm_stem3x3_f32_supported(batch, iv.c, iv.h, iv.w, st.op.cout, st.stem_s, st.stem3_p)) continue;
            st.stem3 = true; stem_f32_ = true;
        }
    if (stem_f32_) launches_ -= 1;

    // ---- 3k. persistent runs: consecutive plain convolutions (1x1 / no padding or 3x3 / padding 1, stride 1, groups 1, ReLU family, dense operands) at ONE resolution of at most
    // 208 pixels per image — the bottleneck blocks of a residual stage — become one launch of sm_stage_fwd: the blocks of an XCD keep batch / 8 images for the whole run and hand
    // tensors over through that XCD's L2 (csrc/kernels/stage.hip).  The steps stay in the list (profiling, SM_FUSE_STAGE off, a run the A/B of `autotune` turns down launch them one
    // by one); every tensor a run touches stays alive for the whole run (pass 4): inside it the images advance independently, there is no run-wide order between its phases.
    runs_.clear(); step_run_.assign(steps_.size(), -1);
    if (cfg_.fuse && !cfg_.keepIntermediates && !cfg_.branchConcurrency && cfg_.allows(SM_FUSE_STAGE)) {
        auto dense = [&](int v) { return v >= 0 && place_[v].root == v && place_[v].c_off == 0 && place_[v].ld == values_[v].c; };
        auto eligible = [&](const Step& st) {
            const Op& op = st.op;
            if (op.kind != OpKind::Conv || st.has_mb || st.has_head || st.has_side || st.stem || st.stem3 || st.pool_out >= 0 || st.pool3_out >= 0 || st.gather || op.x_ups || op.grid_h) return false;
            if (op.groups != 1 || op.sh != 1 || op.sw != 1 || op.dh != 1 || op.dw != 1 || op.act > SM_ACT_PRELU || (op.cout & 7)) return false;
            const bool t1 = op.kh == 1 && op.kw == 1 && !op.pt && !op.pl && !op.pb && !op.pr, t3 = op.kh == 3 && op.kw == 3 && op.pt == 1 && op.pl == 1 && op.pb == 1 && op.pr == 1;
            if (!t1 && !t3) return false;
            const Value& iv = values_[op.inputs[0]]; const Value& ov = values_[op.output];
            if (iv.h != ov.h || iv.w != ov.w || ov.h * ov.w > 208 || !dense(op.inputs[0]) || !dense(op.output) || (op.residual >= 0 && !dense(op.residual))) return false;
            if (st.dual_src >= 0 && (!t1 || !dense(st.dual_src))) return false;
            return true;
        };
        for (size_t t = 0; t < steps_.size();) {
            if (!eligible(steps_[t])) { t++; continue; }
            const Value& v0 = values_[steps_[t].op.output];
            size_t e = t + 1;
            while (e < steps_.size() && e - t < SM_STAGE_MAX_PHASES && eligible(steps_[e]) && values_[steps_[e].op.output].h == v0.h && values_[steps_[e].op.output].w == v0.w) e++;
            // kernel-level check on a synthetic address map (every tensor of the run in a range of its own; the real map comes out of pass 4 with the same property).
            // when the kernel turns the window down, its PREFIXES are tried (the eligible phases in front of one the kernel rejects still form a run), longest first.
            const bool wide = v0.h * v0.w > 112;
            bool ok = false; StageRun r;
            if (!wide || cfg_.allows(SM_FUSE_STAGE_WIDE)) {
                const int keep = batch_; batch_ = batch;
                std::vector<size_t> save(root_offset_); root_offset_.assign(nv, 0);
                size_t o = 0; for (int v = 0; v < nv; v++) { root_offset_[v] = o; o += alignUp((size_t)batch * values_[v].h * values_[v].w * std::max(place_[v].ld, 8) * 2 + 64); }
                size_t wo = 0; for (size_t q = t; q < e; q++) { steps_[q].w_off = wo; wo += alignUp(steps_[q].op.weight.size() * 2); steps_[q].b_off = wo; wo += alignUp(steps_[q].op.bias.size() * 4 + 64); steps_[q].has_bias = !steps_[q].op.bias.empty(); }
                // (a PREFIX must hold at least three layers: a two-layer run saves one launch gap and pays the run's arrival + hand-off protocol for it; the whole window may be two)
                for (size_t e2 = e; e2 >= t + 2 && (e2 == e || e2 >= t + 3) && !ok; e2--) {
                    r = StageRun(); r.first = (int)t; r.count = (int)(e2 - t); r.wide = wide;
                    const std::vector<sm_stage_phase> ph = runPhases(r, (const char*)(uintptr_t)0x100000000ull, (const char*)(uintptr_t)0x4000000000ull, false);
                    ok = sm_stage_supported(ph.data(), (int)ph.size()) != 0;
                    if (ok) e = e2;
                }
                root_offset_ = save; batch_ = keep;
            }
            if (ok) { for (size_t q = t; q < e; q++) step_run_[q] = (int)runs_.size(); runs_.push_back(r); }
            t = ok ? e : t + 1;
        }
    }

    // ---- 4. memory plan ------------------------------------------------------------------------------------
    const int ns = (int)steps_.size();
    std::vector<int> first(nv, INT32_MAX), last(nv, -1);   // per ROOT value: live step interval
    auto touch = [&](int v, int t) { int r = place_[v].root; first[r] = std::min(first[r], t); last[r] = std::max(last[r], t); };
    // ---- side branch: a convolution whose result is read ONLY as the fused residual operand of one later step (ResNet's
    // projection shortcut: conv1x1 on the block input, consumed three launches later) has no dependency on the steps in
    // between -> it runs on a second stream, forked before it and joined before its consumer; its tiles fill the CUs the
    // main branch's small-M layers leave idle.  At most one side step is in flight at a time.
    if (cfg_.branchConcurrency) {
        int busy_until = -1;
        for (int t = 0; t + 2 < ns; t++) {
            Step& st = steps_[t];
            if (t <= busy_until || !isConvLike(st.op.kind) || st.op.kind == OpKind::ConvTranspose || st.op.grid_h || st.op.output == output_value_) continue;
            const int r = place_[st.op.output].root; if (r != st.op.output || place_[st.op.output].c_off) continue;
            int users = 0, join = -1; bool other_use = false;
            for (int j = 0; j < ns; j++) {
                if (j == t) continue;
                const Op& o = steps_[j].op;
                for (int u : o.inputs) if (place_[u].root == r) other_use = true;
                if (place_[o.output].root == r) other_use = true;
                if (o.residual >= 0 && place_[o.residual].root == r) { users++; join = j; }
            }
            if (getenv("SMELTER_DEBUG_PLAN") && users) fprintf(stderr, "[plan] step %d %s: residual users %d, join %d, other_use %d\n", t, st.op.name.c_str(), users, join, (int)other_use);
            if (other_use || users != 1 || join < t + 2) continue;
            st.join = join; steps_[join].waits_side = true; busy_until = join;
        }
    }
    touch(input_value_, -1);
    for (int t = 0; t < ns; t++) {
        const Op& op = steps_[t].op;
        const int until = steps_[t].join >= 0 ? steps_[t].join : t;   // a side step may still be reading its operands at the join
        if (!steps_[t].has_reduce) for (int u : op.inputs) { touch(u, t); touch(u, until); }   // (a whole-block step reads the REDUCE layer's input — touched below; its own input tensor exists in LDS only)
        if (op.residual >= 0) { touch(op.residual, t); touch(op.residual, until); }
        if (steps_[t].pool3_out >= 0) { touch(steps_[t].pool3_out, t); continue; }             // (the stem's own output exists in LDS only)
        if (steps_[t].has_mb) { touch(steps_[t].mb_proj.output, t); continue; }
        if (steps_[t].has_side) { touch(steps_[t].side.output, t); touch(steps_[t].side.inputs[0], t); continue; }                    // (neither the 3x3 nor the expand output exists)
        if (steps_[t].has_reduce) touch(steps_[t].reduce.inputs[0], t);                      // (the block input is what the launch reads; the reduce output exists in LDS only)
        if (steps_[t].has_head) { touch(steps_[t].head.output, t); if (steps_[t].head.residual >= 0) touch(steps_[t].head.residual, t); continue; }             // (the 3x3 output exists in registers only)      // (the hidden tensors exist in LDS only; the residual is the step's own input)
        touch(op.output, t);
        if (steps_[t].pool_out >= 0) touch(steps_[t].pool_out, t);
    }
    for (const StageRun& r : runs_)                          // a run's tensors: all alive from its first to its last step
        for (int t = r.first; t < r.first + r.count; t++) {
            const Op& op = steps_[t].op;
            auto span = [&](int v) { if (v >= 0) { touch(v, r.first); touch(v, r.first + r.count - 1); } };
            for (int u : op.inputs) span(u);
            span(op.residual); span(steps_[t].dual_src); span(op.output);
        }
    touch(output_value_, ns);                               // result image must survive until read back
    if (cfg_.keepIntermediates) for (int v = 0; v < nv; v++) if (last[v] >= 0) { first[v] = -1; last[v] = ns; }
    // storage = the value's root buffer got a lifetime; the image input of a stem launch is never stored as an NHWC fp16 tensor (row-pair image, or nothing at all)
    stored_.assign(nv, 0);
    for (int v = 0; v < nv; v++) stored_[v] = last[place_[v].root] >= 0 && !(stem_ && v == input_value_);
    root_offset_.assign(nv, 0);
    struct Blk { int v; size_t bytes; };
    std::vector<Blk> blks;
    for (int v = 0; v < nv; v++) if (place_[v].root == v && last[v] >= 0) {
        size_t bytes = alignUp((size_t)batch * values_[v].h * values_[v].w * place_[v].ld * 2 + 64);
        if (stem_ && v == input_value_) bytes = stem_f32_ ? alignUp(64) : alignUp((size_t)batch * stem_h_ * stem_w_ * 16 + 64);
        blks.push_back({v, bytes});
    }
    std::sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.bytes > b.bytes; });
    std::vector<Blk> placed; act_bytes_ = 0;
    for (auto& b : blks) {
        std::vector<std::pair<size_t, size_t>> busy;        // byte ranges of already placed, time-overlapping buffers
        for (auto& q : placed) if (!(last[q.v] < first[b.v] || last[b.v] < first[q.v])) busy.push_back({root_offset_[q.v], root_offset_[q.v] + q.bytes});
        std::sort(busy.begin(), busy.end());
        size_t off = 0;
        for (auto& r : busy) { if (off + b.bytes <= r.first) break; off = std::max(off, r.second); }
        root_offset_[b.v] = off; placed.push_back(b); act_bytes_ = std::max(act_bytes_, off + b.bytes);
        if (getenv("SMELTER_DEBUG_PLAN")) fprintf(stderr, "[plan] buffer %-28s %9.2f MB  alive steps %3d .. %3d  at %9.2f MB\n", values_[b.v].name.c_str(), b.bytes / 1e6, first[b.v], last[b.v], off / 1e6);
    }

    // ---- 5. weight arena layout + accounting ---------------------------------------------------------------
    size_t off = 0; ws_bytes_ = 0;
    for (auto& st : steps_) {
        Op& op = st.op;
        if (isConvLike(op.kind)) {
            st.w_off = off; off = alignUp(off + op.weight.size() * 2);
            st.has_bias = !op.bias.empty();
            if (st.has_bias) { st.b_off = off; off = alignUp(off + op.bias.size() * 4); }
            if (step_run_[&st - steps_.data()] >= 0) { st.has_wp = true; st.wp_off = off; off = alignUp(off + (size_t)((op.cout + 255) / 256 * 256) * op.weight.size() / op.cout * 2); }   // fragment-order copy for the run (couts padded to the widest cout tile)
            if (st.stem3) { st.ws_off = off; off = alignUp(off + st.stem_w.size() * 2); }       // the original [Cout][3][3][C] weights (the row-pair packing stays at w_off: unused by this launch)
            if (st.has_mb && st.mb_stem) {
                st.ws_off = off; off = alignUp(off + st.stem_w.size() * 2);
                if (!st.stem_b.empty()) { st.bs_off = off; off = alignUp(off + st.stem_b.size() * 4); }
                const double Ms = (double)batch * st.stem_oh * st.stem_ow; const Value& iv = values_[input_value_];
                flops_ += 2.0 * Ms * (double)st.stem_w.size();                                  // the stem layer of the graph: M x Cout x (kH kW C)
                alg_bytes_ += 2.0 * ((double)batch * iv.c * iv.h * iv.w + Ms * (double)(st.stem_w.size() / 27) + (double)st.stem_w.size());
                // (the depthwise layer below is accounted with this step's input, which is now the image: charge it the stem's output it reads in the graph)
                alg_bytes_ += 2.0 * (Ms * (double)(st.stem_w.size() / 27) - (double)batch * iv.c * iv.h * iv.w);
            }
            if (st.has_mb) {
                st.wd_off = off; off = alignUp(off + st.mb_dw.weight.size() * 2);
                st.has_biasd = !st.mb_dw.bias.empty();
                if (st.has_biasd) { st.bd_off = off; off = alignUp(off + st.mb_dw.bias.size() * 4); }
                st.w2_off = off; off = alignUp(off + st.mb_proj.weight.size() * 2);
                st.has_bias2 = !st.mb_proj.bias.empty();
                if (st.has_bias2) { st.b2_off = off; off = alignUp(off + st.mb_proj.bias.size() * 4); }
                // accounted as the three layers the graph holds (flops and algorithmic bytes do not depend on the fusion)
                const Value& dv = values_[st.mb_dw.output]; const Value& pv = values_[st.mb_proj.output];
                const double Md = (double)batch * dv.h * dv.w;
                flops_ += (st.mb_noexp ? 0.0 : 2.0 * Md * dv.c * 9.0) + 2.0 * Md * pv.c * (double)dv.c;         // (no expand layer: the depthwise layer is this step's own op, counted below)
                alg_bytes_ += (st.mb_noexp ? 0.0 : 2.0 * ((double)st.M * op.cout + Md * dv.c + dv.c * 9.0)) + 2.0 * (Md * dv.c + Md * pv.c * (st.mb_proj.residual >= 0 ? 2.0 : 1.0) + (double)pv.c * dv.c);
            }
            if (st.has_side) {
                st.wd_off = off; off = alignUp(off + st.side.weight.size() * 2);
                flops_ += 2.0 * st.M * st.side.cout * (double)st.side.cin;
                alg_bytes_ += 2.0 * ((double)st.M * st.side.cin + (double)st.M * st.side.cout * 2 + (double)st.side.cout * st.side.cin);      // (the shortcut layer of the graph: input, residual, output, weights)
            }
            if (st.has_reduce) {
                st.w0_off = off; off = alignUp(off + st.reduce.weight.size() * 2);
                st.has_bias0 = !st.reduce.bias.empty();
                if (st.has_bias0) { st.b0_off = off; off = alignUp(off + (size_t)st.reduce.cout * 4); }
                flops_ += 2.0 * st.M * st.reduce.cout * (double)st.reduce.cin;
                alg_bytes_ += 2.0 * ((double)st.M * st.reduce.cin + (double)st.M * st.reduce.cout + (double)st.reduce.cout * st.reduce.cin);      // (accounted as the layer the graph holds)
            }
            if (st.has_head) {
                st.w2_off = off; off = alignUp(off + st.head.weight.size() * 2);
                st.has_bias2 = !st.head.bias.empty() || (st.has_side && !st.side.bias.empty());
                if (st.has_bias2) { st.b2_off = off; off = alignUp(off + (size_t)st.head.cout * 4); }
                flops_ += 2.0 * st.M * st.head.cout * (double)st.head.cin;
                alg_bytes_ += 2.0 * ((double)st.M * st.head.cin + (double)st.M * st.head.cout * (st.head.residual >= 0 ? 2 : 1) + (double)st.head.cout * st.head.cin);      // (accounted as the layer the graph holds)
            }
            const Value& ov = values_[op.output];
            const double K = (double)st.real_k;
            flops_ += 2.0 * st.M * op.cout * K;
            const Value& iv = values_[op.inputs[0]];
            alg_bytes_ += 2.0 * ((double)batch * iv.c * iv.h * iv.w + (double)st.M * ov.c + (op.residual >= 0 ? (double)st.M * ov.c : 0.0) + (double)op.cout * K);
            sm_conv_desc d{}; d.N = batch; d.H = op.kind == OpKind::Conv ? iv.h : 1; d.W = op.kind == OpKind::Conv ? iv.w : 1; d.Cin = op.cin; d.Cout = op.cout;
            d.kH = op.kh; d.kW = op.kw; d.groups = op.groups; d.Ho = ov.h; d.Wo = ov.w;
            ws_bytes_ = std::max(ws_bytes_, sm_conv2d_workspace_bytes(&d));
        } else if (op.kind == OpKind::BatchNorm || op.kind == OpKind::GroupNorm) {
            st.s_off = off; off = alignUp(off + op.scale.size() * 4);
            st.t_off = off; off = alignUp(off + op.shift.size() * 4);
        }
        if (st.kcat) alg_bytes_ += 2.0 * st.kcat_extra_elems;                                 // (accounted as the two layers the graph holds)
        if (!isConvLike(op.kind) && !st.gather) {
            const Value& ov = values_[op.output]; double in_elems = 0;
            for (int u : op.inputs) in_elems += (double)batch * values_[u].c * values_[u].h * values_[u].w;
            alg_bytes_ += 2.0 * (in_elems + (double)batch * (op.kind == OpKind::Concat ? values_[op.inputs[0]].c : ov.c) * ov.h * ov.w);
        }
        launches_ += st.launches;
        char buf[256];
        if (isConvLike(op.kind))
            snprintf(buf, sizeof buf, "%s %s M=%d N=%d K=%d k=%dx%d s=%d g=%d%s%s%s", kindName(op.kind), op.name.c_str(), st.M, op.cout, op.kh * op.kw * (op.cin / op.groups),
                     st.real_kh, st.real_kw, op.sh, op.groups, op.residual >= 0 ? " +res" : "", op.act ? " +act" : "", st.stem3 ? " stem (fp32 image, staged 3x3)" : st.stem ? " stem" : op.grid_h ? " convT-phase" : op.x_ups ? " +ups2x" : "");
        else snprintf(buf, sizeof buf, "%s %s pixels=%d C=%d%s", kindName(op.kind), op.name.c_str(), st.M, values_[op.output].c, op.act ? " +act" : "");
        st.desc = buf;
        if (st.pool3_out >= 0) { const Value& pv = values_[st.pool3_out]; st.desc += " >> MaxPool 3x3/2 (same launch)";
                                 alg_bytes_ += 2.0 * ((double)st.M * pv.c + (double)batch * pv.h * pv.w * pv.c); }       // (the pooling layer of the graph: input read, output written)
        if (st.pool_out >= 0) { st.desc += " >> MaxPool 2x2/2 (same launch)"; alg_bytes_ += 2.0 * (double)(st.M / 4) * values_[st.pool_out].c; }
        if (st.has_reduce) { snprintf(buf, sizeof buf, "Conv %s N=%d K=%d k=1x1%s (same launch: whole bottleneck block) >> ", st.reduce.name.c_str(), st.reduce.cout, st.reduce.cin, st.reduce.act ? " +act" : ""); st.desc = buf + st.desc; }
        if (st.has_head) { snprintf(buf, sizeof buf, " >> Conv %s N=%d K=%d k=1x1%s%s (same launch)", st.head.name.c_str(), st.head.cout, st.head.cin, st.head.residual >= 0 ? " +res" : "", st.head.act ? " +act" : ""); st.desc += buf; }
        if (st.has_side) { snprintf(buf, sizeof buf, " + Conv %s N=%d K=%d k=1x1 projection%s (same launch)", st.side.name.c_str(), st.side.cout, st.side.cin, st.side.act ? " +act" : ""); st.desc += buf; }
        if (st.has_mb && st.mb_noexp) { snprintf(buf, sizeof buf, " >> Conv %s N=%d K=%d k=1x1%s (same launch)", st.mb_proj.name.c_str(), st.mb_proj.cout, st.mb_proj.cin, st.mb_proj.act ? " +act" : ""); st.desc += buf; }
        if (st.has_mb && st.mb_stem) st.desc = "Conv " + st.stem_name + " k=3x3 stem (fp32 image, same launch) >> " + st.desc;
        else if (st.has_mb) { snprintf(buf, sizeof buf, " >> Conv %s k=3x3 s=%d g=%d%s >> Conv %s N=%d K=%d k=1x1%s%s (same launch)", st.mb_dw.name.c_str(), st.mb_dw.sh, st.mb_dw.groups, st.mb_dw.act ? " +act" : "",
                                  st.mb_proj.name.c_str(), st.mb_proj.cout, st.mb_proj.cin, st.mb_proj.residual >= 0 ? " +res" : "", st.mb_proj.act ? " +act" : ""); st.desc += buf; }
        if (st.dual_src >= 0) { snprintf(buf, sizeof buf, " [K = %d + %d: second source %s read in place, pixel stride %d]", op.cin - values_[st.dual_src].c, values_[st.dual_src].c, values_[st.dual_src].name.c_str(), st.dual_stride); st.desc += buf; }
        if (st.join >= 0) st.desc += " side-stream";
    }
    arena_bytes_ = std::max(off, (size_t)kAlign);
    // ---- post-fusion compulsory bytes: per launch, every distinct tensor it reads + every tensor it writes + its weights, once each.  Tensors that exist in
    // LDS / registers only (the hidden tensors of a fused block, a head's 3x3 output, the stem's output under the fused pool, E's output under pass 3h) do
    // not appear.  This is the byte model a counter-measured HBM traffic (rocprofv3 FETCH_SIZE + WRITE_SIZE) is to be read against; `alg_bytes_` above
    // charges every layer of the GRAPH its own operands and therefore exceeds it wherever layers were fused.
    {
        auto elems = [&](int v) { return (double)batch * values_[v].c * values_[v].h * values_[v].w; };
        const Value& inv = values_[input_value_];
        launch_bytes_ = stem_f32_ ? 0.0 : (4.0 + 2.0 * (stem_ ? 8.0 / inv.c : (double)input_ld_ / inv.c)) * elems(input_value_);     // conversion launch: fp32 in, fp16 image out
        for (auto& st : steps_) {
            const Op& op = st.op;
            std::vector<int> rd, wr;
            auto add = [](std::vector<int>& a, int v) { if (v >= 0 && std::find(a.begin(), a.end(), v) == a.end()) a.push_back(v); };
            if (st.has_reduce) add(rd, st.reduce.inputs[0]); else for (int u : op.inputs) add(rd, u);
            add(rd, op.residual);
            if (st.has_head) add(rd, st.head.residual);
            if (st.has_side) for (int u : st.side.inputs) add(rd, u);
            if (st.has_mb && st.mb_proj.residual >= 0) add(rd, st.mb_proj.residual);
            if (st.has_side) add(wr, st.side.output);
            else if (st.has_head) add(wr, st.head.output);
            else if (st.has_mb) add(wr, st.mb_proj.output);
            else if (st.pool3_out >= 0) add(wr, st.pool3_out);
            else { add(wr, op.output); add(wr, st.pool_out); }
            double b = 0;
            for (int v : rd) b += (v == input_value_ && (stem_f32_ || st.mb_stem)) ? 4.0 * elems(v) : 2.0 * elems(v);      // (the fp32 batch is read directly)
            for (int v : wr) b += 2.0 * ((op.kind == OpKind::Concat) ? elems(op.inputs[0]) : elems(v));
            if (op.x_ups) b -= 0.0;                                                                                   // (the half-resolution tensor is op.inputs[0]: already its true size)
            double wb = 2.0 * (double)op.weight.size() + 4.0 * (double)op.bias.size();
            if (st.has_head) wb += 2.0 * (double)st.head.weight.size();
            if (st.has_reduce) wb += 2.0 * (double)st.reduce.weight.size() + 4.0 * (double)st.reduce.bias.size();
            if (st.has_side) wb += 2.0 * (double)st.side.weight.size();
            if (st.has_mb) wb += 2.0 * ((double)st.mb_dw.weight.size() + (double)st.mb_proj.weight.size()) + (st.mb_stem ? 2.0 * (double)st.stem_w.size() : 0.0);
            launch_bytes_ += b + wb;
        }
    }
    value_names_.resize(nv); for (int v = 0; v < nv; v++) value_names_[v] = values_[v].name;
    materialized_.clear(); materialized_.push_back(values_[input_value_].name);
    for (auto& st : steps_) {
        if (st.has_side) { materialized_.push_back(values_[st.op.output].name); materialized_.push_back(values_[st.side.output].name); continue; }   // (the expand output is summed in fp32, never rounded)
        if (st.has_reduce) materialized_.push_back(values_[st.reduce.output].name);                                                          // (rounded to fp16 on its way into the LDS patch)
        if (st.has_head) { materialized_.push_back(values_[st.op.output].name); materialized_.push_back(values_[st.head.output].name); continue; }   // (rounded to fp16 in registers)
        if (st.has_mb) {   // (the two hidden tensors live in LDS only, but are rounded to fp16 there exactly as a stored tensor would be: listed for the parity oracle)
            if (!st.mb_noexp) materialized_.push_back(values_[st.op.output].name);
            if (st.mb_stem) materialized_.push_back(values_[st.stem_out].name);                   // (the stem's output: the block's first hidden tensor)
            materialized_.push_back(values_[st.mb_dw.output].name); materialized_.push_back(values_[st.mb_proj.output].name); continue;
        }
        if (st.op.kind != OpKind::Concat && materialized_.back() != values_[st.op.output].name) materialized_.push_back(values_[st.op.output].name);
        if (st.pool_out >= 0) materialized_.push_back(values_[st.pool_out].name);
        if (st.pool3_out >= 0) materialized_.push_back(values_[st.pool3_out].name);      // (the stem's output is rounded to fp16 in LDS exactly as a stored tensor would be: it stays in the list above)
    }
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------------------
int Executor::packWeights(std::vector<uint8_t>* host) const {
    host->assign(arena_bytes_, 0);
    for (auto& st : steps_) {
        const Op& op = st.op;
        if (isConvLike(op.kind)) {
            uint16_t* dst = (uint16_t*)(host->data() + st.w_off);
            if (op.kind == OpKind::FullyConnected && st.fc_h * st.fc_w > 1) {
                // columns arrive in NCHW-flatten order (c*HW + p); our buffer is NHWC (p*C + c)
                const int hw = st.fc_h * st.fc_w, C = op.cin / hw;
                for (int o = 0; o < op.cout; o++)
                    for (int p = 0; p < hw; p++)
                        for (int c = 0; c < C; c++) dst[(size_t)o * op.cin + (size_t)p * C + c] = float_to_half_bits(op.weight[(size_t)o * op.cin + (size_t)c * hw + p]);
            } else {
                sm_float_to_half(op.weight.data(), dst, op.weight.size());   // fp16 cast, RNE (ONNX2MPS.py:27 numpy astype(float16))
            }
            if (st.has_bias) memcpy(host->data() + st.b_off, op.bias.data(), op.bias.size() * 4);
            if (st.stem3) sm_float_to_half(st.stem_w.data(), (uint16_t*)(host->data() + st.ws_off), st.stem_w.size());
            if (st.has_mb && st.mb_stem) {
                sm_float_to_half(st.stem_w.data(), (uint16_t*)(host->data() + st.ws_off), st.stem_w.size());
                if (!st.stem_b.empty()) memcpy(host->data() + st.bs_off, st.stem_b.data(), st.stem_b.size() * 4);
            }
            if (st.has_mb) {
                sm_float_to_half(st.mb_dw.weight.data(), (uint16_t*)(host->data() + st.wd_off), st.mb_dw.weight.size());
                if (st.has_biasd) memcpy(host->data() + st.bd_off, st.mb_dw.bias.data(), st.mb_dw.bias.size() * 4);
                sm_float_to_half(st.mb_proj.weight.data(), (uint16_t*)(host->data() + st.w2_off), st.mb_proj.weight.size());
                if (st.has_bias2) memcpy(host->data() + st.b2_off, st.mb_proj.bias.data(), st.mb_proj.bias.size() * 4);
            }
            if (st.has_reduce) {
                sm_float_to_half(st.reduce.weight.data(), (uint16_t*)(host->data() + st.w0_off), st.reduce.weight.size());
                if (st.has_bias0) memcpy(host->data() + st.b0_off, st.reduce.bias.data(), st.reduce.bias.size() * 4);
            }
            if (st.has_head) {
                sm_float_to_half(st.head.weight.data(), (uint16_t*)(host->data() + st.w2_off), st.head.weight.size());
                if (st.has_bias2) {
                    float* b2 = (float*)(host->data() + st.b2_off);
                    for (int i = 0; i < st.head.cout; i++) b2[i] = (st.head.bias.empty() ? 0.f : st.head.bias[i]) + ((st.has_side && !st.side.bias.empty()) ? st.side.bias[i] : 0.f);
                }
                if (st.has_side) sm_float_to_half(st.side.weight.data(), (uint16_t*)(host->data() + st.wd_off), st.side.weight.size());
            }
        } else if (op.kind == OpKind::BatchNorm || op.kind == OpKind::GroupNorm) {
            memcpy(host->data() + st.s_off, op.scale.data(), op.scale.size() * 4);
            memcpy(host->data() + st.t_off, op.shift.data(), op.shift.size() * 4);
        }
    }
    // the runs' weights once more, in the fragment order of the tile form the kernel-level planner gives each phase (the same deterministic choice it makes at launch:
    // it depends on the shapes and the batch only — the addresses here are stand-ins with the real layout's offsets)
    for (const StageRun& r : runs_) {
        const std::vector<sm_stage_phase> ph = runPhases(r, (const char*)(uintptr_t)0x100000000ull, (const char*)(uintptr_t)0x4000000000ull, false);
        for (int i = 0; i < r.count; i++) {
            const Step& st = steps_[r.first + i];
            const size_t need = sm_stage_packed_weight_bytes(ph.data(), r.count, i);
            if (!need || st.wp_off + need > host->size()) { sm_set_error("persistent run at step %d: packed weights of phase %d need %zu bytes", r.first, i, need); return SM_ERR_INCONSISTENT_STATE; }
            const int rc = sm_stage_pack_weights(ph.data(), r.count, i, host->data() + st.w_off, host->data() + st.wp_off);
            if (rc) return rc;
        }
    }
    return SM_OK;
}

int Executor::weightBytes(ONNXGraph& g, int batch, size_t* bytes) {
    int rc = g.build(); if (rc) return rc;
    Executor e; rc = e.plan(g, batch); if (rc) return rc;
    *bytes = e.arena_bytes_; return SM_OK;
}

int Executor::packWeightsHost(ONNXGraph& g, int batch, void* dst, size_t cap) {
    int rc = g.build(); if (rc) return rc;
    Executor e; rc = e.plan(g, batch); if (rc) return rc;
    if (cap < e.arena_bytes_) { sm_set_error("smelter_graph_pack_weights: buffer holds %zu bytes, arena needs %zu", cap, e.arena_bytes_); return SM_ERR_INVALID_ARGUMENT; }
    std::vector<uint8_t> host; rc = e.packWeights(&host); if (rc) return rc;
    memcpy(dst, host.data(), host.size());
    return SM_OK;
}

static const int32_t kPlanMagic = 0x4e414c50;   // "PLAN"
// fingerprint of the step list a plan was made for: two compiles of one graph with different fusion masks, batches or library builds can agree in the NUMBER of steps
static int32_t stepsFingerprint(const std::vector<Step>& steps) {
    uint32_t h = 2166136261u;
    auto mix = [&](uint32_t v) { h = (h ^ v) * 16777619u; };
    for (const Step& st : steps) {
        mix((uint32_t)st.op.kind); mix((uint32_t)st.M); mix((uint32_t)st.op.cout); mix((uint32_t)st.op.cin); mix((uint32_t)(st.op.kh * 16 + st.op.kw)); mix((uint32_t)(st.op.sh * 16 + st.op.groups));
        mix((uint32_t)((st.has_reduce ? 0x10000 : 0) | (st.has_mb ? 1 : 0) | (st.has_head ? 2 : 0) | (st.has_side ? 4 : 0) | (st.stem ? 8 : 0) | (st.stem3 ? 16 : 0) | (st.dual_src >= 0 ? 32 : 0) | (st.pool_out >= 0 ? 64 : 0) | (st.pool3_out >= 0 ? 128 : 0) | (st.op.residual >= 0 ? 256 : 0)));
    }
    return (int32_t)h;
}
// [0] 'PLAN', [1] tile configurations in this build, [2] steps S, [3 .. 3 + S) conv_cfg per step, then R = persistent runs, R x (first step, steps, 1 = launched as a run), then the
// fingerprint of the step list
std::vector<int32_t> Executor::exportPlan() const {
    std::vector<int32_t> p{kPlanMagic, (int32_t)sm_conv2d_num_configs(), (int32_t)steps_.size()};
    for (const Step& st : steps_) p.push_back(st.conv_cfg);
    p.push_back((int32_t)runs_.size());
    for (const StageRun& r : runs_) { p.push_back(r.first); p.push_back(r.count); p.push_back(r.on ? 1 : 0); }
    p.push_back(stepsFingerprint(steps_));
    return p;
}
int Executor::importPlan(const int32_t* plan, size_t words) {
    if (!plan || words < 3 || plan[0] != kPlanMagic) { sm_set_error("execution plan: not a plan (bad header)"); return SM_ERR_INVALID_ARGUMENT; }
    if (plan[1] != sm_conv2d_num_configs()) { sm_set_error("execution plan: made by a library with %d tile configurations, this one has %d", plan[1], sm_conv2d_num_configs()); return SM_ERR_INVALID_ARGUMENT; }
    const size_t nst = steps_.size(), want = 3 + nst + 1 + 3 * runs_.size() + 1;
    if ((size_t)plan[2] != nst || words != want) {
        sm_set_error("execution plan: %d steps (%zu words), this graph / batch compiles to %zu steps and %zu persistent runs (%zu words)", plan[2], words, nst, runs_.size(), want); return SM_ERR_INVALID_ARGUMENT; }
    if (plan[words - 1] != stepsFingerprint(steps_)) { sm_set_error("execution plan: made for another step list (fusion mask, batch or graph differ)"); return SM_ERR_INVALID_ARGUMENT; }
    const int32_t* q = plan + 3 + nst;
    if ((size_t)q[0] != runs_.size()) { sm_set_error("execution plan: %d persistent runs, this compile has %zu", q[0], runs_.size()); return SM_ERR_INVALID_ARGUMENT; }
    for (size_t i = 0; i < runs_.size(); i++)
        if (q[1 + 3 * i] != runs_[i].first || q[2 + 3 * i] != runs_[i].count) { sm_set_error("execution plan: persistent run %zu covers other steps", i); return SM_ERR_INVALID_ARGUMENT; }
    for (size_t i = 0; i < nst; i++) {
        const int c = plan[3 + i];
        if (c < -1 || c >= sm_conv2d_num_configs()) { sm_set_error("execution plan: step %zu names tile configuration %d", i, c); return SM_ERR_INVALID_ARGUMENT; }
        steps_[i].conv_cfg = c;
        if (c >= 0) steps_[i].desc += " cfg=" + std::to_string(c);
    }
    // (a rank on a smaller / partitioned device must not take rank 0's "on" — it would run every stage in the slower claimed mode on every step)
    const bool fast = sm_stage_device_fast() != 0;
    for (size_t i = 0; i < runs_.size(); i++) {
        const bool want = q[3 + 3 * i] != 0;
        runs_[i].on = want && fast; runs_[i].timed = false;
        runs_[i].why_off = runs_[i].on ? 0 : (want ? SM_RUN_OFF_DEVICE : SM_RUN_OFF_PLAN);
    }
    annotateRuns();
    return SM_OK;
}
// the step descriptions say which steps leave in one launch (the same text whether the decision was measured here, imported with a plan or is the default)
void Executor::annotateRuns() {
    for (const StageRun& r : runs_)
        for (int t = r.first; t < r.first + r.count; t++) {
            std::string& d = steps_[t].desc;
            const size_t at = d.find(" [persistent run");
            if (at != std::string::npos) d.erase(at, d.find(']', at) - at + 1);
            char buf[96];
            if (r.on) snprintf(buf, sizeof buf, t == r.first ? " [persistent run: this and the next %d layers are ONE launch]" : " [persistent run: launched with step %d]", t == r.first ? r.count - 1 : r.first);
            else snprintf(buf, sizeof buf, " [persistent run turned down: layer by layer]");
            d += buf;
        }
}
int Executor::defaultPlanHost(ONNXGraph& g, int batch, std::vector<int32_t>* out) {
    if (batch <= 0) { sm_set_error("batch must be positive"); return SM_ERR_INVALID_ARGUMENT; }
    int rc = g.build(); if (rc) return rc;
    Executor e; rc = e.plan(g, batch); if (rc) return rc;
    *out = e.exportPlan();
    return SM_OK;
}

int Executor::create(ONNXGraph& g, int device, int batch, void* ext_arena, bool upload, Executor** out, const int32_t* plan, size_t plan_words) {
    if (batch <= 0) { sm_set_error("batch must be positive"); return SM_ERR_INVALID_ARGUMENT; }
    int rc = g.build(); if (rc) return rc;
    std::unique_ptr<Executor> e(new Executor());
    e->device_ = device;
    rc = e->plan(g, batch); if (rc) return rc;
    SM_HIP_CHECK(hipSetDevice(device));
    SM_HIP_CHECK(hipMalloc((void**)&e->act_, e->act_bytes_ + kAlign));
    SM_HIP_CHECK(hipMemset(e->act_, 0, e->act_bytes_ + kAlign));
    if (e->ws_bytes_) SM_HIP_CHECK(hipMalloc((void**)&e->ws_, e->ws_bytes_));
    bool any_side = false; for (auto& st : e->steps_) any_side |= st.join >= 0;
    if (any_side) {
        SM_HIP_CHECK(hipStreamCreateWithFlags(&e->side_, hipStreamNonBlocking));
        SM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork_, hipEventDisableTiming));
        SM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join_, hipEventDisableTiming));
    }
    if (ext_arena) { e->arena_ = (char*)ext_arena; e->own_arena_ = false; }
    else { SM_HIP_CHECK(hipMalloc((void**)&e->arena_, e->arena_bytes_)); e->own_arena_ = true; upload = true; }
    if (upload) {
        std::vector<uint8_t> host; rc = e->packWeights(&host); if (rc) return rc;
        SM_HIP_CHECK(hipMemcpy(e->arena_, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    const Value& ov = e->values_[e->output_value_];
    e->out_f32_count_ = std::max(sm_float_array_count(batch, ov.c, ov.h, ov.w), (size_t)batch * ov.c * ov.h * ov.w);
    SM_HIP_CHECK(hipMalloc((void**)&e->out_f32_, e->out_f32_count_ * 4));
    for (auto& st : e->steps_) { st.op.weight.clear(); st.op.weight.shrink_to_fit();
                                 st.head.weight.clear(); st.head.weight.shrink_to_fit(); st.reduce.weight.clear(); st.reduce.weight.shrink_to_fit(); st.side.weight.clear(); st.side.weight.shrink_to_fit(); st.mb_dw.weight.clear(); st.mb_dw.weight.shrink_to_fit(); st.mb_proj.weight.clear(); st.mb_proj.weight.shrink_to_fit(); }   // host masters no longer needed
    if (!e->runs_.empty()) {
        if (!sm_stage_device_fast()) for (StageRun& r : e->runs_) { r.on = false; r.why_off = SM_RUN_OFF_DEVICE; }   // fewer than 256 compute units: the runs would complete (claimed mode) but slower than their layers
        size_t ctl = 0;
        for (StageRun& r : e->runs_) {                                // the real address map: the kernel-level planner must take every run (pass 4 keeps a run's tensors apart)
            const std::vector<sm_stage_phase> ph = e->runPhases(r, e->arena_, e->act_, true);
            if (!sm_stage_supported(ph.data(), (int)ph.size())) { sm_set_error("persistent run at step %d: rejected on the real address map", r.first); return SM_ERR_INCONSISTENT_STATE; }
            ctl = std::max(ctl, sm_stage_ctl_bytes(ph.data(), (int)ph.size()));
        }
        SM_HIP_CHECK(hipMalloc(&e->stage_ctl_, ctl)); SM_HIP_CHECK(hipMemset(e->stage_ctl_, 0, ctl));
    }
    e->annotateRuns();
    if (plan) { rc = e->importPlan(plan, plan_words); if (rc) return rc; }   // another executor's picks (rank 0 of a batch-shard job): identical kernels on every rank
    else if (e->cfg_.autotune) {   // with an external arena and upload == 0 the caller filled it before compiling
        hipStream_t s; SM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        rc = e->autotune(s);
        (void)hipStreamDestroy(s);
        if (rc) return rc;
    }
    *out = e.release();
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------------------
std::vector<sm_stage_phase> Executor::runPhases(const StageRun& r, const char* arena, const char* act, bool packed) const {
    std::vector<sm_stage_phase> ph(r.count);
    auto ptr = [&](int v) -> const void* { return act + root_offset_[place_[v].root] + (size_t)place_[v].c_off * 2; };
    for (int i = 0; i < r.count; i++) {
        const Step& st = steps_[r.first + i]; const Op& op = st.op; sm_stage_phase& q = ph[i];
        memset(&q, 0, sizeof q);
        q.d = convDesc(st);
        q.x = ptr(op.inputs[0]); q.w_ohwi = arena + st.w_off; q.bias = st.has_bias ? (const float*)(arena + st.b_off) : nullptr;
        q.residual = op.residual >= 0 ? ptr(op.residual) : nullptr; q.y = (void*)ptr(op.output);
        q.w_packed = (packed && st.has_wp) ? arena + st.wp_off : nullptr;
        if (st.dual_src >= 0) { const Value& xv = values_[st.dual_src]; q.x2 = ptr(st.dual_src); q.K2 = xv.c; q.x2_ld = valueLd(st.dual_src); q.x2_H = xv.h; q.x2_W = xv.w; q.x2_stride = st.dual_stride; }
    }
    return ph;
}

int Executor::launchRun(const StageRun& r, hipStream_t s) const {
    const std::vector<sm_stage_phase> ph = runPhases(r, arena_, act_, true);
    return sm_stage_fwd(ph.data(), (int)ph.size(), stage_ctl_, (sm_stream_t)s);
}

int Executor::numLaunchesNow() const {
    int n = launches_;
    for (const StageRun& r : runs_) if (r.on) n -= r.count - 1;
    return n;
}

// ------------------------------------------------------------------------------------------------------------
int Executor::launchStep(const Step& st, hipStream_t s) const {
    const Op& op = st.op; const Value& ov = values_[op.output];
    void* y = valuePtr(op.output); const int y_ld = valueLd(op.output);
    sm_stream_t ss = (sm_stream_t)s;
    switch (op.kind) {
    case OpKind::Conv: case OpKind::FullyConnected: {
        if (st.has_mb && st.mb_stem) {
            if (!cur_in_) { sm_set_error("stem block launched before any input was bound"); return SM_ERR_INCONSISTENT_STATE; }
            const Value& iv = values_[input_value_]; const Value& pv = values_[st.mb_proj.output];
            sm_mbconv_desc m{}; m.N = batch_; m.H = st.stem_oh; m.W = st.stem_ow; m.Ho = pv.h; m.Wo = pv.w; m.stride = st.mb_dw.sh; m.Cin = 27; m.Ch = op.cout; m.Cout = st.mb_proj.cout; m.expand = 1;
            m.act1 = st.stem_act; m.slope1 = st.stem_slope; m.act2 = st.mb_dw.act; m.slope2 = st.mb_dw.slope; m.act3 = st.mb_proj.act; m.slope3 = st.mb_proj.slope; m.y_ld = valueLd(st.mb_proj.output);
            return sm_mbconv_stem_fwd(&m, cur_in_, iv.c, iv.h, iv.w, st.stem_s, st.stem_p, arena_ + st.ws_off, st.stem_b.empty() ? nullptr : (const float*)(arena_ + st.bs_off), arena_ + st.wd_off,
                                      st.has_biasd ? (const float*)(arena_ + st.bd_off) : nullptr, arena_ + st.w2_off, st.has_bias2 ? (const float*)(arena_ + st.b2_off) : nullptr,
                                      valuePtr(st.mb_proj.output), ss);
        }
        if (st.has_mb) {
            const Value& iv = values_[op.inputs[0]]; const Value& pv = values_[st.mb_proj.output];
            sm_mbconv_desc m{}; m.N = batch_; m.H = iv.h; m.W = iv.w; m.Ho = pv.h; m.Wo = pv.w; m.stride = st.mb_dw.sh; m.Cin = op.cin; m.Ch = op.cout; m.Cout = st.mb_proj.cout; m.expand = st.mb_noexp ? 0 : 1;
            m.residual = st.mb_proj.residual >= 0 ? 1 : 0; m.act1 = st.mb_noexp ? SM_ACT_NONE : op.act; m.slope1 = op.slope; m.act2 = st.mb_dw.act; m.slope2 = st.mb_dw.slope; m.act3 = st.mb_proj.act; m.slope3 = st.mb_proj.slope;
            m.x_ld = valueLd(op.inputs[0]); m.y_ld = valueLd(st.mb_proj.output);
            return sm_mbconv_fwd(&m, valuePtr(op.inputs[0]), st.mb_noexp ? nullptr : arena_ + st.w_off, (st.has_bias && !st.mb_noexp) ? (const float*)(arena_ + st.b_off) : nullptr, arena_ + st.wd_off,
                                 st.has_biasd ? (const float*)(arena_ + st.bd_off) : nullptr, arena_ + st.w2_off, st.has_bias2 ? (const float*)(arena_ + st.b2_off) : nullptr,
                                 valuePtr(st.mb_proj.output), ss);
        }
        const sm_conv_desc d = convDesc(st);
        if (st.stem && stem_f32_) {
            if (!cur_in_) { sm_set_error("stem step launched before any input was bound"); return SM_ERR_INCONSISTENT_STATE; }
            const Value& iv = values_[input_value_];
            if (st.stem3)
                return sm_stem3x3_f32_fwd(cur_in_, batch_, iv.c, iv.h, iv.w, arena_ + st.ws_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, op.cout, st.stem_s, st.stem3_p,
                                          op.act, op.slope, y, y_ld, ss);
            if (st.pool3_out >= 0) {
                const Value& pv = values_[st.pool3_out];
                return sm_conv2d_stem_f32_pool_fwd(&d, cur_in_, iv.c, iv.h, iv.w, stem_pt_, stem_pl_, arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr,
                                                   valuePtr(st.pool3_out), pv.h, pv.w, valueLd(st.pool3_out), ss);
            }
            return sm_conv2d_stem_f32_fwd_cfg(&d, st.conv_cfg, cur_in_, iv.c, iv.h, iv.w, stem_pt_, stem_pl_, arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, y, ss);
        }
        if (st.has_side)
            return sm_conv2d_xhead_proj_fwd_cfg(&d, st.conv_cfg, valuePtr(op.inputs[0]), arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, arena_ + st.w2_off,
                                                st.has_bias2 ? (const float*)(arena_ + st.b2_off) : nullptr, st.head.cout, st.side.act, st.side.slope, valuePtr(st.side.inputs[0]),
                                                valueLd(st.side.inputs[0]), arena_ + st.wd_off, valuePtr(st.side.output), valueLd(st.side.output), ss);
        if (st.has_reduce) {
            const int xin = st.reduce.inputs[0];
            sm_bneck_desc bd{}; bd.N = batch_; bd.H = values_[st.head.output].h; bd.W = values_[st.head.output].w; bd.C0 = st.reduce.cin; bd.Cm = op.cout; bd.C3 = st.head.cout;
            bd.x_ld = valueLd(xin); bd.y_ld = valueLd(st.head.output); bd.res_ld = st.head.residual >= 0 ? valueLd(st.head.residual) : 0;
            bd.act1 = st.reduce.act; bd.slope1 = st.reduce.slope; bd.act2 = op.act; bd.slope2 = op.slope; bd.act3 = st.head.act; bd.slope3 = st.head.slope;
            return sm_bottleneck_fwd(&bd, valuePtr(xin), arena_ + st.w0_off, st.has_bias0 ? (const float*)(arena_ + st.b0_off) : nullptr, arena_ + st.w_off,
                                     st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, arena_ + st.w2_off, st.has_bias2 ? (const float*)(arena_ + st.b2_off) : nullptr,
                                     st.head.residual >= 0 ? valuePtr(st.head.residual) : nullptr, valuePtr(st.head.output), ss);
        }
        if (st.has_head && (st.head.residual >= 0 || st.head.cout > 32))
            return sm_conv2d_xhead_fwd_cfg(&d, st.conv_cfg, valuePtr(op.inputs[0]), arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, nullptr, arena_ + st.w2_off,
                                           st.has_bias2 ? (const float*)(arena_ + st.b2_off) : nullptr, st.head.cout, st.head.act, st.head.slope,
                                           st.head.residual >= 0 ? valuePtr(st.head.residual) : nullptr, st.head.residual >= 0 ? valueLd(st.head.residual) : 0,
                                           valuePtr(st.head.output), valueLd(st.head.output), ss);
        if (st.has_head)
            return sm_conv2d_head_fwd_cfg(&d, st.conv_cfg, valuePtr(op.inputs[0]), arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, nullptr, arena_ + st.w2_off,
                                          st.has_bias2 ? (const float*)(arena_ + st.b2_off) : nullptr, st.head.cout, st.head.act, st.head.slope, valuePtr(st.head.output), valueLd(st.head.output), ss);
        if (st.dual_src >= 0) {
            const Value& xv = values_[st.dual_src];
            return sm_conv2d_dual_fwd_cfg(&d, st.conv_cfg, valuePtr(op.inputs[0]), valuePtr(st.dual_src), xv.c, valueLd(st.dual_src), xv.h, xv.w, st.dual_stride, arena_ + st.w_off,
                                          st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, y, ss);
        }
        if (st.pool_out >= 0)
            return sm_conv2d_pool_fwd_cfg(&d, st.conv_cfg, valuePtr(op.inputs[0]), arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, y,
                                          valuePtr(st.pool_out), valueLd(st.pool_out), ss);
        return sm_conv2d_fwd_cfg(&d, st.conv_cfg, valuePtr(op.inputs[0]), arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr,
                                 op.residual >= 0 ? valuePtr(op.residual) : nullptr, y, st.join >= 0 ? nullptr : ws_, st.join >= 0 ? 0 : ws_bytes_, ss);   // the split-K scratch belongs to the main branch
    }
    case OpKind::ConvTranspose: {   // shapes the phase decomposition does not take: direct gather kernel
        const Value& iv = values_[op.inputs[0]];
        sm_conv_desc d{};
        d.N = batch_; d.H = iv.h; d.W = iv.w; d.Cin = op.cin; d.Cout = op.cout; d.kH = op.kh; d.kW = op.kw; d.sH = op.sh; d.sW = op.sw; d.dH = op.dh; d.dW = op.dw;
        d.pt = op.pt; d.pl = op.pl; d.pb = op.pb; d.pr = op.pr; d.groups = 1; d.Ho = ov.h; d.Wo = ov.w; d.act = op.act; d.slope = op.slope;
        d.x_ld = valueLd(op.inputs[0]); d.y_ld = y_ld;
        return sm_conv_transpose2d_fwd(&d, valuePtr(op.inputs[0]), arena_ + st.w_off, st.has_bias ? (const float*)(arena_ + st.b_off) : nullptr, y, ss);
    }
    case OpKind::BatchNorm:
        return sm_batchnorm(valuePtr(op.inputs[0]), y, st.M, ov.c, valueLd(op.inputs[0]), y_ld, (const float*)(arena_ + st.s_off), (const float*)(arena_ + st.t_off), op.act, op.slope, ss);
    case OpKind::Neuron:
        return sm_batchnorm(valuePtr(op.inputs[0]), y, st.M, ov.c, valueLd(op.inputs[0]), y_ld, nullptr, nullptr, op.act, op.slope, ss);
    case OpKind::Add:
        return sm_binary(op.binop, valuePtr(op.inputs[0]), valuePtr(op.inputs[1]), y, st.M, ov.c, valueLd(op.inputs[0]), valueLd(op.inputs[1]), y_ld, op.act, op.slope, ss);
    case OpKind::Softmax:
        return sm_softmax(valuePtr(op.inputs[0]), y, st.M, ov.c, valueLd(op.inputs[0]), y_ld, op.logSoftmax ? 1 : 0, ss);
    case OpKind::GroupNorm: {
        const Value& iv = values_[op.inputs[0]];
        return sm_group_norm(valuePtr(op.inputs[0]), y, batch_, iv.h * iv.w, iv.c, op.normGroups, (const float*)(arena_ + st.s_off), (const float*)(arena_ + st.t_off),
                             op.eps, valueLd(op.inputs[0]), y_ld, ss);
    }
    case OpKind::Pad: {
        const Value& iv = values_[op.inputs[0]];
        return sm_pad2d(valuePtr(op.inputs[0]), y, batch_, iv.h, iv.w, iv.c, op.pt, op.pl, op.pb, op.pr, op.padMode, valueLd(op.inputs[0]), y_ld, ss);
    }
    case OpKind::MaxPool: case OpKind::AvgPool: {
        const Value& iv = values_[op.inputs[0]];
        sm_pool_desc d{batch_, iv.h, iv.w, iv.c, op.kh, op.kw, op.sh, op.sw, op.y0, op.x0, ov.h, ov.w, valueLd(op.inputs[0]), y_ld};
        return op.kind == OpKind::MaxPool ? sm_maxpool2d(&d, valuePtr(op.inputs[0]), y, ss) : sm_avgpool2d(&d, valuePtr(op.inputs[0]), y, ss);
    }
    case OpKind::GlobalAvgPool: {
        const Value& iv = values_[op.inputs[0]];
        return sm_global_avgpool(valuePtr(op.inputs[0]), y, batch_, iv.h * iv.w, iv.c, valueLd(op.inputs[0]), y_ld, ss);
    }
    case OpKind::Upsample: {
        const Value& iv = values_[op.inputs[0]];
        if (op.bilinear) return sm_upsample_bilinear(valuePtr(op.inputs[0]), y, batch_, iv.h, iv.w, iv.c, op.sy, op.sx, op.alignCorners, valueLd(op.inputs[0]), y_ld, ss);
        return sm_upsample_nearest(valuePtr(op.inputs[0]), y, batch_, iv.h, iv.w, iv.c, op.sy, op.sx, valueLd(op.inputs[0]), y_ld, ss);
    }
    case OpKind::Concat: {   // copy fallback for one operand: op.pl = channel offset inside the concat buffer
        const Value& iv = values_[op.inputs[0]];
        return sm_copy_channels(valuePtr(op.inputs[0]), (char*)y + (size_t)op.pl * 2, st.M, iv.c, valueLd(op.inputs[0]), y_ld, ss);
    }
    case OpKind::Reshape: {
        const Value& iv = values_[op.inputs[0]];
        return sm_reshape_nchw_order(valuePtr(op.inputs[0]), y, batch_, iv.c, iv.h * iv.w, valueLd(op.inputs[0]), ov.c, ov.h * ov.w, y_ld, ss);
    }
    case OpKind::Alias: return SM_OK;
    }
    return SM_OK;
}

// the kernel-level descriptor of a Conv / FullyConnected step
sm_conv_desc Executor::convDesc(const Step& st) const {
    const Op& op = st.op; const Value& ov = values_[op.output]; const Value& iv = values_[op.inputs[0]];
    sm_conv_desc d{};
    d.N = batch_; d.Cin = op.cin; d.Cout = op.cout; d.kH = op.kh; d.kW = op.kw; d.sH = op.sh; d.sW = op.sw; d.dH = op.dh; d.dW = op.dw;
    d.pt = op.pt; d.pl = op.pl; d.pb = op.pb; d.pr = op.pr; d.groups = op.groups; d.act = op.act; d.slope = op.slope;
    if (op.kind == OpKind::Conv) { d.H = iv.h; d.W = iv.w; d.Ho = ov.h; d.Wo = ov.w; d.x_ld = valueLd(op.inputs[0]); }
    else { d.H = d.W = d.Ho = d.Wo = 1; d.x_ld = st.fc_h * st.fc_w > 1 ? op.cin : valueLd(op.inputs[0]); }   // flattened map: one dense row per image
    if (op.x_ups) { d.H = ov.h; d.W = ov.w; d.x_ups = 1; }                                                   // input value = the half-resolution tensor
    if (st.stem) { d.H = stem_h_; d.W = stem_w_; d.x_ld = 8; }                                             // row-pair image
    if (op.grid_h) { d.Ho = op.grid_h; d.Wo = op.grid_w; d.y_sH = op.y_sh; d.y_sW = op.y_sw; d.y_oH = op.y_oh; d.y_oW = op.y_ow; d.y_H = op.y_h; d.y_W = op.y_w; }   // ConvTranspose phase
    d.y_ld = valueLd(op.output); d.res_ld = op.residual >= 0 ? valueLd(op.residual) : 0;
    return d;
}

// application image (NCHW fp32) -> the layout the first layer reads
int Executor::convertInput(hipStream_t s, const float* in) const {
    const Value& iv = values_[input_value_];
    if (stem_f32_) return SM_OK;                                     // the stem kernel reads `in` itself
    if (stem_) return sm_nchw_f32_to_rowpair_f16(in, valuePtr(input_value_), batch_, iv.c, iv.h, iv.w, stem_h_, stem_w_, stem_pt_, stem_pl_, (sm_stream_t)s);
    return sm_nchw_f32_to_nhwc_f16(in, valuePtr(input_value_), batch_, iv.c, iv.h, iv.w, input_ld_, 0, 0, 0, 0, (sm_stream_t)s);
}

int Executor::launchAll(hipStream_t s, const float* in) const {
    static const bool debug_sync = getenv("SMELTER_DEBUG_SYNC") != nullptr;   // debugging switch, read once
    cur_in_ = in;
    const Roctx& rx = roctx();
    if (rx.on) rx.push("input NCHW f32 -> first-layer layout");
    int rc = convertInput(s, in);
    if (rx.on) rx.pop();
    if (rc) return rc;
    struct Range { const Roctx& r; Range(const Roctx& r_, const char* n) : r(r_) { if (r.on) r.push(n); } ~Range() { if (r.on) r.pop(); } };
    for (size_t si = 0; si < steps_.size(); si++) {
        const Step& st = steps_[si];
        if (step_run_[si] >= 0 && runs_[step_run_[si]].on) {      // a persistent run: one launch at its first step
            const StageRun& r = runs_[step_run_[si]];
            if ((int)si == r.first) { Range range(rx, ("persistent run: " + st.desc).c_str()); rc = launchRun(r, s); if (rc) return rc; }
            continue;
        }
        Range range(rx, st.desc.c_str());
        if (st.join >= 0 && side_) {                            // fork: the side stream sees everything enqueued so far
            SM_HIP_CHECK(hipEventRecord(ev_fork_, s));
            SM_HIP_CHECK(hipStreamWaitEvent(side_, ev_fork_, 0));
            rc = launchStep(st, side_); if (rc) return rc;
            SM_HIP_CHECK(hipEventRecord(ev_join_, side_));
            continue;
        }
        if (st.waits_side && side_) SM_HIP_CHECK(hipStreamWaitEvent(s, ev_join_, 0));
        rc = launchStep(st, s); if (rc) return rc;
        if (debug_sync && !cfg_.useHipGraph) {                  // fault localisation: name the step before its kernel's fault aborts the process
            fprintf(stderr, "[sync] %s\n", st.desc.c_str()); fflush(stderr);
            SM_HIP_CHECK(hipStreamSynchronize(s));
        }
    }
    return SM_OK;
}

int Executor::encode(hipStream_t s, const float* in) {
    if (!in) { sm_set_error("smelter_exec_encode: null input"); return SM_ERR_INVALID_ARGUMENT; }
    if (!cfg_.useHipGraph || roctx().on) return launchAll(s, in);      // (trace ranges bracket real launches, not a graph replay)
    if (!graph_exec_ || captured_in_ != in) {
        if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
        if (graph_) { (void)hipGraphDestroy(graph_); graph_ = nullptr; }
        hipStream_t cs; SM_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        SM_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc = launchAll(cs, in);
        hipError_t ee = hipStreamEndCapture(cs, &graph_);
        (void)hipStreamDestroy(cs);
        if (rc) return rc;
        if (ee != hipSuccess) { sm_set_error("hipStreamEndCapture: %s", hipGetErrorString(ee)); return SM_ERR_HIP; }
        SM_HIP_CHECK(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
        captured_in_ = in;
    }
    SM_HIP_CHECK(hipGraphLaunch(graph_exec_, s));
    return SM_OK;
}

int Executor::encodeSized(hipStream_t s, const float* in, int in_h, int in_w) {
    const Value& iv = values_[input_value_];
    if (in_h == iv.h && in_w == iv.w) return encode(s, in);
    if (cfg_.inputConstraint == Configuration::InputConstraint::none) {
        sm_set_error("source image is %dx%d, the model takes %dx%d and Configuration.inputConstraint is .none", in_h, in_w, iv.h, iv.w);
        return SM_ERR_UNSUPPORTED_INPUT;
    }
    if (!in || in_h <= 0 || in_w <= 0) { sm_set_error("smelter_exec_encode_sized: bad argument"); return SM_ERR_INVALID_ARGUMENT; }
    if (!resized_) SM_HIP_CHECK(hipMalloc((void**)&resized_, (size_t)batch_ * iv.c * iv.h * iv.w * sizeof(float)));
    const int mode = cfg_.inputConstraint == Configuration::InputConstraint::forceInputScaleLanczos ? SM_INPUT_FORCE_SCALE_LANCZOS : SM_INPUT_FORCE_SCALE_BILINEAR;
    int rc = sm_resize_nchw_f32(in, resized_, batch_ * iv.c, in_h, in_w, iv.h, iv.w, mode, (sm_stream_t)s);   // MPSNN{Lanczos,Bilinear}ScaleNode (:222-239)
    if (rc) return rc;
    return encode(s, resized_);                                       // the captured hipGraph is keyed on this fixed buffer
}

int Executor::readFloat(hipStream_t s, float* dst, size_t cap, int layout) {
    const Value& ov = values_[output_value_];
    size_t count = layout == 1 ? sm_float_array_count(batch_, ov.c, ov.h, ov.w) : (size_t)batch_ * ov.c * ov.h * ov.w;
    if (cap < count) { sm_set_error("smelter_exec_read_float: destination holds %zu floats, result has %zu", cap, count); return SM_ERR_INVALID_ARGUMENT; }
    int rc = layout == 1 ? sm_nhwc_f16_to_float_array(valuePtr(output_value_), out_f32_, batch_, ov.c, ov.h, ov.w, valueLd(output_value_), (sm_stream_t)s)
                         : sm_nhwc_f16_to_nchw_f32(valuePtr(output_value_), out_f32_, batch_, ov.c, ov.h, ov.w, valueLd(output_value_), (sm_stream_t)s);
    if (rc) return rc;
    SM_HIP_CHECK(hipMemcpyAsync(dst, out_f32_, count * 4, hipMemcpyDeviceToHost, s));
    SM_HIP_CHECK(hipStreamSynchronize(s));
    return checkRuns(s);
}

// A persistent run whose hand-off wait ran out left undefined tensors behind: say so where the host first looks at a result (the encode itself only enqueues)
int Executor::checkRuns(hipStream_t s) const {
    if (!stage_ctl_) return SM_OK;
    bool any = false; for (const StageRun& r : runs_) any = any || r.on;
    if (!any) return SM_OK;                                          // (no run is launched — nothing to ask, no extra copy / synchronise per read)
    int32_t err = 0, mode = 0;
    const int rc = sm_stage_status(stage_ctl_, (sm_stream_t)s, &err, &mode);
    if (rc) return rc;
    if (err) { sm_set_error("a persistent run (sm_stage_fwd) reported a timed-out hand-off (bits %d, modes %d): the results of this encode are undefined", err, mode); return SM_ERR_INCONSISTENT_STATE; }
    return SM_OK;
}

int Executor::materializedStored(int i) const {
    if (i < 0 || i >= (int)materialized_.size()) return 0;
    for (size_t v = 0; v < value_names_.size(); v++) if (value_names_[v] == materialized_[i]) return stored_[v];
    return 0;
}
int Executor::readValue(hipStream_t s, const char* name, float* dst, size_t cap, int* c, int* h, int* w) {
    { const int rc0 = checkRuns(s); if (rc0) return rc0; }
    int v = -1; for (size_t i = 0; i < value_names_.size(); i++) if (value_names_[i] == name) v = (int)i;
    if (v < 0) { sm_set_error("no value named '%s' survives fusion", name); return SM_ERR_NO_SUCH_OUTPUT; }
    if (!stored_[v]) { sm_set_error("value '%s' has no storage: a fused launch keeps it in registers / LDS only", name); return SM_ERR_NO_SUCH_OUTPUT; }
    const Value& val = values_[v]; size_t count = (size_t)batch_ * val.c * val.h * val.w;
    *c = val.c; *h = val.h; *w = val.w;
    if (cap < count) { sm_set_error("smelter_exec_read_value: destination too small"); return SM_ERR_INVALID_ARGUMENT; }
    float* tmp; SM_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
    int rc = sm_nhwc_f16_to_nchw_f32(valuePtr(v), tmp, batch_, val.c, val.h, val.w, valueLd(v), (sm_stream_t)s);
    if (!rc) { hipError_t e1 = hipMemcpyAsync(dst, tmp, count * 4, hipMemcpyDeviceToHost, s); hipError_t e2 = hipStreamSynchronize(s); if (e1 != hipSuccess || e2 != hipSuccess) { sm_set_error("read_value copy failed"); rc = SM_ERR_HIP; } }
    (void)hipFree(tmp);
    return rc;
}

int Executor::timeOp(hipStream_t s, int idx, int iters, float* ms) {
    if (idx < 0 || idx >= (int)steps_.size() || iters <= 0) { sm_set_error("smelter_exec_time_op: bad index"); return SM_ERR_INVALID_ARGUMENT; }
    hipEvent_t a, b; SM_HIP_CHECK(hipEventCreate(&a)); SM_HIP_CHECK(hipEventCreate(&b));
    int rc = launchStep(steps_[idx], s);   // warm
    SM_HIP_CHECK(hipEventRecord(a, s));
    for (int i = 0; i < iters && !rc; i++) rc = launchStep(steps_[idx], s);
    SM_HIP_CHECK(hipEventRecord(b, s));
    SM_HIP_CHECK(hipEventSynchronize(b));
    float t = 0; SM_HIP_CHECK(hipEventElapsedTime(&t, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms = t / iters;
    return rc;
}

int Executor::profile(hipStream_t s, const float* in, int iters, float* ms, int cap) {
    const int n = (int)steps_.size() + 1;
    if (!in || !ms || cap < n || iters <= 0) { sm_set_error("smelter_exec_profile: bad argument"); return SM_ERR_INVALID_ARGUMENT; }
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) SM_HIP_CHECK(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    int rc = SM_OK;
    for (int it = 0; it < iters + 1 && !rc; it++) {          // pass 0 = warm-up, not accumulated
        SM_HIP_CHECK(hipEventRecord(ev[0], s));
        rc = convertInput(s, in);
        SM_HIP_CHECK(hipEventRecord(ev[1], s));
        for (int i = 0; i < n - 1 && !rc; i++) {
            const int ri = step_run_[i];
            if (ri >= 0 && runs_[ri].on) { if (i == runs_[ri].first) rc = launchRun(runs_[ri], s); }   // (the whole run under its first step, zeros under the others)
            else rc = launchStep(steps_[i], s);
            SM_HIP_CHECK(hipEventRecord(ev[i + 2], s));
        }
        SM_HIP_CHECK(hipStreamSynchronize(s));
        if (it > 0) for (int i = 0; i < n; i++) { float t = 0; SM_HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1])); acc[i] += t; }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < n; i++) ms[i] = (float)(acc[i] / iters);
    return rc;
}

int Executor::autotune(hipStream_t s) {
    // Tile configs are ranked under COLD caches: inside the network a layer finds its weights in HBM and its input in
    // HBM / Infinity Cache, never in an L2 warmed by the previous launch of the same kernel (L2 is invalidated at kernel
    // boundaries).  Ranking on back-to-back warm launches picked configs that were up to 1.5x slower in the graph.
    int ncfg = sm_conv2d_num_configs();
    if (cfg_.autotuneMaxCfg > 0) ncfg = std::min(ncfg, cfg_.autotuneMaxCfg);   // A/B runs on one box
    // Default (round 3): IN SITU — a config is timed behind the two real predecessor launches, i.e. with its operands where the graph leaves them (the
    // producer just wrote them: Infinity Cache, L2 flushed at the boundary).  The all-cold ranking (640 MB memset before every launch, SMELTER_AUTOTUNE=cold)
    // mis-ranks the streaming kernels of conv1x1_rw.hpp: it charged them HBM-cold operands the network never shows them and kept the tiled kernels on the
    // 256 -> 128 and 128 -> 512 layers (39 vs 30 us, 27 vs 22 us in the graph; same-box step 1.172 -> 1.157 ms).
    const bool insitu = cfg_.autotuneMode != SM_TUNE_COLD;
    char* flush = nullptr; const size_t flush_bytes = (size_t)640 << 20;
    if (hipMalloc((void**)&flush, flush_bytes) != hipSuccess) { flush = nullptr; (void)hipGetLastError(); }
    hipEvent_t a, b; SM_HIP_CHECK(hipEventCreate(&a)); SM_HIP_CHECK(hipEventCreate(&b));
    int rc = SM_OK;
    float* tune_in = nullptr;                                        // the loader-wave stem reads the fp32 batch: a zero image while its configurations are timed
    if (stem_f32_ && !cur_in_) {
        const Value& iv = values_[input_value_]; const size_t bytes = (size_t)batch_ * iv.c * iv.h * iv.w * sizeof(float);
        SM_HIP_CHECK(hipMalloc((void**)&tune_in, bytes)); SM_HIP_CHECK(hipMemsetAsync(tune_in, 0, bytes, s));
        cur_in_ = tune_in;
    }
    for (size_t i = 0; i < steps_.size() && !rc; i++) {
        Step& st = steps_[i];
        if (!isConvLike(st.op.kind) || st.op.groups != 1 || (st.op.cin % 8) || st.has_mb || st.has_reduce || st.pool3_out >= 0 || st.stem3) continue;   // (whole-block steps run one dedicated kernel)
        if (st.op.kind != OpKind::ConvTranspose) { const sm_conv_desc cd = convDesc(st); if (!sm_conv2d_uses_tile_config(&cd)) continue; }   // dedicated kernel, nothing to tune
        // one config: `reps` timed launches after an untimed first one (code fetch); returns false if the config does not apply
        auto timeCfg = [&](int c, int reps, std::vector<float>& samples) -> bool {
            st.conv_cfg = c; samples.clear();
            for (int it = 0; it < reps + 1 && !rc; it++) {
                if (insitu && i >= 2) {                                    // the cache state the graph leaves: run the two predecessors
                    for (size_t q = i - 2; q < i && !rc; q++) rc = launchStep(steps_[q], s);
                    if (rc) return false;
                } else if (flush && hipMemsetAsync(flush, it, flush_bytes, s) != hipSuccess) { rc = SM_ERR_HIP; return false; }
                if (hipEventRecord(a, s) != hipSuccess) { rc = SM_ERR_HIP; return false; }
                int r = launchStep(st, s);
                if (hipEventRecord(b, s) != hipSuccess) { rc = SM_ERR_HIP; return false; }
                if (r == SM_ERR_INVALID_ARGUMENT) return false;            // tile config not applicable to this layer
                if (r) { rc = r; return false; }
                float t = 0;
                if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t, a, b) != hipSuccess) { rc = SM_ERR_HIP; return false; }
                if (it > 0) samples.push_back(t);
            }
            return !rc;
        };
        // Phase 1: every config, best of 3 (cheap screen).  Phase 2: the 4 best again, MEDIAN of 9 — the minimum of a few
        // launches rewards lucky outliers and made the chosen set (and the step time, +-2 %) differ from compile to compile.
        std::vector<std::pair<float, int>> screen; std::vector<float> smp;
        for (int c = 0; c < ncfg && !rc; c++)
            if (timeCfg(c, 3, smp) && !smp.empty()) screen.push_back({*std::min_element(smp.begin(), smp.end()), c});
        std::sort(screen.begin(), screen.end());
        float best = 1e30f; int best_cfg = screen.empty() ? -1 : screen[0].second;
        for (size_t k = 0; k < screen.size() && k < 4 && !rc; k++) {
            if (!timeCfg(screen[k].second, 9, smp) || smp.empty()) continue;
            std::sort(smp.begin(), smp.end());
            const float med = smp[smp.size() / 2];
            if (med < best) { best = med; best_cfg = screen[k].second; }
        }
        st.conv_cfg = best_cfg;
        st.desc += " cfg=" + std::to_string(best_cfg);
    }
    // Persistent runs (pass 3k): ONE launch against its layers launched one by one with the configurations just chosen, both behind the two real predecessor launches.
    // VERDICT r5 item 2 — the admission must not depend on the luck of a compile: both forms are launched twice untimed first (code fetch; the run kernel's first launch in a
    // process finds its blocks arriving apart and falls to the claimed mode), then timed ALTERNATELY (a clock or cache drift hits both forms alike), 15 samples each, medians;
    // the run is kept unless it measures more than 3 % slower than its layers (inside the noise band the decision is the same on every compile: the run), a hand-off wait ran
    // out, or no timed launch ran in the static mode.  What was measured stays with the run (smelter_exec_run_info).
    for (size_t ri = 0; ri < runs_.size() && !rc; ri++) {
        StageRun& r = runs_[ri];
        if (!sm_stage_device_fast()) { r.on = false; r.why_off = SM_RUN_OFF_DEVICE; continue; }
        constexpr int kWarm = 2, kSamples = 15;
        auto launchForm = [&](bool as_run, bool timed, float* ms) {
            for (int q = std::max(0, r.first - 2); q < r.first && !rc; q++) rc = launchStep(steps_[q], s);
            if (rc) return;
            if (timed && hipEventRecord(a, s) != hipSuccess) { rc = SM_ERR_HIP; return; }
            if (as_run) rc = launchRun(r, s);
            else for (int q = r.first; q < r.first + r.count && !rc; q++) rc = launchStep(steps_[q], s);
            if (rc || !timed) return;
            if (hipEventRecord(b, s) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(ms, a, b) != hipSuccess) rc = SM_ERR_HIP;
        };
        int32_t err = 0, mode = 0, werr = 0, wmode = 0;
        for (int it = 0; it < kWarm && !rc; it++) { float t = 0; launchForm(false, false, &t); launchForm(true, false, &t); }
        if (!rc) rc = sm_stage_status(stage_ctl_, (sm_stream_t)s, &werr, &wmode);   // (synchronises; clears the words: the warm-up's modes are not the run's)
        std::vector<float> tl, tr;
        for (int it = 0; it < kSamples && !rc; it++) {
            float t = 0;
            launchForm(false, true, &t); if (!rc) tl.push_back(t);
            launchForm(true, true, &t); if (!rc) tr.push_back(t);
        }
        if (rc) break;
        rc = sm_stage_status(stage_ctl_, (sm_stream_t)s, &err, &mode); if (rc) break;
        std::sort(tl.begin(), tl.end()); std::sort(tr.begin(), tr.end());
        r.timed = true; r.samples = kSamples; r.t_layers_us = tl[tl.size() / 2] * 1e3f; r.t_run_us = tr[tr.size() / 2] * 1e3f; r.mode_bits = mode; r.err_bits = err | werr;
        r.why_off = ((err | werr) ? SM_RUN_OFF_TIMEOUT : 0) | (!(mode & 1) ? SM_RUN_OFF_CLAIMED : 0) | (r.t_run_us > 1.03f * r.t_layers_us ? SM_RUN_OFF_SLOWER : 0);
        r.on = r.why_off == 0;
        if (getenv("SMELTER_DEBUG_PLAN")) fprintf(stderr, "[plan] persistent run at step %d (%d layers): %.1f us as one launch, %.1f us layer by layer, modes %d, errors %d -> %s (why_off %d)\n", r.first, r.count, r.t_run_us, r.t_layers_us, mode, err | werr, r.on ? "run" : "layers", r.why_off);
    }
    annotateRuns();
    (void)hipStreamSynchronize(s);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (tune_in) { (void)hipFree(tune_in); cur_in_ = nullptr; }
    if (flush) (void)hipFree(flush);
    return rc;
}

}  // namespace smelter
