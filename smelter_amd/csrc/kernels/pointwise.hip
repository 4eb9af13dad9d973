// pointwise.hip — HBM-bound NHWC fp16 kernels: add(+neuron), relu/prelu, batch-norm (scale/shift), pooling,
// global average pooling, upsampling, channel-slice copy.  These replace the MPS node classes Smelter
// instantiates in /root/reference/Sources/Smelter/Converters.swift (cited per kernel).
//
// Common shape: the channel axis is innermost, so one thread owns 8 consecutive channels (one 16-byte
// load/store, guide G13) of one pixel; a wave covers 512 contiguous channels-bytes -> fully coalesced.
// Grid-stride loops over (pixel, channel-vector) with at most 8 blocks per CU.  Scalar variants cover
// C % 8 != 0 or unaligned pixel strides.
#include "common.hpp"

namespace {

__device__ __forceinline__ bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- MPSNNAdditionNode (Converters.swift:440) [+ fused MPSCNNNeuronReLUNode] --------------------------
__device__ __forceinline__ float sm_binop(float a, float b, int op) {   // MPSNNAddition/Subtraction/Multiplication/DivisionNode
    return op == SM_BIN_ADD ? a + b : op == SM_BIN_SUB ? a - b : op == SM_BIN_MUL ? a * b : a / b;
}
__global__ void add_vec_kernel(const half_t* a, const half_t* b, half_t* y, long pixels, int cv, int a_ld, int b_ld, int y_ld, int act, float slope, int op) {
    long total = pixels * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / cv; int c = (int)(i - m * cv) * 8;
        half8_t va = *(const half8_t*)(a + m * a_ld + c), vb = *(const half8_t*)(b + m * b_ld + c), o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (half_t)sm_act(sm_binop((float)va[j], (float)vb[j], op), act, slope);
        *(half8_t*)(y + m * y_ld + c) = o;
    }
}
__global__ void add_scalar_kernel(const half_t* a, const half_t* b, half_t* y, long pixels, int C, int a_ld, int b_ld, int y_ld, int act, float slope, int op) {
    long total = pixels * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / C; int c = (int)(i - m * C);
        y[m * y_ld + c] = (half_t)sm_act(sm_binop((float)a[m * a_ld + c], (float)b[m * b_ld + c], op), act, slope);
    }
}

// ---- MPSCNNBatchNormalizationNode (:812) as y = x*scale[c]+shift[c]; scale==nullptr -> pure neuron (:351,:376)
__global__ void affine_vec_kernel(const half_t* x, half_t* y, long pixels, int cv, int x_ld, int y_ld, const float* scale, const float* shift, int act, float slope) {
    long total = pixels * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / cv; int c = (int)(i - m * cv) * 8;
        half8_t v = *(const half8_t*)(x + m * x_ld + c), o;
#pragma unroll
        for (int j = 0; j < 8; j++) { float t = (float)v[j]; if (scale) t = t * scale[c + j] + shift[c + j]; o[j] = (half_t)sm_act(t, act, slope); }
        *(half8_t*)(y + m * y_ld + c) = o;
    }
}
__global__ void affine_scalar_kernel(const half_t* x, half_t* y, long pixels, int C, int x_ld, int y_ld, const float* scale, const float* shift, int act, float slope) {
    long total = pixels * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / C; int c = (int)(i - m * C);
        float t = (float)x[m * x_ld + c]; if (scale) t = t * scale[c] + shift[c];
        y[m * y_ld + c] = (half_t)sm_act(t, act, slope);
    }
}

// ---- MPSCNNPoolingMaxNode / MPSCNNPoolingAverageNode (:662, :617) -------------------------------------
// One block row per output image row (wave-uniform n, oy: no per-thread 64-bit division), threads over (ox, 8-channel
// vector).  Block b runs on XCD b % 8 and the XCD L2s are private, so XCD x takes the CONTIGUOUS band of output rows
// [x*R/8, (x+1)*R/8): the input rows that overlapping windows share are then fetched into one L2 instead of two or three
// (a 3x3/2 max-pool re-reads every input pixel 2.25 times).  Max pooling of fp16 data is exact in fp16 (v_pk_max_f16);
// averages accumulate in fp32.
// Max-pool fast path for small windows (2x2/2 of the U-Net, 3x3/2 of ResNet): same block -> row map as pool_kernel, but the KH*KW
// 16-byte loads of an output vector are ALL issued before the first max (clamped addresses, out-of-window taps replaced by -inf
// afterwards: no branch between loads), and a thread takes OPT output vectors — the generic kernel's nested loops with `continue`
// waited for every load before issuing the next one (2 TB/s on the 512x512 U-Net pool: latency-, not bandwidth-bound).
template <int KH, int KW, int OPT>
__global__ __launch_bounds__(256) void maxpool_small_kernel(sm_pool_desc d, const half_t* x, half_t* y) {
    const int cv = d.C >> 3;
    const int rows = d.N * d.Ho, rpx = (rows + SM_NUM_XCD - 1) / SM_NUM_XCD;
    const int row = (blockIdx.x & (SM_NUM_XCD - 1)) * rpx + (blockIdx.x >> 3);
    if (row >= rows || (int)(blockIdx.x >> 3) >= rpx) return;
    const int n = row / d.Ho, oy = row - n * d.Ho, per_row = d.Wo * cv;
    const half_t* img = x + (long)n * d.H * d.W * d.x_ld;
    half8_t v[OPT][KH * KW]; bool ok[OPT][KH * KW]; int idx[OPT];
#pragma unroll
    for (int o = 0; o < OPT; o++) {
        idx[o] = (blockIdx.y * OPT + o) * 256 + threadIdx.x;
        const int id = min(idx[o], per_row - 1), ox = id / cv, c = (id - ox * cv) * 8;
#pragma unroll
        for (int a = 0; a < KH; a++)
#pragma unroll
            for (int b = 0; b < KW; b++) {
                const int iy = oy * d.sH + d.y0 + a, ix = ox * d.sW + d.x0 + b;
                ok[o][a * KW + b] = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                v[o][a * KW + b] = *(const half8_t*)(img + ((long)min(max(iy, 0), d.H - 1) * d.W + min(max(ix, 0), d.W - 1)) * d.x_ld + c);
            }
    }
#pragma unroll
    for (int o = 0; o < OPT; o++) {
        if (idx[o] >= per_row) continue;
        const int ox = idx[o] / cv, c = (idx[o] - ox * cv) * 8;
        half8_t m;
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = (half_t)(-__builtin_inff());
#pragma unroll
        for (int t = 0; t < KH * KW; t++) if (ok[o][t]) m = __builtin_elementwise_max(m, v[o][t]);
        *(half8_t*)(y + ((long)row * d.Wo + ox) * d.y_ld + c) = m;
    }
}

template <bool MAX, int V>
__global__ __launch_bounds__(256) void pool_kernel(sm_pool_desc d, const half_t* x, half_t* y) {
    const int cv = (d.C + V - 1) / V;
    const int rows = d.N * d.Ho, rpx = (rows + SM_NUM_XCD - 1) / SM_NUM_XCD;
    const int row = (blockIdx.x & (SM_NUM_XCD - 1)) * rpx + (blockIdx.x >> 3);
    if (row >= rows || (int)(blockIdx.x >> 3) >= rpx) return;
    const int n = row / d.Ho, oy = row - n * d.Ho;
    const int idx = blockIdx.y * 256 + threadIdx.x;
    if (idx >= d.Wo * cv) return;
    const int ox = idx / cv, c = (idx - ox * cv) * V;
    const int ix0 = ox * d.sW + d.x0;
    half_t* py = y + ((long)row * d.Wo + ox) * d.y_ld + c;
    if (MAX && V == 8) {
        half8_t m;
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = (half_t)(-__builtin_inff());
        for (int a = 0; a < d.kH; a++) {
            const int iy = oy * d.sH + d.y0 + a; if ((unsigned)iy >= (unsigned)d.H) continue;
            const half_t* prow = x + ((long)n * d.H + iy) * d.W * d.x_ld + c;
            for (int b = 0; b < d.kW; b++) {
                const int ix = ix0 + b; if ((unsigned)ix >= (unsigned)d.W) continue;
                m = __builtin_elementwise_max(m, *(const half8_t*)(prow + (long)ix * d.x_ld));
            }
        }
        *(half8_t*)py = m;
        return;
    }
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) acc[j] = MAX ? -3.0e38f : 0.f;
    for (int a = 0; a < d.kH; a++) {
        const int iy = oy * d.sH + d.y0 + a; if ((unsigned)iy >= (unsigned)d.H) continue;
        for (int b = 0; b < d.kW; b++) {
            const int ix = ix0 + b; if ((unsigned)ix >= (unsigned)d.W) continue;
            const half_t* px = x + (((long)n * d.H + iy) * d.W + ix) * d.x_ld + c;
            if (V == 8) { half8_t v = *(const half8_t*)px;
#pragma unroll
                for (int j = 0; j < V; j++) acc[j] = MAX ? fmaxf(acc[j], (float)v[j]) : acc[j] + (float)v[j];
            } else acc[0] = MAX ? fmaxf(acc[0], (float)px[0]) : acc[0] + (float)px[0];
        }
    }
    const float inv = 1.f / (float)(d.kH * d.kW);
    if (V == 8) { half8_t o;
#pragma unroll
        for (int j = 0; j < V; j++) o[j] = (half_t)(MAX ? acc[j] : acc[j] * inv);
        *(half8_t*)py = o;
    } else py[0] = (half_t)(MAX ? acc[0] : acc[0] * inv);
}

// ---- MPSCNNSoftMaxNode / MPSCNNLogSoftMaxNode (:697-714, :1213-1232): over the channels of one pixel.  One wave per pixel,
// lanes stride over C; max and sum by cross-lane butterflies; fp32 math, one rounding to fp16. -------------------------------
__global__ __launch_bounds__(256) void softmax_kernel(const half_t* x, half_t* y, long pixels, int C, int x_ld, int y_ld, int logsm) {
    const int lane = threadIdx.x & 63;
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < pixels; m += (long)gridDim.x * 4) {
        const half_t* px = x + m * x_ld; half_t* py = y + m * y_ld;
        float mx = -3.0e38f;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, (float)px[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf((float)px[c] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum, lse = logf(sum);
        for (int c = lane; c < C; c += 64) { const float t = (float)px[c] - mx; py[c] = (half_t)(logsm ? t - lse : expf(t) * inv); }
    }
}

// ---- MPSCNNInstanceNormalizationNode (:992-1011) / MPSCNNGroupNormalizationNode (:1273-1300): one block per (image, group);
// two-pass mean / biased variance in fp32 over HW x (C/groups) elements, then normalise + per-channel gamma/beta. ------------
__global__ __launch_bounds__(256) void group_norm_kernel(const half_t* x, half_t* y, int HW, int C, int groups, const float* gamma, const float* beta,
                                                         float eps, int x_ld, int y_ld) {
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups, cg = C / groups, c0 = g * cg;
    const long base = (long)n * HW; const int total = HW * cg;
    __shared__ float red[4]; __shared__ float s_mean, s_rstd;
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float s = 0.f;
    for (int i = threadIdx.x; i < total; i += 256) { const int p = i / cg, c = i - p * cg; s += (float)x[(base + p) * x_ld + c0 + c]; }
    const float mean = block_sum(s) / (float)total;
    float q = 0.f;
    for (int i = threadIdx.x; i < total; i += 256) { const int p = i / cg, c = i - p * cg; const float d = (float)x[(base + p) * x_ld + c0 + c] - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(block_sum(q) / (float)total + eps);
    if (threadIdx.x == 0) { s_mean = mean; s_rstd = rstd; }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += 256) {
        const int p = i / cg, c = i - p * cg;
        const float v = ((float)x[(base + p) * x_ld + c0 + c] - s_mean) * s_rstd;
        y[(base + p) * y_ld + c0 + c] = (half_t)(v * (gamma ? gamma[c0 + c] : 1.f) + (beta ? beta[c0 + c] : 0.f));
    }
}

// ---- MPSNNPadNode (:944-990): constant 0 / reflect (mirror without repeating the edge) / edge (clamp) ----------------------
__global__ void pad_kernel(const half_t* x, half_t* y, int N, int H, int W, int C, int pt, int pl, int Ho, int Wo, int mode, int x_ld, int y_ld) {
    const long total = (long)N * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long m = i / C;
        const int ox = (int)(m % Wo); m /= Wo; const int oy = (int)(m % Ho), n = (int)(m / Ho);
        int iy = oy - pt, ix = ox - pl; bool zero = false;
        if (mode == SM_PAD_EDGE) { iy = min(max(iy, 0), H - 1); ix = min(max(ix, 0), W - 1); }
        else if (mode == SM_PAD_REFLECT) {
            if (H > 1) { const int per = 2 * (H - 1); iy = ((iy % per) + per) % per; if (iy >= H) iy = per - iy; } else iy = 0;
            if (W > 1) { const int per = 2 * (W - 1); ix = ((ix % per) + per) % per; if (ix >= W) ix = per - ix; } else ix = 0;
        } else zero = (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W;
        y[(((long)n * Ho + oy) * Wo + ox) * y_ld + c] = zero ? (half_t)0.f : x[(((long)n * H + iy) * W + ix) * x_ld + c];
    }
}

// ---- GlobalAveragePoolConverter (:578-605): mean over H*W.  One block per (image, 64-channel group):
// 4 waves stride over the pixels (lane = 8-channel vector ... C small) then combine through LDS. -----------
__global__ __launch_bounds__(256) void gap_kernel(const half_t* x, half_t* y, int HW, int C, int x_ld, int y_ld) {
    // block = (n, channel block of 64); thread t: channel = cb*64 + (t & 63), pixel lane = t >> 6 (0..3)
    const int cblocks = (C + 63) / 64, n = blockIdx.x / cblocks, cb = blockIdx.x % cblocks;
    const int c = cb * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    __shared__ float part[4][64];
    float s = 0.f;
    if (c < C) for (int p = pl; p < HW; p += 4) s += (float)x[((long)n * HW + p) * x_ld + c];
    part[pl][threadIdx.x & 63] = s;
    __syncthreads();
    if (pl == 0 && c < C) y[(long)n * y_ld + c] = (half_t)((part[0][c & 63] + part[1][c & 63] + part[2][c & 63] + part[3][c & 63]) / (float)HW);
}

// same, 8 channels (16 bytes) per thread: block = (image, group of 64 channel vectors = 512 channels), 4 pixel lanes; the sums are
// formed in the same order per channel as in gap_kernel (pixel lanes p = pl, pl+4, ..., then lane 0+1+2+3), so results are identical
__global__ __launch_bounds__(256) void gap_vec8_kernel(const half_t* x, half_t* y, int HW, int C, int x_ld, int y_ld) {
    const int cv = C >> 3, vblocks = (cv + 63) / 64, n = blockIdx.x / vblocks, vb = blockIdx.x % vblocks;
    const int v = vb * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    __shared__ float part[4][64][8];
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (v < cv) {
        const half_t* src = x + (long)n * HW * x_ld + v * 8;
        // 8 loads in flight per thread (a rolled loop pays one memory latency per pixel: 13 x ~0.5 us for a 7x7 map); the additions
        // stay in pixel order
        for (int p0 = pl; p0 < HW; p0 += 32) {
            half8_t t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int p = p0 + 4 * u; t[u] = *(const half8_t*)(src + (long)(p < HW ? p : pl) * x_ld); }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (p0 + 4 * u < HW) {
#pragma unroll
                    for (int j = 0; j < 8; j++) s[j] += (float)t[u][j];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) part[pl][threadIdx.x & 63][j] = s[j];
    __syncthreads();
    if (pl == 0 && v < cv) {
        half8_t o; const int l = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (half_t)((part[0][l][j] + part[1][l][j] + part[2][l][j] + part[3][l][j]) / (float)HW);
        *(half8_t*)(y + (long)n * y_ld + v * 8) = o;
    }
}

// ---- MPSCNNUpsamplingNearestNode / MPSCNNUpsamplingBilinearNode (:524, :531) ---------------------------
template <int V>
__global__ void upsample_kernel(const half_t* x, half_t* y, int N, int H, int W, int C, int sy, int sx, int bilinear, int align, int x_ld, int y_ld) {
    const int Ho = H * sy, Wo = W * sx, cv = (C + V - 1) / V;
    const long total = (long)N * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / cv; int c = (int)(i - m * cv) * V;
        int n = (int)(m / ((long)Ho * Wo)), r = (int)(m - (long)n * Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
        half_t* py = y + m * y_ld + c;
        const half_t* xb = x + (long)n * H * W * x_ld + c;
        if (!bilinear) {
            const half_t* px = xb + ((long)(oy / sy) * W + ox / sx) * x_ld;
            if (V == 8) *(half8_t*)py = *(const half8_t*)px; else py[0] = px[0];
            continue;
        }
        float fy, fx;
        if (align) { fy = Ho > 1 ? (float)oy * (float)(H - 1) / (float)(Ho - 1) : 0.f; fx = Wo > 1 ? (float)ox * (float)(W - 1) / (float)(Wo - 1) : 0.f; }
        else { fy = fmaxf(((float)oy + 0.5f) / (float)sy - 0.5f, 0.f); fx = fmaxf(((float)ox + 0.5f) / (float)sx - 0.5f, 0.f); }
        int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1), y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        float ly = fy - (float)y0, lx = fx - (float)x0;
        const half_t *p00 = xb + ((long)y0 * W + x0) * x_ld, *p01 = xb + ((long)y0 * W + x1) * x_ld;
        const half_t *p10 = xb + ((long)y1 * W + x0) * x_ld, *p11 = xb + ((long)y1 * W + x1) * x_ld;
        if (V == 8) {
            half8_t a = *(const half8_t*)p00, b = *(const half8_t*)p01, cc = *(const half8_t*)p10, dd = *(const half8_t*)p11, o;
#pragma unroll
            for (int j = 0; j < 8; j++) { float top = (float)a[j] + ((float)b[j] - (float)a[j]) * lx, bot = (float)cc[j] + ((float)dd[j] - (float)cc[j]) * lx; o[j] = (half_t)(top + (bot - top) * ly); }
            *(half8_t*)py = o;
        } else { float top = (float)p00[0] + ((float)p01[0] - (float)p00[0]) * lx, bot = (float)p10[0] + ((float)p11[0] - (float)p10[0]) * lx; py[0] = (half_t)(top + (bot - top) * ly); }
    }
}

// ---- MPSNNConcatenationNode (:564): strided channel-slice copy -------------------------------------------
template <int V>
__global__ void copy_channels_kernel(const half_t* x, half_t* y, long pixels, int C, int x_ld, int y_ld) {
    const int cv = (C + V - 1) / V; const long total = pixels * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / cv; int c = (int)(i - m * cv) * V;
        if (V == 8) *(half8_t*)(y + m * y_ld + c) = *(const half8_t*)(x + m * x_ld + c); else y[m * y_ld + c] = x[m * x_ld + c];
    }
}

bool vec8(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (C % 8) return false;
    for (int l : lds) if (l % 8) return false;
    for (const void* p : ptrs) if (p && (((uintptr_t)p) & 15)) return false;
    return true;
}

}  // namespace

extern "C" {

int sm_binary(int32_t op, const void* a, const void* b, void* y, int64_t pixels, int32_t C, int32_t a_ld, int32_t b_ld, int32_t y_ld, int32_t act, float slope, sm_stream_t st) {
    SM_REQUIRE(a && b && y && pixels > 0 && C > 0 && op >= SM_BIN_ADD && op <= SM_BIN_DIV, "sm_binary: bad argument");
    a_ld = a_ld ? a_ld : C; b_ld = b_ld ? b_ld : C; y_ld = y_ld ? y_ld : C; if (act == SM_ACT_RELU) slope = 0.f;
    if (vec8(C, {a_ld, b_ld, y_ld}, {a, b, y}))
        hipLaunchKernelGGL(add_vec_kernel, dim3(sm_stream_grid(pixels * (C / 8), 256)), dim3(256), 0, sm_s(st), (const half_t*)a, (const half_t*)b, (half_t*)y, (long)pixels, C / 8, a_ld, b_ld, y_ld, act, slope, op);
    else
        hipLaunchKernelGGL(add_scalar_kernel, dim3(sm_stream_grid(pixels * C, 256)), dim3(256), 0, sm_s(st), (const half_t*)a, (const half_t*)b, (half_t*)y, (long)pixels, C, a_ld, b_ld, y_ld, act, slope, op);
    SM_LAUNCH_CHECK(); return SM_OK;
}
int sm_add(const void* a, const void* b, void* y, int64_t pixels, int32_t C, int32_t a_ld, int32_t b_ld, int32_t y_ld, int32_t act, float slope, sm_stream_t st) {
    return sm_binary(SM_BIN_ADD, a, b, y, pixels, C, a_ld, b_ld, y_ld, act, slope, st);
}

int sm_batchnorm(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, const float* scale, const float* shift, int32_t act, float slope, sm_stream_t st) {
    SM_REQUIRE(x && y && pixels > 0 && C > 0 && ((scale == nullptr) == (shift == nullptr)), "sm_batchnorm: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C; if (act == SM_ACT_RELU) slope = 0.f;
    if (vec8(C, {x_ld, y_ld}, {x, y}))
        hipLaunchKernelGGL(affine_vec_kernel, dim3(sm_stream_grid(pixels * (C / 8), 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, (long)pixels, C / 8, x_ld, y_ld, scale, shift, act, slope);
    else
        hipLaunchKernelGGL(affine_scalar_kernel, dim3(sm_stream_grid(pixels * C, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, (long)pixels, C, x_ld, y_ld, scale, shift, act, slope);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_relu(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, float slope, sm_stream_t st) {
    return sm_batchnorm(x, y, pixels, C, x_ld, y_ld, nullptr, nullptr, SM_ACT_PRELU, slope, st);
}
int sm_neuron(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, int32_t act, float slope, sm_stream_t st) {
    SM_REQUIRE(act >= SM_ACT_NONE && act <= SM_ACT_CLIP, "sm_neuron: unknown activation");
    return sm_batchnorm(x, y, pixels, C, x_ld, y_ld, nullptr, nullptr, act, slope, st);
}

int sm_softmax(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, int32_t logsm, sm_stream_t st) {
    SM_REQUIRE(x && y && pixels > 0 && C > 0, "sm_softmax: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C;
    hipLaunchKernelGGL(softmax_kernel, dim3(sm_stream_grid(pixels * 64, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, (long)pixels, C, x_ld, y_ld, logsm);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_group_norm(const void* x, void* y, int32_t N, int32_t HW, int32_t C, int32_t groups, const float* gamma, const float* beta, float eps,
                  int32_t x_ld, int32_t y_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "sm_group_norm: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C;
    hipLaunchKernelGGL(group_norm_kernel, dim3(N * groups), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, HW, C, groups, gamma, beta, eps, x_ld, y_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_pad2d(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pt, int32_t pl, int32_t pb, int32_t pr, int32_t mode,
             int32_t x_ld, int32_t y_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && pt >= 0 && pl >= 0 && pb >= 0 && pr >= 0 && mode >= SM_PAD_CONSTANT && mode <= SM_PAD_EDGE,
               "sm_pad2d: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C;
    const int Ho = H + pt + pb, Wo = W + pl + pr;
    hipLaunchKernelGGL(pad_kernel, dim3(sm_stream_grid((long)N * Ho * Wo * C, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, N, H, W, C, pt, pl, Ho, Wo, mode, x_ld, y_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

static int pool_common(const sm_pool_desc* d0, const void* x, void* y, bool is_max, sm_stream_t st) {
    SM_REQUIRE(d0 && x && y && d0->N > 0 && d0->C > 0 && d0->Ho > 0 && d0->Wo > 0 && d0->kH > 0 && d0->kW > 0 && d0->sH > 0 && d0->sW > 0, "sm_pool: bad argument");
    sm_pool_desc d = *d0; d.x_ld = d.x_ld ? d.x_ld : d.C; d.y_ld = d.y_ld ? d.y_ld : d.C;
    SM_REQUIRE((long)d.N * d.Ho < (1L << 31), "sm_pool2d: too many output rows");
    if (is_max && vec8(d.C, {d.x_ld, d.y_ld}, {x, y}) && ((d.kH == 2 && d.kW == 2) || (d.kH == 3 && d.kW == 3))) {
        const long per_row = (long)d.Wo * (d.C / 8);
        const int opt = d.kH == 2 ? (per_row >= 1024 ? 4 : 1) : (per_row >= 512 ? 2 : 1);
        dim3 g(sm_ceil_div((long)d.N * d.Ho, SM_NUM_XCD) * SM_NUM_XCD, sm_ceil_div(per_row, 256 * opt));
        if (d.kH == 2 && opt == 4) hipLaunchKernelGGL((maxpool_small_kernel<2, 2, 4>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
        else if (d.kH == 2) hipLaunchKernelGGL((maxpool_small_kernel<2, 2, 1>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
        else if (opt == 2) hipLaunchKernelGGL((maxpool_small_kernel<3, 3, 2>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
        else hipLaunchKernelGGL((maxpool_small_kernel<3, 3, 1>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (vec8(d.C, {d.x_ld, d.y_ld}, {x, y})) {
        dim3 g(sm_ceil_div((long)d.N * d.Ho, SM_NUM_XCD) * SM_NUM_XCD, sm_ceil_div((long)d.Wo * (d.C / 8), 256));
        if (is_max) hipLaunchKernelGGL((pool_kernel<true, 8>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
        else hipLaunchKernelGGL((pool_kernel<false, 8>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
    } else {
        dim3 g(sm_ceil_div((long)d.N * d.Ho, SM_NUM_XCD) * SM_NUM_XCD, sm_ceil_div((long)d.Wo * d.C, 256));
        if (is_max) hipLaunchKernelGGL((pool_kernel<true, 1>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
        else hipLaunchKernelGGL((pool_kernel<false, 1>), g, dim3(256), 0, sm_s(st), d, (const half_t*)x, (half_t*)y);
    }
    SM_LAUNCH_CHECK(); return SM_OK;
}
int sm_maxpool2d(const sm_pool_desc* d, const void* x, void* y, sm_stream_t st) { return pool_common(d, x, y, true, st); }
int sm_avgpool2d(const sm_pool_desc* d, const void* x, void* y, sm_stream_t st) { return pool_common(d, x, y, false, st); }

int sm_global_avgpool(const void* x, void* y, int32_t N, int32_t HW, int32_t C, int32_t x_ld, int32_t y_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "sm_global_avgpool: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C;
    if (vec8(C, {x_ld, y_ld}, {x, y})) {                            // 16-byte loads: 64 channel vectors x 4 pixel lanes per block
        hipLaunchKernelGGL(gap_vec8_kernel, dim3(N * ((C / 8 + 63) / 64)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, HW, C, x_ld, y_ld);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    hipLaunchKernelGGL(gap_kernel, dim3(N * ((C + 63) / 64)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, HW, C, x_ld, y_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

static int upsample_common(const void* x, void* y, int N, int H, int W, int C, int sy, int sx, int bil, int align, int x_ld, int y_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && sy > 0 && sx > 0, "sm_upsample: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C;
    long pix = (long)N * H * sy * W * sx;
    if (vec8(C, {x_ld, y_ld}, {x, y}))
        hipLaunchKernelGGL((upsample_kernel<8>), dim3(sm_stream_grid(pix * (C / 8), 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, N, H, W, C, sy, sx, bil, align, x_ld, y_ld);
    else
        hipLaunchKernelGGL((upsample_kernel<1>), dim3(sm_stream_grid(pix * C, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, N, H, W, C, sy, sx, bil, align, x_ld, y_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}
int sm_upsample_nearest(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t sy, int32_t sx, int32_t x_ld, int32_t y_ld, sm_stream_t st) {
    return upsample_common(x, y, N, H, W, C, sy, sx, 0, 0, x_ld, y_ld, st);
}
int sm_upsample_bilinear(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t sy, int32_t sx, int32_t align, int32_t x_ld, int32_t y_ld, sm_stream_t st) {
    return upsample_common(x, y, N, H, W, C, sy, sx, 1, align, x_ld, y_ld, st);
}

int sm_copy_channels(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && pixels > 0 && C > 0, "sm_copy_channels: bad argument");
    x_ld = x_ld ? x_ld : C; y_ld = y_ld ? y_ld : C;
    if (vec8(C, {x_ld, y_ld}, {x, y}))
        hipLaunchKernelGGL((copy_channels_kernel<8>), dim3(sm_stream_grid(pixels * (C / 8), 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, (long)pixels, C, x_ld, y_ld);
    else
        hipLaunchKernelGGL((copy_channels_kernel<1>), dim3(sm_stream_grid(pixels * C, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, (long)pixels, C, x_ld, y_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

}  // extern "C"
