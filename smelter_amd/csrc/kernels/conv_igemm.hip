// conv_igemm.hip — implicit-GEMM convolution / fully-connected on MFMA for gfx950 (CDNA4).
//
// Replaces MPSCNNConvolutionNode / MPSCNNFullyConnectedNode (/root/reference/Sources/Smelter/Converters.swift
// :253, :297) together with the epilogue nodes MPS would fuse or run separately: bias (biasTerms(), :166),
// residual MPSNNAdditionNode (:440) and MPSCNNNeuronReLUNode (:351).
//
// GEMM view (per group = 1):   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],   k = (kh*kW + kw)*Cin + ci
//   * W is OHWI fp16 = [Cout][K] with K contiguous — the layout Smelter's preprocessing produces
//     (Array+Extensions.swift:78-83), so weights are the MFMA "A" operand as stored.
//   * X rows are gathered on the fly from the NHWC fp16 input: for Cin % 8 == 0 every 16-byte vector of a
//     row lies inside one (kh,kw) tap, i.e. is 8 contiguous channels of one input pixel (or zero padding).
//   * operands are swapped w.r.t. the usual "pixels = rows" so that each lane ends up holding 4 consecutive
//     output channels of one pixel (32x32 C/D layout: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)),
//     which are contiguous in the NHWC output -> 8-byte stores, no LDS transpose.
//
// Tile: BM pixels x BN couts x BK=32, 256 threads = 4 waves arranged WGM x WGN, each wave TM x TN MFMA tiles of
// 32x32 (v_mfma_f32_32x32x16_f16, fp32 accumulate).  Global->register->LDS staging with the next tile's loads
// in flight during the MFMAs (one barrier per K tile, double-buffered LDS).  LDS rows are 64 B; 16-byte chunks
// are XOR-swizzled (chunk ^ f((row>>2)&3), f = 0,2,3,1) so that every 16-lane service group of ds_read_b128
// touches 16 distinct 16-byte bank slots (MI355X_MICROARCH.md §LDS).
//
// Block -> tile map is XCD aware (8 XCDs with private L2, block b runs on XCD b%8): all BN-tiles of one pixel
// tile are issued back-to-back on the same XCD so the gathered activation tile is fetched from HBM once.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "conv_shared.hpp"

namespace {

// pixel index of output row m (= (n, oy, ox) of the Ho x Wo grid) inside the destination image
__device__ __forceinline__ long sm_out_pixel(const ConvParams& p, int m) {
    if (!p.yplace) return m;
    const int n = (int)sm_fastdiv((unsigned)m, p.dhw_mul, p.dhw_sh), r = m - n * (p.Ho * p.Wo);
    const int oy = (int)sm_fastdiv((unsigned)r, p.dw_mul, p.dw_sh), ox = r - oy * p.Wo;
    return ((long)n * p.yH + oy * p.ysH + p.yoH) * p.yW + ox * p.ysW + p.yoW;
}

constexpr int BK = 32;

// Block -> (pixel tile, cout tile, k split).  MI355X has 8 XCDs with private 4 MiB L2s and block b runs on XCD b % 8,
// so the map decides what each L2 holds:
//   order 0 (activations dominate): XCD x owns the CONTIGUOUS pixel-tile range [x*mt/8, (x+1)*mt/8) — neighbouring
//            tiles share their 3x3 halo rows inside one L2 — and walks all cout tiles of a pixel tile back to back, so
//            the gathered activation tile is fetched from HBM once.
//   order 1 (weights dominate, e.g. 512x4608 at 7x7): XCD x owns a contiguous range of (cout tile, k split) combos and
//            walks every pixel tile for it, so its weight slice stays L2-resident while the small activation tensor is
//            shared by all XCDs.
__device__ __forceinline__ bool map_block(const ConvParams& p, int& mtile, int& ntile, int& ksplit) {
    const int bid = blockIdx.x, xcd = bid & (SM_NUM_XCD - 1), slot = bid >> 3;
    const int per_m = p.nt * p.splitk;
    int combo;
    if (p.order == 0) {
        const int q = (int)sm_fastdiv((unsigned)slot, p.pm_mul, p.pm_sh);
        mtile = xcd * p.mt_per_xcd + q; combo = slot - q * per_m; if (q >= p.mt_per_xcd) return false;
    } else {
        const int q = (int)sm_fastdiv((unsigned)slot, p.mt_mul, p.mt_sh);
        combo = xcd * p.cpx + q; mtile = slot - q * p.mt; if (q >= p.cpx || combo >= per_m) return false;
    }
    if (mtile >= p.mt) return false;
    ksplit = (int)sm_fastdiv((unsigned)combo, p.nt_mul, p.nt_sh); ntile = combo - ksplit * p.nt;
    return true;
}

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
    constexpr int A_IT = BM * 4 / 256, B_IT = BN * 4 / 256;
    static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "tile too small");
    // the ring (2 stages of (BM+BN) x 32 halfs) and the fp32 staging tile of the coalesced epilogue share the LDS
    constexpr int RING_HALFS = 2 * (BM + BN) * BK, EPI_LD = BN + 4, EPI_HALFS = BM * EPI_LD * 2;
    __shared__ __attribute__((aligned(16))) half_t smem_all[RING_HALFS > EPI_HALFS ? RING_HALFS : EPI_HALFS];
    half_t (*smem)[(BM + BN) * BK] = (half_t (*)[(BM + BN) * BK])smem_all;

    // ---- block -> (pixel tile, cout tile, k split) -------------------------------------------------
    int mtile, ntile, ksplit;
    if (!map_block(p, mtile, ntile, ksplit)) return;
    const int kt_begin = ksplit * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    const int ktiles = (p.K + BK - 1) / BK;
    if (kt_end > ktiles) kt_end = ktiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int chunk = tid & 3;              // 16-byte chunk of the 64-byte K row this thread stages

    // ---- per-thread gather state for the activation rows it stages ---------------------------------
    int iy0[A_IT], ix0[A_IT]; long nb[A_IT]; bool rv[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        int row = (tid >> 2) + i * 64;
        int m = mtile * BM + row;
        rv[i] = m < p.M;
        int mm = rv[i] ? m : 0;
        int n = mm / (p.Ho * p.Wo), r = mm - n * (p.Ho * p.Wo);
        int oy = r / p.Wo, ox = r - oy * p.Wo;
        iy0[i] = oy * p.sH - p.pt; ix0[i] = ox * p.sW - p.pl; nb[i] = (long)n * p.H * p.W;
    }
    // position of this thread's chunk inside K: k = kt*32 + chunk*8 -> (kh, kw, ci)
    int k = kt_begin * BK + chunk * 8;
    int tap = k / p.Cin, ci = k - tap * p.Cin, kh = tap / p.kW, kw = tap - kh * p.kW;

    const half_t* wrow[B_IT]; bool wv[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        int n = ntile * BN + (tid >> 2) + i * 64;
        wv[i] = n < p.Cout;
        wrow[i] = p.w + (long)(wv[i] ? n : 0) * p.K;
    }

    half8_t ra[A_IT], rb[B_IT];
    auto load_tile = [&]() {
        const bool kin = k < p.K;
        const int dy = kh * p.dH, dx = kw * p.dW;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            int iy = iy0[i] + dy, ix = ix0[i] + dx;
            bool ok = rv[i] && kin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) v = *(const half8_t*)(p.x + (nb[i] + (long)iy * p.W + ix) * p.x_ld + ci);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (wv[i] && kin) v = *(const half8_t*)(wrow[i] + k);
            rb[i] = v;
        }
        // advance to the next K tile
        k += BK; ci += BK;
        while (ci >= p.Cin) { ci -= p.Cin; if (++kw == p.kW) { kw = 0; ++kh; } }
    };
    auto store_tile = [&](int buf) {
        half_t* sa = smem[buf]; half_t* sb = smem[buf] + BM * BK;
#pragma unroll
        for (int i = 0; i < A_IT; i++) { int row = (tid >> 2) + i * 64; *(half8_t*)(sa + row * BK + swz(row, chunk) * 8) = ra[i]; }
#pragma unroll
        for (int i = 0; i < B_IT; i++) { int row = (tid >> 2) + i * 64; *(half8_t*)(sb + row * BK + swz(row, chunk) * 8) = rb[i]; }
    };

    floatx16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // ---- main loop --------------------------------------------------------------------------------
    load_tile(); store_tile(0); __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; kt++) {
        const bool more = kt + 1 < kt_end;
        if (more) load_tile();
        const half_t* sa = smem[cur]; const half_t* sb = smem[cur] + BM * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int c = ks * 2 + (lane >> 5);
            half8_t fx[TM], fw[TN];
#pragma unroll
            for (int b = 0; b < TM; b++) { int row = (wm * TM + b) * 32 + (lane & 31); fx[b] = *(const half8_t*)(sa + row * BK + swz(row, c) * 8); }
#pragma unroll
            for (int a = 0; a < TN; a++) { int row = (wn * TN + a) * 32 + (lane & 31); fw[a] = *(const half8_t*)(sb + row * BK + swz(row, c) * 8); }
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a], fx[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    // Fast path (dense 16-byte output, no placement / split-K, ReLU family): accumulators -> LDS as fp32 [BM][BN+4], then every
    // thread takes 8 consecutive couts of one pixel: + bias + residual, one rounding, ONE 16-byte store, 8 lanes per 128-byte
    // line.  (The row-per-lane 8-byte stores below put 32 partial lines into every store instruction: the pointwise layers of
    // MobileNet-v2, all epilogue, ran at 2.1 TB/s.)
    if (!p.yplace && p.splitk == 1 && p.act <= SM_ACT_PRELU && (p.Cout & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) &&
        ((((uintptr_t)p.bias) & 15) == 0) && (p.res == nullptr || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0))) {
        float* stg = (float*)smem_all;                            // (the last main-loop barrier already separates it from the ring reads)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = (wm * TM + b) * 32 + (lane & 31), col = (wn * TN + a) * 32 + 8 * q + 4 * (lane >> 5);
                    *(floatx4*)(stg + row * EPI_LD + col) = floatx4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                }
        __syncthreads();
        const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
        constexpr int CH = BN / 8, ITEMS = BM * CH, IPT = ITEMS / 256;
        static_assert(ITEMS % 256 == 0 && 256 % CH == 0, "epilogue items");
        const int ch = tid % CH, co = ntile * BN + ch * 8;
        if (co < p.Cout) {
            floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (p.bias) { b0 = *(const floatx4*)(p.bias + co); b1 = *(const floatx4*)(p.bias + co + 4); }
            floatx4 s0[IPT], s1[IPT]; half8_t rr[IPT];
#pragma unroll
            for (int k2 = 0; k2 < IPT; k2++) {
                const int row = (tid + k2 * 256) / CH;
                s0[k2] = *(const floatx4*)(stg + row * EPI_LD + ch * 8); s1[k2] = *(const floatx4*)(stg + row * EPI_LD + ch * 8 + 4);
                const int m = min(mtile * BM + row, p.M - 1);
                if (p.res) rr[k2] = *(const half8_t*)(p.res + (long)m * p.res_ld + co);
            }
#pragma unroll
            for (int k2 = 0; k2 < IPT; k2++) {
                const int m = mtile * BM + (tid + k2 * 256) / CH;
                float v[8] = {s0[k2][0] + b0[0], s0[k2][1] + b0[1], s0[k2][2] + b0[2], s0[k2][3] + b0[3],
                              s1[k2][0] + b1[0], s1[k2][1] + b1[1], s1[k2][2] + b1[2], s1[k2][3] + b1[3]};
                if (p.res) {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] += (float)rr[k2][j]; }
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = (half_t)(v[j] < 0.f ? v[j] * sl : v[j]);
                if (m < p.M) *(half8_t*)(p.y + (long)m * p.y_ld + co) = o;
            }
        }
        return;
    }
    const bool vec_ok = ((p.y_ld & 3) == 0) && ((((uintptr_t)p.y) & 7) == 0) && ((((uintptr_t)p.bias) & 15) == 0) &&
                        (p.res == nullptr || ((p.res_ld & 3) == 0 && (((uintptr_t)p.res) & 7) == 0));
#pragma unroll
    for (int b = 0; b < TM; b++) {
        const int m = mtile * BM + (wm * TM + b) * 32 + (lane & 31);
        if (m >= p.M) continue;
        const long opix = sm_out_pixel(p, m);
#pragma unroll
        for (int a = 0; a < TN; a++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = ntile * BN + (wn * TN + a) * 32 + 8 * q + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                float v[4] = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (p.splitk > 1) {   // fp32 partials, reduced by splitk_reduce_kernel
                    float* dst = p.ws + ((long)ksplit * p.M + m) * p.Cout + co;
                    if (co + 3 < p.Cout && (p.Cout & 3) == 0) *(floatx4*)dst = floatx4{v[0], v[1], v[2], v[3]};
                    else for (int j = 0; j < 4 && co + j < p.Cout; j++) dst[j] = v[j];
                    continue;
                }
                if (co + 3 < p.Cout && vec_ok) {
                    if (p.bias) { floatx4 bb = *(const floatx4*)(p.bias + co); v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3]; }
                    if (p.res) { half4_t rr = *(const half4_t*)(p.res + (long)m * p.res_ld + co); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
                    half4_t o;
#pragma unroll
                    for (int j = 0; j < 4; j++) o[j] = (half_t)sm_act(v[j], p.act, p.slope);
                    *(half4_t*)(p.y + opix * p.y_ld + co) = o;
                } else {
                    for (int j = 0; j < 4 && co + j < p.Cout; j++) {
                        float t = v[j];
                        if (p.bias) t += p.bias[co + j];
                        if (p.res) t += (float)p.res[(long)m * p.res_ld + co + j];
                        p.y[opix * p.y_ld + co + j] = (half_t)sm_act(t, p.act, p.slope);
                    }
                }
            }
        }
    }
}

// split-K second pass: sum fp32 partials, bias/residual/activation, fp16 store.  Vector form: 8 consecutive couts per
// thread (two 16-byte loads per split, one 16-byte store); scalar form for odd shapes.
__global__ __launch_bounds__(256) void splitk_reduce_vec8_kernel(ConvParams p) {
    const int cv = p.Cout >> 3;
    const long items = (long)p.M * cv, total = (long)p.M * p.Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        const long m = i / cv; const int co = (int)(i - m * cv) * 8;
        const float* src = p.ws + m * p.Cout + co;
        floatx4 a0 = *(const floatx4*)src, a1 = *(const floatx4*)(src + 4);
        // four splits' loads in flight at a time (a rolled loop pays one memory latency per split); additions stay in split order
        int sp = 1;
        for (; sp + 3 < p.splitk; sp += 4) {
            floatx4 u0[4], u1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { u0[q] = *(const floatx4*)(src + (long)(sp + q) * total); u1[q] = *(const floatx4*)(src + (long)(sp + q) * total + 4); }
#pragma unroll
            for (int q = 0; q < 4; q++) { a0 += u0[q]; a1 += u1[q]; }
        }
        for (; sp < p.splitk; sp++) { a0 += *(const floatx4*)(src + (long)sp * total); a1 += *(const floatx4*)(src + (long)sp * total + 4); }
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (p.bias) { const floatx4 b0 = *(const floatx4*)(p.bias + co), b1 = *(const floatx4*)(p.bias + co + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) { v[j] += b0[j]; v[4 + j] += b1[j]; } }
        if (p.res) { const half8_t r = *(const half8_t*)(p.res + m * p.res_ld + co);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += (float)r[j]; }
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (half_t)sm_act(v[j], p.act, p.slope);
        *(half8_t*)(p.y + m * p.y_ld + co) = o;
    }
}
__global__ void splitk_reduce_kernel(ConvParams p) {
    long total = (long)p.M * p.Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / p.Cout; int co = (int)(i - m * p.Cout);
        float t = 0.f;
        for (int s = 0; s < p.splitk; s++) t += p.ws[(long)s * total + i];
        if (p.bias) t += p.bias[co];
        if (p.res) t += (float)p.res[m * p.res_ld + co];
        p.y[m * p.y_ld + co] = (half_t)sm_act(t, p.act, p.slope);
    }
}

// Direct convolution: correctness fallback for shapes the MFMA path does not take (Cin % 8 != 0, grouped
// non-depthwise).  One thread per output element, fp32 accumulate in (kh, kw, ci) order.
__global__ void conv_direct_kernel(ConvParams p, int groups) {
    const long total = (long)p.M * p.Cout;
    const int Cg = p.Cin / groups, Kg = p.Cout / groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / p.Cout; int co = (int)(i - m * p.Cout);
        int n = (int)(m / (p.Ho * p.Wo)), r = (int)(m - (long)n * p.Ho * p.Wo), oy = r / p.Wo, ox = r - oy * p.Wo;
        int g = co / Kg;
        float acc = 0.f;
        for (int kh = 0; kh < p.kH; kh++) {
            int iy = oy * p.sH - p.pt + kh * p.dH; if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.kW; kw++) {
                int ix = ox * p.sW - p.pl + kw * p.dW; if ((unsigned)ix >= (unsigned)p.W) continue;
                const half_t* xp = p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld + g * Cg;
                const half_t* wp = p.w + (((long)co * p.kH + kh) * p.kW + kw) * Cg;
                for (int c = 0; c < Cg; c++) acc += (float)xp[c] * (float)wp[c];
            }
        }
        if (p.bias) acc += p.bias[co];
        if (p.res) acc += (float)p.res[m * p.res_ld + co];
        p.y[sm_out_pixel(p, (int)m) * p.y_ld + co] = (half_t)sm_act(acc, p.act, p.slope);
    }
}

// Transposed convolution (MPSCNNConvolutionTransposeNode, Converters.swift:277): direct gather over the data source's
// flipped OHWI weights, one thread per output element, fp32 accumulate.  Input row iy feeds output row oy through tap kh
// iff iy*sH == oy + pt - (kH-1-kh)*dH.  Any shape; the compiled graph uses per-phase MFMA convolutions where it can.
__global__ void convt_direct_kernel(ConvParams p) {
    const long total = (long)p.M * p.Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / p.Cout; int co = (int)(i - m * p.Cout);
        int n = (int)(m / (p.Ho * p.Wo)), r = (int)(m - (long)n * p.Ho * p.Wo), oy = r / p.Wo, ox = r - oy * p.Wo;
        float acc = 0.f;
        for (int kh = 0; kh < p.kH; kh++) {
            int ty = oy + p.pt - (p.kH - 1 - kh) * p.dH; if (ty < 0 || ty % p.sH) continue;
            int iy = ty / p.sH; if (iy >= p.H) continue;
            for (int kw = 0; kw < p.kW; kw++) {
                int tx = ox + p.pl - (p.kW - 1 - kw) * p.dW; if (tx < 0 || tx % p.sW) continue;
                int ix = tx / p.sW; if (ix >= p.W) continue;
                const half_t* xp = p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld;
                const half_t* wp = p.w + (((long)co * p.kH + kh) * p.kW + kw) * p.Cin;
                for (int c = 0; c < p.Cin; c++) acc += (float)xp[c] * (float)wp[c];
            }
        }
        if (p.bias) acc += p.bias[co];
        p.y[m * p.y_ld + co] = (half_t)sm_act(acc, p.act, p.slope);
    }
}

// Depthwise convolution (groups == Cin == Cout, MPSCNNDepthWiseConvolutionDescriptor, Converters.swift:57-66).
// HBM-bound VALU kernel: each thread owns 8 consecutive channels (16-byte loads/stores) of one output pixel;
// the kH*kW per-channel taps are read from the OHWI weight [C][kH][kW][1] through L1/L2.
__global__ __launch_bounds__(256) void dwconv_kernel(ConvParams p) {
    const int cv = p.Cout >> 3;                       // channel vectors per pixel
    const long total = (long)p.M * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / cv; int c0 = (int)(i - m * cv) * 8;
        int n = (int)(m / (p.Ho * p.Wo)), r = (int)(m - (long)n * p.Ho * p.Wo), oy = r / p.Wo, ox = r - oy * p.Wo;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = p.bias ? p.bias[c0 + j] : 0.f;
        const int taps = p.kH * p.kW;
        for (int kh = 0; kh < p.kH; kh++) {
            int iy = oy * p.sH - p.pt + kh * p.dH; if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.kW; kw++) {
                int ix = ox * p.sW - p.pl + kw * p.dW; if ((unsigned)ix >= (unsigned)p.W) continue;
                half8_t xv = *(const half8_t*)(p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld + c0);
                const half_t* wp = p.w + (long)c0 * taps + kh * p.kW + kw;
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += (float)xv[j] * (float)wp[j * taps];
            }
        }
        half8_t o;
        if (p.res) { half8_t rr = *(const half8_t*)(p.res + m * p.res_ld + c0);
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += (float)rr[j]; }
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (half_t)sm_act(acc[j], p.act, p.slope);
        *(half8_t*)(p.y + m * p.y_ld + c0) = o;
    }
}

// 3x3 depthwise, stride S (MobileNet-v2: 17 layers, all HBM-bound: 2 bytes in + 2 bytes out per 18 FLOPs).
// One thread = 8 consecutive channels x a strip of PX output pixels of one row.  The 72 weights of its 8 channels are
// 144 CONTIGUOUS bytes of the OHWI tensor [C][3][3][1] -> nine 16-byte loads (the generic kernel issued 72 scalar ones);
// the (PX-1)*S+3 input columns of each of the 3 rows are loaded once (16 B each, zero outside the image) and shared by
// the PX outputs; consecutive threads take consecutive channel vectors of the same strip -> fully coalesced.
template <int S, int PX>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(ConvParams p) {
    constexpr int IW = (PX - 1) * S + 3;
    const int cv = p.Cout >> 3, strips = (p.Wo + PX - 1) / PX;
    const long total = (long)p.N * p.Ho * strips * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * 8; long t = i / cv;
        const int st = (int)(t % strips); t /= strips;
        const int oy = (int)(t % p.Ho), n = (int)(t / p.Ho);
        const int ox0 = st * PX, iy0 = oy * S - p.pt, ix0 = ox0 * S - p.pl;
        half8_t wv[9];
        const half8_t* wp = (const half8_t*)(p.w + (long)c0 * 9);
#pragma unroll
        for (int q = 0; q < 9; q++) wv[q] = wp[q];
        float acc[PX][8];
#pragma unroll
        for (int o = 0; o < PX; o++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[o][j] = p.bias ? p.bias[c0 + j] : 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int iy = iy0 + kh;
            const bool rowok = (unsigned)iy < (unsigned)p.H;
            half8_t xin[IW];
#pragma unroll
            for (int c = 0; c < IW; c++) {
                const int ix = ix0 + c;
                half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (rowok && (unsigned)ix < (unsigned)p.W) v = *(const half8_t*)(p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld + c0);
                xin[c] = v;
            }
#pragma unroll
            for (int o = 0; o < PX; o++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        constexpr int dummy = 0; (void)dummy;
                        const int widx = j * 9 + kh * 3 + kw;                 // flat index into the 72 weights of this thread
                        acc[o][j] += (float)xin[o * S + kw][j] * (float)wv[widx >> 3][widx & 7];
                    }
        }
#pragma unroll
        for (int o = 0; o < PX; o++) {
            const int ox = ox0 + o;
            if (ox >= p.Wo) break;
            const long m = ((long)n * p.Ho + oy) * p.Wo + ox;
            if (p.res) { half8_t rr = *(const half8_t*)(p.res + m * p.res_ld + c0);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[o][j] += (float)rr[j]; }
            half8_t ov;
#pragma unroll
            for (int j = 0; j < 8; j++) ov[j] = (half_t)sm_act(acc[o][j], p.act, p.slope);
            *(half8_t*)(p.y + m * p.y_ld + c0) = ov;
        }
    }
}

#include "conv_igemm64.hpp"
#include "conv3x3_strip.hpp"
#include "pw_thin.hpp"
#include "conv3x3_ws.hpp"
#include "conv3x3_wsx.hpp"
#include "fc_small.hpp"
#include "conv3x3_p9.hpp"
#include "conv1x1_ps.hpp"
#include "conv1x1_rw.hpp"
#include "stem_pool.hpp"

// 3x3 depthwise through LDS (the MobileNet-v2 path): the register-strip kernel above re-reads every input pixel 4.5x through
// L1/L2 and needs 227 VGPRs (1.2..2 TB/s of algorithmic bytes).  Here a block owns a tile of TH x TW output pixels x `cbv`
// channel vectors (8 channels = 16 B each; all channels of the layer whenever the patch fits, otherwise a multiple of 8
// vectors so that a block reads whole 128-byte lines): the (TH*S+2) x (TW*S+2) input patch goes to LDS by the LDS-DMA (slot
// L = tid + 256 k holds patch pixel L / cbv, vector L % cbv: lane-linear; zero page outside the image; all loads of a
// thread in flight at once), then every thread owns ONE channel vector (its 72 weights live in registers, regrouped per tap)
// and walks the output pixels of the tile with stride 256/cbv: nine ds_read_b128 + 72 v_fma_mix_f32 per 16-byte store.
// Lanes are adjacent in channel, so LDS and global accesses are contiguous 16-byte runs.  Blocks of one spatial tile are
// adjacent in the grid (channel block fastest).  dynamic LDS = IH*IW*cbv*16 bytes rounded up to whole 1 KB wave images.
template <int S, int TH, int TW, int CBMAX>
__global__ __launch_bounds__(256) void dwconv3x3_lds_kernel(ConvParams p, int cbv, int tiles_x, int tiles_y, int cblocks) {
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    int b = blockIdx.x;
    const int cb = b % cblocks; b /= cblocks;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int tid = threadIdx.x, npt = 256 / cbv;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool active = tid < npt * cbv;                          // 256 % cbv threads have no output column (they still load and reach the barrier)
    const int pt = active ? tid / cbv : 0, v = active ? tid - pt * cbv : 0;
    const int c0 = (cb * cbv + v) * 8;
    {
        const int iy0 = ty * TH * S - p.pt, ix0 = tx * TW * S - p.pl;
        const half_t* img = p.x + (long)n * p.H * p.W * p.x_ld + cb * cbv * 8;
        const int slots = IH * IW * cbv;
        constexpr int NL = (IH * IW * CBMAX + 255) / 256;
        int idx = tid / cbv, vv = tid - idx * cbv;
        const int didx = 256 / cbv, dv = 256 - didx * cbv;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (k * 256 + wave * 64 < slots) {                    // wave-uniform: waves beyond the patch skip
                const int r = idx / IW, c = idx - r * IW, iy = iy0 + r, ix = ix0 + c;
                const bool in = idx < IH * IW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const half_t* src = in ? img + ((long)iy * p.W + ix) * p.x_ld + vv * 8 : (const half_t*)g_zero_page;
                SM_GLDS16(src, dw_smem + (k * 256 + wave * 64) * 16);
            }
            idx += didx; vv += dv; if (vv >= cbv) { vv -= cbv; idx++; }
        }
    }
    // weights [c][3][3] of the thread's 8 channels: 72 contiguous halfs -> one vector of 8 channels per tap
    half8_t wt[9];
    {
        half8_t wq[9];
        const half8_t* wp = (const half8_t*)(p.w + (long)c0 * 9);
#pragma unroll
        for (int q = 0; q < 9; q++) wq[q] = wp[q];
#pragma unroll
        for (int tp = 0; tp < 9; tp++)
#pragma unroll
            for (int j = 0; j < 8; j++) wt[tp][j] = wq[(j * 9 + tp) >> 3][(j * 9 + tp) & 7];
    }
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; j++) bias[j] = p.bias ? p.bias[c0 + j] : 0.f;
    const bool relu_family = p.act <= SM_ACT_PRELU;
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const half8_t* tile = (const half8_t*)dw_smem;
    for (int o = active ? pt : TH * TW; o < TH * TW; o += npt) {
        const int oy = o / TW, ox = o - oy * TW, gy = ty * TH + oy, gx = tx * TW + ox;
        if (gy >= p.Ho || gx >= p.Wo) continue;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = bias[j];
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                // acc[j] += x[j] * w[j]: v_fma_mix_f32 (fp16 operands, fp32 accumulate; the product of two halfs is exact in
                // fp32, so this equals mul + add bit for bit).  Written out because hipcc otherwise converts the 72 loop-invariant
                // weights to fp32 registers: 249 VGPRs, two waves per SIMD.
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 xu = *(const u32x4*)&tile[((oy * S + kh) * IW + ox * S + kw) * cbv + v];
                const u32x4 wu = *(const u32x4*)&wt[kh * 3 + kw];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * d]) : "v"(xu[d]), "v"(wu[d]));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * d + 1]) : "v"(xu[d]), "v"(wu[d]));
                }
            }
        const long m = ((long)n * p.Ho + gy) * p.Wo + gx;
        if (p.res) { const half8_t rr = *(const half8_t*)(p.res + m * p.res_ld + c0);
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += (float)rr[j]; }
        half8_t ov;
        if (relu_family) {
#pragma unroll
            for (int j = 0; j < 8; j++) ov[j] = (half_t)(acc[j] < 0.f ? acc[j] * sl : acc[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) ov[j] = (half_t)sm_act(acc[j], p.act, p.slope);
        }
        *(half8_t*)(p.y + m * p.y_ld + c0) = ov;
    }
}

// kind 0: generic (Cin % 8 == 0, BK = 32, register staging); kind 1: Cin % 64 == 0 fast path (BK = 64, LDS-DMA)
// flags / p0..p2 carry the geometry the launcher needs (never parsed out of `name`, which is a label only):
//   kind 4: p0 = Cin of the instantiation; CF_HEAD / CF_XHEAD / CF_W44 (two-role kernel: 4 of the 8 waves issue the patch DMA) / CF_SIDE
//   kind 6: p0 = taps (kH' of the row-pair form), p1 = KB between the two patch buffers (0 = as large as the weights allow), p2 = loader waves
//   kinds 5, 7: p0 = ring stages
//   kind 9: flags = patch buffers (1 | 2) | CF_KQ_S2 (the stride-2 instantiations), p0 = KG, p1 = NG, p2 = NF, BM / 16 = NFR
//   kind 8: BM = pixels per step (32 FS), BN = channels per block, p0 = Cin of the instantiation; CF_RES = the instantiation that stages a residual tile;
//           ring form: p1 = stages, p2 = parts the weight slice passes the LDS in (0 = 1); CF_DUAL = takes (only) launches with a second source
enum { CF_HEAD = 1, CF_XHEAD = 2, CF_W44 = 4, CF_SIDE = 8, CF_RES = 16, CF_DUAL = 32, CF_POOL = 64, CF_KQ_S2 = 128 };
struct TileCfg { int BM, BN, kind, threads; void (*fn)(ConvParams); const char* name; int flags = 0, p0 = 0, p1 = 0, p2 = 0; };
const TileCfg kCfgs[] = {
    {128, 128, 0, 256, conv_igemm_kernel<128, 128, 2, 2>, "128x128"},
    {128, 64, 0, 256, conv_igemm_kernel<128, 64, 2, 2>, "128x64"},
    {64, 128, 0, 256, conv_igemm_kernel<64, 128, 2, 2>, "64x128"},
    {64, 64, 0, 256, conv_igemm_kernel<64, 64, 2, 2>, "64x64"},
    {256, 64, 0, 256, conv_igemm_kernel<256, 64, 4, 1>, "256x64"},
    {256, 128, 0, 256, conv_igemm_kernel<256, 128, 2, 2>, "256x128"},
    {128, 128, 1, 256, conv_igemm64_kernel<128, 128, 2, 2>, "k64:128x128"},          // 6
    {128, 64, 1, 256, conv_igemm64_kernel<128, 64, 2, 2>, "k64:128x64"},            // 7
    {64, 128, 1, 256, conv_igemm64_kernel<64, 128, 2, 2>, "k64:64x128"},            // 8
    {64, 64, 1, 256, conv_igemm64_kernel<64, 64, 2, 2>, "k64:64x64"},               // 9
    {256, 64, 1, 256, conv_igemm64_kernel<256, 64, 4, 1>, "k64:256x64"},            // 10
    {256, 128, 1, 256, conv_igemm64_kernel<256, 128, 2, 2>, "k64:256x128"},         // 11
    // 3-deep LDS rings with counted vmcnt
    {128, 64, 1, 256, conv_igemm64_kernel<128, 64, 2, 2, 3>, "k64s3:128x64"},       // 12
    {64, 64, 1, 256, conv_igemm64_kernel<64, 64, 2, 2, 3>, "k64s3:64x64"},          // 13
    // 8 waves per block: two waves per SIMD even when the layer gives only one block per CU
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2>, "k64w8:128x128(32x64)"},  // 14
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 2, 4>, "k64w8:128x128(64x32)"},  // 15
    {256, 128, 1, 512, conv_igemm64_kernel<256, 128, 4, 2>, "k64w8:256x128(64x64)"},  // 16
    {256, 64, 1, 512, conv_igemm64_kernel<256, 64, 4, 2>, "k64w8:256x64(64x32)"},     // 17
    {128, 64, 1, 512, conv_igemm64_kernel<128, 64, 4, 2>, "k64w8:128x64(32x32)"},     // 18
    {256, 128, 1, 512, conv_igemm64_kernel<256, 128, 4, 2, 3>, "k64w8s3:256x128"},    // 19
    // 4-deep rings: every K tile of a K <= 256 layer in flight at once
    {64, 64, 1, 256, conv_igemm64_kernel<64, 64, 2, 2, 4>, "k64s4:64x64"},            // 21
    // single LDS stage for K == 64: half the LDS, twice the resident blocks (memory-level parallelism)
    {64, 64, 2, 256, conv_igemm64_kernel<64, 64, 2, 2, 1>, "k64s1:64x64"},            // 27
    // narrow outputs (Cout <= 32: U-Net full-resolution layers): no MFMA work wasted on zero cout columns
    {128, 32, 1, 256, conv_igemm64_kernel<128, 32, 4, 1>, "k64:128x32"},              // 30
    {256, 32, 1, 256, conv_igemm64_kernel<256, 32, 4, 1>, "k64:256x32"},              // 31
    // DMA issue interleaved with the MFMAs (8-wave tiles)
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 2, 4, 3, 0, 1>, "k64w8s3i:128x128(64x32)"},  // 32
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2, 3, 0, 1>, "k64w8s3i:128x128(32x64)"},  // 33
    {256, 128, 1, 512, conv_igemm64_kernel<256, 128, 4, 2, 3, 0, 1>, "k64w8s3i:256x128"},         // 35
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 2, 4, 2, 0, 1>, "k64w8i:128x128(64x32)"},    // 38
    // (round 5: the loader/consumer wave-specialised, K-split-in-block, 4-stage and most single-stage / DMA-interleaved instantiations of rounds 1-2 are gone: no layer of the three BASELINE
    //  models ever picked them in an autotune log — profiles/r04_layers_*.txt — and each cost ~40 KB of code object and ~25 test parametrisations; DESIGN_APPENDIX.md A4.2 keeps what they measured)
    // K split inside the block: 2 groups of 2x2 waves, group g multiplies K tile 2i+g of stage i (64x64 / 64x32 / 32x64 wave tiles)
    {64, 128, 1, 512, conv_igemm64_kernel<64, 128, 2, 2, 3, 0, 0, 0, 2>, "k64ks2s3:64x128"},          // 50
    // kind 3 (conv3x3_strip.hpp): 3x3 / stride 1 / pad 1, one activation strip per kernel row serves its three taps
    {128, 128, 3, 512, conv3x3_strip_kernel<128, 128, 2, 4, 2>, "strip:128x128(64x32)"},              // 52
    {128, 128, 3, 512, conv3x3_strip_kernel<128, 128, 4, 2, 2>, "strip:128x128(32x64)"},              // 53
    {128, 64, 3, 256, conv3x3_strip_kernel<128, 64, 2, 2, 3>, "strip s3:128x64"},                     // 54
    {256, 64, 3, 512, conv3x3_strip_kernel<256, 64, 4, 2, 2>, "strip:256x64"},                        // 55
    {64, 128, 3, 256, conv3x3_strip_kernel<64, 128, 2, 2, 2>, "strip w4:64x128"},                     // 56
    {128, 64, 3, 512, conv3x3_strip_kernel<128, 64, 4, 2, 3>, "strip w8s3:128x64"},                   // 57
    {256, 32, 3, 256, conv3x3_strip_kernel<256, 32, 4, 1, 3>, "strip s3:256x32"},                     // 58
    {128, 128, 3, 256, conv3x3_strip_kernel<128, 128, 2, 2, 2>, "strip w4:128x128"},                  // 59
    {128, 32, 3, 256, conv3x3_strip_kernel<128, 32, 4, 1, 2>, "strip w4:128x32"},                     // 60  (59 KB: two blocks per CU)
    // kind 4 (conv3x3_ws.hpp): weight-stationary persistent 3x3 for Cin, Cout in {32, 64}; BM = pixels per tile, BN = Cout
    {256, 64, 4, 256, conv3x3_ws_kernel<64, 64, 4, 2>, "ws:64->64 w4x2", 0, 64},                             // 61
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2>, "ws:64->32 w4x2", 0, 64},                             // 62
    {256, 64, 4, 256, conv3x3_ws_kernel<32, 64, 4, 2>, "ws:32->64 w4x2", 0, 32},                             // 63
    {256, 32, 4, 256, conv3x3_ws_kernel<32, 32, 4, 2>, "ws:32->32 w4x2", 0, 32},                             // 64
    {224, 64, 4, 448, conv3x3_ws_kernel<64, 64, 7, 1>, "ws:64->64 w7x1", 0, 64},                             // 65  (56x56: 8 x 28 = 224-pixel tiles, no idle fragment)
    {512, 32, 4, 512, conv3x3_ws_kernel<64, 32, 8, 2>, "ws:64->32 w8x2", 0, 64},                             // 66
    {512, 32, 4, 512, conv3x3_ws_kernel<32, 32, 8, 2>, "ws:32->32 w8x2", 0, 32},                             // 67
    // kind 5 (conv3x3_p9.hpp): 3x3 / stride 1 / pad 1, Cin % 64 == 0: one patch per cin block serves all nine taps, weights ring as kind 1
    {256, 128, 5, 512, conv3x3_p9_kernel<256, 128, 4, 2, 3>, "p9:256x128 s3", 0, 3},                        // 68
    {256, 128, 5, 512, conv3x3_p9_kernel<256, 128, 4, 2, 2>, "p9:256x128 s2", 0, 2},                        // 69
    {128, 128, 5, 512, conv3x3_p9_kernel<128, 128, 4, 2, 3>, "p9:128x128(32x64) s3", 0, 3},                 // 70
    {128, 128, 5, 512, conv3x3_p9_kernel<128, 128, 2, 4, 3>, "p9:128x128(64x32) s3", 0, 3},                 // 71
    {256, 64, 5, 512, conv3x3_p9_kernel<256, 64, 4, 2, 3>, "p9:256x64 s3", 0, 3},                           // 72
    {128, 64, 5, 512, conv3x3_p9_kernel<128, 64, 4, 2, 3>, "p9:128x64 s3", 0, 3},                           // 73
    // kind 6 (conv3x3_ws.hpp, STEM form): the image stem on the row-pair image (Cin = 64 = 8 overlapping 16-byte pixels, kH' = 2 | 4 taps,
    // kW' = 1, row dilation 2, stride 1 | 2); name = "wst:<taps>-><cout> ... [p<KB between the two patch buffers>]"
    {256, 64, 6, 256, conv3x3_ws_kernel<64, 64, 4, 2, 6, 0, 4, 23 * 1024>, "wst:4->64 w4x2 p23", 0, 4, 23},     // 74  (ResNet-50 7x7/2: two blocks per CU)
    {512, 64, 6, 512, conv3x3_ws_kernel<64, 64, 8, 2, 6, 0, 4>, "wst:4->64 w8x2", 0, 4},                    // 75
    {256, 64, 6, 256, conv3x3_ws_kernel<64, 64, 4, 2, 6, 0, 2, 20 * 1024>, "wst:2->64 w4x2 p20", 0, 2, 20},     // 76  (U-Net 3x3/1)
    {256, 32, 6, 256, conv3x3_ws_kernel<64, 32, 4, 2, 6, 0, 2, 20 * 1024>, "wst:2->32 w4x2 p20", 0, 2, 20},     // 77  (MobileNet-v2 3x3/2)
    {256, 32, 6, 256, conv3x3_ws_kernel<64, 32, 4, 2, 6, 0, 4, 24 * 1024>, "wst:4->32 w4x2 p24", 0, 4, 24},     // 78
    {512, 64, 6, 512, conv3x3_ws_kernel<64, 64, 8, 2, 6, 0, 2>, "wst:2->64 w8x2", 0, 2},                    // 79
    {512, 32, 6, 512, conv3x3_ws_kernel<64, 32, 8, 2, 6, 0, 2>, "wst:2->32 w8x2", 0, 2},                    // 80
    // kind 7 (conv1x1_ps.hpp): persistent pointwise GEMM, 8 MFMA waves + 4 loader waves, one block per CU
    {128, 128, 7, 768, conv1x1_ps_kernel<128, 128, 4, 2, 4>, "ps:128x128(32x64) s4", 0, 4},                 // 81
    {128, 128, 7, 768, conv1x1_ps_kernel<128, 128, 2, 4, 4>, "ps:128x128(64x32) s4", 0, 4},                 // 82
    {128, 64, 7, 768, conv1x1_ps_kernel<128, 64, 4, 2, 4>, "ps:128x64 s4", 0, 4},                           // 83
    {256, 128, 7, 768, conv1x1_ps_kernel<256, 128, 4, 2, 3, false>, "ps:256x128 s3", 0, 3},                        // 84
    {128, 256, 7, 768, conv1x1_ps_kernel<128, 256, 2, 4, 3, false>, "ps:128x256 s3", 0, 3},                        // 85
    {128, 128, 7, 768, conv1x1_ps_kernel<128, 128, 4, 2, 3>, "ps:128x128(32x64) s3", 0, 3},                 // 86
    // 256x256 (kind 1): for layers with <= 256 such tiles (ResNet-50 layer 3, N = 1024: 196) one round of blocks moves half the L1 bytes of 128x128
    {256, 256, 1, 512, conv_igemm64_kernel<256, 256, 4, 2, 2, 0, 0, 0, 1, 1, true>, "k64w8:256x256(64x128)"},    // 87
    {256, 256, 1, 512, conv_igemm64_kernel<256, 256, 4, 2, 2, 0, 1, 0, 1, 1, true>, "k64w8i:256x256(64x128)"},   // 88
    // kind 4 with the fused 1x1 head (sm_conv2d_head_fwd_cfg): separate instantiations, chosen only for launches that carry a head
    {256, 64, 4, 256, conv3x3_ws_kernel<64, 64, 4, 2, 12, 0, 0, 0, true>, "ws:64->64 w4x2 head", CF_HEAD, 64},     // 89
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 0, 0, 0, true>, "ws:64->32 w4x2 head", CF_HEAD, 64},     // 90
    {256, 64, 4, 256, conv3x3_ws_kernel<32, 64, 4, 2, 6, 0, 0, 0, true>, "ws:32->64 w4x2 head", CF_HEAD, 32},      // 91
    {256, 32, 4, 256, conv3x3_ws_kernel<32, 32, 4, 2, 6, 0, 0, 0, true>, "ws:32->32 w4x2 head", CF_HEAD, 32},      // 92
    {512, 32, 4, 512, conv3x3_ws_kernel<32, 32, 8, 2, 6, 0, 0, 0, true>, "ws:32->32 w8x2 head", CF_HEAD, 32},      // 93
    {128, 64, 4, 512, conv3x3_wsx_kernel<6>, "ws:64->64 w4+4 xhead", CF_XHEAD | CF_W44, 64},                                 // 94  expand head (bottleneck tail: + 1x1 to <= 256 channels + residual) with the two layers on different waves (conv3x3_wsx.hpp)
    {128, 64, 4, 512, conv3x3_wsx_kernel<6, true>, "ws:64->64 w4+4 xhead side", CF_XHEAD | CF_W44 | CF_SIDE, 64},                      // 96  ... and a second GEMM (the projection shortcut) instead of the residual tensor
    // stem form with loader waves (conv3x3_ws.hpp NLW): the patches are built from the fp32 NCHW input inside the launch (sm_conv2d_stem_f32_fwd_cfg)
    {256, 64, 6, 512, conv3x3_ws_kernel<64, 64, 4, 2, 6, 0, 4, 0, false, false, 4>, "wst:4->64 w4x2 ld4", 0, 4, 0, 4},            // 97  (ResNet-50 7x7/2)
    {256, 32, 6, 512, conv3x3_ws_kernel<64, 32, 4, 2, 6, 0, 2, 20 * 1024, false, false, 4>, "wst:2->32 w4x2 p20 ld4", 0, 2, 20, 4},  // 98  (MobileNet-v2 3x3/2, U-Net 3x3/1: the autotuner's pick, first = the default)
    {256, 64, 6, 384, conv3x3_ws_kernel<64, 64, 4, 2, 6, 0, 2, 20 * 1024, false, false, 2>, "wst:2->64 w4x2 p20 ld2", 0, 2, 20, 2}, // 99
    {256, 32, 6, 512, conv3x3_ws_kernel<64, 32, 4, 2, 6, 0, 4, 0, false, false, 4>, "wst:4->32 w4x2 ld4", 0, 4, 0, 4},            // 100
    {256, 64, 6, 512, conv3x3_ws_kernel<64, 64, 4, 2, 6, 0, 4, 23 * 1024, false, false, 4>, "wst:4->64 w4x2 p23 ld4", 0, 4, 23, 4},  // 101 (178 registers: still one resident block; forcing <= 170 for two blocks of 4 + 2 waves spilled and did not win)
    {256, 32, 6, 384, conv3x3_ws_kernel<64, 32, 4, 2, 6, 0, 2, 20 * 1024, false, false, 2>, "wst:2->32 w4x2 p20 ld2", 0, 2, 20, 2}, // 102 (two blocks of 4 + 2 waves per CU)
    // kind 8 (conv1x1_rw.hpp): pointwise convolution with Cin = 64 | 128 | 256, weights resident in registers, pixel steps streaming through two LDS buffer pairs
    {64, 256, 8, 512, conv1x1_rw_kernel<16, 2, 4, 2, true>, "rw:k256 n256 s64 +res", CF_RES, 256},         // 103 (ResNet-50 256 -> 1024 + block input + ReLU at 14x14)
    {128, 256, 8, 512, conv1x1_rw_kernel<16, 2, 4, 4, false>, "rw:k256 n256 s128", 0, 256},                // 104
    {128, 128, 8, 512, conv1x1_rw_kernel<16, 2, 2, 4, false>, "rw:k256 n128 s128", 0, 256},                // 105 (256 -> 128 at 56x56)
    {128, 64, 8, 512, conv1x1_rw_kernel<16, 1, 2, 4, false>, "rw:k256 n64 s128", 0, 256},                  // 106 (256 -> 64 at 56x56)
    {64, 256, 8, 512, conv1x1_rw_kernel<8, 2, 4, 2, true>, "rw:k128 n256 s64 +res", CF_RES, 128},          // 107 (128 -> 512 + block input + ReLU at 28x28)
    {128, 256, 8, 512, conv1x1_rw_kernel<8, 2, 4, 4, false>, "rw:k128 n256 s128", 0, 128},                 // 108
    {256, 64, 8, 512, conv1x1_rw_kernel<4, 1, 2, 8, false>, "rw:k64 n64 s256", 0, 64},                     // 109 (64 -> 64 at 56x56)
    {64, 256, 8, 512, conv1x1_rw_kernel<4, 2, 4, 2, true>, "rw:k64 n256 s64 +res", CF_RES, 64},            // 110 (64 -> 256 + block input + ReLU when the expand head is off)
    // ... with 4 loader waves and a ring of NST stages (conv1x1_rwl_kernel): p1 = stages, BM = pixels per stage, 8 consumer waves x 32 channels = 256 per block
    {32, 256, 8, 768, conv1x1_rwl_kernel<16, 1, 4, true>, "rwl:k256 n256 s32x4 +res", CF_RES, 256, 4},    // 111
    {64, 256, 8, 768, conv1x1_rwl_kernel<8, 2, 3, true>, "rwl:k128 n256 s64x3 +res", CF_RES, 128, 3},     // 112
    {32, 256, 8, 768, conv1x1_rwl_kernel<8, 1, 4, true>, "rwl:k128 n256 s32x4 +res", CF_RES, 128, 4},     // 113
    {64, 256, 8, 768, conv1x1_rwl_kernel<16, 2, 4, false>, "rwl:k256 n256 s64x4", 0, 256, 4},             // 114
    // ... over two sources (sm_conv2d_dual_fwd_cfg: K = K1 + K2 = 384, ResNet-50 stage 2's first block): the slice passes the LDS in two parts
    {32, 256, 8, 768, conv1x1_rwl_kernel<24, 1, 4, false, 2, true>, "rwl:k384 n256 s32x4 dual", CF_DUAL, 384, 4, 2},   // 115
    // ... with 4 consumer waves (128 channels per block, 128 W registers per consumer at K = 512): ResNet-50's 512 -> 128 | 256 reduce layers at 28 x 28
    {32, 128, 8, 512, conv1x1_rwl_kernel<32, 1, 3, false, 1, false, 4>, "rwl:k512 n128 s32x3", 0, 512, 3},                 // 116
    {32, 128, 8, 512, conv1x1_rwl_kernel<32, 1, 3, true, 1, false, 4>, "rwl:k512 n128 s32x3 +res", CF_RES, 512, 3},        // 117 (512 -> 2048 + block input + ReLU at 7 x 7)
    {32, 128, 8, 512, conv1x1_rwl_kernel<48, 1, 3, false, 2, true, 4>, "rwl:k768 n128 s32x3 dual", CF_DUAL, 768, 3, 2},    // 118 (stage 3's first block: K = 256 + 512)
    {32, 128, 8, 512, conv1x1_rwl_kernel<20, 1, 4, false, 1, false, 4>, "rwl:k320 n128 s32x4", 0, 320, 4},                 // 119 (MobileNet-v2's last pointwise layer 320 -> 1280)
    // ... K = 256 with narrow outputs (256 -> 128 | 64 at 56 x 56: one slice of 4 | 2 consumer waves)
    {64, 128, 8, 512, conv1x1_rwl_kernel<16, 2, 4, false, 1, false, 4>, "rwl:k256 n128 s64x4", 0, 256, 4},                 // 120
    {64, 64, 8, 384, conv1x1_rwl_kernel<16, 2, 4, false, 1, false, 2>, "rwl:k256 n64 s64x4", 0, 256, 4},                   // 121
    // kind 9 (conv3x3_kq.hpp): 3x3 / stride 1 / pad 1 on few small images with many channels: 196-pixel tiles (one 14x14 image | 7 rows of 28 | four 7x7
    // images) x BN couts, the K reduction split over the block's four waves (p0 = KG channel slices x p1 = NG cout groups, p2 = NF 16-cout fragments per wave),
    // weights straight from global memory into registers
    {208, 64, 9, 256, nullptr, "kq:k4n1 196x64", 1, 4, 1, 4},
    {208, 128, 9, 256, nullptr, "kq:k2n2 196x128", 1, 2, 2, 4},
    {208, 32, 9, 256, nullptr, "kq:k4n1 196x32", 1, 4, 1, 2},
    {208, 64, 9, 256, nullptr, "kq:k2n2 196x64", 1, 2, 2, 2},
    // ... flags = 2: two patch buffers, the next super-block's patch is copied while this one is multiplied (layers with Cin > 64 KG)
    {208, 64, 9, 256, nullptr, "kq:k2n2 196x64 b2", 2, 2, 2, 2},
    {208, 128, 9, 256, nullptr, "kq:k1n4 196x128 b2", 2, 1, 4, 2},
    // kind 10 (conv3x3_kqp.hpp): persistent wave-split 3x3 with the whole weight tensor in registers, 16 x 14-pixel tiles; p0 = KG (Cin = 64 KG), p1 = NG (Cout = 32 NG)
    {224, 64, 10, 256, nullptr, "kqp:k2n2 128->64", 0, 2, 2},
    {224, 128, 10, 256, nullptr, "kqp:k1n4 64->128", 0, 1, 4},
    // kind 5 with the 2x2 / stride-2 max-pool of the output as a second output of the launch (conv3x3_p9.hpp POOL: the U-Net's 128-channel level; the weight-stationary kernel has it for Cin, Cout <= 64)
    {256, 128, 5, 512, conv3x3_p9_kernel<256, 128, 4, 2, 3, true>, "p9:256x128 s3 pool", CF_POOL, 3},
    // ---- build-conditional entries stay BEHIND the product pool so that product indices do not depend on the build ----
    // kind 9, flags & CF_KQ_S2: 3x3 / STRIDE 2 / pad 1 (the first 3x3 of a ResNet stage), the stride in the patch layout; BM = 16 x the tile's fragments (13 | 7)
    {208, 128, 9, 256, nullptr, "kq/2:k1n4 196x128", 1 | CF_KQ_S2, 1, 4, 2},
    {112, 128, 9, 256, nullptr, "kq/2:k2n2 112x128", 1 | CF_KQ_S2, 2, 2, 4},
    {112, 64, 9, 256, nullptr, "kq/2:k2n2 112x64", 1 | CF_KQ_S2, 2, 2, 2},
    {112, 128, 9, 256, nullptr, "kq/2:k1n4 112x128", 1 | CF_KQ_S2, 1, 4, 2},
#ifdef SM_ABLATION   // the single-role expand head (conv3x3_ws.hpp HX): 16 of 55 us added to the 3x3; the two-role kernel replaced it
    {128, 64, 4, 256, conv3x3_ws_kernel<64, 64, 4, 1, 12, 0, 0, 0, false, true>, "ws:64->64 w4x1 xhead", CF_XHEAD, 64},
#endif
#ifdef SM_ABLATION   // two K tiles per ring stage (KT = 2, BK = 128 per barrier; every wave multiplies both): 24 vs 21.5 us on 14x14 1024->256,
                     // equal to the K-group configs on the 7x7 layers — halving the barriers buys nothing, not in the product pool
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 2, 4, 2, 0, 0, 0, 1, 2>, "k64kt2:128x128(64x32)"},
    {128, 64, 1, 512, conv_igemm64_kernel<128, 64, 4, 2, 3, 0, 0, 0, 1, 2>, "k64kt2s3:128x64"},
#endif
#ifdef SM_ABLATION   // 81 KB variants (two blocks per CU on the N = 64 layers): 45..50 us against 41 us for config 18 on 56x56x64 — not in the product pool
    {128, 64, 3, 256, conv3x3_strip_kernel<128, 64, 2, 2, 2, 1>, "strip lean w4:128x64"},
    {128, 64, 3, 512, conv3x3_strip_kernel<128, 64, 4, 2, 2, 1>, "strip lean w8:128x64"},
#endif
#ifdef SM_ABLATION   // conv3x3_ws ablations (64->32, 4 waves x 2 fragments)
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 0>, "ws:64->32 abl ref", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 1>, "ws:64->32 abl noDMA", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 2>, "ws:64->32 abl noMFMA", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 4>, "ws:64->32 abl noReads", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 8>, "ws:64->32 abl noStores", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 9>, "ws:64->32 abl noDMA noStores", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 15>, "ws:64->32 abl nothing", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 6>, "ws:64->32 abl DMA+stores only", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 13>, "ws:64->32 abl MFMA only", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 11>, "ws:64->32 abl reads only", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 14>, "ws:64->32 abl DMA only", 0, 64},
    {256, 32, 4, 256, conv3x3_ws_kernel<64, 32, 4, 2, 12, 7>, "ws:64->32 abl stores only", 0, 64},
#endif
#ifdef SM_ABLATION   // tools/conv_bench.py only: never built into the product library
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2, 2, 1>, "abl:w8 noMFMA"},
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2, 2, 2>, "abl:w8 noGlobal"},
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2, 2, 3>, "abl:w8 no A traffic"},
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2, 3, 3>, "abl:w8s3 no A traffic"},
    {128, 128, 1, 512, conv_igemm64_kernel<128, 128, 4, 2, 3, 0>, "abl-ref:w8s3"},
#endif
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// Geometry of the weight-stationary kernel for tile config `c` (BM = pixel capacity of a tile = 32 * waves * TM, threads = 64 * waves):
// TW = W split into the fewest column bands of <= 32 pixels (56 -> 28, 512 -> 32), TH = the row count that minimises the number of
// tiles while TH*TW <= BM and weights + two patches fit 160 KiB.  Returns the dynamic LDS size in bytes, 0 if the config does not fit.
size_t ws_setup(ConvParams& p, const TileCfg& c) {
    const int cin = p.Cin;
    if (cin != 32 && cin != 64) return 0;
    if (c.p0 != cin) return 0;
    const bool w44 = (c.flags & CF_W44) != 0;                        // conv3x3_wsx.hpp: 4 of the 8 waves issue the patch DMA, 6 pieces per thread
    const int cp = cin / 8, nt = w44 ? 256 : c.threads;
    const int bands = sm_ceil_div(p.W, 32);
    int TW = sm_ceil_div(p.W, bands);
    if (p.ups && (TW & 1)) TW++;                                       // fused upsampling: even tile origins
    const bool xhead = (c.flags & CF_XHEAD) != 0;                    // head weights + bias in LDS too (conv3x3_ws.hpp HX)
    const size_t fixed = (size_t)9 * p.Cout * cin * 2 + (size_t)p.Cout * 4 + WS_DUMMY + (xhead ? (size_t)ws_hx_bytes(p.Cout) : 0);
    const int nic = w44 ? 6 : cin == 64 ? 12 : 6;                    // the kernels' NIC
    int best_th = 0; long best_tiles = 0; size_t best_lds = 0, best_patch = 0;
    const int thstep = (p.ups || p.y2) ? 2 : 1;                        // fused upsampling / pooling: even tile origins
    for (int th = thstep; th * TW <= c.BM && th <= p.H; th += thstep) {
        const size_t slots = (size_t)(th + 2) * (TW + 2) * cp, patch = (slots * 16 + 1023) / 1024 * 1024;
        if (sm_ceil_div((long)slots, nt) > nic) break;
        const size_t stride = (size_t)ws_pstride((int)fixed);         // distance between the two patch buffers (compile-time in the kernel)
        if (patch > stride) break;
        const size_t lds = fixed + stride + patch;
        const long tiles = (long)sm_ceil_div(p.H, th) * sm_ceil_div(p.W, TW);
        if (!best_th || tiles < best_tiles) { best_th = th; best_tiles = tiles; best_lds = lds; best_patch = patch; }
    }
    if (!best_th) return 0;
    p.ws_TH = best_th; p.ws_TW = TW; p.ws_tiles_x = sm_ceil_div(p.W, TW); p.ws_tiles_y = sm_ceil_div(p.H, best_th);
    const long nt_all = (long)p.N * p.ws_tiles_x * p.ws_tiles_y;
    if (nt_all >= (1L << 31)) return 0;
    p.ws_ntiles = (int)nt_all; p.ws_patch_bytes = (int)best_patch;
    sm_fastdiv_make((unsigned)(TW + 2), &p.ws_pw_mul, &p.ws_pw_sh); sm_fastdiv_make((unsigned)TW, &p.ws_tw_mul, &p.ws_tw_sh);
    sm_fastdiv_make((unsigned)(p.ws_tiles_x * p.ws_tiles_y), &p.ws_txy_mul, &p.ws_txy_sh); sm_fastdiv_make((unsigned)p.ws_tiles_y, &p.ws_ty_mul, &p.ws_ty_sh);
    return best_lds;
}

// Dynamic LDS of the persistent pointwise kernel: ring ("... s<stages>") + the layer's bias vector
size_t ps_lds_bytes(const ConvParams& p, const TileCfg& c) {
    return (size_t)(c.p0 ? c.p0 : 3) * (c.BM + c.BN) * 128 + ((size_t)p.Cout * 4 + 15) / 16 * 16;
}

// Geometry of the stem form of the weight-stationary kernel (kind 6): the same search over the OUTPUT image; a patch is
// ((TH-1)s + 2(taps-1) + 1) x ((TW-1)s + 8) 16-byte row-pair pixels.  Returns the dynamic LDS size in bytes, 0 if the config does not fit.
size_t wst_setup(ConvParams& p, const TileCfg& c) {
    const int taps = c.p0, s = p.sH, nic = 6;
    const int nt = c.threads - c.p2 * 64;                               // loader waves: c.threads counts them, the patch is sized for the MFMA waves' NIC * NT slots
    if (p.kH != taps || p.Cin != 64) return 0;
    const size_t fixed = (size_t)taps * p.Cout * 128 + (size_t)p.Cout * 4 + WS_DUMMY;
    const size_t stride = c.p1 ? (size_t)c.p1 * 1024 : (size_t)ws_pstride((int)fixed);
    const int bands = sm_ceil_div(p.Wo, 32), TW = sm_ceil_div(p.Wo, bands), PW = (TW - 1) * s + 8;
    int best_th = 0; long best_tiles = 0; size_t best_patch = 0;
    for (int th = 1; th * TW <= c.BM && th <= p.Ho; th++) {
        const size_t slots = (size_t)((th - 1) * s + 2 * (taps - 1) + 1) * PW, patch = (slots * 16 + 1023) / 1024 * 1024;
        if (sm_ceil_div((long)slots, nt) > nic || patch > stride) break;
        const long tiles = (long)sm_ceil_div(p.Ho, th) * sm_ceil_div(p.Wo, TW);
        if (!best_th || tiles < best_tiles) { best_th = th; best_tiles = tiles; best_patch = patch; }
    }
    if (!best_th) return 0;
    p.ws_TH = best_th; p.ws_TW = TW; p.ws_tiles_x = sm_ceil_div(p.Wo, TW); p.ws_tiles_y = sm_ceil_div(p.Ho, best_th);
    const long nt_all = (long)p.N * p.ws_tiles_x * p.ws_tiles_y;
    if (nt_all >= (1L << 31)) return 0;
    p.ws_ntiles = (int)nt_all; p.ws_patch_bytes = (int)best_patch;
    sm_fastdiv_make((unsigned)PW, &p.ws_pw_mul, &p.ws_pw_sh); sm_fastdiv_make((unsigned)TW, &p.ws_tw_mul, &p.ws_tw_sh);
    sm_fastdiv_make((unsigned)(p.ws_tiles_x * p.ws_tiles_y), &p.ws_txy_mul, &p.ws_txy_sh); sm_fastdiv_make((unsigned)p.ws_tiles_y, &p.ws_ty_mul, &p.ws_ty_sh);
    return fixed + stride + best_patch;
}

// Geometry of the nine-tap patch kernel for tile config `c`: the column-band width TW (the whole row for small images, else ~32)
// that moves the fewest patch pixels per tile while two patches + the weight ring fit 160 KiB and a patch needs at most NIC*NT
// 16-byte DMA slots.  Returns the dynamic LDS size in bytes, 0 if the config does not fit the layer.
size_t p9_setup(ConvParams& p, const TileCfg& c) {
    if ((p.Cin % 64) != 0) return 0;
    const int nsb = c.p0 == 2 ? 2 : 3, nic = 11 - nsb, nt = c.threads, BM = c.BM;
    const size_t ring = (size_t)nsb * c.BN * 128;
    int best_tw = 0; long best_px = 0; size_t best_patch = 0;
    const int cand[2] = {p.W, sm_ceil_div(p.W, sm_ceil_div(p.W, 32))};
    for (int k = 0; k < 2; k++) {
        const int TW = cand[k]; if (TW <= 0 || TW > 126 || (k == 1 && TW == cand[0])) continue;
        const int rows = (BM % TW) == 0 ? BM / TW : (BM - 1 + TW - 1) / TW + 1;        // output rows a tile can touch
        const int imgs = (rows + p.H - 2) / p.H + 1;                                    // images those rows can lie in
        const long px = (long)(rows + 2 * imgs) * (TW + 2);
        const size_t patch = ((size_t)px * 128 + 1023) / 1024 * 1024;
        if (px * 8 > (long)nic * nt || WS_DUMMY + 2 * patch + ring > 160 * 1024) continue;
        const long cost = px * sm_ceil_div(p.W, TW);                                    // (ragged last band counts as a whole one)
        if (!best_tw || cost < best_px) { best_tw = TW; best_px = cost; best_patch = patch; p.p9_patch_px = (int)px; }
    }
    if (!best_tw) return 0;
    p.p9_TW = best_tw; p.p9_patch_bytes = (int)best_patch;
    const long tpb = ((long)p.N * p.H * best_tw + BM - 1) / BM, blocks = tpb * sm_ceil_div(p.W, best_tw) * sm_ceil_div(p.Cout, c.BN);
    if (blocks >= (1L << 31) || (long)p.N * p.H >= (1L << 31) / (best_tw + 2)) return 0;
    p.p9_tiles_per_band = (int)tpb; p.nt = sm_ceil_div(p.Cout, c.BN);
    sm_fastdiv_make((unsigned)(best_tw + 2), &p.p9_pw_mul, &p.p9_pw_sh); sm_fastdiv_make((unsigned)(p.H + 2), &p.p9_h2_mul, &p.p9_h2_sh);
    return WS_DUMMY + 2 * best_patch + ring;
}

// Geometry of the wave-split 3x3 kernel (conv3x3_kq.hpp) for tile config `c` (p0 = KG, p1 = NG, p2 = NF): whole images per tile when an image has
// <= 208 pixels, else a band of TH full rows.  Returns the dynamic LDS size in bytes, 0 if the config does not fit the layer.
size_t kq_setup(ConvParams& p, const TileCfg& c) {
    const int KG = c.p0, NG = c.p1, NF = c.p2, NB = c.flags & 3, PT = c.BM;   // (PT: output pixels a tile may hold, 16 per fragment)
    const bool s2 = (c.flags & CF_KQ_S2) != 0;
    if (p.W < 7 || p.W > 208 || (p.Cin % (KG * 64)) != 0) return 0;
    const int HW = p.H * p.W, Ho = s2 ? p.Ho : p.H, Wo = s2 ? p.Wo : p.W, HWo = Ho * Wo;
    if (Wo < 4 || Wo > PT) return 0;
    // slots per patch row, a whole number of 8-slot DMA instructions.  Stride 1: W pixels + >= 1 zero column.  Stride 2: two halves of >= Wo + 1 slots (columns 0, 2, .. 2 Wo | 1, 3, .. 2 Wo - 1)
    const int RS = s2 ? 2 * ((Wo + 1 + 3) / 4 * 4) : (p.W + 1 + 7) / 8 * 8;
    int rows;                                                        // patch rows of a tile
    if (HWo <= PT) { p.kq_TH = 0; p.kq_TI = PT / HWo; p.kq_bands = 1; p.kq_mt = sm_ceil_div(p.N, p.kq_TI); rows = p.kq_TI * (p.H + 1) + 1; }   // images H + 1 rows apart: the zero row between two of them is bottom halo of one, top halo of the next
    else { p.kq_TH = PT / Wo; p.kq_TI = 1; p.kq_bands = sm_ceil_div(Ho, p.kq_TH); if ((long)p.N * p.kq_bands >= (1L << 30)) return 0; p.kq_mt = p.N * p.kq_bands; rows = s2 ? 2 * p.kq_TH + 1 : p.kq_TH + 2; }
    // stride 1: when the row has no spare zero column (RS == W + 1) the bottom-right tap of the last pixel is slot 0 of one more row
    p.kq_RS = RS; p.kq_NI = rows * (RS / 8) + (!s2 && RS == p.W + 1 ? 1 : 0); p.kq_slice_bytes = (p.kq_NI + 1) / 2 * 2048; p.kq_nsb = p.Cin / (KG * 64);
    if ((long)p.N * HW * p.x_ld * 2 >= 0x7fff0000L || (long)p.N * HWo * p.y_ld * 2 >= 0x7fff0000L) return 0;   // (32-bit buffer offsets on both sides)
    p.nt = sm_ceil_div(p.Cout, NG * NF * 16); p.kq_mpx = sm_ceil_div(p.kq_mt, SM_NUM_XCD);
    if ((long)p.kq_mpx * SM_NUM_XCD * p.nt >= (1L << 31)) return 0;
    sm_fastdiv_make((unsigned)RS, &p.kq_rs_mul, &p.kq_rs_sh); sm_fastdiv_make((unsigned)(p.H + 1), &p.kq_h1_mul, &p.kq_h1_sh);
    sm_fastdiv_make((unsigned)p.kq_bands, &p.kq_bands_mul, &p.kq_bands_sh); sm_fastdiv_make((unsigned)p.nt, &p.nt_mul, &p.nt_sh);
    if (NB == 2 && (p.kq_nsb < 2 || sm_ceil_div(p.kq_NI, NG) > 17)) return 0;   // two buffers: something to prefetch, one patch instruction per unit and wave
    // epilogue area: two K slices with >= 2 fragments per wave trade half of their sums in accumulator layout (1 KB per fragment); one such slice has nothing to add; else every partial sum passes the LDS
    const bool xch = KG <= 2 && NF >= 2;
    const int nfr = PT / 16;
    const size_t patch = (size_t)KG * p.kq_slice_bytes * NB + (NB == 2 ? 1024 : 0);
    const size_t red = xch ? (KG == 2 ? (size_t)4 * ((nfr + 1) / 2) * NF * 1024 : 0) : (size_t)4 * (NF == 4 ? 112 : 208) * (NF * 16 + 4) * 4;
    const size_t lds = std::max(patch, red) + 2048;                  // (+ 2 KB: the kernel aligns its slices to 2 KB)
    return lds <= 160 * 1024 ? lds : 0;
}

// kind 2 = single LDS stage: only for K == 64 (one K tile)
// kinds 1/2 (conv_igemm64.hpp): Cin % 64 == 0 and a tap mask of <= 32 bits
bool cfg_eligible(const TileCfg& c, const ConvParams& p) {
    if (p.x2 && c.kind != 7 && !(c.kind == 8 && (c.flags & CF_DUAL))) return false;     // a second source: the persistent pointwise kernel and the dual ring instantiations only
    if (!p.x2 && (c.flags & CF_DUAL)) return false;
    if (c.kind == 0) return p.y2 == nullptr && p.y3 == nullptr;
    const bool same3x3 = p.kH == 3 && p.kW == 3 && p.sH == 1 && p.sW == 1 && p.dH == 1 && p.dW == 1 && p.pt == 1 && p.pl == 1 && p.Ho == p.H && p.Wo == p.W;
    if (c.kind == 4) {   // weight-stationary: the template's Cin / Cout exactly, no residual, dense 16-byte rows on both sides, ReLU family
        ConvParams q = p;
        const bool ok = same3x3 && !p.yplace && !p.res && p.splitk <= 1 && p.act <= SM_ACT_PRELU && (p.y_ld & 7) == 0 && (p.x_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) &&
               ((((uintptr_t)p.x) & 15) == 0) && ((((uintptr_t)p.w) & 15) == 0) && p.Cout == c.BN && ws_setup(q, c) > 0;
        // fused max-pool: two fragments per wave that are the image rows 2k, 2k+1 of one 32-pixel column band
        const bool is_head = (c.flags & CF_HEAD) != 0, is_xhead = (c.flags & CF_XHEAD) != 0;
        const bool want_x = p.y3 && (p.n3 > 32 || p.res3 || p.w3s);                                       // wide or residual heads: the expand-head instantiations
        if (ok && ((p.y3 != nullptr && !want_x) != is_head || want_x != is_xhead)) return false;   // head launches <-> head instantiations
        if (ok && p.y3 && !want_x && !(p.n3 > 0 && p.n3 <= 32 && (p.n3 & 7) == 0 && (p.y3_ld & 3) == 0 && ((((uintptr_t)p.y3) & 7) == 0) && ((((uintptr_t)p.w3) & 15) == 0) && p.act3 <= SM_ACT_PRELU && !p.y2)) return false;
        if (ok && is_xhead && (c.flags & CF_W44) && (p.y || p.ups)) return false;            // the two-role kernel never stores the 3x3 output
        if (ok && is_xhead && ((c.flags & CF_SIDE) != 0) != (p.w3s != nullptr)) return false;
        if (ok && p.w3s && !(p.res3 && ((((uintptr_t)p.w3s) & 15) == 0))) return false;
        if (ok && want_x && !(p.n3 >= 32 && p.n3 <= WS_HX_N3 && (p.n3 & 31) == 0 && (p.y3_ld & 7) == 0 && ((((uintptr_t)p.y3) & 15) == 0) && ((((uintptr_t)p.w3) & 15) == 0) && p.act3 <= SM_ACT_PRELU && !p.y2 &&
                                (!p.res3 || ((p.res3_ld & 7) == 0 && ((((uintptr_t)p.res3) & 15) == 0))) &&
                                (long)p.N * p.H * p.W * std::max(p.y3_ld, p.res3 ? p.res3_ld : 0) * 2 < 0x7fff0000L)) return false;   // (both head streams go through 32-bit buffer offsets)
        if (ok && p.y2) return c.BM / (c.threads / 64) == 64 && q.ws_TW == 32 && (q.ws_TH & 1) == 0 && (p.H & 1) == 0 && (p.W & 1) == 0 && (p.y2_ld & 7) == 0 && ((((uintptr_t)p.y2) & 15) == 0);
        return ok;
    }
    if (p.y3 || (p.y2 && !(c.kind == 5 && (c.flags & CF_POOL)))) return false;   // only the weight-stationary kernel applies a head; it and the pooling form of the patch kernel write the pooled second output
    if (c.kind == 8)     // weights in registers: 1x1 / stride 1 / no padding, Cin = the instantiation's, Cout a multiple of the block's slice, dense 16-byte rows, ReLU family
        return p.kH == 1 && p.kW == 1 && p.pt == 0 && p.pl == 0 && p.sH == 1 && p.sW == 1 && p.Ho == p.H && p.Wo == p.W && p.Cin == c.p0 && (p.Cout % c.BN) == 0 && p.Cout / c.BN <= 32 &&
               !p.ups && !p.yplace && p.splitk <= 1 && p.act <= SM_ACT_PRELU && (p.y_ld & 7) == 0 && (p.x_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) && ((((uintptr_t)p.x) & 15) == 0) &&
               ((((uintptr_t)p.w) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) && ((c.flags & CF_RES) != 0) == (p.res != nullptr) && (p.res == nullptr || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0)) &&
               (long)p.M * std::max(p.y_ld, p.x_ld) * 2 < 0x7fff0000L && (p.res == nullptr || (long)p.M * p.res_ld * 2 < 0x7fff0000L) &&
               (!p.x2 || (p.kt_sw > 0 && p.kt_sw < c.p0 / 64));
    if (c.kind == 7)     // persistent pointwise GEMM: 1x1, no padding, one square stride, dense 16-byte rows, ReLU family, bias vector beside the ring
        return p.kH == 1 && p.kW == 1 && p.pt == 0 && p.pl == 0 && p.sH == p.sW && (p.sH == 1 || p.sH == 2) && (p.Cin % 64) == 0 && !p.ups && !p.yplace &&
               p.splitk <= 1 && p.act <= SM_ACT_PRELU && (p.Cout & 7) == 0 && (p.y_ld & 7) == 0 && (p.x_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) &&
               ((((uintptr_t)p.x) & 15) == 0) && ((((uintptr_t)p.w) & 15) == 0) && (p.res == nullptr || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0)) &&
               (long)(p.Ho - 1) * p.sH < p.H && (long)(p.Wo - 1) * p.sW < p.W && ps_lds_bytes(p, c) <= 160 * 1024;
    if (c.kind == 6) {   // stem form: overlapping 64-half pixels of a row-pair image (x_ld = 8), one kernel column, rows two apart, no padding
        ConvParams q = p;
        if ((c.p2 != 0) != (p.xf != nullptr)) return false;                                    // loader-wave instantiations <-> fp32 input
        if (p.xf && !(p.in_C >= 1 && p.in_C <= 4 && (long)p.N * p.in_C * p.in_H * p.in_W * 4 < 0x7fff0000L)) return false;
        return p.Cin == 64 && p.x_ld == 8 && p.kW == 1 && p.dH == 2 && p.pt == 0 && p.pl == 0 && p.sH == p.sW && (p.sH == 1 || p.sH == 2) && !p.ups &&
               (long)(p.Ho - 1) * p.sH + 2 * (p.kH - 1) < p.H && (long)(p.Wo - 1) * p.sW + 8 <= p.W &&
               !p.yplace && !p.res && p.splitk <= 1 && p.act <= SM_ACT_PRELU && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) &&
               ((((uintptr_t)p.x) & 15) == 0) && ((((uintptr_t)p.w) & 15) == 0) && p.Cout == c.BN && wst_setup(q, c) > 0;
    }
    if (c.kind == 10)    // persistent wave-split 3x3, weights in registers: exactly the instantiation's channel counts; an upsampled input needs even tile origins (14 is)
        return same3x3 && !p.yplace && !p.res && p.splitk <= 1 && p.act <= SM_ACT_PRELU && p.Cin == 64 * c.p0 && p.Cout == 32 * c.p1 && (p.y_ld & 7) == 0 && (p.x_ld & 7) == 0 &&
               ((((uintptr_t)p.y) & 15) == 0) && ((((uintptr_t)p.x) & 15) == 0) && ((((uintptr_t)p.w) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) &&
               (long)p.N * p.H * p.W * p.x_ld * 2 < 0x7fff0000L && (long)p.N * p.H * p.W * p.y_ld * 2 < 0x7fff0000L && (long)p.N * ((p.H + 15) / 16) * ((p.W + 13) / 14) < (1L << 30);
    if (c.kind == 9) {   // wave-split 3x3: no residual, no upsampled input, dense 16-byte rows, ReLU family, 32-bit weight offsets
        ConvParams q = p;
        // stride 2: pad 1 in front, any padding behind it that leaves the last tap at most one row / column beyond the image (the patch holds zeros there)
        const bool half3x3 = p.kH == 3 && p.kW == 3 && p.sH == 2 && p.sW == 2 && p.dH == 1 && p.dW == 1 && p.pt == 1 && p.pl == 1 && 2 * p.Ho - 1 <= p.H && 2 * p.Wo - 1 <= p.W && p.Ho >= 1;
        return ((c.flags & CF_KQ_S2) ? half3x3 : same3x3) && !p.ups && !p.yplace && !p.res && p.splitk <= 1 && p.act <= SM_ACT_PRELU && (p.Cout & 7) == 0 && (p.y_ld & 7) == 0 && (p.x_ld & 7) == 0 &&
               ((((uintptr_t)p.y) & 15) == 0) && ((((uintptr_t)p.x) & 15) == 0) && ((((uintptr_t)p.w) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) &&
               (long)p.Cout * p.K * 2 < 0x7fff0000L && kq_setup(q, c) > 0;
    }
    if (c.kind == 5) {   // nine-tap patch kernel: no residual, dense 16-byte rows, ReLU family
        ConvParams q = p;
        if (((c.flags & CF_POOL) != 0) != (p.y2 != nullptr)) return false;
        const bool ok = same3x3 && !p.yplace && !p.res && p.splitk <= 1 && p.act <= SM_ACT_PRELU && (p.Cout & 7) == 0 && (p.y_ld & 7) == 0 && (p.x_ld & 7) == 0 &&
               ((((uintptr_t)p.y) & 15) == 0) && ((((uintptr_t)p.x) & 15) == 0) && ((((uintptr_t)p.w) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) && p9_setup(q, c) > 0;
        // pooling form: 32-column bands, tiles of BM / 32 whole rows that never straddle two images, even sizes, a stored full-resolution output
        if (ok && p.y2) return !p.ups && p.y && q.p9_TW == 32 && (p.W % 32) == 0 && (p.H % (c.BM / 32)) == 0 && (p.y2_ld & 7) == 0 && ((((uintptr_t)p.y2) & 15) == 0);
        return ok;
    }
    if ((p.Cin % 64) != 0 || p.kH * p.kW > 32) return false;
    if (c.kind == 1) return true;
    if (c.kind == 2) return p.K == 64;
    // kind 3: 3x3 / stride 1 / pad 1 / "same" size, and everything the fast epilogue needs (dense 16-byte output, ReLU family)
    return same3x3 && !p.yplace && p.act <= SM_ACT_PRELU && (p.Cout & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) &&
           (p.res == nullptr || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0));
}

int pick_cfg(const ConvParams& p) {
    // largest tile that still gives >= ~2 blocks per CU; otherwise the one with the most blocks
    const int want_kind = ((p.Cin % 64) == 0 && p.kH * p.kW <= 32) ? 1 : 0;
    int best = want_kind ? 9 : 3; double best_score = -1;
    for (int i = 0; i < kNumCfgs; i++) {
        const TileCfg& c = kCfgs[i];
        if (c.kind != want_kind) continue;
        if (c.BN > 64 && p.Cout <= 64) continue;
        long blocks = (long)sm_ceil_div(p.M, c.BM) * sm_ceil_div(p.Cout, c.BN);
        double waves = (double)blocks / (SM_NUM_CU * 2);                       // 2 resident blocks per CU
        double util = waves >= 1.0 ? waves / (double)(long)(waves + 0.999) : waves;       // tail quantisation
        double pad = ((double)p.M * p.Cout) / ((double)sm_ceil_div(p.M, c.BM) * c.BM * sm_ceil_div(p.Cout, c.BN) * c.BN);
        double eff = (c.BM * c.BN >= 128 * 128) ? 1.0 : (c.BM * c.BN >= 128 * 64 ? 0.8 : 0.55); // LDS-read bound small tiles
        double score = util * pad * eff;
        if (score > best_score) { best_score = score; best = i; }
    }
    return best;
}

int fill_params(const sm_conv_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y, ConvParams* p) {
    SM_REQUIRE(d && x && w && y, "sm_conv2d_fwd: null argument");
    SM_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kH > 0 && d->kW > 0 && d->Ho > 0 && d->Wo > 0,
               "sm_conv2d_fwd: non-positive dimension");
    SM_REQUIRE(d->sH > 0 && d->sW > 0 && d->dH > 0 && d->dW > 0 && d->groups > 0, "sm_conv2d_fwd: bad stride/dilation/groups");
    SM_REQUIRE(d->Cin % d->groups == 0 && d->Cout % d->groups == 0, "sm_conv2d_fwd: channels not divisible by groups");
    p->x = (const half_t*)x; p->w = (const half_t*)w; p->bias = bias; p->res = (const half_t*)res; p->y = (half_t*)y; p->ws = nullptr;
    p->N = d->N; p->H = d->H; p->W = d->W; p->Cin = d->Cin; p->Cout = d->Cout; p->kH = d->kH; p->kW = d->kW;
    p->sH = d->sH; p->sW = d->sW; p->dH = d->dH; p->dW = d->dW; p->pt = d->pt; p->pl = d->pl; p->Ho = d->Ho; p->Wo = d->Wo;
    p->x_ld = d->x_ld ? d->x_ld : d->Cin; p->y_ld = d->y_ld ? d->y_ld : d->Cout; p->res_ld = d->res_ld ? d->res_ld : d->Cout;
    p->act = d->act; p->slope = d->act == SM_ACT_RELU ? 0.f : d->slope;
    p->M = d->N * d->Ho * d->Wo; p->K = d->kH * d->kW * (d->Cin / d->groups);
    p->splitk = 1; p->mt = p->nt = 0; p->ktiles_per_split = 0; p->ups = d->x_ups ? 1 : 0; p->y2 = nullptr; p->y2_ld = 0;
    p->w3 = nullptr; p->b3 = nullptr; p->y3 = nullptr; p->n3 = 0; p->y3_ld = 0; p->act3 = 0; p->sl3 = 0.f; p->res3 = nullptr; p->res3_ld = 0; p->hx_dbg = 0; p->w3s = nullptr; p->xf = nullptr; p->in_C = p->in_H = p->in_W = p->in_pt = p->in_pl = 0; p->pq_OH = p->pq_OW = 0; p->x2 = nullptr; p->x2_ld = p->x2_H = p->x2_W = p->x2_s = p->kt_sw = 0;
    SM_REQUIRE(!d->x_ups || ((d->H & 1) == 0 && (d->W & 1) == 0 && d->groups == 1), "sm_conv2d_fwd: x_ups needs even H and W");
    SM_REQUIRE((long)d->N * d->Ho * d->Wo < (1L << 31), "sm_conv2d_fwd: more than 2^31 output pixels");
    SM_REQUIRE((long)d->N * d->H * d->W < (1L << 31) - (1L << 24), "sm_conv2d_fwd: more than 2^31 input pixels");
    sm_fastdiv_make((unsigned)(d->Ho * d->Wo), &p->dhw_mul, &p->dhw_sh); sm_fastdiv_make((unsigned)d->Wo, &p->dw_mul, &p->dw_sh);
    p->rowpat = 0; if (d->kH * d->kW <= 32) for (int kh = 0; kh < d->kH; kh++) p->rowpat |= 1u << (kh * d->kW);
    p->yplace = (d->y_sH | d->y_sW | d->y_oH | d->y_oW | d->y_H | d->y_W) != 0;
    p->ysH = d->y_sH ? d->y_sH : 1; p->ysW = d->y_sW ? d->y_sW : 1; p->yoH = d->y_oH; p->yoW = d->y_oW;
    p->yH = d->y_H ? d->y_H : d->Ho; p->yW = d->y_W ? d->y_W : d->Wo;
    if (p->yplace) {
        SM_REQUIRE(d->y_sH >= 0 && d->y_sW >= 0 && d->y_oH >= 0 && d->y_oW >= 0 && !res && d->groups == 1, "sm_conv2d_fwd: bad output placement (no residual, groups 1)");
        SM_REQUIRE((long)(d->Ho - 1) * p->ysH + p->yoH < p->yH && (long)(d->Wo - 1) * p->ysW + p->yoW < p->yW, "sm_conv2d_fwd: output placement leaves the destination image");
    }
    return SM_OK;
}

bool mfma_eligible(const sm_conv_desc* d, const ConvParams& p) {
    return d->groups == 1 && (d->Cin % 8) == 0 && (p.x_ld % 8) == 0 && (((uintptr_t)p.x) & 15) == 0 && (((uintptr_t)p.w) & 15) == 0;
}

int launch(const sm_conv_desc* d, ConvParams& p, int cfg, void* ws, size_t ws_bytes, hipStream_t s) {
    if (p.ups && (d->groups != 1 || d->kH != 3 || d->kW != 3)) { sm_set_error("sm_conv2d_fwd: x_ups is only implemented for 3x3 / stride 1 / pad 1 convolutions"); return SM_ERR_INVALID_ARGUMENT; }
    if (d->groups > 1 && d->groups == d->Cin && d->groups == d->Cout && (d->Cout % 8) == 0 && (p.x_ld % 8) == 0 &&
        (p.y_ld % 8) == 0 && (!p.res || (p.res_ld % 8) == 0)) {
        long work = (long)p.M * (p.Cout / 8);
        const bool k3 = p.kH == 3 && p.kW == 3 && p.dH == 1 && p.dW == 1 && p.sH == p.sW && (((uintptr_t)p.w) & 15) == 0;
        static const bool dw_strip = SM_ABL_ENV("SMELTER_DW_STRIP") != nullptr;   // experiment switch, read once (launch() is on the per-launch path)
        const bool lds_ok = k3 && (p.sH == 1 || p.sH == 2) && !dw_strip;
        if (lds_ok) {
            // Channel vectors per block: ALL of them when the patch of some tile fits ~52 KB of LDS; otherwise the largest divisor of
            // C/8 that is a multiple of 8 (whole 128-byte lines per pixel), otherwise the largest divisor at all.
            const int cvt = p.Cout / 8, S = p.sH;
            const bool narrow = p.Wo <= 8;
            // tile variants: wide (TW = 16) up to CB_W vectors, narrow (TW = 8) up to CB_N vectors
            const int CB_W = S == 1 ? 12 : 6, CB_N = S == 1 ? 24 : 18;
            int cbv;
            if (cvt <= CB_N) cbv = cvt;
            else { cbv = 0; for (int q = 8; q <= CB_N; q += 8) if (cvt % q == 0) cbv = q; if (!cbv) for (int q = 1; q <= CB_N; q++) if (cvt % q == 0) cbv = q; }
            const bool wide = cbv <= CB_W && !narrow;
            const bool big = cbv <= 4 && p.Wo >= 32 && p.Ho >= 16;   // few channels (first MobileNet layers): 4x larger tiles, 4x fewer blocks
            const int TH = big ? (S == 1 ? 16 : 8) : (S == 1 ? 8 : 4), TW = big ? (S == 1 ? 32 : 16) : (wide ? 16 : 8);
            const int cblocks = cvt / cbv;
            const int tiles_y = sm_ceil_div(p.Ho, TH), tiles_x = sm_ceil_div(p.Wo, TW);
            const long blocks = (long)p.N * tiles_y * tiles_x * cblocks;
            SM_REQUIRE(blocks < (1L << 31), "sm_conv2d_fwd: depthwise grid too large");
            const int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
            const size_t lds = ((size_t)IH * IW * cbv * 16 + 1023) / 1024 * 1024;
            const dim3 grid((unsigned)blocks);
            if (big && S == 1) hipLaunchKernelGGL((dwconv3x3_lds_kernel<1, 16, 32, 4>), grid, dim3(256), lds, s, p, cbv, tiles_x, tiles_y, cblocks);
            else if (big) hipLaunchKernelGGL((dwconv3x3_lds_kernel<2, 8, 16, 4>), grid, dim3(256), lds, s, p, cbv, tiles_x, tiles_y, cblocks);
            else if (S == 1 && wide) hipLaunchKernelGGL((dwconv3x3_lds_kernel<1, 8, 16, 12>), grid, dim3(256), lds, s, p, cbv, tiles_x, tiles_y, cblocks);
            else if (S == 1) hipLaunchKernelGGL((dwconv3x3_lds_kernel<1, 8, 8, 24>), grid, dim3(256), lds, s, p, cbv, tiles_x, tiles_y, cblocks);
            else if (wide) hipLaunchKernelGGL((dwconv3x3_lds_kernel<2, 4, 16, 6>), grid, dim3(256), lds, s, p, cbv, tiles_x, tiles_y, cblocks);
            else hipLaunchKernelGGL((dwconv3x3_lds_kernel<2, 4, 8, 18>), grid, dim3(256), lds, s, p, cbv, tiles_x, tiles_y, cblocks);
        } else if (k3 && p.sH == 1) {
            long w4 = (long)p.N * p.Ho * sm_ceil_div(p.Wo, 4) * (p.Cout / 8);
            hipLaunchKernelGGL((dwconv3x3_kernel<1, 4>), dim3(sm_stream_grid(w4, 256)), dim3(256), 0, s, p);
        } else if (k3 && p.sH == 2) {
            long w2 = (long)p.N * p.Ho * sm_ceil_div(p.Wo, 2) * (p.Cout / 8);
            hipLaunchKernelGGL((dwconv3x3_kernel<2, 2>), dim3(sm_stream_grid(w2, 256)), dim3(256), 0, s, p);
        } else {
            hipLaunchKernelGGL(dwconv_kernel, dim3(sm_stream_grid(work, 256)), dim3(256), 0, s, p);
        }
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (fc_small_eligible(d, p)) {                                             // a classifier on few rows: 16 x 16 blocks, K over the block's waves (fc_small.hpp), no tile config
        fc_small_launch(p, s);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (!p.ups && pw_thin_eligible(d, p)) {                                    // thin pointwise layers: untiled streaming kernel (pw_thin.hpp), no tile config
        pw_thin_launch(p, s);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (!mfma_eligible(d, p)) {
        long work = (long)p.M * p.Cout;
        hipLaunchKernelGGL(conv_direct_kernel, dim3(sm_stream_grid(work, 256)), dim3(256), 0, s, p, d->groups);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (p.ups && !mfma_eligible(d, p)) { sm_set_error("sm_conv2d_fwd: x_ups needs the MFMA path (Cin %% 8 == 0, aligned operands)"); return SM_ERR_INVALID_ARGUMENT; }
    if (cfg < 0 || cfg >= kNumCfgs) {
        cfg = pick_cfg(p);
        if (p.ups) { cfg = -1; for (int i = 0; i < kNumCfgs && cfg < 0; i++) if ((kCfgs[i].kind == 5 || kCfgs[i].kind == 4 || kCfgs[i].kind == 10) && cfg_eligible(kCfgs[i], p)) cfg = i; }   // only the patch kernels gather through the upsampling
        if (cfg < 0) { sm_set_error("sm_conv2d_fwd: no kernel takes this shape with x_ups"); return SM_ERR_INVALID_ARGUMENT; }
    }
    const TileCfg& c = kCfgs[cfg];
    if (p.ups && c.kind != 5 && c.kind != 4 && c.kind != 10) { sm_set_error("conv tile config %d (%s) does not apply to an upsampled input", cfg, c.name); return SM_ERR_INVALID_ARGUMENT; }
    if (!cfg_eligible(c, p)) { sm_set_error("conv tile config %d (%s) does not apply to Cin=%d", cfg, c.name, p.Cin); return SM_ERR_INVALID_ARGUMENT; }
    if (c.kind == 4 || c.kind == 6) {                                // persistent: one block per CU (two when the LDS allows), contiguous tile ranges
        const size_t lds = c.kind == 4 ? ws_setup(p, c) : wst_setup(p, c);
        const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
        const int grid = std::min(p.ws_ntiles, SM_NUM_CU * per_cu);
        typedef void (*WsFn)(ConvParams);
        hipLaunchKernelGGL((WsFn)c.fn, dim3(grid), dim3(c.threads), lds, s, p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (c.kind == 8) {                                               // persistent: one block per CU; Cout / BN slices x pixel-step ranges, the slices of a range on one XCD
        const int nsl = p.Cout / c.BN;
        p.nt = nsl; p.mt = sm_ceil_div(p.M, c.BM); p.splitk = 1;
        const int unit = SM_NUM_XCD * nsl;
        int groups = SM_NUM_CU / unit;                                // ranges per XCD
        while (groups > 1 && (groups - 1) * SM_NUM_XCD >= p.mt) groups--;          // (no more ranges than steps)
        const bool rres = (c.flags & CF_RES) != 0;
        size_t lds = (size_t)rw_lds_bytes(c.p0 / 16, c.BN, c.BM / 32, rres);
        if (c.p1) {   // ring form (conv1x1_rwl_kernel): E slots behind the weight slice's area, the others inside it
            const size_t wb = (size_t)(c.p0 / 64) / (c.p2 ? c.p2 : 1) * c.BN * 128, stage = (size_t)(c.p0 / 64 + (rres ? c.BN / 64 : 0)) * c.BM * 128;
            const size_t e = std::min<size_t>((160 * 1024 - wb) / stage, (size_t)c.p1 - 1);
            lds = std::max(wb, (c.p1 - e) * stage) + e * stage;
        }
        hipLaunchKernelGGL(c.fn, dim3((unsigned)(groups * unit)), dim3(c.threads), lds, s, p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (c.kind == 7) {                                               // persistent: one block per CU, tiles dealt per XCD inside the kernel
        p.mt = sm_ceil_div(p.M, c.BM); p.nt = sm_ceil_div(p.Cout, c.BN); p.splitk = 1;
        sm_fastdiv_make((unsigned)p.nt, &p.nt_mul, &p.nt_sh);
        const long tiles = (long)p.mt * p.nt;
        // (a whole number of blocks per XCD: XCD x owns tiles [xT/8, (x+1)T/8), at most ceil(T/8) of them)
        hipLaunchKernelGGL(c.fn, dim3((unsigned)std::min<long>((tiles + SM_NUM_XCD - 1) / SM_NUM_XCD * SM_NUM_XCD, SM_NUM_CU)), dim3(c.threads), ps_lds_bytes(p, c), s, p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (c.kind == 10) {
        if (kqp_launch(c.p0, c.p1, p, s)) { sm_set_error("conv tile config %d (%s): no such instantiation", cfg, c.name); return SM_ERR_INVALID_ARGUMENT; }
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (c.kind == 9) {
        const size_t lds = kq_setup(p, c);
        kq_launch(kq_variant(c.p0, c.p1, c.p2, c.flags & 3, c.BM / 16, (c.flags & CF_KQ_S2) ? 1 : 0), p, (unsigned)(p.kq_mpx * SM_NUM_XCD * p.nt), lds, s);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (c.kind == 5) {
        const size_t lds = p9_setup(p, c);
        const long grid = (long)p.p9_tiles_per_band * sm_ceil_div(p.W, p.p9_TW) * p.nt;
        hipLaunchKernelGGL(c.fn, dim3((unsigned)grid), dim3(c.threads), lds, s, p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    p.mt = sm_ceil_div(p.M, c.BM); p.nt = sm_ceil_div(p.Cout, c.BN);
    const int ktiles = sm_ceil_div(p.K, c.kind ? 64 : BK);
    // split-K only for launches that cannot fill the chip and have a long K loop (e.g. the FC layer)
    p.splitk = 1;
    long blocks = (long)p.mt * p.nt;
    if (ws && blocks < SM_NUM_CU / 2 && ktiles >= 16 && !p.yplace && c.kind != 3) {   // (the strip kernel has no split-K form)
        int want = (int)(SM_NUM_CU / blocks);                        // one wave of blocks: a second, partial wave costs a whole K pass
        int maxs = ktiles / 4; if (want > maxs) want = maxs; if (want > 32) want = 32;
        while (want > 1 && (size_t)want * p.M * p.Cout * sizeof(float) > ws_bytes) want--;
        if (want > 1) p.splitk = want;
    }
    p.ktiles_per_split = sm_ceil_div(ktiles, p.splitk);
    p.splitk = sm_ceil_div(ktiles, p.ktiles_per_split);
    p.ws = p.splitk > 1 ? (float*)ws : nullptr;
    // block -> tile order (see map_block): weight-stationary per XCD when the weights outweigh the input tensor
    const int combos = p.nt * p.splitk;
    const double a_bytes = 2.0 * p.N * p.H * p.W * p.Cin, b_bytes = 2.0 * p.Cout * p.K;
    static const int forced_order = SM_ABL_ENV("SM_CONV_ORDER") ? atoi(SM_ABL_ENV("SM_CONV_ORDER")) : -1;
    p.order = (combos >= SM_NUM_XCD && b_bytes > a_bytes) ? 1 : 0;
    if (forced_order == 0 || (forced_order == 1 && combos >= SM_NUM_XCD)) p.order = forced_order;
    p.mt_per_xcd = sm_ceil_div(p.mt, SM_NUM_XCD); p.cpx = sm_ceil_div(combos, SM_NUM_XCD);
    sm_fastdiv_make((unsigned)combos, &p.pm_mul, &p.pm_sh); sm_fastdiv_make((unsigned)p.mt, &p.mt_mul, &p.mt_sh);
    sm_fastdiv_make((unsigned)p.nt, &p.nt_mul, &p.nt_sh);
    const int grid = p.order == 0 ? p.mt_per_xcd * SM_NUM_XCD * combos : p.cpx * SM_NUM_XCD * p.mt;
    hipLaunchKernelGGL(c.fn, dim3(grid), dim3(c.threads), 0, s, p);
    SM_LAUNCH_CHECK();
    if (p.splitk > 1) {
        const bool vec = (p.Cout & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) | ((uintptr_t)p.ws) | ((uintptr_t)p.bias)) & 15) == 0 &&
                         (!p.res || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0));
        if (vec) hipLaunchKernelGGL(splitk_reduce_vec8_kernel, dim3(sm_stream_grid((long)p.M * (p.Cout / 8), 256)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(sm_stream_grid((long)p.M * p.Cout, 256)), dim3(256), 0, s, p);
        SM_LAUNCH_CHECK();
    }
    return SM_OK;
}

}  // namespace

extern "C" {

int sm_conv2d_num_configs(void) { return kNumCfgs; }
const char* sm_conv2d_config_name(int cfg) { return cfg >= 0 && cfg < kNumCfgs ? kCfgs[cfg].name : ""; }
// number of loader waves of a stem-form config that builds its patches from the fp32 batch (sm_conv2d_stem_f32_fwd_cfg takes exactly these), else 0
int sm_conv2d_config_loader_waves(int cfg) { return cfg >= 0 && cfg < kNumCfgs && kCfgs[cfg].kind == 6 ? kCfgs[cfg].p2 : 0; }

size_t sm_conv2d_workspace_bytes(const sm_conv_desc* d) {
    // split-K is considered by launch() when a tile config leaves more than half of the CUs without a block and the K loop is
    // long; the largest tile (256x128) decides whether that can happen, the smallest split count it would use bounds the size
    if (!d || d->groups != 1 || (d->Cin % 8)) return 0;
    long M = (long)d->N * d->Ho * d->Wo, K = (long)d->kH * d->kW * d->Cin;
    long blocks = (long)sm_ceil_div(M, 256) * sm_ceil_div(d->Cout, 128);
    if (blocks >= SM_NUM_CU / 2 || sm_ceil_div(K, 64) < 16) return 0;
    long want = SM_NUM_CU / blocks, maxs = sm_ceil_div(K, 64) / 4;
    if (want > maxs) want = maxs;
    if (want > 32) want = 32;
    return want > 1 ? (size_t)want * M * d->Cout * sizeof(float) : 0;
}

int sm_conv2d_fwd_ws(const sm_conv_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y,
                     void* ws, size_t ws_bytes, sm_stream_t stream) {
    ConvParams p; int rc = fill_params(d, x, w, bias, res, y, &p); if (rc) return rc;
    return launch(d, p, -1, ws, ws_bytes, sm_s(stream));
}
int sm_conv2d_fwd(const sm_conv_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y, sm_stream_t stream) {
    return sm_conv2d_fwd_ws(d, x, w, bias, res, y, nullptr, 0, stream);
}
int sm_conv2d_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w, const float* bias, const void* res, void* y,
                      void* ws, size_t ws_bytes, sm_stream_t stream) {
    ConvParams p; int rc = fill_params(d, x, w, bias, res, y, &p); if (rc) return rc;
    return launch(d, p, cfg, ws, ws_bytes, sm_s(stream));
}

// 3x3 convolution + the 2x2 / stride-2 max-pool of its output in one launch (both tensors are written: the U-Net's skip + pooled pair)
int sm_conv2d_pool_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w, const float* bias, void* y, void* y_pool, int32_t pool_ld, sm_stream_t stream) {
    SM_REQUIRE(y_pool, "sm_conv2d_pool_fwd_cfg: null pooled output");
    SM_REQUIRE(d && d->groups == 1, "sm_conv2d_pool_fwd_cfg: grouped convolutions have no fused max-pool");
    ConvParams p; int rc = fill_params(d, x, w, bias, nullptr, y, &p); if (rc) return rc;
    p.y2 = (half_t*)y_pool; p.y2_ld = pool_ld ? pool_ld : d->Cout;
    if (cfg < 0 || cfg >= kNumCfgs) { cfg = -1; for (int i = 0; i < kNumCfgs && cfg < 0; i++) if ((kCfgs[i].kind == 4 || (kCfgs[i].kind == 5 && (kCfgs[i].flags & CF_POOL))) && cfg_eligible(kCfgs[i], p)) cfg = i; }
    if (cfg < 0) { sm_set_error("sm_conv2d_pool_fwd_cfg: no kernel takes this shape with a fused max-pool (3x3 / s1 / p1; Cin, Cout in {32, 64}, W %% 32 == 0, even H — or Cin %% 64 == 0, W %% 32 == 0, H %% 8 == 0)"); return SM_ERR_INVALID_ARGUMENT; }
    return launch(d, p, cfg, nullptr, 0, sm_s(stream));
}
int sm_conv2d_pool_supported(const sm_conv_desc* d) {
    if (!d || d->groups != 1) return 0;
    alignas(16) static char dummy[16];
    ConvParams p; if (fill_params(d, dummy, dummy, nullptr, nullptr, dummy, &p)) return 0;
    p.y2 = (half_t*)dummy; p.y2_ld = d->Cout;
    for (int i = 0; i < kNumCfgs; i++) if ((kCfgs[i].kind == 4 || (kCfgs[i].kind == 5 && (kCfgs[i].flags & CF_POOL))) && cfg_eligible(kCfgs[i], p)) return 1;
    return 0;
}

// 3x3 / stride 1 / pad 1 convolution (Cin, Cout in {32, 64}) followed by a 1x1 convolution to n3 <= 32 channels, one launch: the 3x3 output
// tile (bias, activation, rounded to fp16 as the stored tensor would be) is the B operand of one more MFMA while it is still in registers.
// y == nullptr: the 3x3 output is not stored at all (it has no other reader).
int sm_conv2d_head_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w, const float* bias, void* y, const void* w3, const float* b3, int32_t n3,
                           int32_t act3, float slope3, void* y3, int32_t y3_ld, sm_stream_t stream) {
    SM_REQUIRE(w3 && y3 && n3 > 0, "sm_conv2d_head_fwd_cfg: null head operand");
    SM_REQUIRE(d && d->groups == 1, "sm_conv2d_head_fwd_cfg: grouped convolutions have no fused head");   // (launch() would route a depthwise layer past the head kernels)
    ConvParams p; int rc = fill_params(d, x, w, bias, nullptr, y ? y : y3, &p); if (rc) return rc;
    if (!y) p.y = nullptr;
    p.w3 = (const half_t*)w3; p.b3 = b3; p.y3 = (half_t*)y3; p.n3 = n3; p.y3_ld = y3_ld ? y3_ld : n3; p.act3 = act3; p.sl3 = act3 == SM_ACT_RELU ? 0.f : act3 == SM_ACT_NONE ? 1.f : slope3;
    if (cfg < 0 || cfg >= kNumCfgs) { cfg = -1; for (int i = 0; i < kNumCfgs && cfg < 0; i++) if (kCfgs[i].kind == 4 && cfg_eligible(kCfgs[i], p)) cfg = i; }
    if (cfg < 0) { sm_set_error("sm_conv2d_head_fwd_cfg: no kernel takes this shape with a fused 1x1 head (3x3 / s1 / p1, Cin, Cout in {32, 64}, n3 <= 32, n3 %% 8 == 0)"); return SM_ERR_INVALID_ARGUMENT; }
    return launch(d, p, cfg, nullptr, 0, sm_s(stream));
}
int sm_conv2d_head_supported(const sm_conv_desc* d, int32_t n3) {
    if (!d || d->groups != 1 || n3 > 32) return 0;                  // (wider heads: sm_conv2d_xhead_supported)
    alignas(16) static char dummy[16];
    ConvParams p; if (fill_params(d, dummy, dummy, nullptr, nullptr, dummy, &p)) return 0;
    p.w3 = (const half_t*)dummy; p.y3 = (half_t*)dummy; p.n3 = n3; p.y3_ld = n3;
    for (int i = 0; i < kNumCfgs; i++) if (kCfgs[i].kind == 4 && cfg_eligible(kCfgs[i], p)) return 1;
    return 0;
}

// The EXPAND head (a bottleneck's last two layers): 3x3 / stride 1 / pad 1 convolution (64 -> 64) followed by a 1x1 convolution to n3 = 32 .. 256
// channels (n3 % 32 == 0) with an optional residual operand, one launch: y3 = act3((w3 . y + b3) + res3), evaluated in the order of the tiled
// kernels' epilogue.  w3 / b3 sit in LDS beside the 3x3 weights.  y == nullptr: the 3x3 output is not stored.
int sm_conv2d_xhead_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w, const float* bias, void* y, const void* w3, const float* b3, int32_t n3,
                            int32_t act3, float slope3, const void* res3, int32_t res3_ld, void* y3, int32_t y3_ld, sm_stream_t stream) {
    SM_REQUIRE(w3 && y3 && n3 > 0, "sm_conv2d_xhead_fwd_cfg: null head operand");
    SM_REQUIRE(d && d->groups == 1, "sm_conv2d_xhead_fwd_cfg: grouped convolutions have no fused head");
    ConvParams p; int rc = fill_params(d, x, w, bias, nullptr, y ? y : y3, &p); if (rc) return rc;
    if (!y) p.y = nullptr;
    p.w3 = (const half_t*)w3; p.b3 = b3; p.y3 = (half_t*)y3; p.n3 = n3; p.y3_ld = y3_ld ? y3_ld : n3; p.act3 = act3; p.sl3 = act3 == SM_ACT_RELU ? 0.f : act3 == SM_ACT_NONE ? 1.f : slope3;
    p.res3 = (const half_t*)res3; p.res3_ld = res3_ld ? res3_ld : n3;
    { static const int dbg = SM_ABL_ENV("SM_HX_DEBUG") ? atoi(SM_ABL_ENV("SM_HX_DEBUG")) : 0; p.hx_dbg = dbg; }   // ablation switch, read once
    const bool small = n3 <= 32 && !res3;                            // (the narrow heads without residual run on the " head" instantiations)
    if (small || cfg < 0 || cfg >= kNumCfgs) {                       // default pick: the two-role kernel where it applies (the faster one), else the first that takes the shape
        cfg = -1;
        for (int i = 0; i < kNumCfgs && cfg < 0; i++) if (kCfgs[i].kind == 4 && (kCfgs[i].flags & CF_W44) && cfg_eligible(kCfgs[i], p)) cfg = i;
        for (int i = 0; i < kNumCfgs && cfg < 0; i++) if (kCfgs[i].kind == 4 && cfg_eligible(kCfgs[i], p)) cfg = i;
    }
    if (cfg < 0) { sm_set_error("sm_conv2d_xhead_fwd_cfg: no kernel takes this shape with a fused expand head (3x3 / s1 / p1, Cin = Cout = 64, n3 <= 256, n3 %% 32 == 0)"); return SM_ERR_INVALID_ARGUMENT; }
    return launch(d, p, cfg, nullptr, 0, sm_s(stream));
}
// ... with a second GEMM in the head instead of the residual tensor: y3 = act3(w3 . y + ws . xs + b3), xs = [N,H,W,64] fp16 (row stride xs_ld), ws = [n3][64];
// b3 = the SUM of the two layers' biases (the caller adds them).  A bottleneck's first block: 3x3, 1x1 expand, projection shortcut + Add + ReLU.
int sm_conv2d_xhead_proj_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w, const float* bias, const void* w3, const float* b3, int32_t n3,
                                 int32_t act3, float slope3, const void* xs, int32_t xs_ld, const void* ws, void* y3, int32_t y3_ld, sm_stream_t stream) {
    SM_REQUIRE(w3 && y3 && n3 > 0 && xs && ws, "sm_conv2d_xhead_proj_fwd_cfg: null head operand");
    SM_REQUIRE(d && d->groups == 1, "sm_conv2d_xhead_proj_fwd_cfg: grouped convolutions have no fused head");
    ConvParams p; int rc = fill_params(d, x, w, bias, nullptr, y3, &p); if (rc) return rc;
    p.y = nullptr;
    p.w3 = (const half_t*)w3; p.b3 = b3; p.y3 = (half_t*)y3; p.n3 = n3; p.y3_ld = y3_ld ? y3_ld : n3; p.act3 = act3; p.sl3 = act3 == SM_ACT_RELU ? 0.f : act3 == SM_ACT_NONE ? 1.f : slope3;
    p.res3 = (const half_t*)xs; p.res3_ld = xs_ld ? xs_ld : 64; p.w3s = (const half_t*)ws;
    { static const int dbg = SM_ABL_ENV("SM_HX_DEBUG") ? atoi(SM_ABL_ENV("SM_HX_DEBUG")) : 0; p.hx_dbg = dbg; }
    if (cfg < 0 || cfg >= kNumCfgs) { cfg = -1; for (int i = 0; i < kNumCfgs && cfg < 0; i++) if (kCfgs[i].kind == 4 && cfg_eligible(kCfgs[i], p)) cfg = i; }
    if (cfg < 0) { sm_set_error("sm_conv2d_xhead_proj_fwd_cfg: no kernel takes this shape (3x3 / s1 / p1, Cin = Cout = 64, n3 <= 256, n3 %% 32 == 0, 64-channel side input)"); return SM_ERR_INVALID_ARGUMENT; }
    return launch(d, p, cfg, nullptr, 0, sm_s(stream));
}
int sm_conv2d_xhead_proj_supported(const sm_conv_desc* d, int32_t n3) {
    if (!d || d->groups != 1) return 0;
    alignas(16) static char dummy[16];
    ConvParams p; if (fill_params(d, dummy, dummy, nullptr, nullptr, dummy, &p)) return 0;
    p.y = nullptr; p.w3 = (const half_t*)dummy; p.y3 = (half_t*)dummy; p.n3 = n3; p.y3_ld = n3; p.res3 = (const half_t*)dummy; p.res3_ld = 64; p.w3s = (const half_t*)dummy;
    for (int i = 0; i < kNumCfgs; i++) if (kCfgs[i].kind == 4 && cfg_eligible(kCfgs[i], p)) return 1;
    return 0;
}
// Image stem with the input conversion inside the launch: `d` is the stem-form descriptor (the row-pair geometry sm_conv2d_fwd takes for the stem:
// H x W = row-pair image, Cin = 64, kW = 1, dH = 2, x_ld = 8), x_nchw the fp32 [N, C <= 4, H_in, W_in] batch, (pt, pl) the convolution's top / left padding.
int sm_conv2d_stem_f32_fwd_cfg(const sm_conv_desc* d, int cfg, const float* x_nchw, int32_t C, int32_t H_in, int32_t W_in, int32_t pt, int32_t pl, const void* w,
                               const float* bias, void* y, sm_stream_t stream) {
    SM_REQUIRE(x_nchw && C >= 1 && C <= 4 && H_in > 0 && W_in > 0 && pt >= 0 && pl >= 0, "sm_conv2d_stem_f32_fwd_cfg: bad input description");
    ConvParams p; int rc = fill_params(d, x_nchw, w, bias, nullptr, y, &p); if (rc) return rc;
    p.xf = x_nchw; p.in_C = C; p.in_H = H_in; p.in_W = W_in; p.in_pt = pt; p.in_pl = pl;
    if (cfg < 0 || cfg >= kNumCfgs) { cfg = -1; for (int i = 0; i < kNumCfgs && cfg < 0; i++) if (kCfgs[i].kind == 6 && cfg_eligible(kCfgs[i], p)) cfg = i; }
    if (cfg < 0) { sm_set_error("sm_conv2d_stem_f32_fwd_cfg: no loader-wave stem kernel takes this shape"); return SM_ERR_INVALID_ARGUMENT; }
    // (the generic kernels would take the stem-form descriptor and read `x_nchw` as a row-pair image)
    if (kCfgs[cfg].kind != 6) { sm_set_error("conv tile config %d (%s) does not apply to an fp32 NCHW input", cfg, kCfgs[cfg].name); return SM_ERR_INVALID_ARGUMENT; }
    return launch(d, p, cfg, nullptr, 0, sm_s(stream));
}
// Image stem + the 3x3 / stride-2 max-pool that consumes it, one launch (stem_pool.hpp): `d` is the stem-form descriptor as above, the convolution's own
// output is never stored; y_pool = [N, pool_H, pool_W, Cout] fp16 with pixel stride pool_ld.  The pooling windows start at row / column 2 p of the
// convolution output (PyTorchPoolPadding's rule gives offset 0 for these shapes) and are clamped at its edge.
namespace {
constexpr int SP_PSB = 24 * 1024;                                    // bytes between the two patch buffers of the instantiations below
size_t stem_pool_setup(ConvParams& p, int pool_H, int pool_W) {
    const int taps = p.kH, s = p.sH;
    if ((taps != 4 && taps != 2) || p.Cout != 64 || p.Cin != 64 || (s != 1 && s != 2)) return 0;
    int best_th = 0, best_tw = 0; long best_tiles = 0, best_px = 0;
    for (int th = 3; th <= 31; th += 2)
        for (int tw = 3; tw <= 63 && th * tw <= 256; tw += 2) {
            const long slots = (long)((th - 1) * s + 2 * (taps - 1) + 1) * ((tw - 1) * s + 8);
            if (slots > 6 * 256 || slots * 16 > SP_PSB) continue;
            const int qh = (th - 1) / 2, qw = (tw - 1) / 2;
            const long tiles = (long)sm_ceil_div(pool_H, qh) * sm_ceil_div(pool_W, qw), px = tiles * th * tw;
            if (!best_th || tiles < best_tiles || (tiles == best_tiles && px < best_px)) { best_th = th; best_tw = tw; best_tiles = tiles; best_px = px; }
        }
    if (!best_th) return 0;
    p.ws_TH = best_th; p.ws_TW = best_tw; p.ws_tiles_y = sm_ceil_div(pool_H, (best_th - 1) / 2); p.ws_tiles_x = sm_ceil_div(pool_W, (best_tw - 1) / 2);
    const long nt_all = (long)p.N * p.ws_tiles_x * p.ws_tiles_y;
    if (nt_all >= (1L << 31)) return 0;
    p.ws_ntiles = (int)nt_all; p.pq_OH = pool_H; p.pq_OW = pool_W;
    sm_fastdiv_make((unsigned)((best_tw - 1) * s + 8), &p.ws_pw_mul, &p.ws_pw_sh); sm_fastdiv_make((unsigned)best_tw, &p.ws_tw_mul, &p.ws_tw_sh);
    sm_fastdiv_make((unsigned)(p.ws_tiles_x * p.ws_tiles_y), &p.ws_txy_mul, &p.ws_txy_sh); sm_fastdiv_make((unsigned)p.ws_tiles_y, &p.ws_ty_mul, &p.ws_ty_sh);
    return (size_t)taps * 64 * 128 + 64 * 4 + 2 * (size_t)(256 * 64 * 2) + 2 * (size_t)SP_PSB;
}
bool stem_pool_ok(const sm_conv_desc* d, const ConvParams& p, int C, int H_in, int W_in, int pool_H, int pool_W) {
    return d->groups == 1 && p.Cin == 64 && p.x_ld == 8 && p.kW == 1 && p.dH == 2 && p.pt == 0 && p.pl == 0 && p.sH == p.sW && (p.sH == 1 || p.sH == 2) && !p.ups && !p.yplace &&
           p.act <= SM_ACT_PRELU && C >= 1 && C <= 4 && (long)p.N * C * H_in * W_in * 4 < 0x7fff0000L && ((((uintptr_t)p.w) & 15) == 0) &&
           (long)(p.Ho - 1) * p.sH + 2 * (p.kH - 1) < p.H && (long)(p.Wo - 1) * p.sW + 8 <= p.W &&
           // (only window row / column 2 may leave the conv image — stem_pool.hpp redirects just those taps: rows / columns 0 and 1 of the LAST window are inside)
           pool_H >= 1 && pool_W >= 1 && 2 * (pool_H - 1) + 1 < p.Ho && 2 * (pool_W - 1) + 1 < p.Wo;
}
}  // namespace
int sm_conv2d_stem_f32_pool_supported(const sm_conv_desc* d, int32_t C, int32_t H_in, int32_t W_in, int32_t pool_H, int32_t pool_W) {
    if (!d) return 0;
    static const int dummy[4] = {0, 0, 0, 0};
    ConvParams p; if (fill_params(d, dummy, dummy, nullptr, nullptr, (void*)dummy, &p)) return 0;
    return stem_pool_ok(d, p, C, H_in, W_in, pool_H, pool_W) && stem_pool_setup(p, pool_H, pool_W) > 0;
}
int sm_conv2d_stem_f32_pool_fwd(const sm_conv_desc* d, const float* x_nchw, int32_t C, int32_t H_in, int32_t W_in, int32_t pt, int32_t pl, const void* w, const float* bias,
                                void* y_pool, int32_t pool_H, int32_t pool_W, int32_t pool_ld, sm_stream_t stream) {
    SM_REQUIRE(x_nchw && y_pool && H_in > 0 && W_in > 0 && pt >= 0 && pl >= 0, "sm_conv2d_stem_f32_pool_fwd: bad input description");
    ConvParams p; int rc = fill_params(d, x_nchw, w, bias, nullptr, y_pool, &p); if (rc) return rc;
    p.xf = x_nchw; p.in_C = C; p.in_H = H_in; p.in_W = W_in; p.in_pt = pt; p.in_pl = pl; p.y = nullptr;
    p.y2 = (half_t*)y_pool; p.y2_ld = pool_ld ? pool_ld : p.Cout;
    { static const int dbg = SM_ABL_ENV("SM_SP_DEBUG") ? atoi(SM_ABL_ENV("SM_SP_DEBUG")) : 0; p.hx_dbg = dbg; }   // ablation switch (1 = no pooling, 2 = no k loop, 4 = no patch building, 8 = no hand-over), read once
    size_t lds = 0;
    if (!stem_pool_ok(d, p, C, H_in, W_in, pool_H, pool_W) || (p.y2_ld & 7) || (((uintptr_t)y_pool) & 15) || (long)p.N * pool_H * pool_W * p.y2_ld * 2 >= 0x7fff0000L ||
        !(lds = stem_pool_setup(p, pool_H, pool_W))) {
        sm_set_error("sm_conv2d_stem_f32_pool_fwd: shape not taken (row-pair stem form with 2 | 4 taps, Cout = 64, pool 3x3 / 2 whose windows start at 2p)"); return SM_ERR_INVALID_ARGUMENT;
    }
    const int grid = std::min(p.ws_ntiles, SM_NUM_CU);
    if (p.kH == 4 && C == 3) hipLaunchKernelGGL((conv_stem_pool_kernel<64, 4, SP_PSB, 3>), dim3(grid), dim3(512), lds, sm_s(stream), p);
    else if (p.kH == 4) hipLaunchKernelGGL((conv_stem_pool_kernel<64, 4, SP_PSB, 0>), dim3(grid), dim3(512), lds, sm_s(stream), p);
    else hipLaunchKernelGGL((conv_stem_pool_kernel<64, 2, SP_PSB, 0>), dim3(grid), dim3(512), lds, sm_s(stream), p);
    SM_LAUNCH_CHECK(); return SM_OK;
}
int sm_conv2d_stem_f32_supported(const sm_conv_desc* d, int32_t C, int32_t H_in, int32_t W_in) {
    if (!d || d->groups != 1) return 0;
    alignas(16) static char dummy[16];
    ConvParams p; if (fill_params(d, dummy, dummy, nullptr, nullptr, dummy, &p)) return 0;
    p.xf = (const float*)dummy; p.in_C = C; p.in_H = H_in; p.in_W = W_in;
    for (int i = 0; i < kNumCfgs; i++) if (kCfgs[i].kind == 6 && cfg_eligible(kCfgs[i], p)) return 1;
    return 0;
}
int sm_conv2d_xhead_supported(const sm_conv_desc* d, int32_t n3, int32_t has_res) {
    if (!d || d->groups != 1) return 0;
    alignas(16) static char dummy[16];
    ConvParams p; if (fill_params(d, dummy, dummy, nullptr, nullptr, dummy, &p)) return 0;
    p.y = nullptr;                                                    // (the expand head's 3x3 output is never stored: what the executor launches)
    p.w3 = (const half_t*)dummy; p.y3 = (half_t*)dummy; p.n3 = n3; p.y3_ld = n3; p.res3 = has_res ? (const half_t*)dummy : nullptr; p.res3_ld = n3;
    for (int i = 0; i < kNumCfgs; i++) if (kCfgs[i].kind == 4 && cfg_eligible(kCfgs[i], p)) return 1;
    return 0;
}

// Pointwise convolution over TWO sources concatenated along K (a ResNet stage's first block: the expand layer and the stride-2 projection shortcut whose
// Add it feeds are one GEMM, y = act([W_E | W_S] . [h ; x_s] + b)): `d` describes the convolution over x = [N, H, W, K1] (1x1 / stride 1 / no padding,
// d->Cin = K1 + K2 = the row length of w), x2 = [N, x2_H, x2_W, K2] is read at pixel (oy * x2_stride, ox * x2_stride); K1 % 64 == K2 % 64 == 0.
int sm_conv2d_dual_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* x2, int32_t K2, int32_t x2_ld, int32_t x2_H, int32_t x2_W, int32_t x2_stride,
                           const void* w, const float* bias, void* y, sm_stream_t stream) {
    SM_REQUIRE(d && x2 && K2 > 0 && (K2 % 64) == 0 && d->Cin > K2 && ((d->Cin - K2) % 64) == 0 && x2_stride >= 1 && x2_H > 0 && x2_W > 0, "sm_conv2d_dual_fwd_cfg: bad second source");
    SM_REQUIRE(d->kH == 1 && d->kW == 1 && d->sH == 1 && d->sW == 1 && d->pt == 0 && d->pl == 0 && d->groups == 1 && d->Ho == d->H && d->Wo == d->W && !d->x_ups,
               "sm_conv2d_dual_fwd_cfg: 1x1 / stride 1 / no padding over the first source");
    SM_REQUIRE((long)(d->Ho - 1) * x2_stride < x2_H && (long)(d->Wo - 1) * x2_stride < x2_W && (long)d->N * x2_H * x2_W < (1L << 31), "sm_conv2d_dual_fwd_cfg: the second source is too small for the output grid");
    const int K1 = d->Cin - K2;
    ConvParams p; int rc = fill_params(d, x, w, bias, nullptr, y, &p); if (rc) return rc;
    if (!d->x_ld) p.x_ld = K1;                                          // (the first source's rows hold K1 channels, not d->Cin)
    p.x2 = (const half_t*)x2; p.x2_ld = x2_ld ? x2_ld : K2; p.x2_H = x2_H; p.x2_W = x2_W; p.x2_s = x2_stride; p.kt_sw = K1 / 64;
    SM_REQUIRE((p.x2_ld & 7) == 0 && ((((uintptr_t)x2) & 15) == 0), "sm_conv2d_dual_fwd_cfg: the second source needs 16-byte rows");
    // (an unaligned first source would send launch() down the direct kernel, which knows no second source: wrong values behind SM_OK)
    SM_REQUIRE((p.x_ld & 7) == 0 && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)w) & 15) == 0), "sm_conv2d_dual_fwd_cfg: the first source and the weights need 16-byte rows");
    if (cfg < 0 || cfg >= kNumCfgs) { cfg = -1; for (int i = 0; i < kNumCfgs && cfg < 0; i++) if (kCfgs[i].kind == 7 && cfg_eligible(kCfgs[i], p)) cfg = i; }
    if (cfg < 0) { sm_set_error("sm_conv2d_dual_fwd_cfg: no kernel takes this shape"); return SM_ERR_INVALID_ARGUMENT; }
    return launch(d, p, cfg, nullptr, 0, sm_s(stream));
}
int sm_conv_transpose2d_fwd(const sm_conv_desc* d, const void* x, const void* w, const float* bias, void* y, sm_stream_t stream) {
    ConvParams p; int rc = fill_params(d, x, w, bias, nullptr, y, &p); if (rc) return rc;
    SM_REQUIRE(d->groups == 1, "sm_conv_transpose2d_fwd: groups != 1");
    SM_REQUIRE(!p.yplace, "sm_conv_transpose2d_fwd: output placement not supported");
    hipLaunchKernelGGL(convt_direct_kernel, dim3(sm_stream_grid((long)p.M * p.Cout, 256)), dim3(256), 0, sm_s(stream), p);
    SM_LAUNCH_CHECK(); return SM_OK;
}

// 1 if the launch of this shape goes through a tile configuration (sm_conv2d_fwd_cfg's `cfg` matters), 0 if a dedicated kernel
// (depthwise, thin pointwise, direct) takes it; operands are assumed 16-byte aligned, as in the executor's arenas.
int sm_conv2d_uses_tile_config(const sm_conv_desc* d) {
    if (!d) return 0;
    ConvParams p; alignas(16) static char dummy[16];
    if (fill_params(d, dummy, dummy, nullptr, nullptr, dummy, &p)) return 0;
    if (d->groups > 1 && d->groups == d->Cin && d->groups == d->Cout) return 0;
    if (fc_small_eligible(d, p)) return 0;
    if (pw_thin_eligible(d, p)) return 0;
    return mfma_eligible(d, p) ? 1 : 0;
}

// Times `iters` back-to-back launches with HIP events on `stream` (host launch cost is hidden once the queue is full).
int sm_conv2d_time(const sm_conv_desc* d, int cfg, const void* x, const void* w, const float* bias, const void* res, void* y,
                   void* ws, size_t ws_bytes, int32_t iters, sm_stream_t stream, float* ms_per_launch) {
    SM_REQUIRE(iters > 0 && ms_per_launch, "sm_conv2d_time: bad argument");
    ConvParams p; int rc = fill_params(d, x, w, bias, res, y, &p); if (rc) return rc;
    hipStream_t s = sm_s(stream);
    rc = launch(d, p, cfg, ws, ws_bytes, s); if (rc) return rc;
    hipEvent_t a, b; SM_HIP_CHECK(hipEventCreate(&a)); SM_HIP_CHECK(hipEventCreate(&b));
    float t = 0;
    if (SM_ABL_ENV("SM_TIME_COLD")) {
        // cold-cache timing: in a network every layer finds its weights in HBM and its input in HBM / Infinity Cache,
        // not in a cache warmed by the previous launch of the same kernel -> evict (768 MB memset) before each launch
        static char* scratch = nullptr; const size_t sbytes = (size_t)768 << 20;
        if (!scratch) SM_HIP_CHECK(hipMalloc((void**)&scratch, sbytes));
        for (int i = 0; i < iters && !rc; i++) {
            SM_HIP_CHECK(hipMemsetAsync(scratch, i & 0xff, sbytes, s));
            SM_HIP_CHECK(hipEventRecord(a, s));
            rc = launch(d, p, cfg, ws, ws_bytes, s);
            SM_HIP_CHECK(hipEventRecord(b, s)); SM_HIP_CHECK(hipEventSynchronize(b));
            float ti = 0; SM_HIP_CHECK(hipEventElapsedTime(&ti, a, b)); t += ti;
        }
    } else {
        SM_HIP_CHECK(hipEventRecord(a, s));
        for (int i = 0; i < iters && !rc; i++) rc = launch(d, p, cfg, ws, ws_bytes, s);
        SM_HIP_CHECK(hipEventRecord(b, s)); SM_HIP_CHECK(hipEventSynchronize(b));
        SM_HIP_CHECK(hipEventElapsedTime(&t, a, b));
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
#ifdef SM_TIMELINE
    {   // one more launch with the per-block timeline recorded (conv_igemm64 kernels only)
        static unsigned long long h[8192 * 16];
        memset(h, 0, sizeof h);
        SM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tl), h, sizeof h));
        if (SM_ABL_ENV("SM_TIME_COLD")) { static char* scratch2 = nullptr; if (!scratch2) SM_HIP_CHECK(hipMalloc((void**)&scratch2, (size_t)768 << 20)); SM_HIP_CHECK(hipMemsetAsync(scratch2, 1, (size_t)768 << 20, s)); }
        rc = launch(d, p, cfg, ws, ws_bytes, s); SM_HIP_CHECK(hipStreamSynchronize(s));
        SM_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tl), sizeof h));
        static const char* names[8] = {"map_block", "rows", "brow+cursor", "prologue issue", "res+bias+init", "first tile", "main loop", "epilogue"};
        double ph[8] = {0}, wall = 0; unsigned long long w0 = ~0ull, w1 = 0; int nb = 0;
        for (int b2 = 0; b2 < 8192; b2++) {
            const unsigned long long* e = h + b2 * 16;
            if (!e[0] || !e[15]) continue;                       // unused slot or a block that map_block() retired
            nb++; for (int q = 0; q < 8; q++) if (e[2 + q] && e[1 + q]) ph[q] += (double)(e[2 + q] - e[1 + q]);
            wall += (double)(e[15] - e[0]); if (e[0] < w0) w0 = e[0]; if (e[15] > w1) w1 = e[15];
        }
        if (const char* dp = SM_ABL_ENV("SM_TL_DUMP")) {              // raw per-block stamps for offline analysis
            char fn[512]; snprintf(fn, sizeof fn, "%s.cfg%d.bin", dp, cfg);
            if (FILE* f = fopen(fn, "wb")) { fwrite(h, 1, sizeof h, f); fclose(f); }
        }
        if (nb) {
            fprintf(stderr, "[timeline cfg %d] %d blocks:", cfg, nb);
            for (int q = 0; q < 8; q++) fprintf(stderr, " %s %.0f |", names[q], ph[q] / nb);
            fprintf(stderr, " block wall avg %.2f us ; first start -> last end %.2f us ; event %.2f us\n", wall / nb / 100.0, (double)(w1 - w0) / 100.0, t / iters * 1e3);
        }
    }
#endif
    *ms_per_launch = t / iters;
    return rc;
}

}  // extern "C"
