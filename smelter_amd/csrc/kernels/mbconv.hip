// mbconv.hip — fused inverted-residual block for gfx950:  [1x1 expand + act] -> depthwise 3x3 (stride 1|2, pad 1) + act -> 1x1 project
// (+ residual) in ONE launch.  Replaces three MPSCNNConvolutionNodes in a row (Converters.swift:253; the depthwise one built from the
// descriptor of :57-66) whose two intermediate "hidden" tensors — 6x the block's input, the bulk of MobileNet-v2's HBM traffic —
// are written once and read once by the separate launches (16->96->24 at 112x112, bs=256: 616 + 616 + 154 + 154 MB around 103 MB of
// input and 38 MB of output).  Here they live in LDS only:
//   phase A  hidden patch = act1(X_patch * W1^T + b1), ((TH-1)s+3) x ((TW-1)s+3) pixels x Ch channels, fp16, ZERO outside the image
//            (the depthwise layer pads its INPUT with zeros, i.e. the hidden tensor, not X): MFMA, A fragments gathered straight from
//            global memory (a pixel row of Cin <= 32 halfs), W1 from LDS; rounded to fp16 once, as the stored tensor would be;
//   phase B  depthwise 3x3 on the LDS patch: one thread = 8 channels of one output pixel, nine 16-byte LDS reads + 72 fp16 x fp16 -> fp32
//            FMAs (weights [tap][Ch] in LDS), + bias, act2, rounded to fp16 once -> hidden tile in LDS;
//   phase C  Y tile = act3(hidden tile * W2^T + b2 (+ X)) : MFMA with both operands from LDS, epilogue in registers
//            (v_permlane32_swap -> 8 consecutive couts per lane -> 16-byte stores).
// Blocks are persistent (weights stay in LDS) and 16 waves wide: every phase is a short dependent chain (gather -> MFMA -> LDS, LDS -> FMA ->
// LDS, LDS -> MFMA -> store), so the tile time is latency, not throughput — 4-wave blocks took 20 k clk per tile, 16 waves spread the
// (row block, channel block) pairs of phase A and the pixel x channel-vector items of phase B over the whole CU.
// Numerics = the three separate launches: same fp32 accumulation order over k, the same two intermediate roundings to fp16.
//
// Kernels in this file, in the order they were built (mb_pick() / sm_mbconv_stem_fwd choose; DESIGN.md 4.4, DESIGN_APPENDIX.md A4.8, A4.14):
//   mbconv_kernel   (v1)  16-wave blocks, whole hidden patch in LDS, block barriers between the phases           — small Cin / Ch where nothing else applies
//   mbconv2_kernel  (v2)  the same chunked over 32 hidden channels in 4-wave blocks                              — shapes nothing else takes
//   mbconv3_kernel  (v3)  every wave its own pipeline over a 4 x 8 tile, hidden chunk in a wave-private LDS region, depthwise on MFMAs against diagonal
//                         weight fragments                                                                       — no-expand blocks, wide blocks with W1 / W2 read from global memory
//   mbconv4_kernel  (v4)  one 14 x 14 tile per 8-wave block, the waves share every 64-channel chunk              — images of at most four tiles (28^2, 14^2, 7^2)
//   mbconv5_kernel  (v5)  v3's pipeline with the hidden patch in REGISTERS (lane = channel, depthwise = register stencil with v_dot2c) — expand blocks on large images
//   mbconv5s_kernel       v5 behind an image stem, the input region staged in the LDS                            — MobileNet-v2's first three layers
#include <algorithm>

#include "common.hpp"
// the block kernels' reads of their input image (X fragments kept in registers per tile): MB_X_NT = 1 makes them non-temporal — build-time knob, ranked in the network
#ifndef MB_X_NT
#define MB_X_NT 0
#endif
#if MB_X_NT
#define MB_X_LOAD(ptr) __builtin_nontemporal_load((const half8_t*)(ptr))
#else
#define MB_X_LOAD(ptr) (*(const half8_t*)(ptr))
#endif

#ifdef SM_TIMELINE   // diagnostic build: clock64() spent per phase by thread 0 of block 0, summed over its tiles (sm_mbconv_debug_timeline)
__device__ unsigned long long g_mb_tl[8];   // (v4: slot 7 = the wait for the weight DMA at the top of phase B)
#define MB_TL(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now = clock64(); g_mb_tl[i] += now - mb_t0; mb_t0 = now; } } while (0)
#else
#define MB_TL(i)
#endif

namespace {

struct MbParams {
    const half_t* x; half_t* y; const half_t *w1, *wd, *w2; const float *b1, *bd, *b2;
    int N, H, W, Ho, Wo, s, Cin, Ch, Cout, x_ld, y_ld;
    int expand, res;                 // expand = 0: no 1x1 in front (the depthwise layer reads X itself, Ch == Cin)
    int act1, act2, act3; float sl1, sl2, sl3, hi1, hi2, hi3;   // hiN = +inf unless actN is SM_ACT_CLIP
    int ChP, hstride;                // Ch rounded up to 32; LDS bytes per hidden pixel = ChP*2 + 16 (an odd number of 16-byte slots: conflict-free b128 columns)
    int tiles_x, tiles_y, ntiles;
    int tile_o;                      // v4: side of the square output tile (14 at stride 1, 7 at stride 2; <= the image)
    unsigned tx_mul, tx_sh, txy_mul, txy_sh, pw_mul, pw_sh;
    // mbconv5s_kernel (the block behind an image stem): the "expand" layer is the stem's 3x3 convolution of the fp32 NCHW image `xf` (3 planes of in_H x in_W,
    // stride st_s, padding st_p): K = 27 taps (kh, kw, c) + the two bias columns; H x W above = the stem's OUTPUT image
    const float* xf; int in_H, in_W, st_s, st_p;
};

inline void fastdiv_make(unsigned d, unsigned* mul, unsigned* sh) { unsigned s = 0; while ((1ull << s) < d) s++; *mul = (unsigned)(((1ull << (31 + s)) / d) + 1); *sh = 31 + s; }
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned mul, unsigned sh) { return (unsigned)(((unsigned long long)n * mul) >> sh); }
// the activations a fused block takes: x >= 0 ? x : sl x (none: sl = 1, ReLU: sl = 0, leaky / PReLU), then min(., hi) — hi = +inf but for SM_ACT_CLIP (ONNX Clip(0, hi), ReLU6 = 6:
// sl = 0).  The packed-max fast paths below serve plain ReLU only (hi = +inf); a clipped activation takes the general path with one v_min more per value.
// CLIP is a compile-time property of the kernel instantiation: the kernels that are NOT instantiated for clipped activations keep exactly the code (and the registers: they sit at
// their spill edge) they had without the bound.
template <bool CLIP> __device__ __forceinline__ float relu_family(float v, float sl, float hi) { const float r = v < 0.f ? v * sl : v; if constexpr (CLIP) return fminf(r, hi); else return r; }
// Barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0): inside the tile loop that made every barrier wait for the
// X-patch prefetch of the NEXT tile (a full HBM round trip) and for the previous tile's output stores.
__device__ __forceinline__ void mb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// TH x TW output pixels per tile (TH*TW = 32 or 64); S = depthwise stride
template <int TH, int TW, int S>
__global__ __launch_bounds__(1024) void mbconv_kernel(MbParams p) {
    constexpr int NT = 1024, NW = 16, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, HP = PH * PW, OP = TH * TW;
    constexpr int MFA = (HP + 31) / 32, MFC = OP / 32;
    static_assert(OP % 32 == 0, "whole MFMA row blocks per tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
    // KP1: the expand GEMM's K = Cin + 2 rounded up to 16.  Columns Cin, Cin+1 of an X-patch row are 1 inside the image (0 outside) and meet
    // hi(b1), lo(b1) = b1 split into two halfs in W1: the bias enters through the MFMA, a pixel outside the image comes out as exactly 0 (what the
    // depthwise layer's zero padding of the HIDDEN tensor needs) and the epilogue has no bias add and no select.
    const int ChP = p.ChP, hs = p.hstride, KP1 = (p.Cin + (p.expand ? 2 : 0) + 15) & ~15, CoP = (p.Cout + 31) & ~31;
    // LDS: X patch [2][MFA*32][KP1] (expand) | hidden patch [HP][hs] (x2 without expand: it IS the X patch) | hidden tile [OP][hs] |
    //      W1 [ChP][KP1] | Wd [9][ChP] | W2 [CoP][hs] | b1[ChP] bd[ChP] b2[CoP]
    // The X patch of tile t+1 is fetched (one 16-byte load per thread) at the START of tile t and written to LDS at its end: no global
    // load latency sits inside a tile's dependent chain (gather -> MFMA -> ... took 14 k clk per tile with the gather in phase A).
    const int XPB = p.expand ? MFA * 32 * KP1 * 2 : 0, HIDB = MFA * 32 * hs;
    unsigned char* const xpl = mb_smem;
    unsigned char* const hid0 = xpl + 2 * XPB;
    unsigned char* const hid2 = hid0 + (p.expand ? 1 : 2) * HIDB;
    half_t* const w1l = (half_t*)(hid2 + OP * hs);
    half_t* const wdl = w1l + ChP * KP1;
    unsigned char* const w2l = (unsigned char*)(wdl + 9 * ChP);
    float* const b1l = (float*)(w2l + CoP * hs);
    float* const bdl = b1l + ChP;
    float* const b2l = bdl + ChP;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights -> LDS once per block, zero-padded (Ch -> ChP, Cin -> KP1, Cout -> CoP) so that padded lanes multiply zeros --------
    if (p.expand) for (int i = tid; i < ChP * KP1; i += NT) {
        const int n = i / KP1, k = i - n * KP1;
        half_t v = (half_t)0.f;
        if (n < p.Ch) {
            if (k < p.Cin) v = p.w1[(long)n * p.Cin + k];
            else if (k < p.Cin + 2 && p.b1) { const float b = p.b1[n]; const half_t bh = (half_t)b; v = k == p.Cin ? bh : (half_t)(b - (float)bh); }
        }
        w1l[i] = v;
    }
    for (int i = tid; i < 9 * ChP; i += NT) { const int t = i / ChP, c = i - t * ChP; wdl[i] = c < p.Ch ? p.wd[(long)c * 9 + t] : (half_t)0.f; }
    for (int i = tid; i < CoP * (hs / 2); i += NT) { const int n = i / (hs / 2), k = i - n * (hs / 2); ((half_t*)w2l)[i] = (n < p.Cout && k < p.Ch) ? p.w2[(long)n * p.Ch + k] : (half_t)0.f; }
    for (int i = tid; i < ChP; i += NT) { b1l[i] = (p.expand && p.b1 && i < p.Ch) ? p.b1[i] : 0.f; bdl[i] = (p.bd && i < p.Ch) ? p.bd[i] : 0.f; }
    for (int i = tid; i < CoP; i += NT) b2l[i] = (p.b2 && i < p.Cout) ? p.b2[i] : 0.f;
    for (int i = tid; i < OP * (hs / 2); i += NT) ((half_t*)hid2)[i] = (half_t)0.f;      // (the pad channels / pad slot of the hidden tile stay zero for good)
    if (!p.expand) for (int i = tid; i < 2 * HIDB / 2; i += NT) ((half_t*)hid0)[i] = (half_t)0.f;   // (pad channels of the copied patch)

    const bool relu1 = p.sl1 == 0.f, relu2 = p.sl2 == 0.f;   // block-uniform: plain ReLU takes the packed-max path; a clipped one (ONNX Clip(0, hi)) the general path below
                                                                      // (a packed min inside the fast path cost these kernels 5 registers each — they sit at their spill edge: +6.6 us on MobileNet-v2's 7 x 7 blocks)            // block-uniform: plain ReLU takes the packed-max path
    const int T = p.ntiles, G = gridDim.x;
    const int t_begin = (int)((long)blockIdx.x * T / G), t_end = (int)((long)(blockIdx.x + 1) * T / G);
    const int nfa = ChP >> 5, ks1 = KP1 >> 4, ks2 = ChP >> 4, nfc = CoP >> 5, cvn = p.Ch >> 3;
    // X patch slot of this thread: patch pixel xs_pp, 16-byte chunk xs_c (slots = MFA*32 pixels x KP1/8 chunks <= NT)
    const int xchunks = KP1 >> 3, xs_pp = tid / xchunks, xs_c = tid - xs_pp * xchunks;
    const bool xs_mine = xs_pp < MFA * 32;
    const int xs_py = (int)fastdiv((unsigned)xs_pp, p.pw_mul, p.pw_sh), xs_px = xs_pp - xs_py * PW;
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        n = (int)fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y), by = (int)fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        oy0 = by * TH; ox0 = (r - by * p.tiles_x) * TW;
    };
    auto fetch_patch = [&](int t) -> half8_t {                       // this thread's slot of tile t's zero-padded X patch
        int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
        const int iy = oy0 * S - 1 + xs_py, ix = ox0 * S - 1 + xs_px;
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (xs_mine && xs_pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
            if (xs_c * 8 < p.Cin) v = *(const half8_t*)(p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld + xs_c * 8);
            else if (p.expand && xs_c * 8 == p.Cin) { v[0] = (half_t)1.f; v[1] = (half_t)1.f; }        // the two bias columns
        }
        return v;
    };
    auto store_patch = [&](int buf, half8_t v) {
        if (!xs_mine) return;
        if (p.expand) *(half8_t*)(xpl + buf * XPB + (xs_pp * KP1 + xs_c * 8) * 2) = v;
        else *(half8_t*)(hid0 + buf * HIDB + xs_pp * hs + xs_c * 16) = v;
    };
    __syncthreads();                                                  // (the zero fills above are done before anyone overwrites a patch slot)
    if (t_begin < t_end) store_patch(0, fetch_patch(t_begin));
    __syncthreads();

#ifdef SM_TIMELINE
    unsigned long long mb_t0 = clock64();
#endif
#pragma unroll 1
    for (int t = t_begin; t < t_end; t++) {
        MB_TL(0);
        const int buf = (t - t_begin) & 1;
        int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
        half8_t xnext = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t + 1 < t_end) xnext = fetch_patch(t + 1);               // in flight for the whole tile
        const unsigned char* const xp_l = xpl + buf * XPB;
        unsigned char* const hid = p.expand ? hid0 : hid0 + buf * HIDB;

        // ---- phase A: hidden patch.  (patch row block, hidden channel block) pairs over the waves; X fragments from the LDS patch ------
        for (int item = wave; item < (p.expand ? MFA * nfa : 0); item += NW) {
            const int mf = item / nfa, nf = item - mf * nfa;
            const int pp = mf * 32 + l31;
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 3; ks++)
                if (ks < ks1) {
                    const int k = ks * 16 + hi * 8;
                    const half8_t fx = *(const half8_t*)(xp_l + (pp * KP1 + k) * 2);     // (zero outside the image and beyond Cin)
                    const half8_t fw = *(const half8_t*)(w1l + (nf * 32 + l31) * KP1 + k);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fx, acc, 0, 0, 0);
                }
            // register 4q+r <-> hidden channel nf*32 + 8q + 4*hi + r of patch pixel pp (bias included, exactly 0 outside the image): round to
            // fp16, activation on the packed halfs (ReLU: v_pk_max_f16 — max commutes with the rounding), four 8-byte LDS stores
#pragma unroll
            for (int q = 0; q < 4; q++) {
                half4_t o;
                if (relu1) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)acc[4 * q + r2];
                    o = __builtin_elementwise_max(o, half4_t{0, 0, 0, 0});
                } else {
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)relu_family<false>(acc[4 * q + r2], p.sl1, p.hi1);
                }
                *(half4_t*)(hid + pp * hs + (nf * 32 + 8 * q + 4 * hi) * 2) = o;
            }
        }
        MB_TL(1);
        mb_lds_barrier();
        MB_TL(2);

        // ---- phase B: depthwise 3x3, one thread = one output pixel x 8 channels --------------------------------------------------
        for (int it = tid; it < OP * cvn; it += NT) {
            const int op = it / cvn, cv = it - op * cvn, oy = op / TW, ox = op - oy * TW;
            float a[8];
            {
                const floatx4 b0 = *(const floatx4*)(bdl + cv * 8), b1v = *(const floatx4*)(bdl + cv * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; j++) { a[j] = b0[j]; a[4 + j] = b1v[j]; }
            }
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    // a[j] += x[j] * w[j] as v_fma_mix_f32 (fp16 operands, fp32 accumulate; the product of two halfs is exact in fp32, so this is
                    // mul + add bit for bit) — written out: hipcc otherwise spends two conversions per multiply-add
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 xu = *(const u32x4*)(hid + ((oy * S + kh) * PW + ox * S + kw) * hs + cv * 16);
                    const u32x4 wu = *(const u32x4*)(wdl + (kh * 3 + kw) * ChP + cv * 8);
#pragma unroll
                    for (int d2 = 0; d2 < 4; d2++) {
                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2]) : "v"(xu[d2]), "v"(wu[d2]));
                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2 + 1]) : "v"(xu[d2]), "v"(wu[d2]));
                    }
                }
            half8_t ov;
            if (relu2) {
#pragma unroll
                for (int j = 0; j < 8; j++) ov[j] = (half_t)a[j];
                ov = __builtin_elementwise_max(ov, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) ov[j] = (half_t)relu_family<false>(a[j], p.sl2, p.hi2);
            }
            *(half8_t*)(hid2 + op * hs + cv * 16) = ov;
        }
        MB_TL(3);
        mb_lds_barrier();
        MB_TL(4);

        // ---- phase C: project.  (row block, cout block) pairs over the waves ------------------------------------------------------
        for (int item = wave; item < MFC * nfc; item += NW) {
            const int mf = item / nfc, nf = item - mf * nfc;
            const int op = mf * 32 + l31, oy = oy0 + op / TW, ox = ox0 + op % TW;
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.f;
            for (int ks = 0; ks < ks2; ks++) {
                const half8_t fy = *(const half8_t*)(hid2 + op * hs + (ks * 2 + hi) * 16);
                const half8_t fw = *(const half8_t*)(w2l + (nf * 32 + l31) * hs + (ks * 2 + hi) * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fy, acc, 0, 0, 0);
            }
            floatx4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const floatx4*)(b2l + nf * 32 + 8 * q + 4 * hi);
            const bool pin = oy < p.Ho && ox < p.Wo;
            const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float o[8];
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++) {
                    unsigned ua = __builtin_bit_cast(unsigned, acc[4 * (2 * j) + r2] + bq[2 * j][r2]), ub = __builtin_bit_cast(unsigned, acc[4 * (2 * j + 1) + r2] + bq[2 * j + 1][r2]);
                    sm_permlane32_swap(ua, ub);
                    o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                }
                const int co = nf * 32 + 16 * j + 8 * hi;
                if (pin && co < p.Cout) {
                    if (p.res) { const half8_t rv = *(const half8_t*)(xp_l + (((op / TW + 1) * PW + op % TW + 1) * KP1 + co) * 2);   // residual = the block input at the same pixel (stride 1, Cin == Cout): the patch centre
#pragma unroll
                        for (int r2 = 0; r2 < 8; r2++) o[r2] += (float)rv[r2]; }
                    half8_t ov;
#pragma unroll
                    for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)relu_family<false>(o[r2], p.sl3, p.hi3);
                    *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                }
            }
        }
        if (t + 1 < t_end) store_patch(buf ^ 1, xnext);              // the other buffer: nobody reads it during this tile
        MB_TL(5);
        mb_lds_barrier();                                              // next tile's patch visible; this tile's reads of the hidden tile are over
        MB_TL(6);
    }
}


// ---- v2: the same block CHUNKED over the hidden channels, in 4-wave blocks ------------------------------------------------------------------
// The v1 kernel above keeps the whole hidden patch (all Ch channels) and every weight in LDS: 60..105 KB, i.e. one or two 16-wave blocks per
// CU, and a tile is a dependent chain (patch -> MFMA -> LDS -> barrier -> FMA -> LDS -> barrier -> MFMA -> store) that those one or two blocks
// cannot hide: 14 k clk per tile for ~2.5 k clk of work.  Here the hidden tensor exists 32 channels at a time:
//     for each chunk c:  hidden[patch][32] = act1(Xpatch * W1[c]^T + b1[c])    (MFMA; X patch in LDS for the whole tile)
//                        dw[tile][32]      = act2(depthwise3x3(hidden) + bd[c]) (one thread = 8 channels of one output pixel)
//                        acc[tile][Cout]  += dw * W2[:, c]^T                    (MFMA; accumulators live across the chunks)
// so a block needs ~20..27 KB of LDS whatever Ch is, 6..7 blocks share a CU and hide each other's chains, weights are read as MFMA
// fragments straight from global memory (L2-resident, a few KB per block), and Ch is unbounded.  Numerics as v1: hidden and depthwise
// outputs rounded to fp16 once; the projection sums k in ascending order (chunks in order, k inside a chunk ascending) in fp32.
template <int TH, int TW, int S>
__global__ __launch_bounds__(256) void mbconv2_kernel(MbParams p) {
    constexpr int NT = 256, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, HP = PH * PW, OP = TH * TW;
    constexpr int MFA = (HP + 31) / 32, MFC = OP / 32, HS = 80;          // HS: bytes per pixel of a 32-channel chunk (+16: conflict-free 8/16-byte columns)
    static_assert(OP % 32 == 0 && MFC <= 4, "whole MFMA row blocks per tile, one (row block, cout block) pair per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
    const int ChP = p.ChP, KP1 = (p.Cin + 15) & ~15, NC = ChP >> 5, ks1 = KP1 >> 4, nfc = (p.Cout + 31) >> 5;
    unsigned char* const xpl = mb_smem;                                  // X patch [MFA*32][KP1] halfs (expand only)
    unsigned char* const hid = xpl + (p.expand ? MFA * 32 * KP1 * 2 : 0);// hidden chunk [MFA*32][HS]
    unsigned char* const dwo = hid + MFA * 32 * HS;                      // depthwise output chunk [OP][HS]
    half_t* const wdl = (half_t*)(dwo + OP * HS);                        // depthwise weights [9][ChP]

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 9 * ChP; i += NT) { const int t = i / ChP, c = i - t * ChP; wdl[i] = c < p.Ch ? p.wd[(long)c * 9 + t] : (half_t)0.f; }

    const int T = p.ntiles, G = gridDim.x;
    const int t_begin = (int)((long)blockIdx.x * T / G), t_end = (int)((long)(blockIdx.x + 1) * T / G);
    // projection: wave -> (row block mf, cout block nf)
    const int c_mf = wave % MFC, c_nf = wave / MFC;
    const bool c_mine = c_nf < nfc;
    const half8_t hz = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef SM_TIMELINE
    unsigned long long mb_t0 = clock64();
#endif
#pragma unroll 1
    for (int t = t_begin; t < t_end; t++) {
        const int n = (int)fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y), by = (int)fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        const int oy0 = by * TH, ox0 = (r - by * p.tiles_x) * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        const half_t* const xim = p.x + (long)n * p.H * p.W * p.x_ld;
        mb_lds_barrier();                                                // the previous tile's last readers of xpl / dwo are done (first tile: wdl is written)
        if (p.expand) {
            // X patch -> LDS, zero outside the image and beyond Cin
            const int xchunks = KP1 >> 3;
            for (int sl = tid; sl < MFA * 32 * xchunks; sl += NT) {
                const int pp = sl / xchunks, c8 = sl - pp * xchunks, py = pp / PW, px = pp - py * PW, iy = iy0 + py, ix = ix0 + px;
                half8_t v = hz;
                if (pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c8 * 8 < p.Cin) v = *(const half8_t*)(xim + ((long)iy * p.W + ix) * p.x_ld + c8 * 8);
                *(half8_t*)(xpl + (pp * KP1 + c8 * 8) * 2) = v;
            }
        }
        floatx16 acc2;
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[i] = 0.f;
        MB_TL(0);
#pragma unroll 1
        for (int c = 0; c < NC; c++) {
            // projection weights of this chunk: requested now, used after two barriers
            half8_t fw2[2] = {hz, hz};
            if (c_mine) {
                const int co = c_nf * 32 + l31;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) { const int k = c * 32 + ks * 16 + hi * 8; if (co < p.Cout && k < p.Ch) fw2[ks] = *(const half8_t*)(p.w2 + (long)co * p.Ch + k); }
            }
            if (p.expand) {
                if (c == 0) mb_lds_barrier();                            // X patch complete
                // ---- phase A: hidden chunk = act1(X patch * W1[c]^T + b1[c]); patch row blocks over the waves ----------------------------------
                const int hch = c * 32 + l31;
                half8_t fw1[4] = {hz, hz, hz, hz};
#pragma unroll
                for (int ks = 0; ks < 4; ks++) { const int k = ks * 16 + hi * 8; if (ks < ks1 && hch < p.Ch && k < p.Cin) fw1[ks] = *(const half8_t*)(p.w1 + (long)hch * p.Cin + k); }
                floatx4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const int ch = c * 32 + 8 * q + 4 * hi; bq[q] = (p.b1 && ch < p.Ch) ? *(const floatx4*)(p.b1 + ch) : floatx4{0.f, 0.f, 0.f, 0.f}; }
                for (int mf = wave; mf < MFA; mf += 4) {
                    const int pp = mf * 32 + l31, py = pp / PW, px = pp - py * PW, iy = iy0 + py, ix = ix0 + px;
                    const bool inimg = pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    floatx16 acc;
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++)
                        if (ks < ks1) {
                            const half8_t fx = *(const half8_t*)(xpl + (pp * KP1 + ks * 16 + hi * 8) * 2);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1[ks], fx, acc, 0, 0, 0);
                        }
                    // register 4q+r <-> hidden channel 8q + 4hi + r of the chunk, patch pixel pp; pixels outside the image are ZERO (the depthwise layer pads the hidden tensor)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        half4_t o;
#pragma unroll
                        for (int r2 = 0; r2 < 4; r2++) o[r2] = inimg ? (half_t)relu_family<false>(acc[4 * q + r2] + bq[q][r2], p.sl1, p.hi1) : (half_t)0.f;
                        *(half4_t*)(hid + pp * HS + (8 * q + 4 * hi) * 2) = o;
                    }
                }
            } else {
                // no expand layer: the hidden chunk IS channels c*32 .. +31 of the X patch
                for (int sl = tid; sl < MFA * 32 * 4; sl += NT) {
                    const int pp = sl >> 2, c8 = sl & 3, py = pp / PW, px = pp - py * PW, iy = iy0 + py, ix = ix0 + px, ch = c * 32 + c8 * 8;
                    half8_t v = hz;
                    if (pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && ch < p.Ch) v = *(const half8_t*)(xim + ((long)iy * p.W + ix) * p.x_ld + ch);
                    *(half8_t*)(hid + pp * HS + c8 * 16) = v;
                }
            }
            MB_TL(1);
            mb_lds_barrier();
            MB_TL(2);
            // ---- phase B: depthwise 3x3 on the chunk, one thread = one output pixel x 8 channels ----------------------------------------------
            for (int it = tid; it < OP * 4; it += NT) {
                const int op = it >> 2, cv = it & 3, oy = op / TW, ox = op - oy * TW, ch = c * 32 + cv * 8;
                float a[8];
                {
                    const bool hb = p.bd && ch < p.Ch;
                    const floatx4 b0 = hb ? *(const floatx4*)(p.bd + ch) : floatx4{0.f, 0.f, 0.f, 0.f}, b1v = hb ? *(const floatx4*)(p.bd + ch + 4) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; j++) { a[j] = b0[j]; a[4 + j] = b1v[j]; }
                }
#pragma unroll
                for (int kh = 0; kh < 3; kh++)
#pragma unroll
                    for (int kw = 0; kw < 3; kw++) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 xu = *(const u32x4*)(hid + ((oy * S + kh) * PW + ox * S + kw) * HS + cv * 16);
                        const u32x4 wu = *(const u32x4*)(wdl + (kh * 3 + kw) * ChP + ch);
#pragma unroll
                        for (int d2 = 0; d2 < 4; d2++) {
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2]) : "v"(xu[d2]), "v"(wu[d2]));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2 + 1]) : "v"(xu[d2]), "v"(wu[d2]));
                        }
                    }
                half8_t ov;
#pragma unroll
                for (int j = 0; j < 8; j++) ov[j] = (half_t)relu_family<false>(a[j], p.sl2, p.hi2);
                *(half8_t*)(dwo + op * HS + cv * 16) = ov;
            }
            MB_TL(3);
            mb_lds_barrier();
            MB_TL(4);
            // ---- phase C: acc2 += dw chunk * W2[:, chunk]^T ---------------------------------------------------------------------------
            if (c_mine) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const half8_t fy = *(const half8_t*)(dwo + (c_mf * 32 + l31) * HS + (ks * 2 + hi) * 16);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw2[ks], fy, acc2, 0, 0, 0);
                }
            }
            MB_TL(5);
        }
        // ---- epilogue: + b2 (+ X), act3, 16-byte stores ------------------------------------------------------------------------------
        if (c_mine) {
            const int op = c_mf * 32 + l31, oy = oy0 + op / TW, ox = ox0 + op % TW;
            const bool pin = oy < p.Ho && ox < p.Wo;
            const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float o[8];
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++) {
                    const int ca = c_nf * 32 + 8 * (2 * j) + 4 * hi + r2, cb = ca + 8;
                    unsigned ua = __builtin_bit_cast(unsigned, acc2[4 * (2 * j) + r2] + ((p.b2 && ca < p.Cout) ? p.b2[ca] : 0.f));
                    unsigned ub = __builtin_bit_cast(unsigned, acc2[4 * (2 * j + 1) + r2] + ((p.b2 && cb < p.Cout) ? p.b2[cb] : 0.f));
                    sm_permlane32_swap(ua, ub);
                    o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                }
                const int co = c_nf * 32 + 16 * j + 8 * hi;
                if (pin && co < p.Cout) {
                    if (p.res) { const half8_t rv = *(const half8_t*)(xpl + (((op / TW + 1) * PW + op % TW + 1) * KP1 + co) * 2);   // the block input at the same pixel: the patch centre
#pragma unroll
                        for (int r2 = 0; r2 < 8; r2++) o[r2] += (float)rv[r2]; }
                    half8_t ov;
#pragma unroll
                    for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)relu_family<false>(o[r2], p.sl3, p.hi3);
                    *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                }
            }
        }
        MB_TL(6);
    }
}


// ---- v3: every WAVE is its own pipeline -----------------------------------------------------------------------------------------------
// PMC and timeline of v1 after its VALU diet: the tile is a chain of short dependent phases separated by block-wide barriers, and the one
// or two 16-wave blocks a CU holds cannot hide it (≈13 k clk per tile for ≈2 k clk of issue).  Here nothing in the tile loop is shared
// between waves: a wave owns a 4x8 output tile, the hidden patch chunk (32 channels) and the depthwise output chunk live in LDS regions
// private to the wave, so the phases need no barrier at all (LDS operations of one wave execute in order) and the 8 waves of a CU are 8
// independent chains for the scheduler to interleave.  Block-shared and read-only after the prologue: W1 (+ the two bias columns), Wd,
// W2, bd, b2.  The X patch never touches LDS: its MFMA B fragments (patch pixel = lane, 16 bytes of channels) are loaded straight from
// global memory once per tile and stay in registers for all chunks; the bias columns (1, 1 inside the image) are made in registers.
// Hidden channels are processed 32 at a time with the projection accumulating across chunks, so Ch is limited only by the weights' LDS.
// KS1 = k steps of the expand GEMM (K = Cin + 2 rounded up to 16; 0 = no expand layer: the hidden chunk is a channel slice of the X patch, copied
// into the wave's region by the wave itself), NW = waves per block (they share only the weights), WREG = depthwise
// weights of the lane's channel vector in registers for the chunk (36 registers: only where the occupancy survives it)
// TH x 8 output pixels per tile: 4 x 8 for stride 1; 2 x 8 for stride 2 (5 x 17 patch = 3 row blocks instead of 9 x 17 = 5: 9 KB of LDS per wave
// instead of 15, i.e. 16 waves per CU instead of 8; half of the projection MFMA's pixel columns are idle, which costs nothing that matters)
// WG: W1 and W2 do not fit the LDS (Ch = 384 / 576): their MFMA fragments are read from global memory (L2-resident) — W1's for chunk c+1
// right behind the MFMAs that consumed chunk c's, W2's at the top of the chunk that multiplies them two phases later; the bias columns of W1
// are made in registers from b1.  NF = cout blocks of 32 (accumulator sets).
#ifndef MB3_BD
#define MB3_BD 1
#endif
template <int S, int KS1, int NW, bool WREG, int TH = (S == 1 ? 4 : 2), bool WG = false, int NF = 2, bool CLIP = false>   // CLIP: bounded activations (instantiated for the no-expand form only)
__global__ __launch_bounds__(NW * 64) void mbconv3_kernel(MbParams p) {
    constexpr int TW = 8, OPX = TH * TW, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, HP = PH * PW, MFA = (HP + 31) / 32, HS = 80, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
    constexpr int KP1 = KS1 * 16;
    const int ChP = p.ChP, NC = ChP >> 5, nfc = (p.Cout + 31) >> 5, CoP = nfc * 32, hs2 = ChP * 2 + 16;
    half_t* const w1l = (half_t*)mb_smem;                                // [ChP][KP1]   (absent with WG)
    unsigned* const wdl = (unsigned*)(w1l + (WG ? 0 : ChP * KP1));       // [10][ChP]: tap t of channel c as {w, w} (two halfs); row 9 = zeros (phase B's fifth k step holds one tap only)
    unsigned char* const w2l = (unsigned char*)(wdl + 10 * ChP);         // [CoP] rows of hs2 bytes   (absent with WG)
    float* const bdl = (float*)(w2l + (WG ? 0 : CoP * hs2));             // [ChP]
    float* const b2l = bdl + ChP;                                        // [CoP]
    unsigned char* const wave_base = (unsigned char*)(b2l + CoP);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const hid = wave_base + wave * ((MFA * 32 + OPX) * HS);// hidden chunk [MFA*32][HS] | depthwise output chunk [OPX][HS]
    unsigned char* const dwo = hid + MFA * 32 * HS;

    // (the loads of a batch are independent of each other: one memory round trip per batch of eight elements, not per element — the plain loops
    //  were 5 + 2 + 6 dependent round trips in front of the first tile of a 144-channel block)
    constexpr int PB = 8;
    if constexpr (KS1 > 0 && !WG) for (int i0 = tid; i0 < ChP * KP1; i0 += NT * PB) {
        half_t v[PB]; float bb[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int i = i0 + j * NT, n = i / KP1, k = i - n * KP1;
            const bool on = i < ChP * KP1 && n < p.Ch;
            v[j] = (on && k < p.Cin) ? p.w1[(long)n * p.Cin + k] : (half_t)0.f;
            bb[j] = (on && k >= p.Cin && k < p.Cin + 2 && p.b1) ? p.b1[n] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int i = i0 + j * NT, n = i / KP1, k = i - n * KP1;
            if (i < ChP * KP1) {
                half_t o = v[j];
                if (n < p.Ch && k >= p.Cin && k < p.Cin + 2) { const half_t bh = (half_t)bb[j]; o = k == p.Cin ? bh : (half_t)(bb[j] - (float)bh); }
                w1l[i] = o;
            }
        }
    }
    for (int i0 = tid; i0 < 10 * ChP; i0 += NT * PB) {
        half_t v[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT, t = i / ChP, c = i - t * ChP; v[j] = (i < 9 * ChP && c < p.Ch) ? p.wd[(long)c * 9 + t] : (half_t)0.f; }
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT; if (i < 10 * ChP) wdl[i] = (unsigned)__builtin_bit_cast(unsigned short, v[j]) * 0x10001u; }
    }
    if constexpr (!WG) for (int i0 = tid; i0 < CoP * (hs2 / 2); i0 += NT * PB) {
        half_t v[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT, n = i / (hs2 / 2), k = i - n * (hs2 / 2); v[j] = (i < CoP * (hs2 / 2) && n < p.Cout && k < p.Ch) ? p.w2[(long)n * p.Ch + k] : (half_t)0.f; }
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT; if (i < CoP * (hs2 / 2)) ((half_t*)w2l)[i] = v[j]; }
    }
    for (int i = tid; i < ChP; i += NT) bdl[i] = (p.bd && i < p.Ch) ? p.bd[i] : 0.f;
    for (int i = tid; i < CoP; i += NT) b2l[i] = (p.b2 && i < p.Cout) ? p.b2[i] : 0.f;
    __syncthreads();                                                     // the only barrier of the kernel

    const bool relu1 = p.sl1 == 0.f && (!CLIP || p.hi1 == INFINITY), relu2 = p.sl2 == 0.f && (!CLIP || p.hi2 == INFINITY);   // block-uniform: plain ReLU takes the packed-max path; a clipped one (ONNX Clip(0, hi)) the general path below
                                                                      // (a packed min inside the fast path cost these kernels 5 registers each — they sit at their spill edge: +6.6 us on MobileNet-v2's 7 x 7 blocks)
    const long T = p.ntiles, GW = (long)gridDim.x * NW, gw = (long)blockIdx.x * NW + wave;
    const int t_begin = (int)(gw * T / GW), t_end = (int)((gw + 1) * T / GW);
    const half8_t hz = {0, 0, 0, 0, 0, 0, 0, 0};
    half8_t ones = hz; ones[0] = (half_t)1.f; ones[1] = (half_t)1.f;
    half8_t fw1[KS1 ? KS1 : 1];
    auto load_w1 = [&](int c) {                                          // A fragments of W1's rows c*32 .. +31 (incl. the two bias columns)
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
            if constexpr (!WG) fw1[ks] = *(const half8_t*)(w1l + (c * 32 + l31) * KP1 + ks * 16 + hi * 8);
            else {
                const int ch = c * 32 + l31, k = ks * 16 + hi * 8;
                half8_t v = hz;
                if (ch < p.Ch) {
                    if (k < p.Cin) v = *(const half8_t*)(p.w1 + (long)ch * p.Cin + k);
                    else if (k == p.Cin && p.b1) { const float b = p.b1[ch]; const half_t bh = (half_t)b; v[0] = bh; v[1] = (half_t)(b - (float)bh); }
                }
                fw1[ks] = v;
            }
        }
    };
    // ---- phase B's lane constants: the depthwise layer as 16x16x32 MFMAs against DIAGONAL weight fragments ----------------------------------------------------
    // D[channel i][pixel j] = sum_k A[i][k] B[k][j] with k = (tap pair member, channel of the 16-group): A[i][k] = wd[tap][i] where k's channel is i, else 0;
    // B[k][j] = hidden value of that channel at pixel j shifted by that tap — a 16-byte LDS read per lane, exactly the slot layout phase A wrote.  Nine taps = five
    // k steps of two taps (the tenth tap row of wdl is zero).  The fp32 accumulation, one rounding to fp16 and act2 are those of the vector-ALU form; only the
    // ORDER of the nine fp32 additions differs.  1/16 of the MFMA's multipliers do work — they were idle: what it buys is 72 v_fma_mix + 18 LDS reads per
    // (pixel, channel vector) turned into 5/16 MFMA + 5/16 reads (profiles/r04_pmc_mbconv_v3.txt: LDS 67 % and VALU 47 % busy before).
    constexpr int NPG = OPX / 16;                                           // pixel groups of 16 = MFMA columns
    const int n16 = lane & 15, g4 = lane >> 4;
    unsigned amask[4];                                                      // lane (row i = n16, k group g4): the one half of its A fragment that is on the diagonal
#pragma unroll
    for (int d2 = 0; d2 < 4; d2++) amask[d2] = ((n16 >> 3) == (g4 & 1) && ((n16 & 7) >> 1) == d2) ? ((n16 & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
    unsigned a_off[5], b_off[NPG][5];                                       // byte offsets into wdl (+ 4 * (chunk base + 16 cg)) / into hid (+ 32 cg)
#pragma unroll
    for (int s5 = 0; s5 < 5; s5++) {
        const int tap = 2 * s5 + (g4 >> 1), tb = tap < 9 ? tap : 8, kh = tb / 3, kw = tb - 3 * kh;
        a_off[s5] = (unsigned)((tap * ChP + n16) * 4);
#pragma unroll
        for (int pg = 0; pg < NPG; pg++) {
            const int op = pg * 16 + n16, oy = op / TW, ox = op - oy * TW;
            b_off[pg][s5] = (unsigned)(((oy * S + kh) * PW + ox * S + kw) * HS + (g4 & 1) * 16);
        }
    }
#pragma unroll 1
    for (int t = t_begin; t < t_end; t++) {
        const int n = (int)fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y), by = (int)fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        const int oy0 = by * TH, ox0 = (r - by * p.tiles_x) * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        const half_t* const xim = p.x + (long)n * p.H * p.W * p.x_ld;
        // ---- X fragments of the whole patch: [row block][k step], 16 bytes per lane, kept for all chunks ------------------------------
        half8_t fx[MFA][KS1 ? KS1 : 1];
#pragma unroll
        for (int mf = 0; mf < (KS1 ? MFA : 0); mf++) {
            const int pp = mf * 32 + l31, py = pp / PW, px = pp - py * PW, iy = iy0 + py, ix = ix0 + px;
            const bool inimg = pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const half_t* src = xim + ((long)iy * p.W + ix) * p.x_ld;
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                const int k = ks * 16 + hi * 8;
                half8_t v = hz;
                if (inimg) { if (k < p.Cin) v = MB_X_LOAD(src + k); else if (k == p.Cin) v = ones; }
                fx[mf][ks] = v;
            }
        }
        floatx16 acc2[NF];
#pragma unroll
        for (int a = 0; a < NF; a++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[a][i] = 0.f;
        if constexpr (KS1 > 0 && WG) load_w1(0);
#pragma unroll 1
        for (int c = 0; c < NC; c++) {
            half8_t fw2g[WG ? NF : 1][2];                                // WG: this chunk's W2 fragments, in flight across phases A and B
            if constexpr (WG) {
#pragma unroll
                for (int a = 0; a < NF; a++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        const int co = a * 32 + l31, k = c * 32 + ks * 16 + hi * 8;
                        fw2g[a][ks] = (co < p.Cout && k < p.Ch) ? *(const half8_t*)(p.w2 + (long)co * p.Ch + k) : hz;
                    }
            }
            // ---- phase A: hidden chunk = act1(W1[c] (incl. bias columns) x patch) -> wave-private LDS ------------------------------------
            if (KS1 == 0) {
                // no expand layer: channels c*32 .. +31 of the zero-padded X patch, 16 bytes per (pixel, channel vector) slot
#pragma unroll
                for (int i2 = 0; i2 < MFA * 2; i2++) {
                    const int sl = lane + 64 * i2, pp = sl >> 2, c8 = sl & 3, py = pp / PW, px = pp - py * PW, iy = iy0 + py, ix = ix0 + px, ch = c * 32 + c8 * 8;
                    half8_t v = hz;
                    if (pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && ch < p.Ch) v = *(const half8_t*)(xim + ((long)iy * p.W + ix) * p.x_ld + ch);
                    *(half8_t*)(hid + pp * HS + c8 * 16) = v;
                }
            }
            // WG: all row blocks multiplied first, so that the next chunk's W1 fragments can be requested behind the MFMAs that read this chunk's
            floatx16 accA[(KS1 && WG) ? MFA : 1];
            if constexpr (KS1 > 0 && WG) {
#pragma unroll
                for (int mf = 0; mf < MFA; mf++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) accA[mf][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS1; ks++) accA[mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1[ks], fx[mf][ks], accA[mf], 0, 0, 0);
                }
                if (c + 1 < NC) load_w1(c + 1);
            } else if constexpr (KS1 > 0) load_w1(c);
#pragma unroll
            for (int mf = 0; mf < (KS1 ? MFA : 0); mf++) {
                floatx16 acc;
                if constexpr (WG) acc = accA[mf];
                else {
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS1; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1[ks], fx[mf][ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    half4_t o;
                    if (relu1) {
#pragma unroll
                        for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)acc[4 * q + r2];
                        o = __builtin_elementwise_max(o, half4_t{0, 0, 0, 0});
                    } else {
#pragma unroll
                        for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)relu_family<CLIP>(acc[4 * q + r2], p.sl1, p.hi1);
                    }
                    *(half4_t*)(hid + (mf * 32 + l31) * HS + (8 * q + 4 * hi) * 2) = o;
                }
            }
            asm volatile("" ::: "memory");                                // (one wave: its LDS operations execute in order — no wait; this orders the compiler)
            // ---- phase B: depthwise 3x3 on the chunk: NPG pixel groups x 2 channel groups of 16, five MFMAs each (lane constants above) -------------------
            {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                floatx4 accB[NPG][2];
#pragma unroll
                for (int cg = 0; cg < 2; cg++) {
                    const floatx4 b4 = *(const floatx4*)(bdl + c * 32 + cg * 16 + 4 * g4);
#pragma unroll
                    for (int pg = 0; pg < NPG; pg++) accB[pg][cg] = b4;
                }
                // (a unit = one k step of one channel group: its reads are issued BD units ahead of the MFMAs that consume them, pinned: left alone, the scheduler
                //  issues every read right in front of its MFMA and the wave sits out an LDS round trip per k step)
                constexpr int BD = KS1 ? MB3_BD : 0, NU = 10;                 // (no read-ahead in the no-expand form: its 16-wave instantiations spill with it)
                unsigned wv[BD + 1]; half8_t fb[BD + 1][NPG];
                auto ld_b = [&](auto uc) {
                    constexpr int u = decltype(uc)::value, s5 = u >> 1, cg = u & 1, buf = u % (BD + 1);
                    wv[buf] = *(const unsigned*)((const unsigned char*)wdl + a_off[s5] + (c * 32 + cg * 16) * 4);
#pragma unroll
                    for (int pg = 0; pg < NPG; pg++) fb[buf][pg] = *(const half8_t*)(hid + b_off[pg][s5] + cg * 32);
                };
                sm_static_for<0, BD>([&](auto uc) { ld_b(uc); });
                sm_static_for<0, NU>([&](auto uc) {
                    constexpr int u = decltype(uc)::value, cg = u & 1, buf = u % (BD + 1);
                    if constexpr (u + BD < NU) ld_b(std::integral_constant<int, u + BD>());
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned w = wv[buf];
                    const half8_t fa = __builtin_bit_cast(half8_t, u32x4{w & amask[0], w & amask[1], w & amask[2], w & amask[3]});
#pragma unroll
                    for (int pg = 0; pg < NPG; pg++) accB[pg][cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[buf][pg], accB[pg][cg], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int pg = 0; pg < NPG; pg++)
#pragma unroll
                    for (int cg = 0; cg < 2; cg++) {
                        half4_t o;
                        if (relu2) {
#pragma unroll
                            for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)accB[pg][cg][r2];
                            o = __builtin_elementwise_max(o, half4_t{0, 0, 0, 0});
                        } else {
#pragma unroll
                            for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)relu_family<CLIP>(accB[pg][cg][r2], p.sl2, p.hi2);
                        }
                        *(half4_t*)(dwo + (pg * 16 + n16) * HS + cg * 32 + g4 * 8) = o;
                    }
            }
            asm volatile("" ::: "memory");
            // ---- phase C: acc2 += W2[:, chunk] x depthwise chunk ---------------------------------------------------------------------
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const half8_t fy = *(const half8_t*)(dwo + l31 * HS + (ks * 2 + hi) * 16);
#pragma unroll
                for (int a = 0; a < NF; a++)
                    if (a < nfc) {
                        half8_t fw2;
                        if constexpr (WG) fw2 = fw2g[a][ks]; else fw2 = *(const half8_t*)(w2l + (a * 32 + l31) * hs2 + (c * 32 + ks * 16 + hi * 8) * 2);
                        acc2[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw2, fy, acc2[a], 0, 0, 0);
                    }
            }
            asm volatile("" ::: "memory");                                // the chunk's reads are issued before the next chunk's writes to hid / dwo (in-order LDS: no wait)
        }
        // ---- epilogue: + b2 (+ X), act3, 16-byte stores; lane = output pixel l31 of the tile ------------------------------------------
        const int oy = oy0 + l31 / TW, ox = ox0 + l31 % TW;
        const bool pin = l31 < OPX && oy < p.Ho && ox < p.Wo;               // (2 x 8 tiles: lanes 16..31 of a half-wave carry no pixel)
        const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int a = 0; a < NF; a++)
            if (a < nfc) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float o[8];
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) {
                        unsigned ua = __builtin_bit_cast(unsigned, acc2[a][4 * (2 * j) + r2] + b2l[a * 32 + 8 * (2 * j) + 4 * hi + r2]);
                        unsigned ub = __builtin_bit_cast(unsigned, acc2[a][4 * (2 * j + 1) + r2] + b2l[a * 32 + 8 * (2 * j + 1) + 4 * hi + r2]);
                        sm_permlane32_swap(ua, ub);
                        o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                    }
                    const int co = a * 32 + 16 * j + 8 * hi;
                    if (pin && co < p.Cout) {
                        if (p.res) { const half8_t rv = *(const half8_t*)(xim + ((long)oy * p.W + ox) * p.x_ld + co);   // stride 1, Cin == Cout: the block input at the same pixel
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) o[r2] += (float)rv[r2]; }
                        half8_t ov;
                        if (p.sl3 == 1.f && (!CLIP || p.hi3 == INFINITY)) {                                  // no activation behind the projection (the usual linear bottleneck): 4 conversions instead of 8 compare / multiply / select
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)o[r2];
                        } else {
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)relu_family<CLIP>(o[r2], p.sl3, p.hi3);
                        }
                        *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                    }
                }
            }
    }
}

// ---- v5: the wave pipeline with the hidden patch in REGISTERS -------------------------------------------------------------------------------------------------------------
// v3's phase B is bound by the LDS the four SIMDs share (profiles/r04_mbconv_v3_phases.txt): every hidden value is read back once per tap.  Here the expand GEMM is
// turned around — D[patch pixel][channel] = X[pixel][k] W1[k][channel], the same two fragments with the MFMA's operands swapped — so a LANE holds one hidden channel and
// its accumulator registers the pixels: the depthwise layer becomes a stencil over the lane's own registers (v_fma_mix_f32 on packed fp16 pairs, nine per output, fp32
// accumulation, the same single rounding of the hidden tensor to fp16), and the hidden patch never touches the LDS at all.  The two halves of the wave (lanes 0..31 /
// 32..63 = the two row groups of a 32x32 accumulator) own the left / right four output columns of the 4 x 8 (stride 2: 2 x 8) tile; the upper half keeps its window
// MIRRORED (window column j = patch column PW-1-j, output column 7-j, taps read from a kw-mirrored copy of the depthwise weights), so both halves run the same
// instructions on the same register numbers; the WC = 6 (stride 2: 9) window columns of a half overlap the other half's by 2 (1) columns that are computed twice —
// MFMA work, which is idle here — instead of exchanged.  Only the depthwise OUTPUT passes the LDS (2 bytes per lane and pixel into the wave's [pixel][channel] region):
// the transposition the projection GEMM's B operand needs; phase C and the epilogue are v3's.  Expand forms with W1 / W2 in the LDS (K = Cin + 2 <= 48), and the image-stem form.
// (plain C, not an asm statement: behind a v_dot2c the hardware wants wait states before another vector-ALU instruction reads the result, and the compiler's
//  hazard recogniser does not look into inline asm — a v_fma_mix_f32 written as asm read the accumulator one instruction too early)
template <int X0, int X1> __device__ __forceinline__ void mb_fma_mix(float& a, unsigned x, unsigned w) {    // a += half X0 of x * half X1 of w, one v_fma_mix_f32
    a = __builtin_fmaf((float)__builtin_bit_cast(half2_t, x)[X0], (float)__builtin_bit_cast(half2_t, w)[X1], a);
}
template <int S, int KS1, int NW, int NF = 1, int RELU = 1>   // RELU: 1 = act1 and act2 are ReLU, 2 = ReLU with an upper bound (ONNX Clip(0, hi): packed max, packed min), 0 = any member of the leaky family (slopes p.sl1 / p.sl2, bounds p.hi1 / p.hi2)
__global__ __launch_bounds__(NW * 64) void mbconv5_kernel(MbParams p) {
    constexpr int TW = 8, TH = S == 1 ? 4 : 2, OPX = TH * TW, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, WC = 3 * S + 3, WPR = (WC + 1) / 2;
    constexpr int PXH = PH * WC, MFA = (PXH + 15) / 16, HS = 80, NT = NW * 64, KP1 = KS1 * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
    const int ChP = p.ChP, NC = ChP >> 5, nfc = (p.Cout + 31) >> 5, CoP = nfc * 32, hs2 = ChP * 2 + 16;
    half_t* const w1l = (half_t*)mb_smem;                                // [ChP][KP1]
    half_t* const wdt = w1l + ChP * KP1;                                 // [ChP][2][3][4]: a channel's taps per kernel row as the pairs (w0, w1), (w1, w2), plain and kw-mirrored (three 8-byte reads)
    unsigned char* const w2l = (unsigned char*)(wdt + ChP * 24);         // [CoP] rows of hs2 bytes
    float* const bdl = (float*)(w2l + CoP * hs2);                        // [ChP]
    float* const b2l = bdl + ChP;                                        // [CoP]
    unsigned char* const wave_base = (unsigned char*)(b2l + CoP);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const dwo = wave_base + wave * (32 * HS);             // depthwise output chunk [32 pixels][HS] (stride 2: 16 pixels used)
    const half8_t hz0 = {0, 0, 0, 0, 0, 0, 0, 0};

    constexpr int PB = 8;
    // W1 rows as 16-byte vectors (Cin % 8 == 0, rows 16-byte aligned: checked by the launcher), the two bias columns behind them: with 2-byte loads the prologue
    // of a 384 x 64 layer was fourteen memory round trips (28 of the block's 65 us)
    for (int i0 = tid; i0 < ChP * (KP1 / 8); i0 += NT * PB) {
        half8_t v[PB]; float bb[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int i = i0 + j * NT, n = i / (KP1 / 8), k = (i - n * (KP1 / 8)) * 8;
            const bool on = i < ChP * (KP1 / 8) && n < p.Ch;
            v[j] = (on && k < p.Cin) ? *(const half8_t*)(p.w1 + (long)n * p.Cin + k) : hz0;
            bb[j] = (on && k == p.Cin && p.b1) ? p.b1[n] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int i = i0 + j * NT, n = i / (KP1 / 8), k = (i - n * (KP1 / 8)) * 8;
            if (i < ChP * (KP1 / 8)) {
                half8_t o = v[j];
                if (n < p.Ch && k == p.Cin) { const half_t bh = (half_t)bb[j]; o[0] = bh; o[1] = (half_t)(bb[j] - (float)bh); }
                *(half8_t*)(w1l + n * KP1 + k) = o;
            }
        }
    }
    for (int i0 = tid; i0 < ChP * 24; i0 += NT * PB) {
        half_t v[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int i = i0 + j * NT, c = i / 24, r = i - c * 24, mir = r >= 12, t = r - 12 * mir, kh = t >> 2, e = t & 3, kw = e < 2 ? e : e - 1;   // (w0, w1 | w1, w2) per kernel row
            v[j] = (i < ChP * 24 && c < p.Ch) ? p.wd[(long)c * 9 + kh * 3 + (mir ? 2 - kw : kw)] : (half_t)0.f;
        }
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT; if (i < ChP * 24) wdt[i] = v[j]; }
    }
    for (int i0 = tid; i0 < CoP * (hs2 / 16); i0 += NT * PB) {             // W2 rows, 16 bytes at a time (the row's last slot is the padding)
        half8_t v[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT, n = i / (hs2 / 16), k = (i - n * (hs2 / 16)) * 8; v[j] = (i < CoP * (hs2 / 16) && n < p.Cout && k < p.Ch) ? *(const half8_t*)(p.w2 + (long)n * p.Ch + k) : hz0; }
#pragma unroll
        for (int j = 0; j < PB; j++) { const int i = i0 + j * NT, n = i / (hs2 / 16), k = (i - n * (hs2 / 16)) * 8; if (i < CoP * (hs2 / 16)) *(half8_t*)(w2l + n * hs2 + k * 2) = v[j]; }
    }
    for (int i = tid; i < ChP; i += NT) bdl[i] = (p.bd && i < p.Ch) ? p.bd[i] : 0.f;
    for (int i = tid; i < CoP; i += NT) b2l[i] = (p.b2 && i < p.Cout) ? p.b2[i] : 0.f;
    __syncthreads();                                                     // the only barrier of the kernel

    const long T = p.ntiles, GW = (long)gridDim.x * NW, gw = (long)blockIdx.x * NW + wave;
    const int t_begin = (int)(gw * T / GW), t_end = (int)((gw + 1) * T / GW);
    const half8_t hz = {0, 0, 0, 0, 0, 0, 0, 0};
    half8_t ones = hz; ones[0] = (half_t)1.f; ones[1] = (half_t)1.f;
    // row m = l31 of X fragment mf is accumulator element idx = 4 (m / 8) + m % 4 of the lanes of half (m / 4) % 2: window pixel q = 16 mf + idx = (row py, window column j)
    int ppy[MFA], ppx[MFA];                                              // patch row / column of this lane's fragment rows; ppy < 0: the row is beyond the window
#pragma unroll
    for (int mf = 0; mf < MFA; mf++) {
        const int hm = (l31 >> 2) & 1, q = mf * 16 + (l31 >> 3) * 4 + (l31 & 3), py = q / WC, j = q - py * WC;
        ppy[mf] = q < PXH ? py : -1000000; ppx[mf] = hm ? PW - 1 - j : j;
    }
    unsigned woff[4];                                                    // depthwise-output slots of this lane's four output columns (+ row * TW * HS)
#pragma unroll
    for (int oxl = 0; oxl < 4; oxl++) woff[oxl] = (unsigned)((hi ? 7 - oxl : oxl) * HS + l31 * 2);
    half8_t fw1[KS1];
#pragma unroll 1
    for (int t = t_begin; t < t_end; t++) {
        const int n = (int)fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y), by = (int)fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        const int oy0 = by * TH, ox0 = (r - by * p.tiles_x) * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        const half_t* const xim = p.x + (long)n * p.H * p.W * p.x_ld;
        // ---- X fragments of the window pixels: [fragment][k step], 16 bytes per lane, kept for all chunks ----------------------------------------------------------
        half8_t fx[MFA][KS1];
#pragma unroll
        for (int mf = 0; mf < MFA; mf++) {
            const int iy = iy0 + ppy[mf], ix = ix0 + ppx[mf];
            const bool inimg = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const half_t* src = xim + ((long)iy * p.W + ix) * p.x_ld;
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                const int k = ks * 16 + hi * 8;
                half8_t v = hz;
                if (inimg) { if (k < p.Cin) v = MB_X_LOAD(src + k); else if (k == p.Cin) v = ones; }
                fx[mf][ks] = v;
            }
        }
    
        floatx16 acc2[NF];
#pragma unroll
        for (int a = 0; a < NF; a++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[a][i] = 0.f;
#pragma unroll 1
        for (int c = 0; c < NC; c++) {
            // ---- phase A: hidden[window pixel][channel] = act1(X W1[c]^T): lane = channel c*32 + l31 of its half's window -----------------------------------------------
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) fw1[ks] = *(const half8_t*)(w1l + (c * 32 + l31) * KP1 + ks * 16 + hi * 8);
            const int chl = c * 32 + l31;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            unsigned wqv[6];                                             // this lane's taps (its half's orientation): [2 kh] = (w0, w1), [2 kh + 1] = (w1, w2)
#pragma unroll
            for (int i = 0; i < 3; i++) { const u32x2 w2 = *(const u32x2*)(wdt + (chl * 2 + hi) * 12 + i * 4); wqv[2 * i] = w2[0]; wqv[2 * i + 1] = w2[1]; }
            const float bdv = bdl[chl];
            floatx16 accA[MFA];
#pragma unroll
            for (int mf = 0; mf < MFA; mf++) {
#pragma unroll
                for (int i = 0; i < 16; i++) accA[mf][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS1; ks++) accA[mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[mf][ks], fw1[ks], accA[mf], 0, 0, 0);
            }
            unsigned hw[PH][WPR];                                        // the window: row py, columns (2 q, 2 q + 1) as an fp16 pair
#pragma unroll
            for (int py = 0; py < PH; py++)
#pragma unroll
                for (int q = 0; q < WPR; q++) {
                    const int q0 = py * WC + 2 * q, q1 = q0 + 1;
                    const float a = accA[q0 >> 4][q0 & 15], b = (2 * q + 1 < WC) ? accA[q1 >> 4][q1 & 15] : 0.f;
                    half2_t o;
                    if constexpr (RELU != 0) { o[0] = (half_t)a; o[1] = (half_t)b; o = __builtin_elementwise_max(o, half2_t{0, 0}); if constexpr (RELU == 2) o = __builtin_elementwise_min(o, half2_t{(half_t)p.hi1, (half_t)p.hi1}); }
                    else { o[0] = (half_t)relu_family<true>(a, p.sl1, p.hi1); o[1] = (half_t)relu_family<true>(b, p.sl1, p.hi1); }   // (<true> — act1 = Clip next to a non-ReLU act2 lands here; a bound of +inf is a no-op min)
                    hw[py][q] = __builtin_bit_cast(unsigned, o);
                }
            // ---- phase B: depthwise 3x3 over the lane's window: TH x 4 outputs, nine fp32 FMAs each -------------------------------------------------------------------
            asm volatile("" ::: "memory");                                // (phase C's reads of the previous chunk are issued before these writes: in-order LDS, no wait)
            // (per kernel row one v_dot2c_f32_f16 — the two window columns that share a register against (w0, w1) or (w1, w2) — and one v_fma_mix_f32 for the third
            //  column: 6 instead of 9 vector-ALU instructions per output, all accumulating in fp32)
            sm_static_for<0, TH * 2>([&](auto oc) {
                constexpr int oy = decltype(oc)::value >> 1, op2 = decltype(oc)::value & 1;
                float a2[2];
                sm_static_for<0, 2>([&](auto ec) {
                    constexpr int e = decltype(ec)::value, j0 = (2 * op2 + e) * S;                       // output column 2 op2 + e of the half: window columns j0 .. j0 + 2
                    float a = bdv;
                    sm_static_for<0, 3>([&](auto kc) {
                        constexpr int kh = decltype(kc)::value, row = oy * S + kh;
                        if constexpr ((j0 & 1) == 0) {
                            a = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, hw[row][j0 >> 1]), __builtin_bit_cast(half2_t, wqv[2 * kh]), a, false);
                            mb_fma_mix<0, 1>(a, hw[row][(j0 >> 1) + 1], wqv[2 * kh + 1]);
                        } else {
                            a = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, hw[row][(j0 + 1) >> 1]), __builtin_bit_cast(half2_t, wqv[2 * kh + 1]), a, false);
                            mb_fma_mix<1, 0>(a, hw[row][j0 >> 1], wqv[2 * kh]);
                        }
                    });
                    a2[e] = a;
                });
                half2_t o;
                if constexpr (RELU != 0) { o[0] = (half_t)a2[0]; o[1] = (half_t)a2[1]; o = __builtin_elementwise_max(o, half2_t{0, 0}); if constexpr (RELU == 2) o = __builtin_elementwise_min(o, half2_t{(half_t)p.hi2, (half_t)p.hi2}); }
                else { o[0] = (half_t)relu_family<true>(a2[0], p.sl2, p.hi2); o[1] = (half_t)relu_family<true>(a2[1], p.sl2, p.hi2); }
                *(half_t*)(dwo + woff[2 * op2] + oy * TW * HS) = o[0];
                *(half_t*)(dwo + woff[2 * op2 + 1] + oy * TW * HS) = o[1];
            });
            asm volatile("" ::: "memory");
            // ---- phase C: acc2 += W2[:, chunk] x depthwise chunk -----------------------------------------------------------------------------------------------------
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const half8_t fy = *(const half8_t*)(dwo + l31 * HS + (ks * 2 + hi) * 16);
#pragma unroll
                for (int a = 0; a < NF; a++)
                    if (a < nfc) {
                        const half8_t fw2 = *(const half8_t*)(w2l + (a * 32 + l31) * hs2 + (c * 32 + ks * 16 + hi * 8) * 2);
                        acc2[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw2, fy, acc2[a], 0, 0, 0);
                    }
            }
        }
        // ---- epilogue: + b2 (+ X), act3, 16-byte stores; lane = output pixel l31 of the tile --------------------------------------------------------------------------
        const int oy = oy0 + l31 / TW, ox = ox0 + l31 % TW;
        const bool pin = l31 < OPX && oy < p.Ho && ox < p.Wo;
        const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int a = 0; a < NF; a++)
            if (a < nfc) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float o[8];
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) {
                        unsigned ua = __builtin_bit_cast(unsigned, acc2[a][4 * (2 * j) + r2] + b2l[a * 32 + 8 * (2 * j) + 4 * hi + r2]);
                        unsigned ub = __builtin_bit_cast(unsigned, acc2[a][4 * (2 * j + 1) + r2] + b2l[a * 32 + 8 * (2 * j + 1) + 4 * hi + r2]);
                        sm_permlane32_swap(ua, ub);
                        o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                    }
                    const int co = a * 32 + 16 * j + 8 * hi;
                    if (pin && co < p.Cout) {
                        if (p.res) { const half8_t rv = *(const half8_t*)(xim + ((long)oy * p.W + ox) * p.x_ld + co);
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) o[r2] += (float)rv[r2]; }
                        half8_t ov;
                        if (p.sl3 == 1.f) {                                  // no activation behind the projection (the usual linear bottleneck): 4 conversions instead of 8 compare / multiply / select
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)o[r2];
                        } else {
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)relu_family<true>(o[r2], p.sl3, p.hi3);
                        }
                        *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                    }
                }
            }
    }
}

// ---- v5 behind an image stem, the input patch STAGED: mbconv5s_kernel ---------------------------------------------------------------------------------------------------
// The image-stem forms above gather the 27 taps of every patch pixel from the fp32 planes with 4-byte loads (32..48 per lane and tile, ~6 vector-ALU instructions of
// address arithmetic each): MobileNet-v2's first block spent most of its 215 us there.  Here a block of eight waves owns a BLOCK TILE of 2 x 4 wave tiles ((2 TH) x 32
// output pixels); the waves copy the input region under it — rows as coalesced segments, one (plane, row) per wave and step: scalar address arithmetic — into the LDS as
// fp16 PIXEL SLOTS of 8 bytes (c0, c1, c2, -), so the three taps of a kernel row are 24 contiguous bytes and the expand GEMM's A fragment of a patch pixel is ONE
// ds_read2_b64 per k step at a lane-constant address: K is laid out as (kh, kw in 0..3, c in 0..3) = 48 = three k steps, with zeros in W1 at the padding positions, and
// the fourth "kw" slot of kernel row 1 — overwritten in registers — carries the two bias columns (1, 1 inside the hidden image: a hidden pixel outside is exactly 0, the
// depthwise layer's padding).  Two blocks per CU (launch bounds): one stages while the other computes.  Phases A (operands swapped: lane = channel), B (register
// stencil), C and the epilogue are mbconv5_kernel's.
template <int S, int NF, int RELU>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void mbconv5s_kernel(MbParams p) {   // (128 registers: two blocks per CU)
    constexpr int NW = 8, TW = 8, TH = S == 1 ? 4 : 2, OPX = TH * TW, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, WC = 3 * S + 3, WPR = (WC + 1) / 2;
    constexpr int PXH = PH * WC, MFA = (PXH + 15) / 16, HS = 80, NT = NW * 64, KP1 = 48, BWY = 2, BWX = 4;
    constexpr int HR = (BWY * TH - 1) * S + 3, HC = (BWX * TW - 1) * S + 3;                 // hidden region of a block tile (rows, columns)
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
    const int ChP = p.ChP, nfc = (p.Cout + 31) >> 5, CoP = nfc * 32, hs2 = ChP * 2 + 16;
    constexpr int NSG = 8, NPS = S == 1 ? 2 : 3, pitch = 64 * NPS + 2;     // staging: segments per wave (3 IR <= 64), 64-column passes (IC <= 64 NPS: checked by the launcher); LDS row pitch in pixel slots
    const int st = p.st_s, IR = (HR - 1) * st + 3, IC = (HC - 1) * st + 3;              // input region (rows, columns); columns IC .. 64 NPS - 1 are staged as zeros, the last two stay zero
    half_t* const w1l = (half_t*)mb_smem;                                // [ChP][48]: k = 16 kh + 4 kw + c
    half_t* const wdt = w1l + ChP * KP1;                                 // [ChP][2][3][4] (mbconv5_kernel)
    unsigned char* const w2l = (unsigned char*)(wdt + ChP * 24);         // [CoP] rows of hs2 bytes
    float* const bdl = (float*)(w2l + CoP * hs2);                        // [ChP]
    float* const b2l = bdl + ChP;                                        // [CoP]
    unsigned char* const reg = (unsigned char*)(b2l + CoP);              // input region: [IR][pitch] pixel slots of 8 bytes
    unsigned char* const wave_base = reg + (size_t)IR * pitch * 8;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wy = wave >> 2, wx = wave & 3;
    unsigned char* const dwo = wave_base + wave * (32 * HS);

    for (int i = tid; i < ChP * KP1; i += NT) {
        const int n = i / KP1, k = i - n * KP1, kh = k >> 4, kw = (k >> 2) & 3, c = k & 3;
        half_t v = (half_t)0.f;
        if (n < p.Ch) {
            if (kw < 3 && c < 3) v = p.w1[(long)n * 27 + (kh * 3 + kw) * 3 + c];
            else if (kh == 1 && kw == 3 && c < 2 && p.b1) { const float b = p.b1[n]; const half_t bh = (half_t)b; v = c == 0 ? bh : (half_t)(b - (float)bh); }
        }
        w1l[i] = v;
    }
    for (int i = tid; i < ChP * 24; i += NT) {
        const int c = i / 24, r = i - c * 24, mir = r >= 12, t = r - 12 * mir, kh = t >> 2, e = t & 3, kw = e < 2 ? e : e - 1;
        wdt[i] = c < p.Ch ? p.wd[(long)c * 9 + kh * 3 + (mir ? 2 - kw : kw)] : (half_t)0.f;
    }
    for (int i = tid; i < CoP * (hs2 / 2); i += NT) { const int n = i / (hs2 / 2), k = i - n * (hs2 / 2); ((half_t*)w2l)[i] = (n < p.Cout && k < p.Ch) ? p.w2[(long)n * p.Ch + k] : (half_t)0.f; }
    for (int i = tid; i < ChP; i += NT) bdl[i] = (p.bd && i < p.Ch) ? p.bd[i] : 0.f;
    for (int i = tid; i < CoP; i += NT) b2l[i] = (p.b2 && i < p.Cout) ? p.b2[i] : 0.f;
    for (int i = tid; i < IR * pitch * 2; i += NT) ((unsigned*)reg)[i] = 0u;          // (the slots' fourth halfs and the spare column are never written again)

    // window pixel (row py, patch column px) of this lane's fragment rows (mbconv5_kernel); rows beyond the window read pixel (0, 0) and are never used
    int ppy[MFA], ppx[MFA]; bool pon[MFA];
    unsigned abase[MFA][3];                                              // byte offset in `reg` of the lane's two slots (kw = 2 hi, 2 hi + 1) of kernel row kh, per fragment
#pragma unroll
    for (int mf = 0; mf < MFA; mf++) {
        const int hm = (l31 >> 2) & 1, q = mf * 16 + (l31 >> 3) * 4 + (l31 & 3), py = q / WC, j = q - py * WC;
        pon[mf] = q < PXH; ppy[mf] = pon[mf] ? py : 0; ppx[mf] = pon[mf] ? (hm ? PW - 1 - j : j) : 0;
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
            abase[mf][kh] = (unsigned)((((wy * TH * S + ppy[mf]) * st + kh) * pitch + (wx * TW * S + ppx[mf]) * st + 2 * hi) * 8);
    }
    unsigned woff[4];
#pragma unroll
    for (int oxl = 0; oxl < 4; oxl++) woff[oxl] = (unsigned)((hi ? 7 - oxl : oxl) * HS + l31 * 2);
    const unsigned plane = (unsigned)p.in_H * (unsigned)p.in_W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xf, 0, (int)((unsigned)p.N * 3u * plane * 4u), 0x00020000);
    const int nseg = 3 * IR;
    __syncthreads();
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    half8_t fw1[3];                                                      // one chunk (Ch <= 32): the lane's weights stay in registers for the whole kernel
#pragma unroll
    for (int ks = 0; ks < 3; ks++) fw1[ks] = *(const half8_t*)(w1l + l31 * KP1 + ks * 16 + hi * 8);
    unsigned wqv[6];
#pragma unroll
    for (int i = 0; i < 3; i++) { const u32x2 w2 = *(const u32x2*)(wdt + (l31 * 2 + hi) * 12 + i * 4); wqv[2 * i] = w2[0]; wqv[2 * i + 1] = w2[1]; }
    const float bdv = bdl[l31];

#pragma unroll 1
    for (int bt = blockIdx.x; bt < p.ntiles; bt += gridDim.x) {
        const int n = (int)fastdiv((unsigned)bt, p.txy_mul, p.txy_sh);
        const int r = bt - n * (p.tiles_x * p.tiles_y), by = (int)fastdiv((unsigned)r, p.tx_mul, p.tx_sh), bx = r - by * p.tiles_x;
        const int hy0 = by * (BWY * TH) * S - 1, hx0 = bx * (BWX * TW) * S - 1;           // hidden pixel of the region's corner
        const int iry0 = hy0 * st - p.st_p, irx0 = hx0 * st - p.st_p;                     // input pixel of the region's corner
        // ---- stage the input region: wave w copies (plane, row) segments w, w + 8, ...; four segments in flight per lane and pass -------------------------------------
        // (all of a lane's loads are issued before the first conversion: one memory round trip per block tile, not one per batch)
        {
            float g[NPS][NSG];
#pragma unroll
            for (int ps = 0; ps < NPS; ps++) {
                const int x = lane + 64 * ps, gx = irx0 + x;
                const bool colok = x < IC && (unsigned)gx < (unsigned)p.in_W;
#pragma unroll
                for (int i = 0; i < NSG; i++) {
                    const int sg = wave + i * NW, c = sg >= 2 * IR ? 2 : sg >= IR ? 1 : 0, rr = sg - c * IR, gy = iry0 + rr;
                    const bool ok = colok && sg < nseg && (unsigned)gy < (unsigned)p.in_H;
                    const unsigned off = ok ? ((((unsigned)n * 3u + (unsigned)c) * (unsigned)p.in_H + (unsigned)gy) * (unsigned)p.in_W + (unsigned)gx) * 4u : 0x80000000u;
                    g[ps][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, SM_STEM_IN_AUX));
                }
            }
#pragma unroll
            for (int ps = 0; ps < NPS; ps++) {
                const int x = lane + 64 * ps;
#pragma unroll
                for (int i = 0; i < NSG; i++) {
                    const int sg = wave + i * NW, c = sg >= 2 * IR ? 2 : sg >= IR ? 1 : 0, rr = sg - c * IR;
                    if (sg < nseg) *(half_t*)(reg + ((rr * pitch + x) * 4 + c) * 2) = (half_t)g[ps][i];   // (wave-uniform condition; columns beyond IC were loaded as zeros)
                }
            }
        }
        __syncthreads();
        const int oy0 = by * (BWY * TH) + wy * TH, ox0 = bx * (BWX * TW) + wx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        if (oy0 < p.Ho && ox0 < p.Wo) {
            // ---- phase A: hidden[window pixel][channel] = act1(X W1^T), X fragments straight from the staged region ------------------------------------------------
            floatx16 accA[MFA];
            const bool border = iy0 < 0 || ix0 < 0 || iy0 + PH > p.H || ix0 + PW > p.W;   // wave-uniform: only tiles on the image's edge pay for the masking
#pragma unroll
            for (int mf = 0; mf < MFA; mf++) {
                const int iy = iy0 + ppy[mf], ix = ix0 + ppx[mf];
                const bool inimg = pon[mf] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned onesv = inimg ? 0x3c003c00u : 0u;            // (1, 1): the bias columns of a pixel inside the hidden image
#pragma unroll
                for (int i = 0; i < 16; i++) accA[mf][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 3; ks++) {
                    const u32x2* const sl = (const u32x2*)(reg + abase[mf][ks]);          // two 8-byte slots: one ds_read2_b64
                    const u32x2 s0v = sl[0], s1v = sl[1];
                    u32x4 d = {s0v[0], s0v[1], s1v[0], s1v[1]};
                    if (hi) { d[2] = ks == 1 ? onesv : 0u; d[3] = 0u; }       // the fourth "kw" slot: the bias columns in kernel row 1, zeros elsewhere (what the LDS holds there is the next pixel)
                    if (border && !inimg) d = u32x4{0u, 0u, 0u, 0u};          // a hidden pixel beyond the right / lower edge still sees input pixels: it must be exactly 0 (the depthwise padding)
                    accA[mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, d), fw1[ks], accA[mf], 0, 0, 0);
                }
            }
            unsigned hw[PH][WPR];
#pragma unroll
            for (int py = 0; py < PH; py++)
#pragma unroll
                for (int q = 0; q < WPR; q++) {
                    const int q0 = py * WC + 2 * q, q1 = q0 + 1;
                    const float a = accA[q0 >> 4][q0 & 15], b = (2 * q + 1 < WC) ? accA[q1 >> 4][q1 & 15] : 0.f;
                    half2_t o;
                    if constexpr (RELU != 0) { o[0] = (half_t)a; o[1] = (half_t)b; o = __builtin_elementwise_max(o, half2_t{0, 0}); if constexpr (RELU == 2) o = __builtin_elementwise_min(o, half2_t{(half_t)p.hi1, (half_t)p.hi1}); }
                    else { o[0] = (half_t)relu_family<true>(a, p.sl1, p.hi1); o[1] = (half_t)relu_family<true>(b, p.sl1, p.hi1); }
                    hw[py][q] = __builtin_bit_cast(unsigned, o);
                }
            // ---- phase B (mbconv5_kernel) ----------------------------------------------------------------------------------------------------------------------------
            sm_static_for<0, TH * 2>([&](auto oc) {
                constexpr int oy = decltype(oc)::value >> 1, op2 = decltype(oc)::value & 1;
                float a2[2];
                sm_static_for<0, 2>([&](auto ec) {
                    constexpr int e = decltype(ec)::value, j0 = (2 * op2 + e) * S;
                    float a = bdv;
                    sm_static_for<0, 3>([&](auto kc) {
                        constexpr int kh = decltype(kc)::value, row = oy * S + kh;
                        if constexpr ((j0 & 1) == 0) {
                            a = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, hw[row][j0 >> 1]), __builtin_bit_cast(half2_t, wqv[2 * kh]), a, false);
                            mb_fma_mix<0, 1>(a, hw[row][(j0 >> 1) + 1], wqv[2 * kh + 1]);
                        } else {
                            a = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, hw[row][(j0 + 1) >> 1]), __builtin_bit_cast(half2_t, wqv[2 * kh + 1]), a, false);
                            mb_fma_mix<1, 0>(a, hw[row][j0 >> 1], wqv[2 * kh]);
                        }
                    });
                    a2[e] = a;
                });
                half2_t o;
                if constexpr (RELU != 0) { o[0] = (half_t)a2[0]; o[1] = (half_t)a2[1]; o = __builtin_elementwise_max(o, half2_t{0, 0}); if constexpr (RELU == 2) o = __builtin_elementwise_min(o, half2_t{(half_t)p.hi2, (half_t)p.hi2}); }
                else { o[0] = (half_t)relu_family<true>(a2[0], p.sl2, p.hi2); o[1] = (half_t)relu_family<true>(a2[1], p.sl2, p.hi2); }
                *(half_t*)(dwo + woff[2 * op2] + oy * TW * HS) = o[0];
                *(half_t*)(dwo + woff[2 * op2 + 1] + oy * TW * HS) = o[1];
            });
            asm volatile("" ::: "memory");
            // ---- phase C + epilogue (mbconv5_kernel; one chunk: Ch <= 32) -----------------------------------------------------------------------------------------------
            floatx16 acc2[NF];
#pragma unroll
            for (int a = 0; a < NF; a++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc2[a][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const half8_t fy = *(const half8_t*)(dwo + l31 * HS + (ks * 2 + hi) * 16);
#pragma unroll
                for (int a = 0; a < NF; a++)
                    if (a < nfc) {
                        const half8_t fw2 = *(const half8_t*)(w2l + (a * 32 + l31) * hs2 + (ks * 16 + hi * 8) * 2);
                        acc2[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw2, fy, acc2[a], 0, 0, 0);
                    }
            }
            const int oy = oy0 + l31 / TW, ox = ox0 + l31 % TW;
            const bool pin = l31 < OPX && oy < p.Ho && ox < p.Wo;
            const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
            for (int a = 0; a < NF; a++)
                if (a < nfc) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        float o[8];
#pragma unroll
                        for (int r2 = 0; r2 < 4; r2++) {
                            unsigned ua = __builtin_bit_cast(unsigned, acc2[a][4 * (2 * j) + r2] + b2l[a * 32 + 8 * (2 * j) + 4 * hi + r2]);
                            unsigned ub = __builtin_bit_cast(unsigned, acc2[a][4 * (2 * j + 1) + r2] + b2l[a * 32 + 8 * (2 * j + 1) + 4 * hi + r2]);
                            sm_permlane32_swap(ua, ub);
                            o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                        }
                        const int co = a * 32 + 16 * j + 8 * hi;
                        if (pin && co < p.Cout) {
                            half8_t ov;
                            if (p.sl3 == 1.f && p.hi3 == INFINITY) {                                  // no activation behind the projection (the usual linear bottleneck): 4 conversions instead of 8 compare / multiply / select
#pragma unroll
                                for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)o[r2];
                            } else {
#pragma unroll
                                for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)relu_family<true>(o[r2], p.sl3, p.hi3);
                            }
                            *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                        }
                    }
                }
        }
        mb_lds_barrier();                                                 // every wave has read its fragments: the region may be overwritten
    }
}

// ---- v4: one TILE per block, every hidden chunk shared by the block's eight waves ------------------------------------------------------------------
// v3 is one wave per 4 x 8 tile walking its chunks in a single dependent chain (expand MFMAs -> LDS -> depthwise -> LDS -> project MFMAs): nothing in
// a CU is busy, the 6 x 10 patch of a 4 x 8 tile computes the expand layer 1.9 times, and from Ch = 384 on the weights come from global memory per
// chunk.  Here a block owns a tile of TO x TO output pixels (14 x 14 at stride 1, 7 x 7 at stride 2: a patch of 16 x 16 / 15 x 15 hidden pixels = the
// eight waves' eight 32-pixel fragments; a 14 x 14 or 7 x 7 image is one tile) and the waves cut every phase of a 64-channel chunk between them:
//   A  expand: wave w multiplies the X fragments of patch fragment w (registers, loaded once per tile straight from global memory; pixels outside the
//      image are zero fragments AND have a zero in the bias column, so their hidden value is act1(0) = 0: the depthwise layer's padding) with the
//      chunk's W1 rows (two A fragments per k step from LDS; bias = one more k step against a {hi(b), lo(b)} fragment) and writes act1(.) rounded to
//      fp16 into the block's hidden patch [PW x PW][64 channels];
//   B  depthwise 3x3: the 512 lanes are (column, channel vector of 8, row part) tasks over the tile; at stride 1 a task walks its rows with a
//      three-row window in registers (3 LDS reads + 72 fp32 FMAs per output), act2, one rounding -> [output pixel][64 channels], the B operand of
//   C  project: wave (pf, g) accumulates its cout fragments (a = g mod NG) for output fragment pf over the chunk's four k steps, W2's chunk columns
//      from LDS; the accumulators live across the chunks, epilogue (+ b2, + X, act3, 16-byte stores) per tile.
//   The chunk weights arrive by LDS-DMA in MFMA-fragment order (1 KB per fragment = one wave instruction, lane-linear: conflict-free reads, no
//   registers), two buffers each: phase B of chunk g requests W1 of chunk g+2 and W2 of chunk g+1 and first waits for the group requested one chunk
//   earlier — a whole chunk of flight time; the next tile's X fragments and the residual rows are requested under the last chunk.
//   Two barriers per chunk (A | B | C + next A).
// S = stride, TCF = output fragments of a tile that phase C deals to the waves (7: 14 x 14 tiles, one per wave; 2: <= 64 output pixels, waves =
// (fragment, cout group of four)), KSX = k steps of X (ceil(Cin / 16)), NFW = cout fragments per wave.
__device__ __attribute__((aligned(64))) unsigned int g_mb_zero[16];       // zero-initialised at module load: source of the fragments beyond Ch / Cout / Cin
#define MB_GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

// AS: the patch has at most four fragments (a 7 x 7 image: 81 pixels): waves (fragment, row block) — every wave multiplies ONE of the chunk's two row blocks.
template <int S, int TCF, int KSX, int NFW, bool AS = false, bool CLIP = false>   // CLIP: act1 / act2 / act3 may carry an upper bound (ONNX Clip(0, hi))
__global__ __launch_bounds__(512) void mbconv4_kernel(MbParams p) {
    constexpr int NT = 512, CF = 2, TP = TCF > 4 ? 8 : (TCF > 2 ? 4 : 2), NG = 8 / TP, CK = CF * 32, HSB = CK * 2 + 16, NCV = CF * 4, RPTM = 4, HPA = 256;
    constexpr int F1 = CF * KSX, W1B = F1 * 1024, NI1 = (F1 + 7) / 8, NI2 = (NFW * NG * 2 * CF + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
    const int TO = p.tile_o, PW = (TO - 1) * S + 3, HP = PW * PW, TOO = TO * TO;
    const int ChP = p.ChP, NC = ChP / CK, nfc = (p.Cout + 31) >> 5, CoP = nfc * 32, F2 = nfc * 2 * CF, W2B = F2 * 1024;
    unsigned char* const w1b = mb_smem;                                   // [2][CF][KSX] fragments of 1 KB
    unsigned char* const w2b = w1b + 2 * W1B;                             // [2][nfc][2 CF] fragments
    unsigned char* const hid = w2b + 2 * W2B;                             // [256 patch pixels][HSB]
    unsigned char* const dwo = hid + HPA * HSB;                           // [TCF*32 output pixels][HSB]
    half_t* const wdl = (half_t*)(dwo + TCF * 32 * HSB);                  // [9][ChP]
    float* const bdl = (float*)(wdl + 9 * ChP);                           // [ChP]
    float* const b1l = bdl + ChP;                                         // [ChP]
    float* const b2l = b1l + ChP;                                         // [CoP]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const half8_t hz = {0, 0, 0, 0, 0, 0, 0, 0};
    const half_t* const zero = (const half_t*)g_mb_zero;

    // weights of chunk c (index into the hidden channels) -> buffer b; every wave issues its share (fragment f = wave + 8 i)
    auto dma_w1 = [&](int c, int b) {
#pragma unroll
        for (int i = 0; i < NI1; i++) {
            const int f = wave + 8 * i;
            if (f < F1) {
                const int cf = f / KSX, ks = f - cf * KSX, ch = c * CK + cf * 32 + l31, k = ks * 16 + hi * 8;
                MB_GLDS16((ch < p.Ch && k < p.Cin) ? p.w1 + (long)ch * p.Cin + k : zero, w1b + b * W1B + f * 1024);
            }
        }
    };
    auto dma_w2 = [&](int c, int b) {
#pragma unroll
        for (int i = 0; i < NI2; i++) {
            const int f = wave + 8 * i;
            if (f < F2) {
                const int a = f / (2 * CF), ks = f - a * (2 * CF), co = a * 32 + l31, k = c * CK + ks * 16 + hi * 8;
                MB_GLDS16((co < p.Cout && k < p.Ch) ? p.w2 + (long)co * p.Ch + k : zero, w2b + b * W2B + f * 1024);
            }
        }
    };
#ifdef SM_TIMELINE   // per-phase cycles of wave 0 in scalar registers (a stamp that touches memory would itself be waited for by the kernel's vmcnt(0))
    unsigned long long tl4[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mb_t0 = __builtin_amdgcn_s_memtime();
#define MB4_TL(i) do { const unsigned long long now = __builtin_amdgcn_s_memtime(); tl4[i] += now - mb_t0; mb_t0 = now; } while (0)
#else
#define MB4_TL(i)
#endif
    dma_w1(0, 0); dma_w2(0, 0); dma_w1(NC > 1 ? 1 : 0, 1);
    // ---- lane constants: my patch pixel (phase A), my output pixel (phase C, epilogue) ----
    constexpr int CFA = AS ? 1 : 2;                                       // row blocks of a chunk per wave in phase A
    const int pp = (AS ? (wave & 3) : wave) * 32 + l31, ppy = pp / PW, ppx = pp - ppy * PW;   // (pp >= HP: rows beyond the patch, written and never read)
    const int cfa0 = AS ? (wave >> 2) : 0;                                // my first row block
    const int pf = wave % TP, grp = wave / TP;
    const bool act_pf = pf < TCF;
    const int q = pf * 32 + l31, qy = q / TO, qx = q - qy * TO;
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        n = (int)fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y), by = (int)fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        oy0 = by * TO; ox0 = (r - by * p.tiles_x) * TO;
    };
    constexpr bool PREF = KSX <= 4;                                       // the next tile's X fragments are requested under the last chunk (a second register set: short K only)
    half8_t fx[KSX]; bool inimg;                                          // X fragments of my patch fragment + "my patch pixel is inside the image"
    auto load_x = [&](int t) {
        int n, oy0, ox0; tile_origin(t < p.ntiles ? t : 0, n, oy0, ox0);
        const int iy = oy0 * S - 1 + ppy, ix = ox0 * S - 1 + ppx;
        inimg = t < p.ntiles && pp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const half_t* const src = p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld;
#pragma unroll
        for (int ks = 0; ks < KSX; ks++) { const int k = ks * 16 + hi * 8; fx[ks] = (inimg && k < p.Cin) ? MB_X_LOAD(src + k) : hz; }
    };
    load_x(blockIdx.x);
    {   // depthwise weights [Ch][9] -> [9][ChP]: the loads of a batch are independent (one round trip per batch, not per element)
        constexpr int WB = 8;
        for (int i0 = tid; i0 < 9 * ChP; i0 += NT * WB) {
            half_t v[WB];
#pragma unroll
            for (int j = 0; j < WB; j++) { const int i = i0 + j * NT, t = i / ChP, c = i - t * ChP; v[j] = (i < 9 * ChP && c < p.Ch) ? p.wd[(long)c * 9 + t] : (half_t)0.f; }
#pragma unroll
            for (int j = 0; j < WB; j++) { const int i = i0 + j * NT; if (i < 9 * ChP) wdl[i] = v[j]; }
        }
    }
    for (int i = tid; i < ChP; i += NT) { bdl[i] = (p.bd && i < p.Ch) ? p.bd[i] : 0.f; b1l[i] = (p.b1 && i < p.Ch) ? p.b1[i] : 0.f; }
    for (int i = tid; i < CoP; i += NT) b2l[i] = (p.b2 && i < p.Cout) ? p.b2[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MB4_TL(0);

    // depthwise tasks = (column, channel vector, row part) over the TO x TO tile: RP row parts, the first TO % RP of them one row longer.  A wave takes
    // tasks of ONE row part when the parts split evenly over the waves (RP | 8), and waves w, w + 4 — SIMD partners — take parts from opposite ends of
    // the list, so that every SIMD sees the same number of rows; otherwise tasks are dealt lane-linearly.
    int RP = NT / (NCV * TO); if (RP > TO) RP = TO;
    int cv, tx, rp; bool task_on;
    if ((8 % RP) == 0 && ((NCV * TO) % (8 / RP)) == 0 && NCV * TO / (8 / RP) <= 64) {
        const int wpp = 8 / RP, per = NCV * TO / wpp;                     // waves per part, tasks per wave
        const int part = wave < 4 ? wave / wpp : RP - 1 - (wave - 4) / wpp, sub = wave % wpp;
        const int id = sub * per + lane;
        rp = part; cv = id % NCV; tx = id / NCV; task_on = lane < per;
        if (tx >= TO) tx = TO - 1;
    } else {
        const int trest = tid / NCV;
        cv = tid % NCV; tx = trest % TO; rp = trest / TO; task_on = rp < RP;
    }
    const int rlong = TO % RP, rbase = TO / RP;
    const int ty0 = rp * rbase + (rp < rlong ? rp : rlong), ty1 = task_on ? ty0 + rbase + (rp < rlong ? 1 : 0) : 0;
    const bool relu1 = p.sl1 == 0.f, relu2 = p.sl2 == 0.f;            // block-uniform: (clipped) ReLU takes the packed-max path; the CLIP instantiations add a packed min (monotone: commutes with the rounding)   // block-uniform: plain ReLU takes the packed-max path; a clipped one (ONNX Clip(0, hi)) the general path below
                                                                      // (a packed min inside the fast path cost these kernels 5 registers each — they sit at their spill edge: +6.6 us on MobileNet-v2's 7 x 7 blocks)

    int g = 0;                                                            // chunks done since the launch (buffer parity); the chunk's channels = g % NC
#pragma unroll 1
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
        const int oy = oy0 + qy, ox = ox0 + qx;
        const bool valid = act_pf && q < TOO && oy < p.Ho && ox < p.Wo;   // my output pixel exists
        const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
        if (!PREF && t != (int)blockIdx.x) load_x(t);
        half8_t fone = hz; if (inimg && !hi) { fone[0] = (half_t)1.f; fone[1] = (half_t)1.f; }
        half8_t fxc[PREF ? KSX : 1];                                      // PREF: this tile's fragments (fx is refilled with the next tile's under the last chunk)
        if constexpr (PREF) {
#pragma unroll
            for (int ks = 0; ks < KSX; ks++) fxc[ks] = fx[ks];
        }
        half8_t rres[NFW][2];                                             // residual rows of my output pixel (requested under the last chunk)
        floatx16 acc2[NFW];
#pragma unroll
        for (int a = 0; a < NFW; a++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[a][i] = 0.f;
#pragma unroll 1
        for (int c = 0; c < NC; c++, g++) {
            const int gb = g & 1;
            // ---- phase A ----------------------------------------------------------------------------------------------------------------
            {
                half8_t fw[CFA][KSX];                                      // all fragments first: one LDS round trip, then the two row blocks' MFMA chains interleaved
                float bv[CFA];
#pragma unroll
                for (int cf = 0; cf < CFA; cf++) {
#pragma unroll
                    for (int ks = 0; ks < KSX; ks++) fw[cf][ks] = *(const half8_t*)(w1b + gb * W1B + ((cfa0 + cf) * KSX + ks) * 1024 + lane * 16);
                    bv[cf] = b1l[c * CK + (cfa0 + cf) * 32 + l31];
                }
                floatx16 acc[CFA];
#pragma unroll
                for (int cf = 0; cf < CFA; cf++) {
                    half8_t fb = hz;
                    if (!hi) { const half_t bh = (half_t)bv[cf]; fb[0] = bh; fb[1] = (half_t)(bv[cf] - (float)bh); }
                    acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fone, (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < KSX; ks++)
#pragma unroll
                    for (int cf = 0; cf < CFA; cf++) acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[cf][ks], PREF ? fxc[PREF ? ks : 0] : fx[ks], acc[cf], 0, 0, 0);
#pragma unroll
                for (int cf = 0; cf < CFA; cf++)
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        half4_t o;
                        if (relu1) {
#pragma unroll
                            for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)acc[cf][4 * qq + r2];
                            o = __builtin_elementwise_max(o, half4_t{0, 0, 0, 0});
                            if constexpr (CLIP) { if (p.hi1 != INFINITY) { const half_t hh = (half_t)p.hi1; o = __builtin_elementwise_min(o, half4_t{hh, hh, hh, hh}); } }
                        } else {
#pragma unroll
                            for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)relu_family<CLIP>(acc[cf][4 * qq + r2], p.sl1, p.hi1);
                        }
                        *(half4_t*)(hid + pp * HSB + ((cfa0 + cf) * 32 + 8 * qq + 4 * hi) * 2) = o;
                    }
            }
            MB4_TL(1);
            mb_lds_barrier();                                             // the hidden patch of chunk c is complete; C(c-1) is over everywhere
            MB4_TL(2);
            // ---- phase B ----------------------------------------------------------------------------------------------------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my share of W1(g+1), W2(g), requested one chunk ago
            MB4_TL(7);
            {
                const int c2 = (c + 2) % NC, c1 = c + 1 < NC ? c + 1 : 0;
                dma_w1(c2, gb); dma_w2(c1, gb ^ 1);                       // W1(g) was read in A(g), W2(g-1) in C(g-1): both buffers are free
            }
            if (c == NC - 1) {                                            // under the last chunk: the next tile's X fragments, this tile's residual rows
                if constexpr (PREF) load_x(t + (int)gridDim.x);
                if (p.res) {
#pragma unroll
                    for (int i = 0; i < NFW; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const int co = (grp + NG * i) * 32 + 16 * j + 8 * hi;
                            rres[i][j] = (valid && co < p.Cout) ? *(const half8_t*)(p.x + opix * p.x_ld + co) : hz;      // stride 1, Cin == Cout: the block input at the same pixel
                        }
                }
            }
            if (ty0 < ty1) {
                const int chl = c * CK + cv * 8;
                u32x4 wreg[9];
#pragma unroll
                for (int tp = 0; tp < 9; tp++) wreg[tp] = *(const u32x4*)(wdl + tp * ChP + chl);
                const floatx4 bd0 = *(const floatx4*)(bdl + chl), bd1 = *(const floatx4*)(bdl + chl + 4);
                const unsigned char* const hcol = hid + tx * S * HSB + cv * 16;   // patch column of tap kw = 0 for my output column, my channel vector
                auto finish = [&](float* a, int y) __attribute__((always_inline)) {
                    half8_t ov;
                    if (relu2) {
#pragma unroll
                        for (int j = 0; j < 8; j++) ov[j] = (half_t)a[j];
                        ov = __builtin_elementwise_max(ov, hz);
                        if constexpr (CLIP) { if (p.hi2 != INFINITY) { const half_t hh = (half_t)p.hi2; ov = __builtin_elementwise_min(ov, half8_t{hh, hh, hh, hh, hh, hh, hh, hh}); } }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) ov[j] = (half_t)relu_family<CLIP>(a[j], p.sl2, p.hi2);
                    }
                    *(half8_t*)(dwo + (y * TO + tx) * HSB + cv * 16) = ov;
                };
                if constexpr (S == 1) {
                    u32x4 win[3][3];
#pragma unroll
                    for (int k = 0; k < 2; k++)
#pragma unroll
                        for (int kw = 0; kw < 3; kw++) win[k][kw] = *(const u32x4*)(hcol + ((ty0 + k) * PW + kw) * HSB);
#pragma unroll
                    for (int i = 0; i < RPTM; i++) {
                        const int y = ty0 + i;
                        if (y < ty1) {
#pragma unroll
                            for (int kw = 0; kw < 3; kw++) win[(i + 2) % 3][kw] = *(const u32x4*)(hcol + ((y + 2) * PW + kw) * HSB);
                            float a[8];
#pragma unroll
                            for (int j = 0; j < 4; j++) { a[j] = bd0[j]; a[4 + j] = bd1[j]; }
#pragma unroll
                            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                                for (int kw = 0; kw < 3; kw++) {
                                    const u32x4 xu = win[(i + kh) % 3][kw];
                                    const u32x4 wu = wreg[kh * 3 + kw];
#pragma unroll
                                    for (int d2 = 0; d2 < 4; d2++) {
                                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2]) : "v"(xu[d2]), "v"(wu[d2]));
                                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2 + 1]) : "v"(xu[d2]), "v"(wu[d2]));
                                    }
                                }
                            finish(a, y);
                        }
                    }
                } else {
#pragma unroll 1
                    for (int y = ty0; y < ty1; y++) {
                        u32x4 xv[9];
#pragma unroll
                        for (int kh = 0; kh < 3; kh++)
#pragma unroll
                            for (int kw = 0; kw < 3; kw++) xv[kh * 3 + kw] = *(const u32x4*)(hcol + ((y * S + kh) * PW + kw) * HSB);
                        float a[8];
#pragma unroll
                        for (int j = 0; j < 4; j++) { a[j] = bd0[j]; a[4 + j] = bd1[j]; }
#pragma unroll
                        for (int tp = 0; tp < 9; tp++) {
                            const u32x4 xu = xv[tp], wu = wreg[tp];
#pragma unroll
                            for (int d2 = 0; d2 < 4; d2++) {
                                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2]) : "v"(xu[d2]), "v"(wu[d2]));
                                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a[2 * d2 + 1]) : "v"(xu[d2]), "v"(wu[d2]));
                            }
                        }
                        finish(a, y);
                    }
                }
            }
            MB4_TL(3);
            mb_lds_barrier();                                             // depthwise output complete; everybody's share of W1(g+1) and W2(g) has landed
            MB4_TL(4);
            // ---- phase C ----------------------------------------------------------------------------------------------------------------
            if (act_pf) {
                half8_t fy[2 * CF];
#pragma unroll
                for (int ks = 0; ks < 2 * CF; ks++) fy[ks] = *(const half8_t*)(dwo + q * HSB + (2 * ks + hi) * 16);
#pragma unroll
                for (int i = 0; i < NFW; i++) {
                    const int a = grp + NG * i;
                    if (a < nfc) {
                        half8_t fw2[2 * CF];
#pragma unroll
                        for (int ks = 0; ks < 2 * CF; ks++) fw2[ks] = *(const half8_t*)(w2b + gb * W2B + (a * 2 * CF + ks) * 1024 + lane * 16);
#pragma unroll
                        for (int ks = 0; ks < 2 * CF; ks++) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw2[ks], fy[ks], acc2[i], 0, 0, 0);
                    }
                }
            }
        }
        MB4_TL(5);                                                         // (phase C of a chunk is charged to the next chunk's A, the last one's here)
        // ---- epilogue: + b2 (+ X), act3, 16-byte stores; lane = output pixel q -----------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NFW; i++) {
            const int a = grp + NG * i;
            if (a < nfc) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float o[8];
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) {
                        unsigned ua = __builtin_bit_cast(unsigned, acc2[i][4 * (2 * j) + r2] + b2l[a * 32 + 8 * (2 * j) + 4 * hi + r2]);
                        unsigned ub = __builtin_bit_cast(unsigned, acc2[i][4 * (2 * j + 1) + r2] + b2l[a * 32 + 8 * (2 * j + 1) + 4 * hi + r2]);
                        sm_permlane32_swap(ua, ub);
                        o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                    }
                    const int co = a * 32 + 16 * j + 8 * hi;
                    if (valid && co < p.Cout) {
                        if (p.res) {
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) o[r2] += (float)rres[i][j][r2];
                        }
                        half8_t ov;
                        if (p.sl3 == 1.f && (!CLIP || p.hi3 == INFINITY)) {                                  // no activation behind the projection (the usual linear bottleneck): 4 conversions instead of 8 compare / multiply / select
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)o[r2];
                        } else {
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)relu_family<CLIP>(o[r2], p.sl3, p.hi3);
                        }
                        *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                    }
                }
            }
        }
        MB4_TL(6);
    }
#ifdef SM_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 8; i++) g_mb_tl[i] += tl4[i];
#endif
}

}  // namespace

extern "C" {

#ifdef SM_TIMELINE
int sm_mbconv_debug_timeline(unsigned long long* out8, int reset) {
    static unsigned long long z[8];
    if (reset) { SM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mb_tl), z, sizeof z)); return SM_OK; }
    SM_HIP_CHECK(hipDeviceSynchronize());
    SM_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mb_tl), 8 * sizeof(unsigned long long)));
    return SM_OK;
}
#endif

static size_t mb_v1_lds(const sm_mbconv_desc* d) {
    const int ChP = (d->Ch + 31) & ~31, hs = ChP * 2 + 16, KP1 = (d->Cin + (d->expand ? 2 : 0) + 15) & ~15, CoP = (d->Cout + 31) & ~31;
    const int TH = d->stride == 1 ? 8 : 4, TW = 8, PH = (TH - 1) * d->stride + 3, PW = (TW - 1) * d->stride + 3, MFA = (PH * PW + 31) / 32;
    return (size_t)(d->expand ? 2 * MFA * 32 * KP1 * 2 + MFA * 32 * hs : 2 * MFA * 32 * hs) + (size_t)TH * TW * hs + (size_t)ChP * KP1 * 2 + (size_t)9 * ChP * 2 + (size_t)CoP * hs +
           (size_t)(2 * ChP + CoP) * 4;
}
static size_t mb_v2_lds(const sm_mbconv_desc* d) {
    const int ChP = (d->Ch + 31) & ~31, KP1 = (d->Cin + 15) & ~15;
    const int TH = d->stride == 1 ? 8 : 4, TW = 8, PH = (TH - 1) * d->stride + 3, PW = (TW - 1) * d->stride + 3, MFA = (PH * PW + 31) / 32;
    return (size_t)(d->expand ? MFA * 32 * KP1 * 2 : 0) + (size_t)MFA * 32 * 80 + (size_t)TH * TW * 80 + (size_t)9 * ChP * 2;
}
static size_t mb_v3_lds_nw(const sm_mbconv_desc* d, int nw, bool wg = false) {
    const int ChP = (d->Ch + 31) & ~31, KP1 = d->expand ? (d->Cin + 2 + 15) & ~15 : 0, CoP = (d->Cout + 31) & ~31;
    const int TH = d->stride == 1 ? 4 : 2, PH = (TH - 1) * d->stride + 3, PW = 7 * d->stride + 3, MFA = (PH * PW + 31) / 32;
    return (wg ? 0 : (size_t)ChP * KP1 * 2 + (size_t)CoP * (ChP * 2 + 16)) + (size_t)10 * ChP * 4 + (size_t)(ChP + CoP) * 4 + (size_t)nw * (MFA * 32 + TH * 8) * 80;
}
// W1 / W2 as fragments from global memory (8-wave blocks): the expand form whose weights leave no room for 8 wave regions; instantiated for
// K = Cin + 2 <= 80 | 112 (5 | 7 k steps) and Cout <= 64 | 96 | 160 — MobileNet-v2's 14 x 14 blocks
// v5 (hidden patch in registers): the expand form with W1 / W2 in the LDS, Cout <= 64
static size_t mb_v5_lds_nw(const sm_mbconv_desc* d, int nw) {
    // KP1 = the INSTANTIATION's K (two k steps up to Cin = 30, three beyond: the launcher's MB5_PICK) — not Cin + 2 rounded up: with Cin = 8 the kernel's W1 rows are 32 halfs
    // where this function counted 16, and the last wave's depthwise buffer lay behind the allocation (round-6 shape sweep: wave 15's tile wrong from its second row on)
    const int ChP = (d->Ch + 31) & ~31, KP1 = d->Cin <= 30 ? 32 : 48, CoP = (d->Cout + 31) & ~31;
    return (size_t)ChP * KP1 * 2 + (size_t)ChP * 24 * 2 + (size_t)CoP * (ChP * 2 + 16) + (size_t)(ChP + CoP) * 4 + (size_t)nw * 32 * 80;
}
static bool mb_v5_ok(const sm_mbconv_desc* d) {
    static const int off = SM_ABL_ENV("SMELTER_MBCONV_NO_V5") ? 1 : 0;          // experiment switch, read once
    return !off && d->expand && d->Cin <= 46 && d->Cout <= 64 && mb_v5_lds_nw(d, 16) <= 160 * 1024;
}
static bool mb_v3_wg(const sm_mbconv_desc* d) { return d->expand && mb_v3_lds_nw(d, 8) > 160 * 1024; }
// waves per block: 16 (one block per CU; its 128-register budget holds) when the LDS takes it, else 8, else 4
static int mb_v3_waves(const sm_mbconv_desc* d) {
    if (mb_v3_wg(d)) return 8;
    const char* fe = SM_ABL_ENV("SMELTER_MBCONV_WAVES"); const int force = fe ? atoi(fe) : 0;            // experiment switch: 4 | 8 | 16 where the LDS takes it
    if ((force == 4 || force == 8 || force == 16) && mb_v3_lds_nw(d, force) <= 160 * 1024) return force;
    if (mb_v3_lds_nw(d, 16) <= 160 * 1024) return 16;
    if (mb_v3_lds_nw(d, 8) <= 160 * 1024) return 8;
    return 4;
}
static size_t mb_v3_lds(const sm_mbconv_desc* d) { return mb_v3_lds_nw(d, mb_v3_waves(d), mb_v3_wg(d)); }
static bool mb_common_ok(const sm_mbconv_desc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0) return false;
    if (d->stride != 1 && d->stride != 2) return false;
    if ((d->Cin & 7) || (d->Ch & 7) || (d->Cout & 7) || d->Cin <= 0 || d->Ch <= 0 || d->Cout <= 0) return false;
    if (!d->expand && d->Ch != d->Cin) return false;
    if (d->residual && (d->stride != 1 || d->Cin != d->Cout)) return false;
    auto act_ok = [](int a) { return a <= SM_ACT_PRELU || a == SM_ACT_CLIP; };
    if (!act_ok(d->act1) || !act_ok(d->act2) || !act_ok(d->act3)) return false;
    return d->Ho == (d->H + 2 - 3) / d->stride + 1 && d->Wo == (d->W + 2 - 3) / d->stride + 1;
}
// (kernels 1-3 hold no instantiation for bounded activations: SM_ACT_CLIP blocks go to the register-window kernels — template RELU = 2 — or to kernel 4's CLIP instantiations)
static bool mb_clip(const sm_mbconv_desc* d) { return d->act1 == SM_ACT_CLIP || d->act2 == SM_ACT_CLIP || d->act3 == SM_ACT_CLIP; }
// (v1 adds the residual from its X patch, which only the expand form fills: a depthwise-first block WITH a residual — MnasNet / EfficientNet's repeated MBConv1 — is no kernel's;
//  found by the round-6 shape sweep: the launch returned the block without its residual)
static bool mb_v1_ok(const sm_mbconv_desc* d) { return mb_common_ok(d) && !mb_clip(d) && d->Cin <= 32 && d->Ch <= 192 && d->Cout <= 64 && (!d->residual || d->expand) && mb_v1_lds(d) <= 160 * 1024; }
// v2 (chunked): any Ch; Cin <= 64 when there is an expand layer (X patch in LDS, <= 4 k steps); Cout <= 64 (stride 1) / 128 (stride 2): one
// (row block, cout block) pair per wave; the residual is read from the X patch, i.e. needs the expand form
static bool mb_v2_ok(const sm_mbconv_desc* d) {
    return mb_common_ok(d) && !mb_clip(d) && (!d->expand || d->Cin <= 64) && d->Cout <= (d->stride == 1 ? 64 : 128) && (!d->residual || d->expand) && d->Ch <= 4096 && mb_v2_lds(d) <= 64 * 1024;
}
// v3 (wave pipelines): the expand form with K = Cin + 2 <= 48 (three k steps of X fragments in registers), Cout <= 64, weights + four wave regions in LDS
static bool mb_v3_ok(const sm_mbconv_desc* d) {
    if (!mb_common_ok(d)) return false;
    if (mb_clip(d) && !mb_v5_ok(d) && d->expand) return false;        // (bounded activations: the register-window form and the no-expand form of this family)
    if (mb_v3_wg(d)) return d->Cin <= 110 && d->Cout <= 96 && mb_v3_lds(d) <= 160 * 1024;      // (five accumulator sets, Cout = 160, spill)
    return (d->expand ? d->Cin <= 46 : !d->residual) && d->Cout <= 64 && mb_v3_lds(d) <= 160 * 1024;
}
// v4 (tile per block, the waves share every chunk): the expand form at stride 1 | 2 on any image; tiles of 14 x 14 (stride 1) / 7 x 7 (stride 2) output
// pixels, chunks of 64 channels; phase C deals 7 output fragments to 7 waves (cout <= 96) or, for tiles of <= 64 pixels, 2 fragments x 4 cout groups
static int mb_v4_tile(const sm_mbconv_desc* d) { const int m = d->stride == 1 ? 14 : 7, o = d->Ho > d->Wo ? d->Ho : d->Wo; return o < m ? o : m; }
static int mb_v4_tcf(const sm_mbconv_desc* d) { const int t = mb_v4_tile(d); return t * t <= 64 ? 2 : 7; }
static int mb_v4_ksx(const sm_mbconv_desc* d) { const int k = (d->Cin + 15) / 16; return k <= 2 ? 2 : k <= 4 ? 4 : k <= 6 ? 6 : 10; }
static int mb_v4_nfw(const sm_mbconv_desc* d) { const int nfc = (d->Cout + 31) / 32, ng = mb_v4_tcf(d) == 7 ? 1 : 4; return (nfc + ng - 1) / ng; }
static size_t mb_v4_lds(const sm_mbconv_desc* d) {
    const int CF = 2, CK = CF * 32, HSB = CK * 2 + 16, ChP = (d->Ch + CK - 1) / CK * CK, nfc = (d->Cout + 31) / 32;
    return (size_t)2 * CF * mb_v4_ksx(d) * 1024 + (size_t)2 * nfc * 2 * CF * 1024 + (size_t)(256 + mb_v4_tcf(d) * 32) * HSB + (size_t)9 * ChP * 2 + (size_t)(2 * ChP + nfc * 32) * 4;
}
static bool mb_v4_ok(const sm_mbconv_desc* d) {
    if (!mb_common_ok(d) || !d->expand || d->Cin > 160 || d->Ch < 64) return false;
    if (mb_v4_nfw(d) > 3) return false;                               // (three accumulator sets per wave: cout <= 96 on 14 x 14 tiles, <= 384 on small ones)
    return mb_v4_lds(d) <= 160 * 1024;
}
static int mb_pick(const sm_mbconv_desc* d, int force = 0) {          // 4, 3, 1 or 2; force = 1 | 2 | 3 | 4: that kernel where it applies (sm_mbconv_fwd_kind: tests, tuning)
    const bool ok1 = d && mb_v1_ok(d), ok2 = d && mb_v2_ok(d), ok3 = d && mb_v3_ok(d), ok4 = d && mb_v4_ok(d);
    if (force && !(force == 1 ? ok1 : force == 2 ? ok2 : force == 3 ? ok3 : ok4)) force = 0;      // a kernel that does not take the block: the launcher's own choice (NOT the rest of the list below)
    if (force == 1 && ok1) return 1;
    if (force == 2 && ok2) return 2;
    if (force == 3 && ok3) return 3;
    if (force == 4 && ok4) return 4;
    // v4 measured against v3 / the three launches per MobileNet-v2 block (bs 256, cold, tools/mbconv_bench.py): 28^2 45.6 vs 61.5 us, 28^2 -> 14^2 33.7 vs 53, 14^2 30 vs 52 /
    // 36 vs 54 / 51 vs 80, 14^2 -> 7^2 (Ch 576 -> 160) 35 vs 70 as three launches; it LOSES on the large images (56^2: 166 vs 141, 112^2 -> 56^2: 293 vs 122 — sixteen
    // and more tiles per CU, each a chain of barrier-separated phases, against v3's sixteen independent waves) and on the 7^2 blocks (82 vs 61 as three launches:
    // eleven k steps, 15 chunks, spills), which stay where they were
    // => v4 for images of at most four tiles (56^2 -> 28^2 has sixteen 7 x 7 tiles: 107 vs 90 us)
    // (7^2 blocks: with the row blocks of a chunk dealt to different waves — AS — 50 vs 63 us as three launches, cold)
    // (28^2, stride 1: the register-window kernel ties it with the vectorised prologue — 42.6 vs 45.6 us cold, no difference in the graph)
    if (ok4 && !force && sm_ceil_div(d->Ho, mb_v4_tile(d)) * sm_ceil_div(d->Wo, mb_v4_tile(d)) <= 4) return 4;
    // measured on MobileNet-v2 bs = 256 (tools/mbconv_bench.py): the wave pipelines win everywhere (112^2 -> 56^2: 125 vs 307 us, 56^2: 143 vs 245,
    // 56^2 -> 28^2: 92 vs 122, 28^2: 60 vs 93) except on the 28^2 -> 14^2 block (58 vs 53: one or two ragged tiles per wave)
    if (ok3 && !(ok1 && d->stride == 2 && (long)d->Ho * d->Wo <= 196)) return 3;
    return ok1 ? 1 : ok2 ? 2 : 0;
}



static size_t mb_lds_for(const sm_mbconv_desc* d, int k);
size_t sm_mbconv_lds_bytes(const sm_mbconv_desc* d) { return d ? mb_lds_for(d, mb_pick(d)) : 0; }
static size_t mb_lds_for(const sm_mbconv_desc* d, int k) {
    return k == 4 ? mb_v4_lds(d) : k == 3 ? (mb_v5_ok(d) ? mb_v5_lds_nw(d, 16) : mb_v3_lds(d)) : k == 1 ? mb_v1_lds(d) : mb_v2_lds(d);
}

int sm_mbconv_supported(const sm_mbconv_desc* d) { return mb_pick(d) ? 1 : 0; }
int sm_mbconv_kernel_kind(const sm_mbconv_desc* d) { return mb_pick(d); }

int sm_mbconv_fwd(const sm_mbconv_desc* d, const void* x, const void* w1, const float* b1, const void* wd, const float* bd, const void* w2, const float* b2, void* y,
                  sm_stream_t stream) {
    return sm_mbconv_fwd_kind(d, 0, x, w1, b1, wd, bd, w2, b2, y, stream);
}
int sm_mbconv_fwd_kind(const sm_mbconv_desc* d, int32_t kind, const void* x, const void* w1, const float* b1, const void* wd, const float* bd, const void* w2, const float* b2, void* y,
                       sm_stream_t stream) {
    SM_REQUIRE(kind >= 0 && kind <= 4, "sm_mbconv_fwd_kind: kernel %d", kind);
    SM_REQUIRE(sm_mbconv_supported(d) && x && wd && w2 && y && (!d->expand || w1), "sm_mbconv_fwd: unsupported block geometry");
    MbParams p{};
    p.x = (const half_t*)x; p.y = (half_t*)y; p.w1 = (const half_t*)w1; p.wd = (const half_t*)wd; p.w2 = (const half_t*)w2; p.b1 = b1; p.bd = bd; p.b2 = b2;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.s = d->stride; p.Cin = d->Cin; p.Ch = d->Ch; p.Cout = d->Cout;
    p.x_ld = d->x_ld ? d->x_ld : d->Cin; p.y_ld = d->y_ld ? d->y_ld : d->Cout; p.expand = d->expand ? 1 : 0; p.res = d->residual ? 1 : 0;
    SM_REQUIRE((p.x_ld & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0, "sm_mbconv_fwd: operands must be 16-byte aligned, pixel strides % 8 == 0");
    SM_REQUIRE((long)d->N * d->H * d->W < (1L << 31), "sm_mbconv_fwd: too many pixels");
    p.act1 = d->act1; p.act2 = d->act2; p.act3 = d->act3;
    p.sl1 = d->act1 == SM_ACT_NONE ? 1.f : (d->act1 == SM_ACT_RELU || d->act1 == SM_ACT_CLIP) ? 0.f : d->slope1; p.hi1 = d->act1 == SM_ACT_CLIP ? d->slope1 : INFINITY;
    p.sl2 = d->act2 == SM_ACT_NONE ? 1.f : (d->act2 == SM_ACT_RELU || d->act2 == SM_ACT_CLIP) ? 0.f : d->slope2; p.hi2 = d->act2 == SM_ACT_CLIP ? d->slope2 : INFINITY;
    p.sl3 = d->act3 == SM_ACT_NONE ? 1.f : (d->act3 == SM_ACT_RELU || d->act3 == SM_ACT_CLIP) ? 0.f : d->slope3; p.hi3 = d->act3 == SM_ACT_CLIP ? d->slope3 : INFINITY;
    p.ChP = (d->Ch + 31) & ~31; p.hstride = p.ChP * 2 + 16;
    const int TH = d->stride == 1 ? 8 : 4, TW = 8, PW = (TW - 1) * d->stride + 3;
    p.tiles_x = sm_ceil_div(d->Wo, TW); p.tiles_y = sm_ceil_div(d->Ho, TH);
    const long nt = (long)d->N * p.tiles_x * p.tiles_y;
    SM_REQUIRE(nt < (1L << 31), "sm_mbconv_fwd: too many tiles");
    p.ntiles = (int)nt;
    fastdiv_make((unsigned)p.tiles_x, &p.tx_mul, &p.tx_sh); fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh); fastdiv_make((unsigned)PW, &p.pw_mul, &p.pw_sh);
    const int kernel = mb_pick(d, kind);
    const size_t lds = mb_lds_for(d, kernel);
    if (kernel == 4) {                                               // tile per block: persistent over the tiles, one 8-wave block per CU
        SM_REQUIRE(((((uintptr_t)w1) | ((uintptr_t)w2)) & 15) == 0, "sm_mbconv_fwd: weights must be 16-byte aligned");
        p.ChP = (d->Ch + 63) / 64 * 64;
        const int to = mb_v4_tile(d), tcf = mb_v4_tcf(d), ksx = mb_v4_ksx(d), nfw = mb_v4_nfw(d);
        p.tile_o = to; p.tiles_x = sm_ceil_div(d->Wo, to); p.tiles_y = sm_ceil_div(d->Ho, to);
        const long nt4 = (long)d->N * p.tiles_x * p.tiles_y;
        SM_REQUIRE(nt4 < (1L << 31), "sm_mbconv_fwd: too many tiles");
        p.ntiles = (int)nt4;
        fastdiv_make((unsigned)p.tiles_x, &p.tx_mul, &p.tx_sh); fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh);
        void (*fn)(MbParams) = nullptr;
#define MB4_PICKC(S_, T_, A_, C_) (ksx == 2 ? (nfw == 1 ? mbconv4_kernel<S_, T_, 2, 1, A_, C_> : nfw == 2 ? mbconv4_kernel<S_, T_, 2, 2, A_, C_> : mbconv4_kernel<S_, T_, 2, 3, A_, C_>) \
                        : ksx == 4 ? (nfw <= 2 ? mbconv4_kernel<S_, T_, 4, 2, A_, C_> : mbconv4_kernel<S_, T_, 4, 3, A_, C_>) \
                        : ksx == 6 ? (nfw <= 2 ? mbconv4_kernel<S_, T_, 6, 2, A_, C_> : mbconv4_kernel<S_, T_, 6, 3, A_, C_>) \
                                   : (nfw <= 2 ? mbconv4_kernel<S_, T_, 10, 2, A_, C_> : mbconv4_kernel<S_, T_, 10, 3, A_, C_>))
        const bool clip = p.hi1 != INFINITY || p.hi2 != INFINITY || p.hi3 != INFINITY;   // (a bounded activation: the CLIP instantiations; everything else runs the code it ran before the bound existed)
#define MB4_PICK(S_, T_) (clip ? MB4_PICKC(S_, T_, false, true) : MB4_PICKC(S_, T_, false, false))
        const int pw = (to - 1) * d->stride + 3;
        if (d->stride == 1 && tcf == 2 && pw * pw <= 128 && ksx >= 6 && !clip) fn = ksx == 6 ? (nfw <= 2 ? mbconv4_kernel<1, 2, 6, 2, true> : mbconv4_kernel<1, 2, 6, 3, true>)
                                                                                   : (nfw <= 2 ? mbconv4_kernel<1, 2, 10, 2, true> : mbconv4_kernel<1, 2, 10, 3, true>);
        else if (d->stride == 1 && tcf == 2 && pw * pw <= 128 && ksx >= 6) fn = ksx == 6 ? (nfw <= 2 ? mbconv4_kernel<1, 2, 6, 2, true, true> : mbconv4_kernel<1, 2, 6, 3, true, true>)
                                                                                   : (nfw <= 2 ? mbconv4_kernel<1, 2, 10, 2, true, true> : mbconv4_kernel<1, 2, 10, 3, true, true>);
        else if (d->stride == 1) fn = tcf == 7 ? MB4_PICK(1, 7) : MB4_PICK(1, 2);
        else fn = MB4_PICK(2, 2);
#undef MB4_PICK
#undef MB4_PICKC
        hipLaunchKernelGGL(fn, dim3((unsigned)std::min<long>(nt4, SM_NUM_CU)), dim3(512), lds, sm_s(stream), p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (kernel == 3) {                                               // wave pipelines: 4-wave blocks, tiles dealt to waves
        const int th3 = d->stride == 1 ? 4 : 2;
        const long nt3 = (long)d->N * sm_ceil_div(d->Ho, th3) * sm_ceil_div(d->Wo, 8);
        SM_REQUIRE(nt3 < (1L << 31), "sm_mbconv_fwd: too many tiles");
        p.tiles_x = sm_ceil_div(d->Wo, 8); p.tiles_y = sm_ceil_div(d->Ho, th3); p.ntiles = (int)nt3;
        fastdiv_make((unsigned)p.tiles_x, &p.tx_mul, &p.tx_sh); fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh);
        if (mb_v5_ok(d)) {                                           // hidden patch in registers: 16-wave blocks, one per CU
            SM_REQUIRE(((((uintptr_t)w1) | ((uintptr_t)w2)) & 15) == 0, "sm_mbconv_fwd: weights must be 16-byte aligned");
            const int relu = (p.sl1 == 0.f && p.sl2 == 0.f) ? ((p.hi1 == INFINITY && p.hi2 == INFINITY) ? 1 : 2) : 0;   // (a bound on one of the two: the other's is +inf, a no-op min)
            void (*f5)(MbParams) = nullptr;
#define MB5_PICK2(S_, K_) (d->Cout <= 32 ? (relu == 1 ? mbconv5_kernel<S_, K_, 16, 1, 1> : relu == 2 ? mbconv5_kernel<S_, K_, 16, 1, 2> : mbconv5_kernel<S_, K_, 16, 1, 0>) \
                                         : (relu == 1 ? mbconv5_kernel<S_, K_, 16, 2, 1> : relu == 2 ? mbconv5_kernel<S_, K_, 16, 2, 2> : mbconv5_kernel<S_, K_, 16, 2, 0>))
#define MB5_PICK(S_) (d->Cin <= 30 ? MB5_PICK2(S_, 2) : MB5_PICK2(S_, 3))
            if (d->stride == 1) f5 = MB5_PICK(1); else f5 = MB5_PICK(2);
#undef MB5_PICK2
#undef MB5_PICK
            hipLaunchKernelGGL(f5, dim3((unsigned)std::min<long>((nt3 + 15) / 16, (long)SM_NUM_CU)), dim3(1024), mb_v5_lds_nw(d, 16), sm_s(stream), p);
            SM_LAUNCH_CHECK(); return SM_OK;
        }
        const int nw = mb_v3_waves(d), ks1 = d->expand ? ((d->Cin + 2 + 15) & ~15) >> 4 : 0;
        const int per_cu3 = (int)std::max<size_t>(1, std::min<size_t>(16 / nw, (size_t)(160 * 1024) / lds));
        const int grid3 = (int)std::min<long>((nt3 + nw - 1) / nw, (long)SM_NUM_CU * per_cu3);
        void (*fn)(MbParams) = nullptr;
        if (mb_v3_wg(d)) {
            const int nf = (d->Cout + 31) / 32;
            if (d->stride == 1) fn = ks1 <= 5 ? (nf <= 2 ? mbconv3_kernel<1, 5, 8, false, 4, true, 2> : mbconv3_kernel<1, 5, 8, false, 4, true, 3>) : mbconv3_kernel<1, 7, 8, false, 4, true, 3>;
            else fn = ks1 <= 5 ? mbconv3_kernel<2, 5, 8, false, 2, true, 3> : mbconv3_kernel<2, 7, 8, false, 2, true, 3>;
        } else if (ks1 == 0 && mb_clip(d)) fn = d->stride == 2 ? (nw == 16 ? mbconv3_kernel<2, 0, 16, false, 2, false, 2, true> : nw == 8 ? mbconv3_kernel<2, 0, 8, false, 2, false, 2, true> : mbconv3_kernel<2, 0, 4, false, 2, false, 2, true>)
                                          : nw == 16 ? mbconv3_kernel<1, 0, 16, false, 4, false, 2, true> : nw == 8 ? mbconv3_kernel<1, 0, 8, true, 4, false, 2, true> : mbconv3_kernel<1, 0, 4, true, 4, false, 2, true>;
        else if (ks1 == 0) fn = d->stride == 2 ? (nw == 16 ? mbconv3_kernel<2, 0, 16, false> : nw == 8 ? mbconv3_kernel<2, 0, 8, false> : mbconv3_kernel<2, 0, 4, false>)
                                          : nw == 16 ? mbconv3_kernel<1, 0, 16, false> : nw == 8 ? mbconv3_kernel<1, 0, 8, true> : mbconv3_kernel<1, 0, 4, true>;
        else if (d->stride == 2) fn = nw == 16 ? (ks1 <= 2 ? mbconv3_kernel<2, 2, 16, false> : mbconv3_kernel<2, 3, 16, false>)
                                    : nw == 8 ? (ks1 <= 2 ? mbconv3_kernel<2, 2, 8, false> : mbconv3_kernel<2, 3, 8, false>)
                                              : (ks1 <= 2 ? mbconv3_kernel<2, 2, 4, false> : mbconv3_kernel<2, 3, 4, false>);
        else if (nw == 16) fn = ks1 <= 2 ? mbconv3_kernel<1, 2, 16, false> : mbconv3_kernel<1, 3, 16, false>;
        else if (nw == 8) fn = ks1 <= 2 ? mbconv3_kernel<1, 2, 8, true> : mbconv3_kernel<1, 3, 8, true>;
        else fn = ks1 <= 2 ? mbconv3_kernel<1, 2, 4, true> : mbconv3_kernel<1, 3, 4, true>;
        hipLaunchKernelGGL(fn, dim3(grid3), dim3(nw * 64), lds, sm_s(stream), p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    if (kernel == 2) {                                               // v2: 4-wave blocks, as many per CU as the LDS holds (<= 8)
        SM_REQUIRE(((((uintptr_t)b1) | ((uintptr_t)bd) | ((uintptr_t)b2) | ((uintptr_t)w1) | ((uintptr_t)w2)) & 15) == 0, "sm_mbconv_fwd: weights and biases must be 16-byte aligned");
        const int per_cu2 = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / lds));
        const int grid2 = (int)std::min<long>(nt, (long)SM_NUM_CU * per_cu2);
        if (d->stride == 1) hipLaunchKernelGGL((mbconv2_kernel<8, 8, 1>), dim3(grid2), dim3(256), lds, sm_s(stream), p);
        else hipLaunchKernelGGL((mbconv2_kernel<4, 8, 2>), dim3(grid2), dim3(256), lds, sm_s(stream), p);
        SM_LAUNCH_CHECK(); return SM_OK;
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (size_t)(160 * 1024) / lds));
    const int grid = (int)std::min<long>(nt, (long)SM_NUM_CU * std::min(per_cu, 2));      // 16-wave blocks: at most two per CU
    if (d->stride == 1) hipLaunchKernelGGL((mbconv_kernel<8, 8, 1>), dim3(grid), dim3(1024), lds, sm_s(stream), p);
    else hipLaunchKernelGGL((mbconv_kernel<4, 8, 2>), dim3(grid), dim3(1024), lds, sm_s(stream), p);
    SM_LAUNCH_CHECK(); return SM_OK;
}

// The block behind an image stem, stem included: x_nchw [N, 3, in_H, in_W] fp32 -> 3x3 convolution (w1 = the stem's OHWI weights [Ch][3][3][3] fp16; stride st_stride, padding
// st_pad, ReLU-family activation d->act1) -> depthwise 3x3 (d->stride) -> 1x1 project, ONE launch (mbconv5s_kernel: the input region of a block tile staged in the LDS).
// d->H x d->W = the stem's OUTPUT image, d->Ch = its channels (<= 32: one hidden chunk), d->Cin is ignored.
static void mb_stem_geometry(int S, int st, int* IR, int* IC, int* pitch) {                 // input region of a block tile of 2 x 4 wave tiles (mbconv5s_kernel)
    const int TH = S == 1 ? 4 : 2, HR = (2 * TH - 1) * S + 3, HC = 31 * S + 3;
    *IR = (HR - 1) * st + 3; *IC = (HC - 1) * st + 3; *pitch = 64 * (S == 1 ? 2 : 3) + 2;
}
static size_t mb_stem_lds(const sm_mbconv_desc* d, int st) {
    int IR, IC, pitch; mb_stem_geometry(d->stride, st, &IR, &IC, &pitch);
    const int ChP = (d->Ch + 31) & ~31, CoP = (d->Cout + 31) & ~31;
    return (size_t)ChP * 48 * 2 + (size_t)ChP * 24 * 2 + (size_t)CoP * (ChP * 2 + 16) + (size_t)(ChP + CoP) * 4 + (size_t)IR * pitch * 8 + (size_t)8 * 32 * 80;
}
int sm_mbconv_stem_supported(const sm_mbconv_desc* d, int32_t C, int32_t in_H, int32_t in_W, int32_t st_stride, int32_t st_pad) {
    if (!d || C != 3 || d->Ch > 32 || (d->Ch & 7) || d->Cout > 64 || (d->Cout & 7) || d->residual || in_H <= 0 || in_W <= 0 || st_stride < 1 || st_stride > 2 || st_pad < 0 || st_pad > 1) return 0;
    if (d->H != (in_H + 2 * st_pad - 3) / st_stride + 1 || d->W != (in_W + 2 * st_pad - 3) / st_stride + 1) return 0;
    if ((long)d->N * 3 * in_H * in_W * 4 >= 0x7fff0000L) return 0;
    sm_mbconv_desc dl = *d; dl.Cin = 24; dl.expand = 1;               // (the remaining geometry checks are those of an expand block)
    int IR, IC, pitch; mb_stem_geometry(d->stride, st_stride, &IR, &IC, &pitch);
    return mb_common_ok(&dl) && 3 * IR <= 64 && IC <= pitch - 2 && mb_stem_lds(d, st_stride) <= 80 * 1024;     // (eight segments per wave, the passes cover the region's columns, two blocks per CU)
}
int sm_mbconv_stem_fwd(const sm_mbconv_desc* d, const float* x_nchw, int32_t C, int32_t in_H, int32_t in_W, int32_t st_stride, int32_t st_pad, const void* w1, const float* b1,
                       const void* wd, const float* bd, const void* w2, const float* b2, void* y, sm_stream_t stream) {
    SM_REQUIRE(sm_mbconv_stem_supported(d, C, in_H, in_W, st_stride, st_pad) && x_nchw && w1 && wd && w2 && y, "sm_mbconv_stem_fwd: unsupported geometry");
    MbParams p{};
    p.xf = x_nchw; p.in_H = in_H; p.in_W = in_W; p.st_s = st_stride; p.st_p = st_pad;
    p.x = nullptr; p.y = (half_t*)y; p.w1 = (const half_t*)w1; p.wd = (const half_t*)wd; p.w2 = (const half_t*)w2; p.b1 = b1; p.bd = bd; p.b2 = b2;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.s = d->stride; p.Cin = 27; p.Ch = d->Ch; p.Cout = d->Cout;
    p.x_ld = 8; p.y_ld = d->y_ld ? d->y_ld : d->Cout; p.expand = 1; p.res = 0;
    SM_REQUIRE((p.y_ld & 7) == 0 && (((uintptr_t)y) & 15) == 0, "sm_mbconv_stem_fwd: output must be 16-byte aligned, pixel stride % 8 == 0");
    p.act1 = d->act1; p.act2 = d->act2; p.act3 = d->act3;
    p.sl1 = d->act1 == SM_ACT_NONE ? 1.f : (d->act1 == SM_ACT_RELU || d->act1 == SM_ACT_CLIP) ? 0.f : d->slope1; p.hi1 = d->act1 == SM_ACT_CLIP ? d->slope1 : INFINITY;
    p.sl2 = d->act2 == SM_ACT_NONE ? 1.f : (d->act2 == SM_ACT_RELU || d->act2 == SM_ACT_CLIP) ? 0.f : d->slope2; p.hi2 = d->act2 == SM_ACT_CLIP ? d->slope2 : INFINITY;
    p.sl3 = d->act3 == SM_ACT_NONE ? 1.f : (d->act3 == SM_ACT_RELU || d->act3 == SM_ACT_CLIP) ? 0.f : d->slope3; p.hi3 = d->act3 == SM_ACT_CLIP ? d->slope3 : INFINITY;
    p.ChP = (d->Ch + 31) & ~31; p.hstride = p.ChP * 2 + 16;
    const int S = d->stride, TH = S == 1 ? 4 : 2;
    p.tiles_x = sm_ceil_div(d->Wo, 32); p.tiles_y = sm_ceil_div(d->Ho, 2 * TH);                // block tiles of (2 TH) x 32 output pixels
    const long nbt = (long)d->N * p.tiles_x * p.tiles_y;
    SM_REQUIRE(nbt < (1L << 31), "sm_mbconv_stem_fwd: too many tiles");
    p.ntiles = (int)nbt;
    fastdiv_make((unsigned)p.tiles_x, &p.tx_mul, &p.tx_sh); fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh);
    const int relu = (p.sl1 == 0.f && p.sl2 == 0.f) ? ((p.hi1 == INFINITY && p.hi2 == INFINITY) ? 1 : 2) : 0;
    void (*fn)(MbParams) = nullptr;
#define MB5S_PICK(S_) (d->Cout <= 32 ? (relu == 1 ? mbconv5s_kernel<S_, 1, 1> : relu == 2 ? mbconv5s_kernel<S_, 1, 2> : mbconv5s_kernel<S_, 1, 0>) : (relu == 1 ? mbconv5s_kernel<S_, 2, 1> : relu == 2 ? mbconv5s_kernel<S_, 2, 2> : mbconv5s_kernel<S_, 2, 0>))
    if (S == 1) fn = MB5S_PICK(1); else fn = MB5S_PICK(2);
#undef MB5S_PICK
    hipLaunchKernelGGL(fn, dim3((unsigned)std::min<long>(nbt, 2L * SM_NUM_CU)), dim3(512), mb_stem_lds(d, st_stride), sm_s(stream), p);
    SM_LAUNCH_CHECK(); return SM_OK;
}

}  // extern "C"
