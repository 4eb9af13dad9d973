// bneck.hip — a whole ResNet BOTTLENECK BLOCK as ONE tile pass on gfx950:
//
//     T1 = act1(w1 . x  + b1)            1x1 "reduce",  C0 -> 64            (MPSCNNConvolutionNode, /root/reference/Sources/Smelter/Converters.swift:187-338)
//     T2 = act2(w2 * T1 + b2)            3x3 / stride 1 / pad 1, 64 -> 64   (same node class; padding rule ONNXConvolutionPadding.swift:73-79: zeros outside the image)
//     y  = act3(w3 . T2 + b3 + res)      1x1 "expand",  64 -> n3 <= 256, + residual (MPSNNAdditionNode, Converters.swift:430-451), ReLU family (:342-359)
//
// The reference hands MPS three convolution nodes, an addition node and three neuron nodes; layer by layer that is x read twice (reduce, residual), T1 and T2 written and
// read back: 360 MB per block at 64 x 56 x 56 x 256.  Here the two hidden tensors exist in the LDS only (rounded to fp16 there, exactly as the layer-by-layer kernels round
// them on their way to memory — they stay in the parity oracle's rounding list): a block of the grid walks output tiles of TH <= 8 rows x 14 columns and per tile
//   * REDUCER waves (8..11) compute T1 on the tile + its one-pixel halo — a 16-column PATCH of TH + 2 rows, 32-pixel fragments of two patch rows each — straight from
//     global memory: a lane's B operand of k step ks IS 16 bytes of its pixel's row (channels 16 ks + 8 hi ..), so x passes no LDS at all; w1 (their 32-cout half) lives in
//     registers, an 8-deep ring of loads keeps 8 KB per wave in flight across fragments and tiles.  Bias, activation, ONE rounding, lane swaps, then the patch goes to LDS
//     in channel PLANES [8 channels][patch pixel] x 16 bytes: every fragment address of the 3x3 is then "lane base + compile-time offset" (no per-tap address registers)
//     and every access of either side is 16 consecutive 16-byte slots per 16 lanes.  Patch pixels outside the image are written as ZEROS (the padding of the 3x3 — not
//     act1(b1), which is what the GEMM alone would leave there).
//   * PRODUCER waves (0..3): the 3x3 over the patch, one 32-pixel fragment each; w2 [tap][8-channel chunk][cout] resident in LDS (72 KB, loaded once per block by LDS-DMA);
//     hand-counted fragment prefetch as in conv3x3_ws.hpp; the finished fragment (bias, activation, rounded, 8 consecutive channels per lane = the B layout of the expand)
//     goes to a 16 KB hand-over buffer.
//   * CONSUMER waves (4..7): the expand; consumer j owns output channels [64 j, 64 j + 64): its slice of w3 in registers, the residual rows of the NEXT tile requested
//     while this one is computed (they were read by the reducers one period earlier: the XCD's L2 still holds them — blocks of one XCD work on the same images, see
//     the block -> tile map), added through two one-hot MFMAs (exact), + b3, activation, one rounding, 16-byte stores.
// Three tiles are in flight per block (patch of tile t+1 | 3x3 of tile t | expand of tile t-1), ONE s_barrier per tile period; 12 waves = three per SIMD, one of each role,
// <= 168 registers each.  LDS: 72 KB w2 + 1.5 KB biases + 2 x 16 KB hand-over + 2 x 20 KB patch = 145.5 KB.
// Numerics: every layer accumulates in fp32 over k ascending in MFMA steps of 16 from zero, + bias in fp32, activation, one rounding to fp16 — what the per-layer kernels do.
#include <stdlib.h>
#include <math.h>
#include "conv_shared.hpp"

namespace {

struct BnParams {
    const half_t* x; const half_t* w1; const float* b1; const half_t* w2; const float* b2; const half_t* w3; const float* b3; const half_t* res; half_t* y;
    int N, H, W, C0, n3, x_ld, y_ld, res_ld;
    float sl1, sl2, sl3;             // f(v) = v < 0 ? v * sl : v   (1 = no activation, 0 = ReLU)
    int TH, tiles_x, tiles_y, ntiles;
    unsigned txy_mul, txy_sh, ty_mul, ty_sh;
    int dbg;                         // ablation bits (diagnostics): 1 = no output stores, 2 = no residual loads, 4 = no x loads, 8 = no 3x3 MFMAs, 16 = no expand MFMAs
};

constexpr int BN_TW = 14, BN_PW = 16, BN_THMAX = 8, BN_NPX = (BN_THMAX + 2) * BN_PW;       // 160 patch pixels = 5 fragments
constexpr int BN_PLANE = BN_NPX * 16, BN_PATCH = 8 * BN_PLANE;                              // 2560-byte channel planes, 20 KB per patch
constexpr int BN_W2 = 9 * 64 * 128, BN_B1 = BN_W2, BN_B2 = BN_B1 + 256, BN_B3 = BN_B2 + 256, BN_HO = BN_B3 + 1024, BN_HOB = 4 * 4 * 64 * 16;
constexpr int BN_PAT = BN_HO + 2 * BN_HOB, BN_LDS = BN_PAT + 2 * BN_PATCH;
static_assert(BN_LDS <= 160 * 1024, "LDS budget");
constexpr unsigned BN_DEAD = 0x80000000u;   // buffer offset beyond every descriptor: loads return 0, stores are dropped

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KR = C0 / 16 k steps of the reduce (C0 in {64, 128, 256})
template <int KR>
__global__ __launch_bounds__(768) void bneck_kernel(BnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bn_smem[];
    const unsigned lds0 = ws_lds_addr(bn_smem);
    float* const b1l = (float*)(bn_smem + BN_B1); float* const b2l = (float*)(bn_smem + BN_B2); float* const b3l = (float*)(bn_smem + BN_B3);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = p.TH;

    // ---- w2 [cout][tap][cin] -> LDS [tap][chunk c][cout] x 16 bytes (all twelve waves, once), the three biases ----------------------------------------------
    {
        constexpr int WSLOTS = 9 * 8 * 64;
        static_assert(WSLOTS % 768 == 0, "whole block passes");
#pragma unroll 1
        for (int s0 = wave * 64; s0 < WSLOTS; s0 += 768) {
            const int s = s0 + lane, n = s & 63, c = (s >> 6) & 7, tap = s >> 9;
            ws_glds16(p.w2 + ((long)n * 9 + tap) * 64 + c * 8, lds0 + (unsigned)s0 * 16u);
        }
        for (int i = tid; i < 64; i += 768) { b1l[i] = p.b1 ? p.b1[i] : 0.f; b2l[i] = p.b2 ? p.b2[i] : 0.f; }
        for (int i = tid; i < 256; i += 768) b3l[i] = (p.b3 && i < p.n3) ? p.b3[i] : 0.f;
    }
    // ---- this block's contiguous tile range.  Block b runs on XCD b % 8: logical block (b % 8) * G/8 + b / 8 gives every XCD a contiguous range of tiles, i.e. whole
    // images — the column bands of one image run side by side on CUs that share an L2 (their halo columns, and the residual re-read, hit it) --------------------------
    const int T = p.ntiles, G = gridDim.x;
    const int lb = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t_begin = (int)((long)lb * T / G), t_end = (int)((long)(lb + 1) * T / G), nt = t_end - t_begin;
    auto tile_origin = [&](int t, int& n, int& y0, int& x0) {      // tile id -> (image, column band, row band), row band fastest
        n = (int)sm_fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y);
        const int bx = (int)sm_fastdiv((unsigned)r, p.ty_mul, p.ty_sh), by = r - bx * p.tiles_y;
        y0 = by * TH; x0 = bx * BN_TW;
    };
    const unsigned pix_all = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W;

    if (wave < 4) {
        // =========================== PRODUCERS: the 3x3 convolution over the T1 patch, one 32-pixel fragment per wave ===========================
        const int i = wave * 32 + l31, ty = i / BN_TW, tx = i - ty * BN_TW;
        const int slot0 = i < TH * BN_TW ? ty * BN_PW + tx : 0;
        const unsigned a_base = lds0 + BN_PAT + (unsigned)(hi * BN_PLANE + slot0 * 16);
        const unsigned w_base0 = lds0 + (unsigned)(hi * 1024 + l31 * 16), w_base1 = w_base0 + 5 * 8192;
        constexpr int NS = 36, PF = 4, RD = 3;
        static_assert(PF * RD <= 15, "lgkmcnt field");
        auto do_tile = [&](auto bufc, auto relu0c) {
            constexpr int BUF = decltype(bufc)::value;
            constexpr bool RELU0 = decltype(relu0c)::value;
            floatx16 acc[2];
            half8_t fx[PF + 1], fw[PF + 1][2];
            auto fetch = [&](auto sc) {
                constexpr int S2 = decltype(sc)::value, TAP = S2 / 4, KSI = S2 % 4, SLOT = S2 % (PF + 1);
                // (operands of an asm statement are copied to a local first: inside nested generic lambdas hipcc does not capture a variable that only an asm operand names)
                { half8_t& d = fx[SLOT]; const unsigned ad = a_base; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(BUF * BN_PATCH + KSI * 2 * BN_PLANE + ((TAP / 3) * BN_PW + TAP % 3) * 16)); }
                const unsigned wb = TAP >= 5 ? w_base1 : w_base0;         // (ds_read offsets are 16 bits: taps 0..4 from the first base, taps 5..8 from a second one)
                sm_static_for<0, 2>([&](auto ac) {
                    constexpr int A = decltype(ac)::value;
                    half8_t& d = fw[SLOT][A]; const unsigned ad = wb;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"((TAP >= 5 ? TAP - 5 : TAP) * 8192 + KSI * 2048 + A * 512));
                });
            };
            sm_static_for<0, PF>([&](auto sc) { fetch(sc); });
            sm_static_for<0, NS>([&](auto sc) {
                constexpr int S2 = decltype(sc)::value;
                if constexpr (S2 + PF < NS) fetch(std::integral_constant<int, S2 + PF>());
                constexpr int AHEAD = (NS - 1 - S2 < PF ? NS - 1 - S2 : PF) * RD;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    if constexpr (S2 == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)], (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)], acc[a], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // epilogue: bias, activation, one rounding, lane swaps -> k step 2a + j of the expand's B operand -> hand-over buffer BUF
#pragma unroll
            for (int a = 0; a < 2; a++) {
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const floatx4 bq = *(const floatx4*)(b2l + a * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        float v0 = acc[a][4 * q + 2 * h2] + bq[2 * h2], v1 = acc[a][4 * q + 2 * h2 + 1] + bq[2 * h2 + 1];
                        if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * p.sl2 : v0; v1 = v1 < 0.f ? v1 * p.sl2 : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                        d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                    sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                    *(u32x4*)(bn_smem + BN_HO + BUF * BN_HOB + ((wave * 4 + 2 * a + j) * 64 + lane) * 16) = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
                }
            }
        };
        auto prun = [&](auto relu0c) __attribute__((always_inline)) {
            // period q: the 3x3 of tile q - 1 (patch buffer (q-1) & 1 -> hand-over buffer (q-1) & 1); barriers after periods 0 .. nt
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // my share of w2 and of the biases
            __builtin_amdgcn_s_barrier();                                  // (period 0: nothing to multiply yet)
#pragma unroll 1
            for (int q = 1; q <= nt; q += 2) {
                if (!(p.dbg & 8)) do_tile(std::integral_constant<int, 0>(), relu0c);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 > nt) break;
                if (!(p.dbg & 8)) do_tile(std::integral_constant<int, 1>(), relu0c);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (p.sl2 == 0.f) prun(std::true_type()); else prun(std::false_type());
    } else if (wave < 8) {
        // =========================== CONSUMERS: the 1x1 expand + residual, 64 output channels per wave ===========================================
        const int jw = wave - 4;                                           // output channels [64 jw, 64 jw + 64) = chunks 2 jw, 2 jw + 1
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(pix_all * (unsigned)p.y_ld * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res ? (int)(pix_all * (unsigned)p.res_ld * 2u) : 0, 0x00020000);
        half8_t w3r[2][4];                                                 // A fragments of my 64 rows of w3 (row = lane & 31 of chunk cc), resident
        const half8_t hz8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int cc = 0; cc < 2; cc++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int row = (2 * jw + cc) * 32 + l31;
                w3r[cc][ks] = row < p.n3 ? *(const half8_t*)(p.w3 + (long)row * 64 + ks * 16 + hi * 8) : hz8;
            }
        half8_t onehot[2];                                                 // E_j[r][k] = (r == 16j + k): mfma(E_j, R_j, acc) adds residual rows that arrive in the B layout
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) onehot[j][i] = (half_t)((l31 == 16 * j + 8 * hi + i) ? 1.f : 0.f);
        unsigned tyx[4];                                                   // tile pixel of (fragment f, lane & 31)
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int i = f * 32 + l31, ty = i / BN_TW, tx = i - ty * BN_TW;
            tyx[f] = i < TH * BN_TW ? ((unsigned)ty << 16) | (unsigned)tx : 0x7fff7fffu;
        }
        const bool chunk_on[2] = {(2 * jw) * 32 < p.n3, (2 * jw + 1) * 32 < p.n3};      // wave-uniform
        auto pixel_off = [&](int t, int f, int ld, bool on) -> unsigned {               // byte offset of (tile t, fragment f, lane)'s 64-channel slice, or DEAD
            int n, y0, x0; tile_origin(t, n, y0, x0);
            const int ty = (int)(tyx[f] >> 16), oy = y0 + ty, ox = x0 + (int)(tyx[f] & 0xffff);
            const bool live = on && t < t_end && oy < p.H && ox < p.W && ty < TH;
            return live ? (unsigned)(((n * p.H + oy) * p.W + ox) * ld + 64 * jw + 8 * hi) * 2u : BN_DEAD;
        };
        // residual rows of the fragment whose expand runs next but one: a ring of two fragment slots [slot][chunk][j] (a slot is refilled, as soon as its rows are consumed,
        // with the rows of the fragment two further on — this tile's or the next tile's: two fragment times of flight)
        u32x4 rres[2][2][2];
        const bool res_on = p.res != nullptr && !(p.dbg & 2);
#pragma unroll
        for (int f = 0; f < 2; f++) {
            const unsigned off = pixel_off(t_begin, f, p.res_ld, res_on);
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
                for (int j = 0; j < 2; j++) rres[f][cc][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(off + (unsigned)(cc * 64 + 32 * j)), 0, 0);
        }
        auto crun = [&](auto relu0c) __attribute__((always_inline)) {
            constexpr bool RELU0 = decltype(relu0c)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the biases I wrote; w2's DMA share is waited for below, before the first barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // after period 0
            if (nt >= 1) __builtin_amdgcn_s_barrier();                     // after period 1
#pragma unroll 1
            for (int q = 2; q <= nt + 1; q++) {                            // period q: the expand of tile q - 2 from hand-over buffer q & 1
                const int t = t_begin + q - 2;
                const unsigned char* hb = bn_smem + BN_HO + (q & 1) * BN_HOB;
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    half8_t h2[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) h2[ks] = *(const half8_t*)(hb + ((f * 4 + ks) * 64 + lane) * 16);
                    const unsigned offy = pixel_off(t, f, p.y_ld, !(p.dbg & 1));
                    const unsigned offr = f < 2 ? pixel_off(t, f + 2, p.res_ld, res_on) : pixel_off(t + 1, f - 2, p.res_ld, res_on);
#pragma unroll
                    for (int cc = 0; cc < 2; cc++) {
                        floatx16 a3;
#pragma unroll
                        for (int r = 0; r < 16; r++) a3[r] = 0.f;
                        if (!(p.dbg & 16)) {
#pragma unroll
                            for (int ks = 0; ks < 4; ks++) a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3r[cc][ks], h2[ks], a3, 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 2; j++) a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(onehot[j], __builtin_bit_cast(half8_t, rres[f & 1][cc][j]), a3, 0, 0, 0);
                        }
                        // these registers are free again: the rows of the fragment two further on
#pragma unroll
                        for (int j = 0; j < 2; j++) rres[f & 1][cc][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(offr + (unsigned)(cc * 64 + 32 * j)), 0, 0);
                        const int c = 2 * jw + cc;
                        unsigned d[4][2];
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            const floatx4 bq = *(const floatx4*)(b3l + c * 32 + 8 * qq + 4 * hi);
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                float v0 = a3[4 * qq + 2 * h] + bq[2 * h], v1 = a3[4 * qq + 2 * h + 1] + bq[2 * h + 1];
                                if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * p.sl3 : v0; v1 = v1 < 0.f ? v1 * p.sl3 : v1; }
                                half2_t hv = {(half_t)v0, (half_t)v1};
                                if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                                d[qq][h] = __builtin_bit_cast(unsigned, hv);
                            }
                        }
                        const unsigned offc = chunk_on[cc] ? offy : BN_DEAD;
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                            sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]}, rs_y, (int)(offc + (unsigned)(cc * 64 + 32 * j)), 0, SM_ST_AUX);
                        }
                    }
                }
                if (q <= nt) __builtin_amdgcn_s_barrier();                 // after periods 2 .. nt
            }
        };
        if (p.sl3 == 0.f) crun(std::true_type()); else crun(std::false_type());
    } else {
        // =========================== REDUCERS: T1 = act1(w1 . x + b1) on the tile + halo, straight from global memory =============================
        const int rw = wave - 8, chunk = rw & 1, pair = rw >> 1;           // my 32 couts of T1; my patch fragments pair, pair + 2, pair + 4
        const int NF = (TH + 2 + 1) / 2;                                   // fragments of two 16-pixel patch rows
        const int nfr = (NF - pair + 1) / 2;                               // how many of them are mine
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (p.dbg & 4) ? 0 : (int)(pix_all * (unsigned)p.x_ld * 2u), 0x00020000);
        half8_t w1r[KR];                                                   // A fragments of my 32 rows of w1
#pragma unroll
        for (int ks = 0; ks < KR; ks++) w1r[ks] = *(const half8_t*)(p.w1 + (long)(chunk * 32 + l31) * p.C0 + ks * 16 + hi * 8);
        constexpr int RDP = KR >= 8 ? 8 : KR;                              // depth of the load ring (k steps)
        // unit u = (tile index, my fragment j): byte offset of this lane's pixel row (+ 16 hi), DEAD outside the image / beyond my range
        const int pr_l = l31 >> 4, pc_l = l31 & 15;
        auto unit_off = [&](int ti, int j, bool& valid) -> unsigned {
            const int t = t_begin + ti;
            int n, y0, x0; tile_origin(t < t_end ? t : t_begin, n, y0, x0);
            const int fp = pair + 2 * j;
            const int iy = y0 - 1 + 2 * fp + pr_l, ix = x0 - 1 + pc_l;
            valid = t < t_end && j < nfr && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            return valid ? (unsigned)(((n * p.H + iy) * p.W + ix) * p.x_ld + 8 * hi) * 2u : BN_DEAD;
        };
        u32x4 xa[RDP];
        bool v_cur; unsigned off_cur = unit_off(0, 0, v_cur);
#pragma unroll
        for (int ks = 0; ks < RDP; ks++) xa[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_cur + (unsigned)(ks * 32)), 0, 0);
        auto rrun = [&](auto relu0c) __attribute__((always_inline)) {
            constexpr bool RELU0 = decltype(relu0c)::value;
#pragma unroll 1
            for (int q = 0; q < nt; q++) {                                 // period q: the patch of tile q -> patch buffer q & 1
                unsigned char* const pb = bn_smem + BN_PAT + (q & 1) * BN_PATCH;
#pragma unroll 1
                for (int j = 0; j < nfr; j++) {
                    bool v_next; const unsigned off_next = j + 1 < nfr ? unit_off(q, j + 1, v_next) : unit_off(q + 1, 0, v_next);
                    floatx16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.f;
                    sm_static_for<0, KR>([&](auto kc) {
                        constexpr int KS = decltype(kc)::value;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1r[KS], __builtin_bit_cast(half8_t, xa[KS % RDP]), acc, 0, 0, 0);
                        if constexpr (KS + RDP < KR) xa[KS % RDP] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_cur + (unsigned)((KS + RDP) * 32)), 0, 0);
                        else xa[KS % RDP] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_next + (unsigned)((KS + RDP - KR) * 32)), 0, 0);
                    });
                    // bias, activation, ONE rounding; zeros outside the image (the 3x3's padding); lane swaps -> 8 consecutive channels per lane -> channel planes
                    unsigned d[4][2];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const floatx4 bq = *(const floatx4*)(b1l + chunk * 32 + 8 * qq + 4 * hi);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            float v0 = acc[4 * qq + 2 * h] + bq[2 * h], v1 = acc[4 * qq + 2 * h + 1] + bq[2 * h + 1];
                            if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * p.sl1 : v0; v1 = v1 < 0.f ? v1 * p.sl1 : v1; }
                            half2_t hv = {(half_t)v0, (half_t)v1};
                            if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                            d[qq][h] = v_cur ? __builtin_bit_cast(unsigned, hv) : 0u;
                        }
                    }
                    const int qpx = (pair + 2 * j) * 32 + l31;                 // patch pixel of this lane
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        sm_permlane32_swap(d[2 * jj][0], d[2 * jj + 1][0]);
                        sm_permlane32_swap(d[2 * jj][1], d[2 * jj + 1][1]);
                        if (qpx < BN_NPX) *(u32x4*)(pb + (chunk * 4 + 2 * jj + hi) * BN_PLANE + qpx * 16) = u32x4{d[2 * jj][0], d[2 * jj][1], d[2 * jj + 1][0], d[2 * jj + 1][1]};
                    }
                    off_cur = off_next; v_cur = v_next;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my patch writes (the x loads stay in flight across the barrier)
                __builtin_amdgcn_s_barrier();                              // after period q
            }
            __builtin_amdgcn_s_barrier();                                  // after period nt (the producers' last tile)
        };
        // (before the first barrier everybody's share of the w2 DMA and of the biases must have landed: the DMA is an asm statement the compiler does not count)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (nt == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); return; }
        if (p.sl1 == 0.f) rrun(std::true_type()); else rrun(std::false_type());
    }
}

}  // namespace

static BnParams bn_params(const sm_bneck_desc* d) {
    BnParams p{};
    p.N = d->N; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.n3 = d->C3; p.x_ld = d->x_ld; p.y_ld = d->y_ld; p.res_ld = d->res_ld;
    auto sl = [](int act, float s) { return act == SM_ACT_NONE ? 1.f : act == SM_ACT_RELU ? 0.f : s; };
    p.sl1 = sl(d->act1, d->slope1); p.sl2 = sl(d->act2, d->slope2); p.sl3 = sl(d->act3, d->slope3);
    const int bands = sm_ceil_div(d->H, BN_THMAX);
    p.TH = sm_ceil_div(d->H, bands); p.tiles_y = sm_ceil_div(d->H, p.TH); p.tiles_x = sm_ceil_div(d->W, BN_TW);
    p.ntiles = d->N * p.tiles_x * p.tiles_y;
    sm_fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh); sm_fastdiv_make((unsigned)p.tiles_y, &p.ty_mul, &p.ty_sh);
    return p;
}

extern "C" int sm_bottleneck_supported(const sm_bneck_desc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0) return 0;
    if (d->Cm != 64 || (d->C0 != 64 && d->C0 != 128 && d->C0 != 256) || d->C3 < 32 || d->C3 > 256 || (d->C3 % 32) != 0) return 0;
    if (d->act1 > SM_ACT_PRELU || d->act2 > SM_ACT_PRELU || d->act3 > SM_ACT_PRELU) return 0;
    if (d->x_ld < d->C0 || d->y_ld < d->C3 || (d->x_ld & 7) || (d->y_ld & 7) || (d->res_ld & 7)) return 0;
    const long px = (long)d->N * d->H * d->W;
    if (px * d->x_ld * 2 >= 0x7fff0000L || px * d->y_ld * 2 >= 0x7fff0000L || px * (d->res_ld > 0 ? d->res_ld : 1) * 2 >= 0x7fff0000L) return 0;   // 32-bit buffer offsets
    if ((long)d->N * sm_ceil_div(d->W, BN_TW) * sm_ceil_div(d->H, BN_THMAX) >= (1L << 30)) return 0;
    return 1;
}

extern "C" int sm_bottleneck_fwd(const sm_bneck_desc* d, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                 const void* res, void* y, sm_stream_t stream) {
    SM_REQUIRE(d && x && w1 && w2 && w3 && y, "sm_bottleneck_fwd: null argument");
    SM_REQUIRE(sm_bottleneck_supported(d), "sm_bottleneck_fwd: no kernel takes this block (1x1 C0 in {64, 128, 256} -> 64, 3x3 / s1 / p1 64 -> 64, 1x1 64 -> C3 <= 256, C3 %% 32 == 0, ReLU-family activations, tensors < 2 GB)");
    SM_REQUIRE(!res || d->res_ld >= d->C3, "sm_bottleneck_fwd: residual pixel stride %d < C3 %d", d->res_ld, d->C3);
    SM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)w1) | ((uintptr_t)w2) | ((uintptr_t)w3) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) == 0, "sm_bottleneck_fwd: operands must be 16-byte aligned");
    BnParams p = bn_params(d);
    p.x = (const half_t*)x; p.w1 = (const half_t*)w1; p.b1 = b1; p.w2 = (const half_t*)w2; p.b2 = b2; p.w3 = (const half_t*)w3; p.b3 = b3; p.res = (const half_t*)res; p.y = (half_t*)y;
    { static const int dbg = SM_ABL_ENV("SM_BN_DEBUG") ? atoi(SM_ABL_ENV("SM_BN_DEBUG")) : 0; p.dbg = dbg; }   // ablation switch (diagnostic builds only), read once
    void (*fn)(BnParams) = d->C0 == 256 ? bneck_kernel<16> : d->C0 == 128 ? bneck_kernel<8> : bneck_kernel<4>;
    static bool attr_done[3] = {false, false, false};
    const int ai = d->C0 == 256 ? 0 : d->C0 == 128 ? 1 : 2;
    if (!attr_done[ai]) { SM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS)); attr_done[ai] = true; }
    const unsigned grid = (unsigned)(p.ntiles < SM_NUM_CU ? p.ntiles : SM_NUM_CU);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(768), BN_LDS, sm_s(stream), p);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
