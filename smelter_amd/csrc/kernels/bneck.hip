// bneck.hip — a whole ResNet BOTTLENECK BLOCK as ONE tile pass on gfx950:
//
//     T1 = act1(w1 . x  + b1)            1x1 "reduce",  C0 -> 64            (MPSCNNConvolutionNode, /root/reference/Sources/Smelter/Converters.swift:187-338)
//     T2 = act2(w2 * T1 + b2)            3x3 / stride 1 / pad 1, 64 -> 64   (same node class; padding rule ONNXConvolutionPadding.swift:73-79: zeros outside the image)
//     y  = act3(w3 . T2 + b3 + res)      1x1 "expand",  64 -> n3 <= 256, + residual (MPSNNAdditionNode, Converters.swift:430-451), ReLU family (:342-359)
//
// The reference hands MPS three convolution nodes, an addition node and three neuron nodes; layer by layer that is x read twice (reduce, residual), T1 and T2 written and
// read back: 360 MB per block at 64 x 56 x 56 x 256.  Here the two hidden tensors never leave the CU (T1: LDS, T2: registers; both rounded to fp16 exactly as the
// layer-by-layer kernels round them on their way to memory — they stay in the parity oracle's rounding list).  A block of the grid (8 waves, two per SIMD, <= 256 registers)
// walks output tiles of TH <= 8 rows x 14 columns down a column band; per tile
//   * REDUCER waves (4..7) compute T1 on the tile + its one-pixel halo — a 16-column PATCH of TH + 2 rows, 32-pixel fragments of two patch rows — straight from global
//     memory: a lane's B operand of k step ks IS 16 bytes of its pixel's row (channels 16 ks + 8 hi ..), so x passes no LDS.  ALL of w1 lives in a reducer's registers (128),
//     so a fragment of x is loaded ONCE (the first form of this kernel split the couts over two waves and loaded it twice: the CU's vector-memory pipe, ~25 B/clk, was
//     the bottleneck — profiles/r06_bneck_*); the whole NEXT fragment is requested while this one is multiplied (a ring of KR loads).  Bias, activation, ONE rounding, lane
//     swaps, then the patch goes to LDS in channel PLANES [8 channels][patch pixel] x 16 bytes: every fragment address of the 3x3 is "lane base + compile-time offset"
//     (no per-tap address registers), every access 16 consecutive 16-byte slots per 16 lanes.  Patch pixels outside the image are written as ZEROS (the 3x3's padding —
//     not act1(b1), which is what the GEMM alone leaves there).  SLIDING WINDOW: a tile directly below the previous one (TH = 8) inherits patch rows 0, 1 from that
//     tile's rows 8, 9 (an LDS -> LDS copy of 4 KB): four fragments = one per reducer per tile, x is read 1.14 x (the column halo) instead of 1.43 x.
//   * PC waves (0..3): the 3x3 over the patch for one 32-pixel fragment each — w2 [tap][8-channel chunk][cout] resident in LDS (72 KB, loaded once per block by LDS-DMA),
//     hand-counted fragment prefetch as in conv3x3_ws.hpp — and then, on the SAME fragment while it is still in registers (bias, activation, rounded: eight consecutive
//     channels per lane = the MFMA B layout), the expand: w3 fragments from LDS (32 KB), four 32-channel chunks at a time on independent accumulators, the residual rows
//     (requested one tile ahead, re-filled chunk by chunk as they are consumed) added through two one-hot MFMAs (exact), + b3, activation, one rounding, 16-byte stores.
// Two tiles are in flight per block (patch of tile t+1 | 3x3 + expand of tile t), ONE s_barrier per tile.  LDS: 72 KB w2 + 32 KB w3 + 1.5 KB biases + 2 x 20 KB patch.
// Numerics: every layer accumulates in fp32 over k ascending in MFMA steps of 16 from zero, + bias in fp32, activation, one rounding to fp16 — what the per-layer kernels do
// (bit-identical to the 1x1 launch + the 3x3 -> expand-head launch it replaces: tests/test_gpu_bneck.py).
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include "conv_shared.hpp"

namespace {

struct BnParams {
    const half_t* x; const half_t* w1; const float* b1; const half_t* w2; const float* b2; const half_t* w3; const float* b3; const half_t* res; half_t* y;
    int N, H, W, C0, n3, x_ld, y_ld, res_ld;
    float sl1, sl2, sl3;             // f(v) = v < 0 ? v * sl : v   (1 = no activation, 0 = ReLU)
    int TH, tiles_x, tiles_y, ntiles;
    unsigned txy_mul, txy_sh, ty_mul, ty_sh;
    int dbg;                         // ablation bits (diagnostics): 1 = no output stores, 2 = no residual loads, 4 = no x loads, 8 = no 3x3 MFMAs, 16 = no expand MFMAs, 32 = no sliding window
};

// Product builds carry no ablation branches: `dbg` folds to 0 (the accumulators of the first MFMA of a chain then start from the literal 0, no branch sits in a chain).
#ifdef SM_ABLATION
#define BN_DBG(p) ((p).dbg)
#else
#define BN_DBG(p) 0
#endif
// Timeline of a launch (diagnostic build -DSM_BN_TL; tools/bneck_tl.py): s_memtime stamps of wave 0 (PC) and wave 4 (reducer) of every block
#ifdef SM_BN_TL
__device__ unsigned long long g_bn_tl[256 * 2 * 64];
#define BN_STAMP(slot) do { if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256 && (slot) < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_bn_tl[(blockIdx.x * 2 + (wave >> 2)) * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define BN_STAMP(slot) do { } while (0)
#endif
// cache policy of the output stores / of the x and residual loads (aux operand of the raw buffer builtins: 0 default, 2 = nt, 16 = sc1, 17 = sc0 sc1) — build-time knobs for the A/B
// Ranked IN THE NETWORK (bench.py --steps 300, same box, alternating): plain stores -1.1 .. -1.4 % of the ResNet-50 step against sc1; nt on the RESIDUAL rows (read for the last time)
// another -1.1 % (five rounds: 1.0902 -> 1.0786 ms); nt on the x stream as well: faster on one box, slower on another (1.0860), and the residual rows then miss the L2 the x loads
// would have filled (216 instead of 132 MB read per launch, 76 instead of 58 us in the loop) — x stays plain.
// (The chained microbenchmark — two tensors ping-ponging inside the Infinity Cache — ranks them the other way round: sc1 stores best, nt anything 14 .. 28 us worse,
//  profiles/r06_bneck_policy.txt; a block costs 54 us there and 62 .. 77 us between its real neighbours.)
#ifndef BN_ST_AUX
#define BN_ST_AUX 0
#endif
#ifndef BN_LDX_AUX
#define BN_LDX_AUX 0
#endif
#ifndef BN_LDR_AUX
#define BN_LDR_AUX 2
#endif
constexpr int BN_TW = 14, BN_PW = 16, BN_THMAX = 8, BN_NPX = (BN_THMAX + 2) * BN_PW;       // 160 patch pixels = 5 fragments
constexpr int BN_PLANE = BN_NPX * 16, BN_PATCH = 8 * BN_PLANE;                              // 2560-byte channel planes, 20 KB per patch
constexpr int BN_W3 = 9 * 64 * 128, BN_B1 = BN_W3 + 256 * 128, BN_B2 = BN_B1 + 256, BN_B3 = BN_B2 + 256, BN_PAT = BN_B3 + 1024, BN_LDS = BN_PAT + 2 * BN_PATCH;
static_assert(BN_LDS <= 160 * 1024, "LDS budget");
constexpr unsigned BN_DEAD = 0x80000000u;   // buffer offset beyond every descriptor: loads return 0, stores are dropped

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS base pointers the optimiser cannot see through.  The kernel's LDS image is > 64 KB: written as `bn_smem + BN_B3 + ...` an address is "0 + a constant beyond the 16-bit
// offset field of ds_read", and hipcc then keeps EVERY such address in a register of its own (32 bias addresses, 32 fragment addresses: the first pipelined expand spilled
// over them).  A laundered per-lane base in one register + small compile-time offsets is what the hardware wants.
typedef const __attribute__((address_space(3))) unsigned char* bn_lds_cptr;
__device__ __forceinline__ bn_lds_cptr bn_lds_base(unsigned addr) { asm volatile("" : "+v"(addr)); return reinterpret_cast<bn_lds_cptr>(addr); }
template <class T> __device__ __forceinline__ T bn_lds_read(bn_lds_cptr base, int off) { return *(const __attribute__((address_space(3))) T*)(base + off); }

// ROW-PAIR TRANSPOSE of two MFMA-layout registers.  In the B / store layout of the 32x32x16 MFMA lane (p = lane & 31, hi = lane >> 5) holds 16 bytes of pixel p: per memory
// instruction a pixel contributes 2 x 16 bytes, i.e. a wave touches 32 cache lines for 32 bytes each — the CU's vector-memory pipe spends a tag look-up and a 64-byte sector
// on every one of them (measured: ~60 clk per instruction, the bottleneck of the first forms of this kernel).  Two registers a, b that hold the neighbouring 16-byte pieces
// (j = 0, 1: bytes +0 and +32 of a 64-byte group) of 32 pixels are exchanged by v_permlane16_swap (odd 16-lane rows of a <-> even rows of b) into
//     a' = all four pieces of pixels 0..15,  b' = all four pieces of pixels 16..31      lane L: pixel (L & 15) [+ 16], piece at byte ((L >> 4) & 1) * 32 + (L >> 5) * 16
// so that ONE instruction moves 16 pixels x 64 contiguous bytes (16 lines, whole sectors).  The exchange is its own inverse: loads fetch a', b' and swap into a, b;
// stores swap a, b into a', b'.  Pure data movement: the values, and every rounding, are what they were.
__device__ __forceinline__ void bn_swap16(u32x4& a, u32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; e++) { unsigned x = a[e], y = b[e]; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); a[e] = x; b[e] = y; }
}

// KR = C0 / 16 k steps of the reduce (C0 in {64, 128, 256})
template <int KR>
__global__ __launch_bounds__(512) void bneck_kernel(BnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bn_smem[];
    const unsigned lds0 = ws_lds_addr(bn_smem);
    float* const b1l = (float*)(bn_smem + BN_B1); float* const b2l = (float*)(bn_smem + BN_B2); float* const b3l = (float*)(bn_smem + BN_B3);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = p.TH;

    // ---- this block's contiguous tile range.  Block b runs on XCD b % 8: logical block (b % 8) * G/8 + b / 8 gives every XCD a contiguous range of tiles, i.e. whole
    // images — the column bands of one image run side by side on CUs that share an L2 (their halo columns hit it) -------------------------------------------------
    const int T = p.ntiles, G = gridDim.x;
    const int lb = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t_begin = (int)((long)lb * T / G), t_end = (int)((long)(lb + 1) * T / G), nt = t_end - t_begin;
    if (nt <= 0) return;                                               // (grid <= tiles: cannot happen; nobody has touched a barrier yet)
    auto tile_origin = [&](int t, int& n, int& y0, int& x0) {      // tile id -> (image, column band, row band), row band fastest
        n = (int)sm_fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y);
        const int bx = (int)sm_fastdiv((unsigned)r, p.ty_mul, p.ty_sh), by = r - bx * p.tiles_y;
        y0 = by * TH; x0 = bx * BN_TW;
    };
    const unsigned pix_all = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W;

    if (wave < 4) {
        // =========================== PC WAVES: the 3x3 over the T1 patch, then the expand on the same fragment ===========================
        // ---- weights -> LDS (the reducers are already streaming x): w2 [cout][tap][cin] -> [tap][chunk c][cout] x 16 bytes; w3 [n3][64] -> [32-row chunk][k chunk][row] x 16 bytes
        {
            constexpr int W2SLOTS = 9 * 8 * 64;
#pragma unroll 1
            for (int s0 = wave * 64; s0 < W2SLOTS; s0 += 256) {
                const int s = s0 + lane, n = s & 63, c = (s >> 6) & 7, tap = s >> 9;
                ws_glds16(p.w2 + ((long)n * 9 + tap) * 64 + c * 8, lds0 + (unsigned)s0 * 16u);
            }
#pragma unroll 1
            for (int s0 = wave * 64; s0 < 2048; s0 += 256) {
                const int s = s0 + lane, row = s & 31, kc = (s >> 5) & 7, c = s >> 8;
                int r3 = c * 32 + row; if (r3 >= p.n3) r3 = p.n3 - 1;                  // (chunks beyond n3 are never multiplied: any valid address)
                ws_glds16(p.w3 + (long)r3 * 64 + kc * 8, lds0 + BN_W3 + (unsigned)s0 * 16u);
            }
            for (int i = tid; i < 64; i += 256) b2l[i] = p.b2 ? p.b2[i] : 0.f;
            for (int i = tid; i < 256; i += 256) b3l[i] = (p.b3 && i < p.n3) ? p.b3[i] : 0.f;
        }
        const int i = wave * 32 + l31, ty = i / BN_TW, tx = i - ty * BN_TW;
        const bool in_tile = i < TH * BN_TW;
        const int slot0 = in_tile ? ty * BN_PW + tx : 0;
        const unsigned a_base = lds0 + BN_PAT + (unsigned)(hi * BN_PLANE + slot0 * 16);
        const unsigned w_base0 = lds0 + (unsigned)(hi * 1024 + l31 * 16), w_base1 = w_base0 + 5 * 8192;
        const bn_lds_cptr w3f = bn_lds_base(lds0 + BN_W3 + (unsigned)(hi * 512 + l31 * 16));     // A fragment of (chunk c, k step ks) = w3f + c * 4096 + ks * 1024
        const bn_lds_cptr b2f = bn_lds_base(lds0 + BN_B2 + (unsigned)(hi * 16)), b3f = bn_lds_base(lds0 + BN_B3 + (unsigned)(hi * 16));   // bias rows 8 q + 4 hi .. +3 of chunk c at + c * 128 + q * 32
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(pix_all * (unsigned)p.y_ld * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res ? (int)(pix_all * (unsigned)p.res_ld * 2u) : 0, 0x00020000);
        half8_t onehot[2];                                                 // E_j[r][k] = (r == 16j + k): mfma(E_j, R_j, acc) adds residual rows that arrive in the B layout
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 8; e++) onehot[j][e] = (half_t)((l31 == 16 * j + 8 * hi + e) ? 1.f : 0.f);
        const int nch = p.n3 >> 5;                                         // 32-channel chunks of the output
        // memory-side lane map (bn_swap16): instruction 0 of a pair moves pixels 0..15 of the fragment, instruction 1 pixels 16..31; lane L -> pixel (L & 15), 16-byte piece
        // ((L >> 4) & 1) * 32 + (L >> 5) * 16 of each 64-byte channel group
        const int mi0 = wave * 32 + (lane & 15), mty0 = mi0 / BN_TW, mtx0 = mi0 - mty0 * BN_TW, mi1 = mi0 + 16, mty1 = mi1 / BN_TW, mtx1 = mi1 - mty1 * BN_TW;
        const bool min0 = mi0 < TH * BN_TW, min1 = mi1 < TH * BN_TW;
        const unsigned piece = (unsigned)(((lane >> 4) & 1) * 32 + (lane >> 5) * 16);
        auto pixel_off = [&](int t, int ld, bool on, int which) -> unsigned {   // byte offset of (tile t, this lane's pixel of instruction `which`) + its piece, or DEAD
            int n, y0, x0; tile_origin(t < t_end ? t : t_begin, n, y0, x0);
            const int oy = y0 + (which ? mty1 : mty0), ox = x0 + (which ? mtx1 : mtx0);
            const bool live = on && t < t_end && (which ? min1 : min0) && oy < p.H && ox < p.W;
            return live ? (unsigned)(((n * p.H + oy) * p.W + ox) * ld) * 2u + piece : BN_DEAD;
        };
        const bool res_on = p.res != nullptr && !(BN_DBG(p) & 2);
        // residual rows, as loaded (bn_swap16 turns [instruction] into [j]): three PAIR slots [slot][chunk of pair][instruction].  Pairs 0, 1, 2 of a tile are requested ONE TILE
        // AHEAD, each as soon as its slot's previous tenant has been added (during the previous tile's expand: stages 3, 1, 2) — that is the half period in which the reducers
        // read the same rows for the patch, so the XCD's L2 (4 MB against ~2.5 MB of traffic in between) still holds them and the fabric carries them once; pair 3 rides in
        // pair 0's slot inside the tile (64 registers would hold a whole tile ahead; the expand's pipeline leaves 48).
        u32x4 rres[3][2][2];
        auto load_res = [&](int t, auto kc, auto slotc) __attribute__((always_inline)) {       // pair K of tile t -> slot (chunks beyond n3 / tiles beyond the range: dead offset, zeros)
            constexpr int K = decltype(kc)::value, SL = decltype(slotc)::value;
            const unsigned o0 = pixel_off(t, p.res_ld, res_on, 0), o1 = pixel_off(t, p.res_ld, res_on, 1);
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const bool on = 2 * K + cc < nch;
                rres[SL][cc][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)((on ? o0 : BN_DEAD) + (unsigned)((2 * K + cc) * 64)), 0, BN_LDR_AUX);
                rres[SL][cc][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)((on ? o1 : BN_DEAD) + (unsigned)((2 * K + cc) * 64)), 0, BN_LDR_AUX);
            }
        };
        load_res(t_begin, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
        load_res(t_begin, std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
        load_res(t_begin, std::integral_constant<int, 2>(), std::integral_constant<int, 2>());
        constexpr int NS = 36, PF = 4, RD = 3;
        static_assert(PF * RD <= 15, "lgkmcnt field");
        auto do_tile = [&](auto bufc, auto relu2c, auto relu3c, int t) {
            constexpr int BUF = decltype(bufc)::value;
            constexpr bool RELU2 = decltype(relu2c)::value, RELU3 = decltype(relu3c)::value;
            const int tlb = (t - t_begin + 1) * 5; (void)tlb;
            BN_STAMP(tlb);
            floatx16 acc[2];
            {
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
            }
            BN_STAMP(tlb + 1);
            // 3x3 epilogue: bias, activation, one rounding, lane swaps -> outv[a][j] = k step 2a + j of the expand's B operand (channels 32a + 16j + 8hi .. +7 of this lane's pixel)
            u32x4 outv[2][2];
#pragma unroll
            for (int a = 0; a < 2; a++) {
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const floatx4 bq = bn_lds_read<floatx4>(b2f, a * 128 + q * 32);
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        float v0 = acc[a][4 * q + 2 * h2] + bq[2 * h2], v1 = acc[a][4 * q + 2 * h2 + 1] + bq[2 * h2 + 1];
                        if constexpr (!RELU2) { v0 = v0 < 0.f ? v0 * p.sl2 : v0; v1 = v1 < 0.f ? v1 * p.sl2 : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if constexpr (RELU2) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                        d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                    sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                    outv[a][j] = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
                }
            }
            BN_STAMP(tlb + 2);
            // ---- expand, SOFTWARE-PIPELINED over PAIRS of 32-channel chunks.  A wave issues in order: a burst of MFMAs followed by a burst of epilogue VALU work leaves the
            // matrix pipe idle during the second and the wave stalled during the first (the first form of this loop: 7 k clk per tile, more than the 3x3 itself).  Here the twelve
            // MFMAs of pair k+1 (eight w3 . T2 steps, four one-hot residual adds; the two chunks' chains ALTERNATE — a dependent MFMA waits ~100 clk for its predecessor, an
            // independent one 32: one chain at a time measured 115 clk per MFMA) are dealt ONE per slice between the twelve slices of pair k's epilogue (~7 VALU instructions
            // each), the order pinned by sched_barrier; w3 fragments ride a four-deep ring of LDS reads, the bias rows of the next pair are fetched as soon as this pair's are
            // consumed.  Everything is branch-free: chunks beyond n3 compute on whatever the LDS holds and store to a dead offset.
            const unsigned offy0 = pixel_off(t, p.y_ld, !(BN_DBG(p) & 1), 0), offy1 = pixel_off(t, p.y_ld, !(BN_DBG(p) & 1), 1);
            half8_t w3r[4];                                                // ring: w3 fragment F = 8 * pair + slice (chunk of pair = F & 1, k step = (F >> 1) & 3) in slot F & 3
            floatx16 a3[2][2];                                             // [pair & 1][chunk of pair]
            floatx4 bqr[4];                                                // ring: bias rows G = 4 * chunk + q in slot G & 3 (row q of the NEXT chunk is requested when this chunk's is consumed)
            unsigned dd[4][2];                                             // the chunk being converted
            u32x4 o2[2];
            // MFMA number S (0..11) of pair K: S < 8 the w3 . T2 step (chunk S & 1, k step S >> 1), then the residual rows through the two one-hot matrices
            auto mfma_slice = [&](auto kc, auto sc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value, S = decltype(sc)::value, B = K & 1;
                if constexpr (S < 8) {
                    constexpr int CC = S & 1, KS = S >> 1, F = K * 8 + S, F4 = F + 4;
                    if constexpr (KS == 0) a3[B][CC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3r[F & 3], __builtin_bit_cast(half8_t, outv[0][0]), (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else a3[B][CC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3r[F & 3], __builtin_bit_cast(half8_t, outv[KS >> 1][KS & 1]), a3[B][CC], 0, 0, 0);
                    if constexpr (F4 < 32) w3r[F & 3] = bn_lds_read<half8_t>(w3f, ((F4 >> 3) * 2 + (F4 & 1)) * 4096 + ((F4 >> 1) & 3) * 1024);   // (the slot is free: four fragments ahead)
                } else {
                    constexpr int CC = (S - 8) & 1, J = (S - 8) >> 1, C = 2 * K + CC;
                    if constexpr (J == 0) bn_swap16(rres[K % 3][CC][0], rres[K % 3][CC][1]);          // as loaded -> [j] (the B layout); pair 3 sits in slot 0
                    a3[B][CC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(onehot[J], __builtin_bit_cast(half8_t, rres[K % 3][CC][J]), a3[B][CC], 0, 0, 0);
                }
            };
            // slice S (0..11) of pair K's epilogue: per chunk four conversion slices (8 accumulators each: + b3, activation, one rounding), the half-wave swaps, then the
            // row-pair transpose + the two stores
            auto epi_slice = [&](auto kc, auto sc) __attribute__((always_inline)) {
                constexpr int K = decltype(kc)::value, S = decltype(sc)::value, B = K & 1, CC = S / 6, PART = S % 6, C = 2 * K + CC;
                if constexpr (PART < 4) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        float v0 = a3[B][CC][4 * PART + 2 * h] + bqr[PART][2 * h], v1 = a3[B][CC][4 * PART + 2 * h + 1] + bqr[PART][2 * h + 1];
                        if constexpr (!RELU3) { v0 = v0 < 0.f ? v0 * p.sl3 : v0; v1 = v1 < 0.f ? v1 * p.sl3 : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if constexpr (RELU3) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                        dd[PART][h] = __builtin_bit_cast(unsigned, hv);
                    }
                    if constexpr (C + 1 < 8) bqr[PART] = bn_lds_read<floatx4>(b3f, (C + 1) * 128 + PART * 32);   // (consumed: the same row of the next chunk, six slices of flight)
                } else if constexpr (PART == 4) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        sm_permlane32_swap(dd[2 * j][0], dd[2 * j + 1][0]);
                        sm_permlane32_swap(dd[2 * j][1], dd[2 * j + 1][1]);
                        o2[j] = u32x4{dd[2 * j][0], dd[2 * j][1], dd[2 * j + 1][0], dd[2 * j + 1][1]};
                    }
                } else {
                    bn_swap16(o2[0], o2[1]);                                   // [j] -> 16 pixels x 64 contiguous bytes per store
                    const bool on = C < nch;                                   // wave-uniform; folded into the offsets: no branch
                    __builtin_amdgcn_raw_buffer_store_b128(o2[0], rs_y, (int)((on ? offy0 : BN_DEAD) + (unsigned)(C * 64)), 0, BN_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(o2[1], rs_y, (int)((on ? offy1 : BN_DEAD) + (unsigned)(C * 64)), 0, BN_ST_AUX);
                }
            };
            if (!(BN_DBG(p) & 16)) {
                // prologue: the ring, the first pair's bias rows, the first pair's MFMAs
#pragma unroll
                for (int F = 0; F < 4; F++) w3r[F] = bn_lds_read<half8_t>(w3f, (F & 1) * 4096 + (F >> 1) * 1024);
#pragma unroll
                for (int qq = 0; qq < 4; qq++) bqr[qq] = bn_lds_read<floatx4>(b3f, qq * 32);
                sm_static_for<0, 12>([&](auto sc) { mfma_slice(std::integral_constant<int, 0>(), sc); });
                load_res(t, std::integral_constant<int, 3>(), std::integral_constant<int, 0>());      // (pair 0's adds have issued: this tile's pair 3 into its slot)
                __builtin_amdgcn_sched_barrier(0);
                // (all four pairs are walked whatever n3 is — the weights' LDS image and the bias image hold eight chunks, dead chunks store nowhere; narrow heads are not what
                //  this kernel is for: wasted MFMAs at n3 = 64 cost less than a branchy loop costs the 256-channel case)
                sm_static_for<0, 4>([&](auto kc) {
                    constexpr int K = decltype(kc)::value;
                    // the NEXT tile's pairs, each into the slot whose tenant was added in the stage before
                    if constexpr (K == 1) load_res(t + 1, std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
                    if constexpr (K == 2) load_res(t + 1, std::integral_constant<int, 2>(), std::integral_constant<int, 2>());
                    if constexpr (K == 3) load_res(t + 1, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
                    sm_static_for<0, 12>([&](auto sc) {
                        if constexpr (K + 1 < 4) mfma_slice(std::integral_constant<int, K + 1>(), sc);
                        epi_slice(kc, sc);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    if constexpr (K == 1 || K == 3) BN_STAMP(tlb + 3 + (K >> 1));
                });
            }
        };
        auto prun = [&](auto relu2c, auto relu3c) __attribute__((always_inline)) {
            // period q: the 3x3 + expand of tile q - 1 from patch buffer (q-1) & 1; barriers after periods 0 .. nt - 1
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // my share of the weights and of the biases (and the first residual rows)
            __builtin_amdgcn_s_barrier();                                  // after period 0 (nothing to multiply yet)
#pragma unroll 1
            for (int q = 1; q <= nt; q += 2) {
                if (!(BN_DBG(p) & 8)) do_tile(std::integral_constant<int, 0>(), relu2c, relu3c, t_begin + q - 1);
                if (q + 1 > nt) break;
                __builtin_amdgcn_s_barrier();                              // after period q (my patch reads returned: their data went through the MFMAs)
                __builtin_amdgcn_sched_barrier(0);
                if (!(BN_DBG(p) & 8)) do_tile(std::integral_constant<int, 1>(), relu2c, relu3c, t_begin + q);
                if (q + 2 > nt) break;
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (p.sl2 == 0.f && p.sl3 == 0.f) prun(std::true_type(), std::true_type());
        else if (p.sl2 == 0.f) prun(std::true_type(), std::false_type());
        else prun(std::false_type(), std::false_type());                   // (a slope of 0 in the general form is ReLU all the same)
    } else {
        // =========================== REDUCERS: T1 = act1(w1 . x + b1) on the tile + halo, straight from global memory =============================
        const int rw = wave - 4;
        const int NF = (TH + 2 + 1) / 2;                                   // fragments of two 16-pixel patch rows
        const bool can_slide = TH == BN_THMAX && !(BN_DBG(p) & 32);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (BN_DBG(p) & 4) ? 0 : (int)(pix_all * (unsigned)p.x_ld * 2u), 0x00020000);
        half8_t w1r[2][KR];                                                // A fragments of w1: [32-cout chunk][k step], row = lane & 31
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int ks = 0; ks < KR; ks++) w1r[c][ks] = *(const half8_t*)(p.w1 + (long)(c * 32 + l31) * p.C0 + ks * 16 + hi * 8);
        // b1 is read in period 0, BEFORE the first barrier: every reducer wave writes all of it itself (four waves store identical values; a wave sees its own LDS writes in order)
        b1l[lane] = p.b1 ? p.b1[lane] : 0.f;
        const bn_lds_cptr b1f = bn_lds_base(lds0 + BN_B1 + (unsigned)(hi * 16));
        const int pr_l = l31 >> 4, pc_l = l31 & 15;
        // period q computes the patch of tile t_begin + q.  slide(q): the tile is directly below the previous one -> rows 0, 1 are copied, fragments 1 .. 4 computed (one per
        // reducer); else fragments 0 .. NF-1: reducer r takes r and r + 4.  unit (q, j) = my j-th fragment of period q (-1: none)
        auto slide = [&](int q) -> bool {
            if (!can_slide || q <= 0 || q >= nt) return false;
            int n, y0, x0; tile_origin(t_begin + q, n, y0, x0);
            return y0 != 0;                                                // (row band fastest: the previous tile id is the band above unless this one starts a column band)
        };
        auto n_units = [&](int q) -> int { return (q < nt && !slide(q) && rw + 4 < NF) ? 2 : 1; };   // (the first unit of a period is always walked, even empty: the ring keeps its order)
        auto unit_frag = [&](int q, int j) -> int {
            if (q >= nt) return -1;
            if (slide(q)) return rw + 1;
            const int fp = j == 0 ? rw : rw + 4;
            return fp < NF ? fp : -1;
        };
        // compute-side validity of this lane's patch pixel (fragment fp, pixel lane & 31): inside the image?
        auto unit_valid = [&](int q, int fp) -> bool {
            int n, y0, x0; tile_origin(t_begin + (q < nt ? q : 0), n, y0, x0);
            const int iy = y0 - 1 + 2 * fp + pr_l, ix = x0 - 1 + pc_l;
            return fp >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        };
        // memory-side offsets (bn_swap16): instruction `which` of a pair reads patch row 2 fp + which, lane L its pixel (L & 15) and 16-byte piece; DEAD outside the image
        const unsigned piece = (unsigned)(((lane >> 4) & 1) * 32 + (lane >> 5) * 16);
        auto unit_off = [&](int q, int fp, int which) -> unsigned {
            int n, y0, x0; tile_origin(t_begin + (q < nt ? q : 0), n, y0, x0);
            const int iy = y0 - 1 + 2 * fp + which, ix = x0 - 1 + (lane & 15);
            const bool valid = fp >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            return valid ? (unsigned)(((n * p.H + iy) * p.W + ix) * p.x_ld) * 2u + piece : BN_DEAD;
        };
        // the ring of x loads, in PAIRS (k steps 2m, 2m + 1 = one 64-byte channel group, rows 2 fp and 2 fp + 1 of the patch): KR deep = the whole next unit in flight, where the
        // registers allow it (w1 takes 8 KR of them); C0 = 256: eight k steps — the second half of a unit is requested while its first half is multiplied (a reducer has one
        // unit per tile in the steady state: the wait this leaves is inside its slack)
        constexpr int RDP = KR > 8 ? 8 : KR;
        static_assert((RDP & 1) == 0 && (KR % RDP) == 0, "whole pairs");
        u32x4 xa[RDP];
        int fp_cur = unit_frag(0, 0);
        bool v_cur = unit_valid(0, fp_cur); unsigned off_cur0 = unit_off(0, fp_cur, 0), off_cur1 = unit_off(0, fp_cur, 1);
#pragma unroll
        for (int ks = 0; ks < RDP; ks++) xa[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(((ks & 1) ? off_cur1 : off_cur0) + (unsigned)((ks >> 1) * 64)), 0, BN_LDX_AUX);
        auto rrun = [&](auto relu0c) __attribute__((always_inline)) {
            constexpr bool RELU0 = decltype(relu0c)::value;
#pragma unroll 1
            for (int q = 0; q < nt; q++) {                                 // period q: the patch of tile q -> patch buffer q & 1
                unsigned char* const pb = bn_smem + BN_PAT + (q & 1) * BN_PATCH;
                BN_STAMP(q * 5);
                if (slide(q)) {
                    // rows 8, 9 of the previous patch (complete since the last barrier; the PC waves only read it) = rows 0, 1 of this one: planes 2 rw, 2 rw + 1
                    const unsigned char* const po = bn_smem + BN_PAT + ((q & 1) ^ 1) * BN_PATCH;
                    const u32x4 v = *(const u32x4*)(po + (2 * rw + hi) * BN_PLANE + (BN_THMAX * BN_PW + l31) * 16);
                    *(u32x4*)(pb + (2 * rw + hi) * BN_PLANE + l31 * 16) = v;
                }
                const int nu = n_units(q);
#pragma unroll 1
                for (int j = 0; j < nu; j++) {
                    // the unit after this one: my second fragment of this period, else my first of the next
                    const int q_next = j + 1 < nu ? q : q + 1, fp_next = j + 1 < nu ? unit_frag(q, j + 1) : unit_frag(q + 1, 0);
                    const bool v_next = unit_valid(q_next, fp_next); const unsigned off_next0 = unit_off(q_next, fp_next, 0), off_next1 = unit_off(q_next, fp_next, 1);
                    floatx16 acc[2];
#pragma unroll
                    for (int c = 0; c < 2; c++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
                    sm_static_for<0, KR>([&](auto kc) {
                        constexpr int KS = decltype(kc)::value;
                        if constexpr ((KS & 1) == 0) bn_swap16(xa[KS % RDP], xa[(KS + 1) % RDP]);      // the pair as loaded -> k steps KS, KS + 1 in the B layout
#pragma unroll
                        for (int c = 0; c < 2; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1r[c][KS], __builtin_bit_cast(half8_t, xa[KS % RDP]), acc[c], 0, 0, 0);
                        if constexpr ((KS & 1) == 1) {                         // both registers of the pair are free: the pair RDP k steps on (this unit's, or the next unit's)
                            constexpr int K0 = KS - 1 + RDP;
                            if constexpr (K0 < KR) {
                                xa[(KS - 1) % RDP] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_cur0 + (unsigned)((K0 >> 1) * 64)), 0, BN_LDX_AUX);
                                xa[KS % RDP] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_cur1 + (unsigned)((K0 >> 1) * 64)), 0, BN_LDX_AUX);
                            } else {
                                xa[(KS - 1) % RDP] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_next0 + (unsigned)(((K0 - KR) >> 1) * 64)), 0, BN_LDX_AUX);
                                xa[KS % RDP] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off_next1 + (unsigned)(((K0 - KR) >> 1) * 64)), 0, BN_LDX_AUX);
                            }
                        }
                    });
                    BN_STAMP(q * 5 + 1 + 2 * j);
                    // bias, activation, ONE rounding; zeros outside the image (the 3x3's padding); lane swaps -> 8 consecutive channels per lane -> channel planes
                    const int qpx = fp_cur * 32 + l31;                         // patch pixel of this lane
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        unsigned d[4][2];
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            const floatx4 bq = bn_lds_read<floatx4>(b1f, c * 128 + qq * 32);
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                float v0 = acc[c][4 * qq + 2 * h] + bq[2 * h], v1 = acc[c][4 * qq + 2 * h + 1] + bq[2 * h + 1];
                                if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * p.sl1 : v0; v1 = v1 < 0.f ? v1 * p.sl1 : v1; }
                                half2_t hv = {(half_t)v0, (half_t)v1};
                                if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                                d[qq][h] = v_cur ? __builtin_bit_cast(unsigned, hv) : 0u;
                            }
                        }
#pragma unroll
                        for (int jj = 0; jj < 2; jj++) {
                            sm_permlane32_swap(d[2 * jj][0], d[2 * jj + 1][0]);
                            sm_permlane32_swap(d[2 * jj][1], d[2 * jj + 1][1]);
                            if (fp_cur >= 0) *(u32x4*)(pb + (c * 4 + 2 * jj + hi) * BN_PLANE + qpx * 16) = u32x4{d[2 * jj][0], d[2 * jj][1], d[2 * jj + 1][0], d[2 * jj + 1][1]};
                        }
                    }
                    BN_STAMP(q * 5 + 2 + 2 * j);
                    off_cur0 = off_next0; off_cur1 = off_next1; v_cur = v_next; fp_cur = fp_next;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my patch writes (the x loads stay in flight across the barrier)
                __builtin_amdgcn_s_barrier();                              // after period q
            }
        };
        if (p.sl1 == 0.f) rrun(std::true_type()); else rrun(std::false_type());
    }
}

#include "bneck128.hpp"

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

#ifdef SM_BN_TL
extern "C" int sm_bneck_timeline_read(unsigned long long* dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bn_tl), sizeof(unsigned long long) * 256 * 2 * 64) == hipSuccess ? 0 : -1; }
extern "C" int sm_bneck_timeline_clear(void) { static unsigned long long z[256 * 2 * 64]; return hipMemcpyToSymbol(HIP_SYMBOL(g_bn_tl), z, sizeof z) == hipSuccess ? 0 : -1; }
#endif
extern "C" int sm_bottleneck_supported(const sm_bneck_desc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0) return 0;
    if (d->Cm == BN2_CM) {                                             // the 128-channel middle: one band tile per block, three phases (bneck128.hpp)
        if (d->act1 > SM_ACT_PRELU || d->act2 > SM_ACT_PRELU || d->act3 > SM_ACT_PRELU) return 0;
        if (d->x_ld < d->C0 || d->y_ld < d->C3 || (d->x_ld & 7) || (d->y_ld & 7) || (d->res_ld & 7)) return 0;
        const long px2 = (long)d->N * d->H * d->W;
        if (px2 * d->x_ld * 2 >= 0x7fff0000L || px2 * d->y_ld * 2 >= 0x7fff0000L || px2 * (d->res_ld > 0 ? d->res_ld : 1) * 2 >= 0x7fff0000L) return 0;
        if (!bn2_geometry(d, nullptr)) return 0;
        return (long)d->N * sm_ceil_div(d->H, 1) < (1L << 30) ? 1 : 0;
    }
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
    SM_REQUIRE(sm_bottleneck_supported(d), "sm_bottleneck_fwd: no kernel takes this block (Cm = 64: C0 in {64, 128, 256}, C3 <= 256, C3 %% 32 == 0; Cm = 128: C0 %% 64 == 0, C3 %% 128 == 0, 4 <= W <= 62 and a band of rows that fits the LDS; "
               "3x3 / s1 / p1 in the middle, ReLU-family activations, tensors < 2 GB)");
    SM_REQUIRE(!res || d->res_ld >= d->C3, "sm_bottleneck_fwd: residual pixel stride %d < C3 %d", d->res_ld, d->C3);
    SM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)w1) | ((uintptr_t)w2) | ((uintptr_t)w3) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) == 0, "sm_bottleneck_fwd: operands must be 16-byte aligned");
    if (d->Cm == BN2_CM) {
        SM_REQUIRE(!b1 || (((uintptr_t)b1) & 15) == 0, "sm_bottleneck_fwd: biases must be 16-byte aligned");
        SM_REQUIRE((!b2 || (((uintptr_t)b2) & 15) == 0) && (!b3 || (((uintptr_t)b3) & 15) == 0), "sm_bottleneck_fwd: biases must be 16-byte aligned");
        Bn2Params q{};
        q.x = (const half_t*)x; q.w1 = (const half_t*)w1; q.b1 = b1; q.w2 = (const half_t*)w2; q.b2 = b2; q.w3 = (const half_t*)w3; q.b3 = b3; q.res = (const half_t*)res; q.y = (half_t*)y;
        q.N = d->N; q.H = d->H; q.W = d->W; q.C0 = d->C0; q.n3 = d->C3; q.x_ld = d->x_ld; q.y_ld = d->y_ld; q.res_ld = d->res_ld;
        auto sl2 = [](int act, float s) { return act == SM_ACT_NONE ? 1.f : act == SM_ACT_RELU ? 0.f : s; };
        q.sl1 = sl2(d->act1, d->slope1); q.sl2 = sl2(d->act2, d->slope2); q.sl3 = sl2(d->act3, d->slope3);
        bn2_geometry(d, &q);
        q.cap = (res == x && d->res_ld == d->x_ld && d->C3 == d->C0 && (d->C3 % 512) == 0) ? 1 : 0;      // identity block: the first residual passes ride in registers from phase 1 (C3 % 512: a wave's pass p is x stage (C3 / 128) wave + p, in ring slot p)
        const size_t lds = bn2_lds_bytes(q);
        static size_t attr_lds = 0;
        if (lds > attr_lds) { SM_HIP_CHECK(hipFuncSetAttribute((const void*)bneck128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_lds = lds; }
        hipLaunchKernelGGL(bneck128_kernel, dim3((unsigned)(d->N * q.bands)), dim3(256), lds, sm_s(stream), q);
        SM_LAUNCH_CHECK();
        return SM_OK;
    }
    BnParams p = bn_params(d);
    p.x = (const half_t*)x; p.w1 = (const half_t*)w1; p.b1 = b1; p.w2 = (const half_t*)w2; p.b2 = b2; p.w3 = (const half_t*)w3; p.b3 = b3; p.res = (const half_t*)res; p.y = (half_t*)y;
    { static const int dbg = SM_ABL_ENV("SM_BN_DEBUG") ? atoi(SM_ABL_ENV("SM_BN_DEBUG")) : 0; p.dbg = dbg; }   // ablation switch (diagnostic builds only), read once
    void (*fn)(BnParams) = d->C0 == 256 ? bneck_kernel<16> : d->C0 == 128 ? bneck_kernel<8> : bneck_kernel<4>;
    static bool attr_done[3] = {false, false, false};
    const int ai = d->C0 == 256 ? 0 : d->C0 == 128 ? 1 : 2;
    if (!attr_done[ai]) { SM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, BN_LDS)); attr_done[ai] = true; }
    const unsigned grid = (unsigned)(p.ntiles < SM_NUM_CU ? p.ntiles : SM_NUM_CU);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), BN_LDS, sm_s(stream), p);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
