// conv1x1_rw.hpp — pointwise convolution (1x1 / stride 1, MPSCNNConvolutionNode, Converters.swift:253) with a SHORT reduction (Cin = 64 | 128 | 256) whose
// WEIGHTS LIVE IN REGISTERS for the whole launch while the pixels stream through: ResNet-50's expand layers with their residual Add + ReLU
// (128 -> 512 at 28x28, 256 -> 1024 at 14x14; AddConverter :430-451, ReluConverter :342-359) and the 256 -> 64 | 128 reduce layers at 56x56.
// Included by conv_igemm.hip (same ConvParams / launcher / tile-config pool: kind 8).
//
// Why.  These layers have 1..4 K tiles.  In the tiled kernels (conv_igemm64, conv1x1_ps) a block's life is prologue -> a few K tiles -> an LDS-staged
// epilogue that moves more bytes (output + residual) than the main loop did, one phase after the other with one or two blocks per CU: 256 -> 1024 at
// M = 12544 runs 20-25 us for 58 MB of compulsory traffic (2.3-2.9 TB/s) and 6.6 GFLOP (260-330 TFLOP/s); per CU the bytes it pulls through its L1
// (393 KB) would take 7 us at the measured 27 B/clk fill rate.  No resource is the limit: the phases do not overlap.
//
// Here a block owns an N slice (BN = 32 NCW WN output channels) for the whole launch and walks a contiguous range of pixel steps:
//   * every wave keeps its NCW x 32 rows of W as MFMA A fragments in registers (K = 256, NCW = 2: 128 VGPRs), loaded once from global memory: the
//     k loop reads ONE LDS operand (the activation fragment) per NCW MFMAs, and there is no weight traffic after the prologue;
//   * a step = FS fragments of 32 pixels.  Its activation tile [K/64 slabs][FS*32 rows][64 halfs] AND its residual tile [BN/64 slabs][rows][64 halfs]
//     (the BK = 64 layout of the tiled kernels: XOR swizzle on the source side, whole 128-byte lines per group of 8 DMA lanes) are copied by the LDS-DMA
//     into one of two buffer pairs while the previous step is multiplied.  With the residual staged too, the waves issue nothing but LDS-DMA and
//     stores: no wave ever waits for a load in the middle of a step (gfx9-family vmcnt retires loads and stores out of order with respect to each
//     other, so any wait between them would have to be vmcnt(0)), only ONE `s_waitcnt vmcnt(0)` + barrier per step;
//   * the 8 waves are WN along N x WF = 8/WN along the fragments; per fragment KS x NCW MFMAs with hand-counted fragment prefetch (conv3x3_ws.hpp's
//     pattern), then the residual rows — read from LDS in the 8-channels-per-lane layout they are stored in, which is the MFMA B layout — enter
//     through two one-hot MFMAs (conv3x3_wsx.hpp), + bias, activation, one rounding, lane swaps, 16-byte stores through a buffer descriptor
//     (pixels beyond M: offset beyond num_records): a pixel's NCW*64 contiguous bytes come from one wave.
// Numerics = the tiled kernels': fp32 accumulation over k ascending from zero, (acc + residual) + bias in fp32, activation, one rounding.
#pragma once

constexpr int RW_DUMMY = 1024;
// dynamic LDS: two activation tiles, two residual tiles, the slice's bias, a DMA sink
constexpr int rw_lds_bytes(int ks, int bn, int fs, bool res) { return 2 * (ks / 4) * fs * 32 * 128 + (res ? 2 * (bn / 64) * fs * 32 * 128 : 0) + bn * 4 + RW_DUMMY; }

template <int KS, int NCW, int WN, int FS, bool RES>
__global__ __launch_bounds__(512) void conv1x1_rw_kernel(ConvParams p) {
    constexpr int NSLAB = KS / 4, ROWS = FS * 32, WF = 8 / WN, FPW = FS / WF, BN = WN * NCW * 32, K = KS * 16, RSLAB = BN / 64;
    static_assert(KS % 4 == 0 && WN * WF == 8 && FPW * WF == FS && (ROWS % 64) == 0 && (BN % 64) == 0, "geometry");
    constexpr int SLAB_BYTES = ROWS * 128, XT = NSLAB * SLAB_BYTES, RT = RES ? RSLAB * SLAB_BYTES : 0, RB = ROWS / 64;
    constexpr int NPX = NSLAB * RB, NPR = RES ? RSLAB * RB : 0;       // DMA instructions per thread and tile: activations, residual
    static_assert((NSLAB - 1) * SLAB_BYTES < 65536 && (RSLAB - 1) * SLAB_BYTES < 65536, "slab offset = ds_read offset field");
    extern __shared__ __attribute__((aligned(16))) unsigned char rw_smem[];
    unsigned char* const xt0 = rw_smem;                               // [buffer][slab][row][8 chunks x 16 B]
    unsigned char* const rt0 = rw_smem + 2 * XT;
    float* const bl = (float*)(rw_smem + 2 * XT + 2 * RT);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wf = wave / WN;
    const half_t* zero = (const half_t*)g_zero_page;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // ---- block -> (N slice, pixel-step range): the slices of one range sit on the same XCD (block b runs on XCD b % 8), back to back ----
    const int nsl = p.nt, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % nsl, rng = (jb / nsl) * 8 + xcd, R = (int)(gridDim.x >> 3) / nsl * 8;
    const int S = p.mt;                                               // steps of ROWS pixels
    const int t_begin = (int)((long)rng * S / R), t_end = (int)((long)(rng + 1) * S / R);
    if (t_begin >= t_end) return;
    SM_TL(0);
    int tl_i = 1; (void)tl_i;
    const int n0 = slice * BN + wn * NCW * 32;                        // this wave's first output channel

    // ---- per-thread DMA source: piece i = (slab i / RB, row block i % RB) covers rows rb*64 + (tid >> 3), chunk slot tid & 7 <- global chunk slot ^ key(row) ----
    const int drow = tid >> 3, dkey = (drow >> 1) & 7;               // (row + 64: same key)
    const int dchk = ((tid & 7) ^ dkey) << 3;
    const long relx = (long)drow * p.x_ld + dchk, relr = (long)drow * p.res_ld + slice * BN + dchk;
    // ---- weights: global -> LDS by the LDS-DMA in whole 128-byte lines (the tile buffers are not in use yet; same slab layout with BN rows), then each wave
    // lifts its NCW x 32 rows into registers as MFMA A fragments (row = lane & 31 of chunk c, k = 16 ks + 8 hi ..).  Direct 16-byte global loads in the
    // fragment layout touch 32 lines per instruction and re-fetch every line four times through a 32 KB L1 that eight waves thrash: 19 k clk of prologue
    // for a 128 KB slice (timeline), against ~5 k for the DMA ----
    constexpr int WB = BN / 64, NPW = NSLAB * WB;                    // DMA instructions per thread for the slice
    static_assert(NSLAB * BN * 128 <= 2 * XT + 2 * RT, "the weight slice passes through the tile buffers");
    {
        const half_t* wsrc = p.w + ((long)slice * BN + drow) * K + dchk;
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int slab = i / WB, rb = i % WB;
            SM_GLDS16(wsrc + (long)rb * 64 * K + slab * 64, rw_smem + (i * 512 + wave * 64) * 16);
        }
    }
    for (int i = tid; i < BN; i += 512) bl[i] = p.bias ? p.bias[slice * BN + i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    half8_t wr[NCW][KS];
#pragma unroll
    for (int c = 0; c < NCW; c++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int row = wn * NCW * 32 + c * 32 + l31;
            wr[c][ks] = *(const half8_t*)(rw_smem + (ks >> 2) * (BN * 128) + row * 128 + (((2 * (ks & 3) + hi) ^ ((row >> 1) & 7)) << 4));
        }
    __syncthreads();                                                  // everybody holds its fragments: the buffers may be overwritten
    {   // first tiles: ordinary (compiler-visible) DMA statements, waited for at the top of the first step
        const long m0 = (long)t_begin * ROWS;
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            const int slab = i / RB, rb = i % RB;
            const bool in = m0 + rb * 64 + drow < p.M;
            SM_GLDS16_AUX(in ? p.x + (m0 + rb * 64) * p.x_ld + relx + slab * 64 : zero, xt0 + (i * 512 + wave * 64) * 16, SM_RW_X_NT);
        }
#pragma unroll
        for (int i = 0; i < NPR; i++) {
            const int slab = i / RB, rb = i % RB;
            const bool in = m0 + rb * 64 + drow < p.M;
            SM_GLDS16_AUX(in ? p.res + (m0 + rb * 64) * p.res_ld + relr + slab * 64 : zero, rt0 + (i * 512 + wave * 64) * 16, SM_RW_X_NT);
        }
    }

    // ---- fragment addresses (buffer 0; toggled by +-XT / +-RT after every step): fragment j of this wave = tile rows (wf*FPW + j)*32 + l31;
    // activations: k step ks -> slab ks / 4 (instruction offset), chunk 2 (ks % 4) + hi;  residual: channel 32c + 16q + 8hi of this wave's NCW*32 -> slab, chunk ----
    unsigned aoff[FPW][4], roff[FPW][RES ? NCW * 2 : 1];
#pragma unroll
    for (int j = 0; j < FPW; j++) {
        const int row = (wf * FPW + j) * 32 + l31, key = (row >> 1) & 7;
#pragma unroll
        for (int q = 0; q < 4; q++) aoff[j][q] = ws_lds_addr(xt0) + (unsigned)(row * 128 + (((2 * q + hi) ^ key) << 4));
        if constexpr (RES) {
#pragma unroll
            for (int cq = 0; cq < NCW * 2; cq++) {
                const int ch = wn * NCW * 32 + cq * 16 + 8 * hi;       // channel inside the slice
                roff[j][cq] = ws_lds_addr(rt0) + (unsigned)((ch >> 6) * SLAB_BYTES + row * 128 + ((((ch & 63) >> 3) ^ key) << 4));
            }
        }
    }
    // ---- the output stream goes through a buffer descriptor: a dead lane (pixel beyond M) carries an offset beyond num_records ----
    constexpr unsigned DEAD = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((unsigned)p.M * (unsigned)p.y_ld * 2u), 0x00020000);
    half8_t onehot[2];                                                // E_q[r][k] = (r == 16 q + k): mfma(E_q, R_q, acc) adds residual rows that arrive in the store layout
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 8; i++) onehot[q][i] = (half_t)((RES && l31 == 16 * q + 8 * hi + i) ? 1.f : 0.f);
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    const bool relu0 = p.act != SM_ACT_NONE && p.slope == 0.f;

    constexpr int PF = 4;                                             // k steps of fragment prefetch
    auto do_step = [&](auto bufc, int t) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        // the NEXT tiles (the last step re-fetches itself: no branch in the k loop)
        const long m0n = (long)(t + 1 < t_end ? t + 1 : t) * ROWS;
        const unsigned dst_x = ws_lds_addr(xt0) + (BUF ^ 1) * XT + wave * 1024, dst_r = ws_lds_addr(rt0) + (BUF ^ 1) * RT + wave * 1024;
        auto piece = [&](int i) __attribute__((always_inline)) {     // i < NPX: activations, else residual
            const bool isx = i < NPX;
            const int ii = isx ? i : i - NPX, slab = ii / RB, rb = ii % RB;
            const bool in = m0n + rb * 64 + drow < p.M;
            const half_t* src = isx ? p.x + (m0n + rb * 64) * p.x_ld + relx + slab * 64 : p.res + (m0n + rb * 64) * p.res_ld + relr + slab * 64;
            ws_glds16(in ? src : zero, (unsigned)__builtin_amdgcn_readfirstlane((int)((isx ? dst_x : dst_r) + (unsigned)(ii * 512 * 16))));
        };
        sm_static_for<0, FPW>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            floatx16 acc[NCW];
#pragma unroll
            for (int c = 0; c < NCW; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
            half8_t fx[PF + 1];
            auto fetch = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S2 = decltype(sc)::value, SLOT = S2 % (PF + 1);
                half8_t& d = fx[SLOT]; const unsigned ad = aoff[J][S2 & 3];
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"((S2 >> 2) * SLAB_BYTES));
            };
            sm_static_for<0, PF>([&](auto sc) { fetch(sc); });
            u32x4 rr[RES ? NCW * 2 : 1];
            sm_static_for<0, KS>([&](auto sc) {
                constexpr int S2 = decltype(sc)::value;
                // the next tiles' DMA instructions ride between the k steps of the first half of the step
#pragma unroll
                for (int i = 0; i < NPX + NPR; i++) if (i * (FPW * KS) / (2 * (NPX + NPR)) == J * KS + S2) piece(i);
                if constexpr (S2 + PF < KS) fetch(std::integral_constant<int, S2 + PF>());
                if constexpr (RES && S2 == KS - 1) {                  // the residual fragments ride behind the last activation fragment
#pragma unroll
                    for (int cq = 0; cq < NCW * 2; cq++) { u32x4& d = rr[cq]; const unsigned ad = roff[J][cq]; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad)); }
                }
                constexpr int AHEAD = (KS - 1 - S2 < PF ? KS - 1 - S2 : PF) + ((RES && S2 == KS - 1) ? NCW * 2 : 0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < NCW; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[c][S2], fx[S2 % (PF + 1)], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (RES) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int c = 0; c < NCW; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(onehot[q], __builtin_bit_cast(half8_t, rr[c * 2 + q]), acc[c], 0, 0, 0);
            }
            // epilogue in registers: register 4q+r of acc[c] <-> channel n0 + 32c + 8q + 4hi + r of pixel lane & 31
            const long m = (long)t * ROWS + (wf * FPW + J) * 32 + l31;
            const unsigned offy = m < p.M ? (unsigned)((m * p.y_ld + n0 + 8 * hi) * 2) : DEAD;
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const floatx4 bq = *(const floatx4*)(bl + wn * NCW * 32 + c * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        float v0 = acc[c][4 * q + 2 * h2] + bq[2 * h2], v1 = acc[c][4 * q + 2 * h2 + 1] + bq[2 * h2 + 1];
                        if (!relu0) { v0 = v0 < 0.f ? v0 * sl : v0; v1 = v1 < 0.f ? v1 * sl : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if (relu0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});      // max commutes with the rounding
                        d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    sm_permlane32_swap(d[2 * q][0], d[2 * q + 1][0]);
                    sm_permlane32_swap(d[2 * q][1], d[2 * q + 1][1]);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{d[2 * q][0], d[2 * q][1], d[2 * q + 1][0], d[2 * q + 1][1]}, rs_y, (int)(offy + (unsigned)(c * 64 + 32 * q)), 0, SM_ST_AUX);
                }
            }
        });
        // the next step reads the other buffer pair
#pragma unroll
        for (int j = 0; j < FPW; j++) {
#pragma unroll
            for (int q = 0; q < 4; q++) aoff[j][q] += BUF ? (unsigned)-XT : (unsigned)XT;
            if constexpr (RES) {
#pragma unroll
                for (int cq = 0; cq < NCW * 2; cq++) roff[j][cq] += BUF ? (unsigned)-RT : (unsigned)RT;
            }
        }
    };
#pragma unroll 1
    for (int t = t_begin; t < t_end; t += 2) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my share of the tiles of step t, my stores of step t-1, [first step: weights, bias]
        __builtin_amdgcn_s_barrier();                                 // everybody's share landed; everybody is done reading the other buffer pair
        __builtin_amdgcn_sched_barrier(0);
        if (tl_i < 12) { SM_TL(tl_i); tl_i++; }
        do_step(std::integral_constant<int, 0>(), t);
        if (t + 1 >= t_end) break;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (tl_i < 12) { SM_TL(tl_i); tl_i++; }
        do_step(std::integral_constant<int, 1>(), t + 1);
    }
    if (tl_i < 12) { SM_TL(tl_i); tl_i++; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tl_i < 12) { SM_TL(tl_i); tl_i++; }
    SM_TL_END();
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// conv1x1_rwl_kernel — the same operand placement (weights in registers, activation + residual tiles through LDS) with the two jobs on DIFFERENT
// WAVES and a deeper ring.  What the timeline of the kernel above showed (256 -> 1024 + residual at 14x14, bs 64): a step takes ~6.5 k clk for 64 KB
// of LDS-DMA, i.e. one memory round trip under load per step — the `vmcnt(0)` in front of the barrier (the waves also store, so a counted wait is not
// available to them) drains everything in flight every step, and two buffers hold one step of lookahead.  Here
//   * 4 LOADER waves issue every LDS-DMA (loads only: `vmcnt(N)` is exact) into a ring of NST stages, NST-1 stages ahead: ~100 KB in flight per CU;
//   * 8 CONSUMER waves each own 32 output channels of the block's 256 (their W rows: KS A fragments = 64 registers at K = 256) for ALL fragments of a
//     stage; they read fragments, issue MFMAs, add the residual (one-hot MFMAs on the LDS tile), store — and never wait for a load;
//   * one s_barrier per stage for both roles (stage t landed; the slot of stage t-1 is free): conv_igemm64's ring protocol;
//   * a stage is FS fragments of 32 pixels (FS = 1 at K = 256: 16 KB of activations + 16 KB of residual), so pixel ranges are dealt in 32-pixel units
//     (392 of them over 64 ranges per slice at 14x14: 6 or 7 per block instead of 3 or 4 steps of 64).
//
// WP > 1: the weight slice passes the LDS in WP parts (K = 384: 192 KB for 256 rows; each part = NSLAB / WP slabs, lifted between its own pair of barriers).
// DUAL (sm_conv2d_dual_fwd_cfg): K slabs >= p.kt_sw of a stage come from the SECOND source p.x2, pixel (n, oy * x2_s, ox * x2_s) of
// [N, x2_H, x2_W, K2] — the projection shortcut of a bottleneck's first block as the tail of the expand layer's reduction.  Only the loaders know.
// NCONS = consumer waves (x 32 output channels = the block's slice): 8, or 4 for K = 512 — 128 W registers per consumer fit the 256-register budget of an
// 8-wave block (4 + 4), not the 168 of twelve waves: ResNet-50's 512 -> 128 reduce layers at 28 x 28 (N = 128 is one slice).
template <int KS, int FS, int NST, bool RES, int WP = 1, bool DUAL = false, int NCONS = 8>
__global__ __launch_bounds__((NCONS + 4) * 64) void conv1x1_rwl_kernel(ConvParams p) {
    constexpr int NLT = 256, NSLAB = KS / 4, ROWS = FS * 32, BN = NCONS * 32, K = KS * 16, RSLAB = RES ? BN / 64 : 0;
    constexpr int WSLAB = NSLAB / WP, KSP = KS / WP;                  // slabs / k steps of one weight part
    static_assert(NSLAB % WP == 0 && (!DUAL || FS == 1), "weight parts are whole slabs; the second source is addressed per 32-pixel stage");
    constexpr int SLAB_BYTES = ROWS * 128, XT = NSLAB * SLAB_BYTES, STAGE = (NSLAB + RSLAB) * SLAB_BYTES;
    constexpr int LOADS = (NSLAB + RSLAB) * ROWS * 8 / NLT;          // DMA instructions per loader thread and stage
    static_assert(KS % 4 == 0 && ((NSLAB + RSLAB) * ROWS * 8) % NLT == 0 && NST >= 2 && NST <= 4 && (NST - 1) * LOADS <= 63, "geometry / vmcnt range");
    // LDS: the weight slice passes through [0, WB) on its way into the consumers' registers.  E ring slots lie BEHIND that area, so their stages are in flight
    // while the slice lands and is lifted (the first stage used to start its ~4 k clk round trip only after that); the other NST - E slots reuse the slice's area.
    constexpr int WB = WSLAB * BN * 128, E_FIT = (160 * 1024 - WB) / STAGE, E = E_FIT < NST - 1 ? E_FIT : NST - 1;
    constexpr int LDS_TOTAL = (WB > (NST - E) * STAGE ? WB : (NST - E) * STAGE) + E * STAGE;
    static_assert(LDS_TOTAL <= 160 * 1024 && (NST - E) * STAGE <= 160 * 1024, "LDS");
    static_assert((NSLAB - 1) * SLAB_BYTES + (FS - 1) * 4096 < 65536, "ds_read offset field (slab + fragment; the slot and the residual's slab are in the address register)");
    static_assert(E * LOADS <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) unsigned char rw_smem[];
    auto slot_base = [](int s) { return s < E ? WB + s * STAGE : (s - E) * STAGE; };     // byte offset of ring slot s

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= NCONS;
    const half_t* zero = (const half_t*)g_zero_page;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int nsl = p.nt, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % nsl, rng = (jb / nsl) * 8 + xcd, R = (int)(gridDim.x >> 3) / nsl * 8;
    const int S = p.mt;                                               // stages of ROWS pixels
    const int t_begin = (int)((long)rng * S / R), t_end = (int)((long)(rng + 1) * S / R);
    if (t_begin >= t_end) return;
    SM_TL(0);
    const int n_st = t_end - t_begin;

    if (is_loader) {
        // =================================================== loader waves =====================================================
        const int ltid = tid - NCONS * 64, lwave = wave - NCONS;
        const int drow = ltid >> 3, dchunk = ltid & 7;               // a DMA instruction of the four waves covers 32 rows x 8 chunks of one slab
        // ---- weight slice (part wp) -> ring area (slab layout with BN rows) ----
        auto weights = [&](auto wpc) __attribute__((always_inline)) {
            constexpr int WPI = decltype(wpc)::value, NPW = WSLAB * BN * 8 / NLT;
#pragma unroll
            for (int i = 0; i < NPW; i++) {
                const int g = i * 32 + drow, slab = g / BN, row = g % BN;      // (i: compile time, BN a power of two)
                SM_GLDS16(p.w + ((long)slice * BN + row) * K + (WPI * WSLAB + slab) * 64 + ((dchunk ^ ((row >> 1) & 7)) << 3), rw_smem + (i * NLT + lwave * 64) * 16);
            }
        };
        weights(std::integral_constant<int, 0>());
        int issued = 0;
        auto issue = [&]() __attribute__((always_inline)) {          // stage `issued` -> slot issued % NST
            const long m0 = (long)(t_begin + issued) * ROWS;
            unsigned char* sa = rw_smem + slot_base(issued % NST) + lwave * 1024;
            const half_t* row2 = nullptr;                             // DUAL: this thread's pixel of the second source (FS = 1: one row per thread and stage)
            if constexpr (DUAL) {
                long m = m0 + drow; if (m >= p.M) m = p.M - 1;
                const int n = (int)sm_fastdiv((unsigned)m, p.dhw_mul, p.dhw_sh), r = (int)m - n * (p.Ho * p.Wo);
                const int oy = (int)sm_fastdiv((unsigned)r, p.dw_mul, p.dw_sh), ox = r - oy * p.Wo;
                row2 = p.x2 + (long)((n * p.x2_H + oy * p.x2_s) * p.x2_W + ox * p.x2_s) * p.x2_ld;
            }
#pragma unroll
            for (int i = 0; i < LOADS; i++) {
                const int g = i * 32 + drow;                          // row-slab index inside the stage: x slabs first, then the residual's
                const int slab = g / ROWS, row = g % ROWS;
                const bool in = m0 + row < p.M;
                const int sw = (dchunk ^ ((row >> 1) & 7)) << 3;
                const half_t* src = slab < NSLAB ? p.x + (m0 + row) * p.x_ld + slab * 64 + sw
                                                 : p.res + (m0 + row) * p.res_ld + slice * BN + (slab - NSLAB) * 64 + sw;
                if constexpr (DUAL) { if (slab >= p.kt_sw) src = row2 + (slab - p.kt_sw) * 64 + sw; }      // (slab = i: compile time; kt_sw uniform)
                SM_GLDS16_AUX(in ? src : zero, sa + i * NLT * 16, SM_RW_X_NT);
            }
            ++issued;
        };
        // early stages (slots behind the slice's area) ride behind the weight loads; the slice is complete when only they are left in flight
#pragma unroll
        for (int s = 0; s < E; s++) if (issued < n_st) issue();
        if (issued == E) sm_wait_vmcnt<E * LOADS>(); else sm_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                 // P1: the slice (its first part) is in LDS
        __builtin_amdgcn_s_barrier();                                 // P2: every consumer holds its fragments: the slice's area may be overwritten
        sm_static_for<1, WP>([&](auto wpc) {                          // the other parts, one round trip each (the early stages are long in by then)
            weights(wpc);
            sm_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        });
#pragma unroll
        for (int s = E; s < NST - 1; s++) if (issued < n_st) issue();
#pragma unroll 1
        for (int i = 0; i < n_st; i++) {
            const int newer = issued - i - 1;                         // stages behind stage i that may stay in flight
            if (NST >= 4 && newer >= 3) sm_wait_vmcnt<(NST >= 4 ? 3 : 0) * LOADS>();
            else if (NST >= 3 && newer == 2) sm_wait_vmcnt<(NST >= 3 ? 2 : 0) * LOADS>();
            else if (newer == 1) sm_wait_vmcnt<LOADS>();
            else sm_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                             // stage i landed (everybody's share); the slot of stage i-1 is free
            __builtin_amdgcn_sched_barrier(0);
            if (issued < n_st) issue();
        }
        return;
    }

    // ===================================================== consumer waves ======================================================
    const int n0 = slice * BN + wave * 32;                            // this wave's 32 output channels
    floatx4 bq[4];                                                    // this lane's 16 bias values (channels n0 + 8q + 4hi ..): straight from global, once
#pragma unroll
    for (int q = 0; q < 4; q++) bq[q] = p.bias ? *(const floatx4*)(p.bias + n0 + 8 * q + 4 * hi) : floatx4{0.f, 0.f, 0.f, 0.f};
    half8_t wr[KS];
    sm_static_for<0, WP>([&](auto wpc) {
        constexpr int WPI = decltype(wpc)::value;
        __builtin_amdgcn_s_barrier();                                 // P1 (of part WPI)
#pragma unroll
        for (int ks = 0; ks < KSP; ks++) {
            const int row = wave * 32 + l31;
            wr[WPI * KSP + ks] = *(const half8_t*)(rw_smem + (ks >> 2) * (BN * 128) + row * 128 + (((2 * (ks & 3) + hi) ^ ((row >> 1) & 7)) << 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // P2
    });
    SM_TL(1);
    // fragment addresses inside a stage: fragment j = rows 32 j + l31; activations: k step ks -> slab ks / 4 (instruction offset), chunk 2 (ks % 4) + hi;
    // residual: channels 32 wave + 16 q + 8 hi -> slab wave / 2, chunk 4 (wave & 1) + 2 q + hi
    const int key = (l31 >> 1) & 7;
    unsigned aoff[4], roff[2];
#pragma unroll
    for (int q = 0; q < 4; q++) aoff[q] = ws_lds_addr(rw_smem) + (unsigned)(l31 * 128 + (((2 * q + hi) ^ key) << 4));
#pragma unroll
    for (int q = 0; q < 2; q++) roff[q] = ws_lds_addr(rw_smem) + (unsigned)(XT + (wave >> 1) * SLAB_BYTES + l31 * 128 + (((4 * (wave & 1) + 2 * q + hi) ^ key) << 4));
    constexpr unsigned DEAD = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((unsigned)p.M * (unsigned)p.y_ld * 2u), 0x00020000);
    half8_t onehot[2];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 8; i++) onehot[q][i] = (half_t)((RES && l31 == 16 * q + 8 * hi + i) ? 1.f : 0.f);
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    const bool relu0 = p.act != SM_ACT_NONE && p.slope == 0.f;

    constexpr int PF = KS >= 48 ? 2 : 4;                             // (K = 768: 192 W registers leave room for two k steps of prefetch)
    int slot = 0;
#pragma unroll 1
    for (int i = 0; i < n_st; i++) {
        __builtin_amdgcn_s_barrier();                                 // stage i is in slot i % NST
        __builtin_amdgcn_sched_barrier(0);
#ifdef SM_TIMELINE
        if (i < 10) SM_TL(2 + i);
#endif
        const unsigned sb = (unsigned)slot_base(slot);
        const long mb = (long)(t_begin + i) * ROWS;
        sm_static_for<0, FS>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            half8_t fx[PF + 1];
            auto fetch = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S2 = decltype(sc)::value, SLOT = S2 % (PF + 1);
                half8_t& d = fx[SLOT]; const unsigned ad = aoff[S2 & 3] + sb;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"((S2 >> 2) * SLAB_BYTES + J * 4096));
            };
            sm_static_for<0, PF>([&](auto sc) { fetch(sc); });
            u32x4 rr[2];
            sm_static_for<0, KS>([&](auto sc) {
                constexpr int S2 = decltype(sc)::value;
                if constexpr (S2 + PF < KS) fetch(std::integral_constant<int, S2 + PF>());
                if constexpr (RES && S2 == KS - 1) {
#pragma unroll
                    for (int q = 0; q < 2; q++) { u32x4& d = rr[q]; const unsigned ad = roff[q] + sb; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(J * 4096)); }
                }
                constexpr int AHEAD = (KS - 1 - S2 < PF ? KS - 1 - S2 : PF) + ((RES && S2 == KS - 1) ? 2 : 0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[S2], fx[S2 % (PF + 1)], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (RES) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 2; q++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(onehot[q], __builtin_bit_cast(half8_t, rr[q]), acc, 0, 0, 0);
            }
            const long m = mb + J * 32 + l31;
            const unsigned offy = m < p.M ? (unsigned)((m * p.y_ld + n0 + 8 * hi) * 2) : DEAD;
            unsigned d[4][2];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    float v0 = acc[4 * q + 2 * h2] + bq[q][2 * h2], v1 = acc[4 * q + 2 * h2 + 1] + bq[q][2 * h2 + 1];
                    if (!relu0) { v0 = v0 < 0.f ? v0 * sl : v0; v1 = v1 < 0.f ? v1 * sl : v1; }
                    half2_t hv = {(half_t)v0, (half_t)v1};
                    if (relu0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                    d[q][h2] = __builtin_bit_cast(unsigned, hv);
                }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                sm_permlane32_swap(d[2 * q][0], d[2 * q + 1][0]);
                sm_permlane32_swap(d[2 * q][1], d[2 * q + 1][1]);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{d[2 * q][0], d[2 * q][1], d[2 * q + 1][0], d[2 * q + 1][1]}, rs_y, (int)(offy + (unsigned)(32 * q)), 0, SM_ST_AUX);
            }
        });
        if (++slot == NST) slot = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's fragment reads are done before it passes the next barrier
    }
#ifdef SM_TIMELINE
    if (n_st < 10) SM_TL(2 + n_st);
#endif
    SM_TL_END();
}
