// conv_igemm64.hpp — the fast path of the implicit-GEMM convolution: Cin % 64 == 0 (every ResNet-50 layer but the
// stem), BK = 64, operands streamed HBM/L2 -> LDS by the LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane) with no VGPR
// round trip, epilogue transposed through LDS so that every global access is a full 16-byte lane / 128-byte line.
// Included by conv_igemm.hip (same ConvParams, same launcher).
//
//  * K tiling: with Cin % 64 == 0 a 64-wide K tile lies inside ONE (kh,kw) tap, so the tap and its channel base are
//    wave-uniform per tile; a row of the activation tile is 128 contiguous bytes of one input pixel (or zeros).
//    The per-tile bookkeeping is a handful of SALU adds (tap / channel-block counters advanced incrementally); whether a
//    row's tap lies inside the image is ONE bit of a per-row tap mask built once in the prologue (kH*kW <= 32).
//  * LDS-DMA writes lane-linear (wave base + lane*16 B), so the bank swizzle is applied on the SOURCE side: the lane
//    that fills 16-byte slot c' of row r fetches chunk c = c' ^ ((r >> 1) & 7); readers apply the same XOR
//    (cdna_hip_programming.md §5.4 rule 21).  With 128-byte rows this makes every 16-lane service group of
//    ds_read_b128 hit 16 distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md §LDS).
//  * Zero padding: out-of-image taps fetch from a zero page.  Ragged tiles (rows >= M, couts >= Cout) are CLAMPED to the
//    last valid row instead: a GEMM row/column only feeds its own outputs, and those are never stored.
//  * Pipeline: NSTAGE-deep LDS ring, ONE raw s_barrier per K tile, counted `s_waitcnt vmcnt(N)` so that NSTAGE-1 tiles
//    stay in flight across the barrier (cdna_hip_programming.md §5 "Pipelining across barriers").  Inside a tile the
//    fragment reads of k-step s+1 are requested before the MFMAs of k-step s.
//  * Epilogue: the accumulators start at the bias (one value per accumulator register: registers <-> couts), so the
//    epilogue is accumulators -> LDS as fp32 [rows][BN+4] (conflict-free ds_write_b128), then every thread takes 8
//    consecutive couts of one pixel: + residual (16-byte load issued BEFORE the main loop), activation, ONE rounding to
//    fp16, 16-byte coalesced store.  All waves stage at once when the tile fits 68 KB of LDS (two blocks per CU stay
//    resident); 256-row tiles take two passes.
#pragma once
#include <type_traits>

__device__ __attribute__((aligned(64))) unsigned int g_zero_page[64];   // zero-initialised at module load

#define SM_GLDS16(gptr, ldsptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)
// experiment switches (make EXTRA=-DSM_A_NT=2 / -DSM_B_NT=2): cache policy of the activation / weight LDS-DMA (aux 2 = nt)
#ifndef SM_A_NT
#define SM_A_NT 0
#endif
#ifndef SM_B_NT
#define SM_B_NT 0
#endif
// ... per kernel family, for the ACTIVATION streams only (round 6: policies are ranked IN THE NETWORK — a microbenchmark that loops one layer keeps its tensors in the Infinity Cache and
// ranks them differently): SM_RW_X_NT (conv1x1_rw.hpp: x and residual tiles), SM_PS_X_NT (conv1x1_ps.hpp: A rows), SM_WS_X_NT (conv3x3_ws.hpp / conv3x3_wsx.hpp: patches)
#ifndef SM_RW_X_NT
#define SM_RW_X_NT 0
#endif
#ifndef SM_PS_X_NT
#define SM_PS_X_NT 0
#endif
#ifndef SM_WS_X_NT
#define SM_WS_X_NT 0
#endif
#define SM_GLDS16_AUX(gptr, ldsptr, AUX)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, AUX)

#ifdef SM_TIMELINE   // diagnostic build (make EXTRA="-DSM_ABLATION -DSM_TIMELINE"): per-block phase timestamps
// slot 0: wall clock at start, slots 1..13: clock64() at SM_TL(0..12), slot 15: wall clock at the end (SM_TL_END)
__device__ unsigned long long g_tl[8192 * 16];
#define SM_TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { if ((i) == 0) g_tl[blockIdx.x * 16] = wall_clock64(); \
    g_tl[blockIdx.x * 16 + 1 + (i)] = clock64(); } } while (0)
#define SM_TL_END() do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_tl[blockIdx.x * 16 + 15] = wall_clock64(); } while (0)
#else
#define SM_TL(i)
#define SM_TL_END()
#endif


constexpr int sm_pow2_floor(int v) { int r = 1; while (r * 2 <= v) r *= 2; return r; }

// ABL: ablation switch for tools/conv_bench.py (0 = product; 1 = no MFMA; 2 = no global->LDS traffic after the prologue)
// NLW > 0: wave specialisation.  A VMEM instruction stalls the issuing wave while the address queue is full, and one K tile
// is 32..48 KB of LDS-DMA: with every wave issuing its share, the MFMA waves spend about as long stalled on the DMA issue
// as they spend on MFMAs (ablations: LDS->MFMA only 833 clk per 128x128x64 tile, + DMA issue 1300..1450 clk).  With NLW
// extra LOADER waves (one per SIMD) that issue all LDS-DMA of the block and never touch the matrix pipe, the WGM*WGN
// CONSUMER waves only read fragments and issue MFMAs; both roles meet at the ONE barrier per K tile of the ring protocol.
// KS = 2: K split INSIDE the block.  The waves form KS groups of WGM x WGN; a ring stage holds KS consecutive K tiles and group g
// multiplies tile KS*i + g of stage i, so each wave keeps the big 64x64 wave tile of a 4-wave block (fragment reads per
// MFMA as in the 4-wave configs: LDS time 768 instead of 896 clk per 128x128x64 tile by the DESIGN 4.1b model) while 8 waves
// hide each other's LDS latency, and there is ONE barrier per KS tiles.  The KS partial accumulators meet in the fp32 staging
// buffers of the epilogue (one region per group, summed by the store pass in a fixed order).
// KT = K tiles per ring stage (default KS).  KS = 1, KT = 2: every wave multiplies both tiles of a stage — BK = 128 per barrier: half
// the barriers / counted waits per K tile for the one-block-per-CU layers, at twice the stage size.
// LATE_RES: the residual operand is requested by the epilogue pass that adds it instead of before the main loop — for the 256x256 tile,
// whose prefetched residual would be 64 registers per thread beside 128 accumulators.
template <int BM, int BN, int WGM, int WGN, int NSTAGE = 2, int ABL = 0, int ILV = 0, int NLW = 0, int KS = 1, int KT = KS, bool LATE_RES = false>
__global__ __launch_bounds__((WGM * WGN * KS + NLW) * 64) void conv_igemm64_kernel(ConvParams p) {
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 MFMA waves per group");
    static_assert(NSTAGE >= 1 && NSTAGE <= 4, "1..4 stages");
    static_assert(NLW == 0 || (ILV == 0 && NSTAGE >= 2), "loader waves: no interleaved issue, >= 2 stages");
    static_assert(KS == 1 || (KS == 2 && NLW == 0 && ILV == 0 && ABL == 0 && NSTAGE >= 2), "K groups: plain issue only");
    static_assert(KT == KS || (KS == 1 && KT == 2 && ILV == 0 && ABL == 0 && NSTAGE >= 2), "two tiles per stage: plain issue only");
    constexpr int NGW = WGM * WGN;                              // waves per K group
    constexpr int NCW = NGW * KS;                               // consumer (MFMA) waves
    constexpr int NT = (NCW + NLW) * 64;                        // threads per block
    constexpr int NLT = NLW ? NLW * 64 : NT;                    // threads that issue the LDS-DMA
    constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
    constexpr int A_IT = BM * 8 / NLT, B_IT = BN * 8 / NLT;    // 16-byte pieces per loader thread per K tile
    constexpr int RPI = NLT / 8;                                // tile rows covered by one DMA instruction of all loader waves
    static_assert(A_IT >= 1 && B_IT >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave layout");
    constexpr int LOADS = KT * (A_IT + B_IT);                   // LDS-DMA instructions per wave per stage
    constexpr int A_BYTES = BM * 128, SUB_BYTES = (BM + BN) * 128, STAGE_BYTES = KT * SUB_BYTES, PIPE_BYTES = NSTAGE * STAGE_BYTES;
    // epilogue staging (fp32): WPP wave-rows of 32*TM pixels per pass, one region per K group
    constexpr int EP_LD = BN + 4, WROWS = 32 * TM;
    constexpr int EP_MIN = KS == 2 ? 139264 : 69632;
    constexpr int EP_BUDGET = PIPE_BYTES > EP_MIN ? PIPE_BYTES : EP_MIN;
    constexpr int WPP_MAX = EP_BUDGET / (KS * WROWS * EP_LD * 4);
    constexpr int WPP = WPP_MAX >= WGM ? WGM : sm_pow2_floor(WPP_MAX < 1 ? 1 : WPP_MAX);
    constexpr int EP_ROWS = WROWS * WPP, EP_PASSES = WGM / WPP;
    static_assert(WGM % WPP == 0, "epilogue passes");
    // + 8 bytes per tile row at the very end: the row table of the prologue (never touched by the ring or the staging)
    constexpr int EP_REGION = EP_ROWS * EP_LD;                  // floats per K group
    // TABLE_ALIAS: the row table of the prologue lives in the tail of the ring (an extra barrier separates its last reader from the
    // first DMA) instead of behind it, when those BM*8 bytes decide whether a second block fits the CU (256x64, two stages: 80 KB)
    constexpr int RING_OR_EP = PIPE_BYTES > KS * EP_REGION * 4 ? PIPE_BYTES : KS * EP_REGION * 4;
    constexpr bool TABLE_ALIAS = RING_OR_EP <= 80 * 1024 && RING_OR_EP + BM * 8 > 80 * 1024;
    constexpr int LDS_BYTES = RING_OR_EP + (TABLE_ALIAS ? 0 : BM * 8);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert((NSTAGE - 1) * LOADS <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[LDS_BYTES];

    SM_TL(0);
    int mtile, ntile, ksplit;
    if (!map_block(p, mtile, ntile, ksplit)) return;
    SM_TL(1);
    const int kt_begin = ksplit * p.ktiles_per_split;
    const int ktiles = p.K >> 6;
    int kt_end = kt_begin + p.ktiles_per_split; if (kt_end > ktiles) kt_end = ktiles;
    const int nk = kt_end - kt_begin;                           // K tiles of this block
    const int nit = (nk + KT - 1) / KT;                         // ring stages (= loop iterations)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> LDS bases live in SGPRs
    const bool is_loader = NLW == 0 || wave >= NCW, is_consumer = NLW == 0 || wave < NCW;   // wave-uniform roles
    const int kgrp = KS == 1 ? 0 : (wave >= NGW ? 1 : 0), gwave = KS == 1 ? wave : wave - kgrp * NGW;   // K group, wave inside it
    const int wm = (NLW && wave >= NCW) ? 0 : gwave / WGN, wn = (NLW && wave >= NCW) ? 0 : gwave % WGN;
    const int ltid = NLW ? (tid - NCW * 64) & (NLT - 1) : tid;  // index among the loader threads (consumers: unused alias)
    const int lwave = NLW ? (wave >= NCW ? wave - NCW : 0) : wave;
    const int csrc = (ltid & 7) ^ ((ltid >> 4) & 7);           // source chunk for this thread's LDS slot (rows j*RPI + ltid/8)
    const half_t* zero = (const half_t*)g_zero_page;

    // ---- per-thread row state -----------------------------------------------------------------------
    // Setup runs once per block with one wave per SIMD, i.e. every instruction is exposed: no loops, no branches, and
    // each tile row (input pixel index + tap mask, shared by the 8 lanes that fetch its 8 chunks) is derived by ONE thread
    // and handed over through LDS.
    {
        const int HoWo = p.Ho * p.Wo;
        const bool unit_dil = p.dH == 1 && p.dW == 1;
        int2* rowinfo = (int2*)(smem_raw + LDS_BYTES - BM * 8);
        if (BM >= NT || tid < BM) {
#pragma unroll
            for (int rr = 0; rr < (BM + NT - 1) / NT; rr++) {
                const int row = tid + rr * NT;
                int m = mtile * BM + row; if (m >= p.M) m = p.M - 1;
                const int n = (int)sm_fastdiv((unsigned)m, p.dhw_mul, p.dhw_sh), r = m - n * HoWo;
                const int oy = (int)sm_fastdiv((unsigned)r, p.dw_mul, p.dw_sh), ox = r - oy * p.Wo;
                const int iy = oy * p.sH - p.pt, ix = ox * p.sW - p.pl;
                const int pix = (n * p.H + iy) * p.W + ix;          // |pix| < 2^31 (fill_params)
                unsigned msk = 0;
                if (unit_dil) {
                    // taps kw in [lo_w, hi_w] and kh in [lo_h, hi_h] are inside the image: bits kh*kW + kw of the mask
                    const int lo_w = ix < 0 ? -ix : 0, hi_w = min(p.kW - 1, p.W - 1 - ix);
                    const int lo_h = iy < 0 ? -iy : 0, hi_h = min(p.kH - 1, p.H - 1 - iy);
                    const unsigned colm = (0xFFFFFFFFu >> (31 - max(hi_w, 0))) & (0xFFFFFFFFu << min(lo_w, 31));
                    const unsigned rowm = (0xFFFFFFFFu >> (32 - (max(hi_h, 0) + 1) * p.kW)) & (0xFFFFFFFFu << (min(lo_h, p.kH - 1) * p.kW));
                    if (lo_w <= hi_w && lo_h <= hi_h) msk = colm * (p.rowpat & rowm);
                } else {
                    unsigned colm = 0;
                    for (int kw = 0; kw < p.kW; kw++) colm |= ((unsigned)(ix + kw * p.dW) < (unsigned)p.W ? 1u : 0u) << kw;
                    for (int kh = 0; kh < p.kH; kh++) if ((unsigned)(iy + kh * p.dH) < (unsigned)p.H) msk |= colm << (kh * p.kW);
                }
                rowinfo[row] = int2{pix, (int)msk};
            }
        }
        __syncthreads();
    }
    const half_t* arow[A_IT]; unsigned amask[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; j++) {
        const int2 ri = ((const int2*)(smem_raw + LDS_BYTES - BM * 8))[j * RPI + (ltid >> 3)];
        arow[j] = p.x + (long)ri.x * p.x_ld + csrc * 8; amask[j] = (unsigned)ri.y;
    }
    if (TABLE_ALIAS) __syncthreads();                           // every thread has its rows before the ring (which holds the table) fills
    SM_TL(2);
    const half_t* brow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
        int n = ntile * BN + j * RPI + (ltid >> 3); if (n >= p.Cout) n = p.Cout - 1;
        brow[j] = p.w + (long)n * p.K + csrc * 8;
    }
    // ---- K-tile cursor of the NEXT tile to issue: all wave-uniform, advanced with a few SALU ops per tile -------------
    const int cpt = p.Cin >> 6;                                 // K tiles per tap
    int cblk, kw_i;
    long a_off, b_off;                                          // element offsets of the tile inside a pixel row / weight row
    unsigned tapbit;
    if (kt_begin == 0) { cblk = 0; kw_i = 0; a_off = 0; b_off = 0; tapbit = 1u; }
    else {                                                      // split-K blocks start in the middle of the K range
        const int tap = kt_begin / cpt; cblk = kt_begin - tap * cpt;
        const int kh = tap / p.kW; kw_i = tap - kh * p.kW;
        a_off = ((long)(kh * p.dH) * p.W + kw_i * p.dW) * p.x_ld + cblk * 64;
        b_off = (long)kt_begin * 64;
        tapbit = 1u << tap;
    }
    const long a_wrap_c = (long)p.dW * p.x_ld - (long)(cpt - 1) * 64;      // next tap, same kernel row
    const long a_wrap_w = ((long)p.dH * p.W - (long)p.kW * p.dW) * p.x_ld;  // ... and on to the next kernel row

    int issued = 0;                                             // ring stages put in flight so far
    int ktq = 0;                                                // KS > 1: K tiles the cursor has moved past
    auto piece = [&](int slot, int i, int sub = 0) {            // i-th LDS-DMA instruction of the K tile under the cursor
        unsigned char* sa = smem_raw + slot * STAGE_BYTES + sub * SUB_BYTES + lwave * 1024;  // wave-uniform LDS base; lane*16 B added by the DMA
        if (i < A_IT) {
            if (ABL == 3 && issued >= NSTAGE - 1) return;       // ablation: no activation traffic after the prologue
            const half_t* src = (amask[i] & tapbit) ? arow[i] + a_off : zero;
            SM_GLDS16_AUX(src, sa + i * NLT * 16, SM_A_NT);
        } else SM_GLDS16_AUX(brow[i - A_IT] + b_off, sa + A_BYTES + (i - A_IT) * NLT * 16, SM_B_NT);
    };
    auto advance = [&]() {
        b_off += 64;
        if (++cblk == cpt) {
            cblk = 0; a_off += a_wrap_c; tapbit <<= 1;
            if (++kw_i == p.kW) { kw_i = 0; a_off += a_wrap_w; }
        } else a_off += 64;
    };
    auto issue = [&](int slot) {
#pragma unroll
        for (int sub = 0; sub < KT; sub++) {
            if (is_loader) {
#pragma unroll
                for (int i = 0; i < A_IT + B_IT; i++) piece(slot, i, sub);
            }
            if (KT == 1) advance();
            else if (ktq + 1 < nk) { advance(); ++ktq; }        // an odd tail re-fetches the last K tile (never multiplied)
        }
    };

    SM_TL(3);
    // ---- first tiles in flight as early as possible; everything else of the prologue hides behind their latency --------
    // Tile i lives in ring slot i % NSTAGE.  Short K (nk <= NSTAGE, the HBM-streaming 1x1 layers): everything the block
    // will ever read is in flight at once.
    const int pre = nit <= NSTAGE ? nit : NSTAGE - 1;
#pragma unroll
    for (int s = 0; s < NSTAGE; s++) if (s < pre) { issue(s); ++issued; }
    int slot = 0, islot = pre == NSTAGE ? 0 : pre;
    __builtin_amdgcn_sched_barrier(0);
    SM_TL(4);

    // ---- the only ordinary loads of the kernel: EXTRA = 2 + EP_PASSES*IPT 16-byte loads per thread, ALWAYS issued ------
    // (from the zero page when there is nothing to fetch) right behind the first tiles, so that the counted waits of the
    // main loop know exactly how many younger VMEM operations sit behind a prologue tile; consumed in the epilogue.
    //  * residual operand (fused MPSNNAdditionNode): same (row, chunk) ownership as the epilogue stores, clamped addresses
    //    (what a ragged tile reads extra is never stored);
    //  * bias of the thread's 8 output channels for the fast epilogue.  The accumulators then start at ZERO: nothing the
    //    main loop needs comes from an ordinary load, so it starts as soon as tile 0 has landed (bias-initialised
    //    accumulators made every block wait for bias AND residual, the youngest loads, before its first MFMA).
    constexpr int CHUNKS = BN / 8, ITEMS = EP_ROWS * CHUNKS, IPT = (ITEMS + NT - 1) / NT;
    constexpr int EXTRA = 2 + (LATE_RES ? 0 : EP_PASSES * IPT);
    static_assert(NT % CHUNKS == 0, "a thread keeps its channel chunk across items");
    static_assert((NSTAGE - 1) * LOADS + EXTRA <= 63, "vmcnt range");
    const bool vec_ok = ((p.y_ld & 7) == 0) && ((((uintptr_t)p.y) & 15) == 0) && ((p.Cout & 7) == 0) &&
                        (p.res == nullptr || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0));
    const bool fastep = vec_ok && !p.yplace && p.splitk == 1 && p.act <= SM_ACT_PRELU && (((uintptr_t)p.bias) & 15) == 0;
    floatx4 eb0, eb1;
    {
        const float* bp = (fastep && p.bias) ? p.bias + min(ntile * BN + (tid % CHUNKS) * 8, p.Cout - 8) : (const float*)g_zero_page;
        eb0 = *(const floatx4*)bp; eb1 = *(const floatx4*)(bp + 4);
    }
    half8_t rres[LATE_RES ? 1 : EP_PASSES][IPT];
    const bool use_res = p.res && vec_ok && p.splitk == 1;
    auto load_res = [&](int pw) {
#pragma unroll
        for (int k2 = 0; k2 < IPT; k2++) {
            int it = tid + k2 * NT; if (ITEMS % NT != 0 && it >= ITEMS) it = ITEMS - 1;
            const int row = it / CHUNKS, ch = it - row * CHUNKS;
            const int m = min(mtile * BM + pw * EP_ROWS + row, p.M - 1), co = min(ntile * BN + ch * 8, p.Cout - 8);
            rres[LATE_RES ? 0 : pw][k2] = *(const half8_t*)(use_res ? p.res + (long)m * p.res_ld + co : zero);
        }
    };
    if (!LATE_RES) {
#pragma unroll
        for (int pw = 0; pw < EP_PASSES; pw++) load_res(pw);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- accumulators: zero (fast epilogue adds the bias), or the bias for the generic epilogue / split-K partials:
    // register 4q+r of acc[a][*] <-> cout (wn*TN+a)*32 + 8q + 4*(lane>>5) + r
    floatx4 bv[TN][4];
    floatx16 acc[TN][TM];
    {
        const bool bias_vec = (p.Cout & 3) == 0 && ((((uintptr_t)p.bias) & 15) == 0);
#pragma unroll
        for (int a = 0; a < TN; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = ntile * BN + (wn * TN + a) * 32 + 8 * q + 4 * (lane >> 5);
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && p.splitk == 1 && !fastep && kgrp == 0) {      // (K groups: the partial sums are added, the bias enters once)
                    if (bias_vec) v = *(const floatx4*)(p.bias + min(co, p.Cout - 4));   // couts >= Cout are never stored
                    else for (int r = 0; r < 4; r++) if (co + r < p.Cout) v[r] = p.bias[co + r];
                }
                bv[a][q] = v;
            }
    }

    // fragment addresses: row (lane & 31) of a 32-row block, 16-byte chunk (2*ks + (lane >> 5)) ^ swizzle(row)
    unsigned voff[4];
    {
        const unsigned sw = ((lane & 31) >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) voff[ks] = (lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ sw) << 4);
    }
    // ILV = 1: a VMEM instruction stalls its wave while the address queue is full, so the LOADS DMA instructions of the
    // next tile are spread between the MFMAs of this one instead of being issued as one burst (mainloop.hip: +5 % for the
    // 8-wave 128x128 tile, +26 % for 256x128; no gain for 4-wave tiles)
    auto compute = [&](int slot, int dslot, bool dma, int sub = 0) {
        const unsigned sA = slot * STAGE_BYTES + (kgrp + sub) * SUB_BYTES + wm * (TM * 4096), sB = slot * STAGE_BYTES + (kgrp + sub) * SUB_BYTES + A_BYTES + wn * (TN * 4096);
        half8_t fx[2][TM], fw[2][TN];
#pragma unroll
        for (int b = 0; b < TM; b++) fx[0][b] = *(const half8_t*)(smem_raw + sA + voff[0] + b * 4096);
#pragma unroll
        for (int a = 0; a < TN; a++) fw[0][a] = *(const half8_t*)(smem_raw + sB + voff[0] + a * 4096);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int b = 0; b < TM; b++) fx[(ks + 1) & 1][b] = *(const half8_t*)(smem_raw + sA + voff[ks + 1] + b * 4096);
#pragma unroll
                for (int a = 0; a < TN; a++) fw[(ks + 1) & 1][a] = *(const half8_t*)(smem_raw + sB + voff[ks + 1] + a * 4096);
            }
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    if (ABL != 1) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks & 1][a], fx[ks & 1][b], acc[a][b], 0, 0, 0);
                    else { asm volatile("" ::"v"(fw[ks & 1][a]), "v"(fx[ks & 1][b])); }
                    if (ILV) {
                        constexpr int NM = 4 * TM * TN, MPL = NM / LOADS > 0 ? NM / LOADS : 1;   // MFMAs per DMA instruction (ILV: KS == 1)
                        const int cnt = (ks * TN + a) * TM + b + 1;
                        if (cnt % MPL == 0 && cnt / MPL - 1 < LOADS && dma) piece(dslot, cnt / MPL - 1);
                    }
                }
        }
        if (ILV) {
            constexpr int NM = 4 * TM * TN, MPL = NM / LOADS > 0 ? NM / LOADS : 1;
#pragma unroll
            for (int i = NM / MPL; i < LOADS; i++) if (dma) piece(dslot, i);
            if (dma) advance();
        }
    };
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = bv[a][r >> 2][r & 3];
    SM_TL(5);

    // ---- main loop: ONE barrier per K tile; the counted vmcnt leaves the newer tiles (and nothing older) in flight -------
    for (int i = 0; i < nit; i++) {
        const int newer = issued - i - 1;                       // stages issued after stage i, allowed to stay in flight
        if (is_loader) {                                        // (consumer waves of a specialised block have no DMA in flight)
            if (i < pre) {                                      // a prologue tile: the EXTRA ordinary loads are younger than it
                if (NSTAGE >= 4 && newer >= 3) sm_wait_vmcnt<(NSTAGE >= 4 ? 3 : 0) * LOADS + EXTRA>();
                else if (NSTAGE >= 3 && newer == 2) sm_wait_vmcnt<(NSTAGE >= 3 ? 2 : 0) * LOADS + EXTRA>();
                else if (NSTAGE >= 2 && newer == 1) sm_wait_vmcnt<(NSTAGE >= 2 ? 1 : 0) * LOADS + EXTRA>();
                else sm_wait_vmcnt<EXTRA>();
            } else {                                            // issued in the loop: everything older, EXTRA included, has to be back
                if (NSTAGE >= 4 && newer >= 3) sm_wait_vmcnt<(NSTAGE >= 4 ? 3 : 0) * LOADS>();
                else if (NSTAGE >= 3 && newer == 2) sm_wait_vmcnt<(NSTAGE >= 3 ? 2 : 0) * LOADS>();
                else if (NSTAGE >= 2 && newer == 1) sm_wait_vmcnt<(NSTAGE >= 2 ? 1 : 0) * LOADS>();
                else sm_wait_vmcnt<0>();
            }
        }
        __builtin_amdgcn_s_barrier();                           // everybody's share of tile i landed; the slot of tile i-1 is free
        __builtin_amdgcn_sched_barrier(0);
#ifdef SM_TIMELINE
        if (i == 0) SM_TL(6);
#endif
        const bool more = issued < nit && ABL != 2;
        if (!ILV && more) issue(islot);
        if (is_consumer && (KT == 1 || KT * i + kgrp < nk)) compute(slot, islot, more);
        if (KT > KS && is_consumer && KT * i + 1 < nk) compute(slot, islot, more, 1);       // second K tile of the stage (KS = 1, KT = 2)
        if (more) { ++issued; if (++islot == NSTAGE) islot = 0; }
        if (++slot == NSTAGE) slot = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's fragment reads are done before it passes the next barrier
    }
    __syncthreads();
    SM_TL(7);

    // ---- epilogue: LDS transpose, coalesced 16-byte global accesses ---------------------------------------------
    float* stg = (float*)smem_raw;
#pragma unroll
    for (int pw = 0; pw < EP_PASSES; pw++) {
        if (LATE_RES && use_res) load_res(pw);                  // in flight across the staging writes and the barrier
        if (is_consumer && wm / WPP == pw) {
            const int wrow = (wm % WPP) * WROWS + kgrp * EP_ROWS;   // K group g stages into region g (EP_ROWS*EP_LD floats each)
#pragma unroll
            for (int b = 0; b < TM; b++)
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int row = wrow + b * 32 + (lane & 31), col = (wn * TN + a) * 32 + 8 * q + 4 * (lane >> 5);
                        *(floatx4*)(stg + row * EP_LD + col) = floatx4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                    }
        }
        __syncthreads();
        // Fast path (every ResNet / MobileNet / U-Net layer): dense 16-byte output, no placement, no split-K, ReLU-family
        // activation.  Straight-line: all staging reads of the thread requested up front, one select per value, no per-element
        // branches or calls (the generic loop below costs ~1300 clk per 8-value item, this one ~1/3 of that).
        if (fastep) {
            const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;        // f(v) = v < 0 ? v*sl : v
            const int m0 = mtile * BM + pw * EP_ROWS, co0 = ntile * BN;
            floatx4 s0[IPT], s1[IPT]; int rowi[IPT], chi[IPT];
#pragma unroll
            for (int k2 = 0; k2 < IPT; k2++) {
                int it = tid + k2 * NT; if (ITEMS % NT != 0 && it >= ITEMS) it = ITEMS - 1;
                rowi[k2] = it / CHUNKS; chi[k2] = it - rowi[k2] * CHUNKS;
                s0[k2] = *(const floatx4*)(stg + rowi[k2] * EP_LD + chi[k2] * 8); s1[k2] = *(const floatx4*)(stg + rowi[k2] * EP_LD + chi[k2] * 8 + 4);
                if (KS == 2) {                                  // + the other K group's partial sums
                    const floatx4 t0 = *(const floatx4*)(stg + EP_REGION + rowi[k2] * EP_LD + chi[k2] * 8), t1 = *(const floatx4*)(stg + EP_REGION + rowi[k2] * EP_LD + chi[k2] * 8 + 4);
                    s0[k2] += t0; s1[k2] += t1;
                }
            }
            // relu0: plain ReLU as a packed fp16 max AFTER the rounding (max commutes with it): 4 v_pk_max_f16 per item instead of 24 compare / multiply / select
            auto emit = [&](const bool has_res, auto relu0c) {
                constexpr bool RELU0 = decltype(relu0c)::value;
#pragma unroll
                for (int k2 = 0; k2 < IPT; k2++) {
                    const int m = m0 + rowi[k2], co = co0 + chi[k2] * 8;
                    float v[8] = {s0[k2][0] + eb0[0], s0[k2][1] + eb0[1], s0[k2][2] + eb0[2], s0[k2][3] + eb0[3],
                                  s1[k2][0] + eb1[0], s1[k2][1] + eb1[1], s1[k2][2] + eb1[2], s1[k2][3] + eb1[3]};
                    if (has_res) { const half8_t rr = rres[LATE_RES ? 0 : pw][k2];
#pragma unroll
                        for (int j = 0; j < 8; j++) v[j] += (float)rr[j]; }
                    half8_t o;
                    if constexpr (RELU0) {
#pragma unroll
                        for (int j = 0; j < 8; j++) o[j] = (half_t)v[j];
                        o = __builtin_elementwise_max(o, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) o[j] = (half_t)(v[j] < 0.f ? v[j] * sl : v[j]);
                    }
                    const bool live = (ITEMS % NT == 0 || tid + k2 * NT < ITEMS) && m < p.M && co < p.Cout;
                    if (live) *(half8_t*)(p.y + (long)m * p.y_ld + co) = o;
                }
            };
            const bool relu0 = p.act != SM_ACT_NONE && p.slope == 0.f;
            if (p.res) { if (relu0) emit(true, std::true_type()); else emit(true, std::false_type()); }
            else { if (relu0) emit(false, std::true_type()); else emit(false, std::false_type()); }
            if (pw + 1 < EP_PASSES) __syncthreads();
            continue;
        }
#pragma unroll
        for (int k2 = 0; k2 < IPT; k2++) {
            const int it = tid + k2 * NT;
            if (it >= ITEMS) continue;
            const int row = it / CHUNKS, ch = it - row * CHUNKS;
            const int m = mtile * BM + pw * EP_ROWS + row, co = ntile * BN + ch * 8;
            if (m >= p.M || co >= p.Cout) continue;
            const long opix = sm_out_pixel(p, m);
            floatx4 v0 = *(const floatx4*)(stg + row * EP_LD + ch * 8), v1 = *(const floatx4*)(stg + row * EP_LD + ch * 8 + 4);
            if (KS == 2) { v0 += *(const floatx4*)(stg + EP_REGION + row * EP_LD + ch * 8); v1 += *(const floatx4*)(stg + EP_REGION + row * EP_LD + ch * 8 + 4); }
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.splitk > 1) {
                float* dst = p.ws + ((long)ksplit * p.M + m) * p.Cout + co;
                if ((p.Cout & 3) == 0 && co + 7 < p.Cout) { *(floatx4*)dst = v0; *(floatx4*)(dst + 4) = v1; }
                else for (int j = 0; j < 8 && co + j < p.Cout; j++) dst[j] = v[j];
                continue;
            }
            if (vec_ok) {
                if (p.res) { const half8_t rr = rres[LATE_RES ? 0 : pw][k2];     // prefetched before the main loop
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] += (float)rr[j]; }
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = (half_t)sm_act(v[j], p.act, p.slope);
                *(half8_t*)(p.y + opix * p.y_ld + co) = o;
            } else {
                for (int j = 0; j < 8 && co + j < p.Cout; j++) {
                    float t = v[j];
                    if (p.res) t += (float)p.res[(long)m * p.res_ld + co + j];
                    p.y[opix * p.y_ld + co + j] = (half_t)sm_act(t, p.act, p.slope);
                }
            }
        }
        if (pw + 1 < EP_PASSES) __syncthreads();
    }
    SM_TL(8);
    SM_TL_END();
}
