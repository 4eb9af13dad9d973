// conv1x1_ps.hpp — PERSISTENT pointwise convolution (1x1, stride 1 or 2, no padding, Cin % 64 == 0): y[M][N] = x[M][K] w[N][K]^T as a
// GEMM whose blocks live for the whole launch.  Included by conv_igemm.hip (same ConvParams / launcher / LDS image as conv_igemm64).
//
// Why: ResNet-50's layer-3/4 pointwise layers have K = 256..2048, i.e. 4..32 K tiles per output tile, and a
// conv_igemm64 block pays ≈2.4 k clk of prologue + ≈2.2 k clk until its first tile lands + ≈3.4 k clk of epilogue around ≈1.3 k clk per
// K tile: with K = 256 more than half of a block's life is fixed cost, and 784 tiles on 256 CUs run as whole rounds of blocks.  Here
//  * ONE block per CU walks its list of output tiles; the operand ring never drains: (tile, K tile) pairs form ONE sequence of ring
//    items, so the first K tiles of output tile t+1 are in flight while tile t is multiplied and written back;
//  * 4 LOADER waves (one per SIMD) issue every LDS-DMA, 8 CONSUMER waves read fragments, issue MFMAs and run the epilogue.  The split is
//    what makes the cross-tile pipeline legal: `vmcnt` counts loads and stores in one counter that retires out of order between the two
//    kinds, so a counted wait is only exact in a wave that issues loads alone — the loaders; the consumers' stores / residual loads are
//    ordinary compiler-tracked accesses in waves that never wait for a DMA;
//  * both roles meet at ONE s_barrier per ring item (item i landed; slot of item i-1 free), exactly conv_igemm64's protocol;
//  * epilogue in registers (no LDS staging, no block-wide barrier): fp32 accumulators of a 32x32 fragment -> v_permlane32_swap gives
//    each lane 8 consecutive output channels of its pixel -> + bias + residual (16-byte load) in fp32, activation, ONE rounding,
//    16-byte store.  The stores of tile t are issued between the MFMAs of tile t+1's first K tile (a burst of VMEM instructions stalls
//    its wave on the address queue), the residual of tile t+1 is requested there as well;
//  * block b runs on XCD b % 8: XCD x owns the contiguous tile range [xT/8, (x+1)T/8) (tile = (m tile, n tile), n fastest), its blocks
//    take tiles round-robin inside that range — the n tiles of one m tile run at the same time behind the same L2.
//  * numerics = conv_igemm64's fast epilogue bit for bit: fp32 MFMA chain over k ascending from zero, (acc + bias) + residual, activation,
//    one rounding.
#pragma once

// DEFER = false (the 64x64 wave tiles: 64 accumulator registers): the epilogue requests the residual and issues the stores itself, so
// neither occupies registers during the K loop.
template <int BM, int BN, int WGM, int WGN, int NSB, bool DEFER = true>
__global__ __launch_bounds__((WGM * WGN + 4) * 64) void conv1x1_ps_kernel(ConvParams p) {
    static_assert(WGM * WGN == 8, "8 consumer waves");
    constexpr int NCW = 8, NLW = 4, NLT = NLW * 64;
    constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
    constexpr int A_IT = BM * 8 / NLT, B_IT = BN * 8 / NLT, LOADS = A_IT + B_IT;
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "tile too small");
    static_assert(NSB >= 2 && NSB <= 4 && (NSB - 1) * LOADS <= 63, "ring depth / vmcnt range");
    static_assert(NSB * STAGE_BYTES <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= NCW;
    SM_TL(0);                                                           // (SM_TIMELINE builds: stamps of consumer wave 0, see sm_conv2d_time)

    // ---- this block's tiles --------------------------------------------------------------------------------------------
    const int nt = p.nt, T = p.mt * nt, G = gridDim.x, KT = p.K >> 6;
    const int xcd = blockIdx.x & (SM_NUM_XCD - 1), bslot = blockIdx.x >> 3;
    const int gx = (G - xcd + SM_NUM_XCD - 1) / SM_NUM_XCD;            // blocks on this XCD
    const int t0 = (int)((long)xcd * T / SM_NUM_XCD), t1 = (int)((long)(xcd + 1) * T / SM_NUM_XCD);
    const int nmine = bslot < t1 - t0 ? (t1 - t0 - bslot + gx - 1) / gx : 0;
    if (nmine == 0) return;
    const int n_items = nmine * KT;
    // the layer's bias vector sits behind the ring (the launcher checks the room); ordered before its first reader by the loop's barriers
    float* const bl = (float*)(ps_smem + NSB * STAGE_BYTES);
    if (!is_loader) {                                                   // (by the consumers: the loaders' vmcnt sees nothing but LDS-DMA)
        for (int i = tid; i < p.Cout; i += NCW * 64) bl[i] = p.bias ? p.bias[i] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto tile_of = [&](int j, int& mtile, int& ntile) {                 // j-th tile of this block
        const int t = t0 + bslot + j * gx;
        mtile = (int)sm_fastdiv((unsigned)t, p.nt_mul, p.nt_sh); ntile = t - mtile * nt;
    };

    if (is_loader) {
        // =================================================== loader waves =====================================================
        const int ltid = tid - NCW * 64, lwave = wave - NCW;
        const int csrc = (ltid & 7) ^ ((ltid >> 4) & 7);               // source chunk of this thread's LDS slot (conv_igemm64's swizzle)
        const int HoWo = p.Ho * p.Wo;
        const bool unit = p.sH == 1 && p.sW == 1;
        const half_t* arow[A_IT]; const half_t* brow[B_IT];
        const half_t* arow2[A_IT];                                     // second source (p.x2): the rows K tiles >= p.kt_sw are read from
        const bool dual = p.x2 != nullptr;
        auto rows_of = [&](int j) {
            int mtile, ntile; tile_of(j < nmine ? j : nmine - 1, mtile, ntile);
#pragma unroll
            for (int i = 0; i < A_IT; i++) {
                int m = mtile * BM + i * (NLT / 8) + (ltid >> 3); if (m >= p.M) m = p.M - 1;   // ragged rows only feed outputs that are never stored
                int pix = m;
                if (!unit || dual) {
                    const int n = (int)sm_fastdiv((unsigned)m, p.dhw_mul, p.dhw_sh), r = m - n * HoWo;
                    const int oy = (int)sm_fastdiv((unsigned)r, p.dw_mul, p.dw_sh), ox = r - oy * p.Wo;
                    if (!unit) pix = (n * p.H + oy * p.sH) * p.W + ox * p.sW;
                    arow2[i] = dual ? p.x2 + (long)((n * p.x2_H + oy * p.x2_s) * p.x2_W + ox * p.x2_s) * p.x2_ld + csrc * 8 : nullptr;
                }
                arow[i] = p.x + (long)pix * p.x_ld + csrc * 8;
            }
#pragma unroll
            for (int i = 0; i < B_IT; i++) {
                int n = ntile * BN + i * (NLT / 8) + (ltid >> 3); if (n >= p.Cout) n = p.Cout - 1;
                brow[i] = p.w + (long)n * p.K + csrc * 8;
            }
        };
        int ld_tile = 0, ld_kt = 0, issued = 0;
        rows_of(0);
        auto issue = [&]() {                                            // ring item `issued` -> slot issued % NSB
            unsigned char* sa = ps_smem + (issued % NSB) * STAGE_BYTES + lwave * 1024;
            const int koff = ld_kt * 64;
            if (dual && ld_kt >= p.kt_sw) {                             // (wave-uniform)
                const int koff2 = (ld_kt - p.kt_sw) * 64;
#pragma unroll
                for (int i = 0; i < A_IT; i++) SM_GLDS16_AUX(arow2[i] + koff2, sa + i * NLT * 16, SM_PS_X_NT);
            } else {
#pragma unroll
                for (int i = 0; i < A_IT; i++) SM_GLDS16_AUX(arow[i] + koff, sa + i * NLT * 16, SM_PS_X_NT);
            }
#pragma unroll
            for (int i = 0; i < B_IT; i++) SM_GLDS16(brow[i] + koff, sa + A_BYTES + i * NLT * 16);
            ++issued;
            if (++ld_kt == KT) { ld_kt = 0; ++ld_tile; rows_of(ld_tile); }
        };
#pragma unroll
        for (int s = 0; s < NSB - 1; s++) if (issued < n_items) issue();
#pragma unroll 1
        for (int i = 0; i < n_items; i++) {
            const int newer = issued - i - 1;                           // items behind item i that may stay in flight
            if (NSB >= 4 && newer >= 3) sm_wait_vmcnt<(NSB >= 4 ? 3 : 0) * LOADS>();
            else if (NSB >= 3 && newer == 2) sm_wait_vmcnt<(NSB >= 3 ? 2 : 0) * LOADS>();
            else if (newer == 1) sm_wait_vmcnt<LOADS>();
            else sm_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                               // item i landed (everybody's share); the slot of item i-1 is free
            __builtin_amdgcn_sched_barrier(0);
            if (issued < n_items) issue();
        }
        return;
    }

    // ===================================================== consumer waves ======================================================
    const int wm = wave / WGN, wn = wave % WGN;
    unsigned voff[4];
    {
        const unsigned sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) voff[ks] = l31 * 128 + (((ks * 2 + hi) ^ sw) << 4);
    }
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;              // f(v) = v < 0 ? v*sl : v  (ReLU family only: launcher)
    const bool relu0 = p.act != SM_ACT_NONE && sl == 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    floatx16 acc[TN][TM];
    u32x4 outv[TN][TM][2];                                              // finished tile waiting for its stores: [a][b][j] = couts a*32 + 16j + 8hi .. +7 of pixel b
    half8_t rres[TN][TM][2];                                            // residual of the tile being multiplied, same ownership
    int pm0 = 0, pn0 = 0; bool pending = false;
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < TN; a++)
#pragma unroll
            for (int b = 0; b < TM; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    };
    auto store_one = [&](int a, int b, int j) {                         // one 16-byte store of the pending tile
        const int m = pm0 + b * 32 + l31, co = pn0 + a * 32 + 16 * j + 8 * hi;
        if (m < p.M && co < p.Cout) *(u32x4*)(p.y + (long)m * p.y_ld + co) = outv[a][b][j];
    };
    auto load_res = [&](int m0, int n0, int a, int b, int j) {
        const int m = min(m0 + b * 32 + l31, p.M - 1), co = min(n0 + a * 32 + 16 * j + 8 * hi, p.Cout - 8);
        rres[a][b][j] = *(const half8_t*)(p.res + (long)m * p.res_ld + co);
    };
    constexpr int NST = TN * TM * 2;                                    // 16-byte stores (and residual loads) per wave per tile

    int mtile, ntile; tile_of(0, mtile, ntile);
    int ctile = 0, ckt = 0;
    int m0 = 0, n0 = 0;
    // one ring item.  PHASE 1 (first K tile of an output tile, DEFER): one deferred store of the previous tile and one residual request of
    // this one ride behind each of the first NST MFMAs; the other K tiles (PHASE 0) are a straight MFMA chain the compiler can pipeline.
    auto compute = [&](auto phasec, int slot) {
        constexpr int PHASE = decltype(phasec)::value;
        const unsigned sA = slot * STAGE_BYTES + wm * (TM * 4096), sB = slot * STAGE_BYTES + A_BYTES + wn * (TN * 4096);
        half8_t fx[2][TM], fw[2][TN];
#pragma unroll
        for (int b = 0; b < TM; b++) fx[0][b] = *(const half8_t*)(ps_smem + sA + voff[0] + b * 4096);
#pragma unroll
        for (int a = 0; a < TN; a++) fw[0][a] = *(const half8_t*)(ps_smem + sB + voff[0] + a * 4096);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int b = 0; b < TM; b++) fx[(ks + 1) & 1][b] = *(const half8_t*)(ps_smem + sA + voff[ks + 1] + b * 4096);
#pragma unroll
                for (int a = 0; a < TN; a++) fw[(ks + 1) & 1][a] = *(const half8_t*)(ps_smem + sB + voff[ks + 1] + a * 4096);
            }
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks & 1][a], fx[ks & 1][b], acc[a][b], 0, 0, 0);
                    const int c = (ks * TN + a) * TM + b;               // 0 .. 4*TN*TM - 1 (>= NST)
                    if (PHASE != 0 && c < NST) {
                        const int ca = c / (TM * 2), cb = (c / 2) % TM, cj = c & 1;
                        if (pending) store_one(ca, cb, cj);
                        if (p.res) load_res(m0, n0, ca, cb, cj);
                    }
                }
        }
    };
    zero_acc();
    SM_TL(1);
#pragma unroll 1
    for (int i = 0; i < n_items; i++) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef SM_TIMELINE
        if (i == 0) SM_TL(2);
#endif
        if (ckt == 0) { m0 = mtile * BM + wm * (TM * 32); n0 = ntile * BN + wn * (TN * 32); }
        if (DEFER && ckt == 0) { compute(std::integral_constant<int, 1>(), i % NSB); pending = false; }
        else compute(std::integral_constant<int, 0>(), i % NSB);
        if (++ckt == KT) {
#ifdef SM_TIMELINE
            if (ctile == 0) SM_TL(3);
#endif
            // ---- epilogue of this output tile, in registers ----------------------------------------------------------------
            if (!DEFER && p.res) {
#pragma unroll
                for (int c = 0; c < NST; c++) load_res(m0, n0, c / (TM * 2), (c / 2) % TM, c & 1);
            }
#pragma unroll
            for (int a = 0; a < TN; a++) {
                floatx4 bq[2][2];                                       // bias of couts a*32 + 16j + 8hi + 4h .. +3
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int co = min(n0 + a * 32 + 16 * j + 8 * hi + 4 * h, p.Cout - 4);
                        bq[j][h] = *(const floatx4*)(bl + co);
                    }
#pragma unroll
                for (int b = 0; b < TM; b++) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        // registers 4q + r of the fragment <-> cout 8q + 4hi + r; q = 2j, 2j+1: after the swap the low half-wave holds couts
                        // 16j .. 16j+7 of its pixel as {lo[0..3], up[0..3]}, the high half-wave couts 16j+8 .. 16j+15
                        unsigned lo[4], up[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            // (by value first: __builtin_bit_cast applied to a vector ELEMENT lvalue reads element 0 of the vector)
                            const float flo = acc[a][b][8 * j + r], fup = acc[a][b][8 * j + 4 + r];
                            lo[r] = __builtin_bit_cast(unsigned, flo); up[r] = __builtin_bit_cast(unsigned, fup);
                            sm_permlane32_swap(lo[r], up[r]);
                        }
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; r++) { v[r] = __builtin_bit_cast(float, lo[r]) + bq[j][0][r]; v[4 + r] = __builtin_bit_cast(float, up[r]) + bq[j][1][r]; }
                        if (p.res) {
                            const half8_t rr = rres[a][b][j];
#pragma unroll
                            for (int k = 0; k < 8; k++) v[k] += (float)rr[k];
                        }
                        half8_t o;
#pragma unroll
                        for (int k = 0; k < 8; k++) o[k] = relu0 ? (half_t)v[k] : (half_t)(v[k] < 0.f ? v[k] * sl : v[k]);
                        if (relu0) o = __builtin_elementwise_max(o, half8_t{0, 0, 0, 0, 0, 0, 0, 0});   // plain ReLU: round, then four packed maxima (the same values)
                        outv[a][b][j] = __builtin_bit_cast(u32x4, o);
                        if (!DEFER) { pm0 = m0; pn0 = n0; store_one(a, b, j); }
                    }
                }
            }
            pm0 = m0; pn0 = n0; pending = DEFER;
            zero_acc();
            ckt = 0; ++ctile;
#ifdef SM_TIMELINE
            if (ctile == 1) SM_TL(4);
            if (ctile == 2) SM_TL(5);
            if (ctile == 3) SM_TL(6);
#endif
            if (ctile < nmine) tile_of(ctile, mtile, ntile);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this wave's fragment reads are done before it passes the next barrier
    }
    SM_TL(7);
    if (pending) {
#pragma unroll
        for (int c = 0; c < NST; c++) store_one(c / (TM * 2), (c / 2) % TM, c & 1);
    }
    SM_TL(8);
    SM_TL_END();
}
