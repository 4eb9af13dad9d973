// conv1x1_pair.hpp — two chained pointwise convolutions in ONE persistent launch:
//     Y = act1(X * W1^T + b1 + R)      [M x N1]   (the 1x1 "expand" that closes a ResNet bottleneck: + residual + ReLU)
//     Z = act2(Y * W2^T + b2)          [M x N2]   (the 1x1 "reduce" that opens the next bottleneck)
// Both outputs are stored: Y is the next block's residual.  Included by conv_igemm.hip.
//
// Why: at 56x56 (M = 200,704 pixels) these layers are pure HBM streaming — 232 MB + 129 MB of compulsory traffic for 6.6 + 6.6
// GFLOP — and the second one spends its whole life re-reading the 103 MB the first has just written.  Fused, Y goes from the
// accumulators through LDS (where it is the A operand of the second GEMM) to HBM once and is never read back: 251 MB instead of
// 361 MB, one launch instead of two.
//  * both weight matrices (N1 x K1 and N2 x N1, fp16: 32 + 32..64 KB) stay on the CU for the whole launch — as MFMA fragments in
//    REGISTERS when they fit (WREG: N2 = 64, 4-wave blocks: 32 + 64 VGPRs per lane; the block then needs only 40 KB of LDS and two
//    blocks share a CU, one streaming while the other multiplies), else in LDS; a block walks a contiguous range of BM-pixel tiles;
//  * per tile: X tile (BM x K1) registers -> LDS, GEMM1 (8 waves, 64 x 64 wave tiles), epilogue in registers (+ bias, v_permlane32_swap
//    -> 8 consecutive couts per lane, + residual, activation, ONE rounding), Y tile -> LDS [BM][N1] (16-byte chunks XOR-swizzled by
//    row & 15: 512-byte rows), then (a) every thread copies whole 512-byte rows LDS -> HBM with full-line 16-byte stores and (b)
//    GEMM2 reads the same tile as its A operand; its N2-wide result leaves through registers;
//  * the X tile and the residual of tile t+1 are fetched with ordinary loads while tile t is in its second half: nothing here needs
//    a hand-counted wait, and stores are never waited for;
//  * numerics: Y is rounded to fp16 exactly once (what the unfused pair stores and re-reads); Z accumulates in fp32 over k ascending.
#pragma once

template <int K1, int N1, int N2, int BM, int NW, bool WREG>
__global__ __launch_bounds__(NW * 64) void conv1x1_pair_kernel(ConvParams p, const half_t* w2, const float* bias2, half_t* z, int z_ld, int act2, float slope2) {
    static_assert(K1 == 64 && N1 == 256 && (N2 == 64 || N2 == 128) && BM == 64 && (NW == 4 || NW == 8), "ResNet stage-1 geometry");
    constexpr int NT = NW * 64;
    constexpr int WGM1 = BM / 64, WGN1 = NW / WGM1, TM1 = 2, TN1 = N1 / (WGN1 * 32);         // GEMM1: wave tile 64 pixels x (TN1*32) couts
    constexpr int WGM2 = BM / 32, WGN2 = NW / WGM2, TN2 = N2 / (WGN2 * 32);                   // GEMM2: wave tile 32 pixels x (TN2*32) couts
    static_assert(TN1 >= 1 && TN2 >= 1 && WGM1 * WGN1 == NW && WGM2 * WGN2 == NW, "wave layout");
    constexpr int W1_BYTES = WREG ? 0 : N1 * K1 * 2, W2_BYTES = WREG ? 0 : N2 * N1 * 2, X_BYTES = BM * K1 * 2, Y_BYTES = BM * N1 * 2;
    constexpr int C2 = N1 / 8;                                                                // 16-byte chunks per Y / W2 row
    extern __shared__ __attribute__((aligned(16))) unsigned char pr_smem[];
    unsigned char* const w1l = pr_smem;
    unsigned char* const w2l = w1l + W1_BYTES;
    unsigned char* const xl = w2l + W2_BYTES;
    unsigned char* const yl = xl + X_BYTES;
    float* const b1l = (float*)(yl + Y_BYTES);
    float* const b2l = b1l + N1;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const half_t* zero = (const half_t*)g_zero_page;
    const int wm1 = wave / WGN1, wn1 = wave % WGN1, wm2 = wave / WGN2, wn2 = wave % WGN2;

    // ---- weights: MFMA fragments in registers (WREG), or LDS images.  W1 rows are 128 B (chunk ^ ((row >> 1) & 7), as conv_igemm64.hpp),
    // W2 rows 512 B (chunk ^ (row & 15)); biases -> LDS
    half8_t w1f[WREG ? TN1 : 1][K1 / 16], w2f[WREG ? TN2 : 1][N1 / 16];
    if constexpr (WREG) {
#pragma unroll
        for (int a = 0; a < TN1; a++)
#pragma unroll
            for (int ks = 0; ks < K1 / 16; ks++) w1f[a][ks] = *(const half8_t*)(p.w + (long)((wn1 * TN1 + a) * 32 + l31) * K1 + (ks * 2 + hi) * 8);
#pragma unroll
        for (int a = 0; a < TN2; a++)
#pragma unroll
            for (int ks = 0; ks < N1 / 16; ks++) w2f[a][ks] = *(const half8_t*)(w2 + (long)((wn2 * TN2 + a) * 32 + l31) * N1 + (ks * 2 + hi) * 8);
    } else {
        for (int s0 = wave * 64; s0 < N1 * 8; s0 += NT) {
            const int s = s0 + lane, n = s >> 3, c = s & 7;
            SM_GLDS16(p.w + (long)n * K1 + ((c ^ ((n >> 1) & 7)) << 3), w1l + s0 * 16);
        }
        for (int s0 = wave * 64; s0 < N2 * C2; s0 += NT) {
            const int s = s0 + lane, n = s / C2, c = s % C2;
            SM_GLDS16(w2 + (long)n * N1 + ((c ^ (n & 15)) << 3), w2l + s0 * 16);
        }
    }
    for (int i = tid; i < N1; i += NT) b1l[i] = p.bias ? p.bias[i] : 0.f;
    for (int i = tid; i < N2; i += NT) b2l[i] = bias2 ? bias2[i] : 0.f;

    const int T = (p.M + BM - 1) / BM, G = gridDim.x;
    const int t_begin = (int)((long)blockIdx.x * T / G), t_end = (int)((long)(blockIdx.x + 1) * T / G);
    const float sl1 = p.act == SM_ACT_NONE ? 1.f : p.slope, sl2 = act2 == SM_ACT_NONE ? 1.f : slope2;

    // X tile staging: BM rows x 8 chunks = BM*8 slots, XPT per thread; residual: this lane's 8 output couts of (a, b, j)
    constexpr int XPT = BM * 8 / NT, NRR = TN1 * TM1 * 2;
    static_assert(XPT >= 1, "X tile");
    half8_t xr[XPT], rr[NRR];
    auto load_tile = [&](int t) {
        const int m0 = t * BM;
#pragma unroll
        for (int i = 0; i < XPT; i++) {
            const int s = i * NT + tid, row = s >> 3, c = s & 7;
            xr[i] = *(const half8_t*)(p.x + (long)min(m0 + row, p.M - 1) * p.x_ld + c * 8);
        }
#pragma unroll
        for (int a = 0; a < TN1; a++)
#pragma unroll
            for (int b = 0; b < TM1; b++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int m = min(m0 + (wm1 * TM1 + b) * 32 + l31, p.M - 1), co = (wn1 * TN1 + a) * 32 + 16 * j + 8 * hi;
                    rr[(a * TM1 + b) * 2 + j] = *(const half8_t*)(p.res ? p.res + (long)m * p.res_ld + co : zero);
                }
    };
    if (t_begin < t_end) load_tile(t_begin);
    if (!WREG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the weight DMAs (the compiler does not wait for LDS-DMA on its own)
    __syncthreads();

#pragma unroll 1
    for (int t = t_begin; t < t_end; t++) {
        const int m0 = t * BM;
        // ---- X tile -> LDS (swizzled like a conv_igemm64 A tile)
#pragma unroll
        for (int i = 0; i < XPT; i++) {
            const int s = i * NT + tid, row = s >> 3, c = s & 7;
            *(half8_t*)(xl + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = xr[i];
        }
        __syncthreads();
        // ---- GEMM1: acc1[a][b] over K1 = 64
        floatx16 acc1[TN1][TM1];
#pragma unroll
        for (int a = 0; a < TN1; a++)
#pragma unroll
            for (int b = 0; b < TM1; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc1[a][b][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K1 / 16; ks++) {
            const int co = (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
            half8_t fx[TM1], fw[TN1];
#pragma unroll
            for (int b = 0; b < TM1; b++) fx[b] = *(const half8_t*)(xl + ((wm1 * TM1 + b) * 32 + l31) * 128 + co);
#pragma unroll
            for (int a = 0; a < TN1; a++) { if constexpr (WREG) fw[a] = w1f[a][ks]; else fw[a] = *(const half8_t*)(w1l + ((wn1 * TN1 + a) * 32 + l31) * 128 + co); }
#pragma unroll
            for (int a = 0; a < TN1; a++)
#pragma unroll
                for (int b = 0; b < TM1; b++) acc1[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a], fx[b], acc1[a][b], 0, 0, 0);
        }
        // ---- epilogue 1 in registers -> Y tile in LDS
#pragma unroll
        for (int a = 0; a < TN1; a++) {
            floatx4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const floatx4*)(b1l + (wn1 * TN1 + a) * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int b = 0; b < TM1; b++) {
                const int row = (wm1 * TM1 + b) * 32 + l31;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float o[8];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        unsigned ua = __builtin_bit_cast(unsigned, acc1[a][b][4 * (2 * j) + r] + bq[2 * j][r]), ub = __builtin_bit_cast(unsigned, acc1[a][b][4 * (2 * j + 1) + r] + bq[2 * j + 1][r]);
                        sm_permlane32_swap(ua, ub);
                        o[r] = __builtin_bit_cast(float, ua); o[4 + r] = __builtin_bit_cast(float, ub);
                    }
                    const half8_t rv = rr[(a * TM1 + b) * 2 + j];
                    half8_t ov;
#pragma unroll
                    for (int r = 0; r < 8; r++) { float v = o[r]; if (p.res) v += (float)rv[r]; ov[r] = (half_t)(v < 0.f ? v * sl1 : v); }
                    const int chunk = (wn1 * TN1 + a) * 4 + 2 * j + hi;
                    *(half8_t*)(yl + row * (N1 * 2) + ((chunk ^ (row & 15)) << 4)) = ov;
                }
            }
        }
        if (t + 1 < t_end) load_tile(t + 1);                          // in flight during the second half of this tile (and the other block's first half)
        __syncthreads();
        // ---- (a) Y tile LDS -> HBM: whole 512-byte rows, 16 bytes per thread
        constexpr int YPT = BM * C2 / NT;
#pragma unroll
        for (int i = 0; i < YPT; i++) {
            const int s = i * NT + tid, row = s / C2, c = s % C2;
            const half8_t v = *(const half8_t*)(yl + row * (N1 * 2) + ((c ^ (row & 15)) << 4));
            if (m0 + row < p.M) *(half8_t*)(p.y + (long)(m0 + row) * p.y_ld + c * 8) = v;
        }
        // ---- (b) GEMM2: Z tile = Y tile x W2^T over N1 = 256
        floatx16 acc2[TN2];
#pragma unroll
        for (int a = 0; a < TN2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[a][r] = 0.f;
        const int row2 = wm2 * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < N1 / 16; ks++) {
            const int c = ks * 2 + hi;
            const half8_t fy = *(const half8_t*)(yl + row2 * (N1 * 2) + ((c ^ (row2 & 15)) << 4));
#pragma unroll
            for (int a = 0; a < TN2; a++) {
                half8_t fw;
                if constexpr (WREG) fw = w2f[a][ks];
                else { const int n = (wn2 * TN2 + a) * 32 + l31; fw = *(const half8_t*)(w2l + n * (N1 * 2) + ((c ^ (n & 15)) << 4)); }
                acc2[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fy, acc2[a], 0, 0, 0);
            }
        }
        // ---- epilogue 2 in registers -> Z
        const int mz = m0 + row2;
#pragma unroll
        for (int a = 0; a < TN2; a++) {
            floatx4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const floatx4*)(b2l + (wn2 * TN2 + a) * 32 + 8 * q + 4 * hi);
            unsigned d[4][2];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    float v0 = acc2[a][4 * q + 2 * h2] + bq[q][2 * h2], v1 = acc2[a][4 * q + 2 * h2 + 1] + bq[q][2 * h2 + 1];
                    v0 = v0 < 0.f ? v0 * sl2 : v0; v1 = v1 < 0.f ? v1 * sl2 : v1;
                    const half2_t hv = {(half_t)v0, (half_t)v1};
                    d[q][h2] = __builtin_bit_cast(unsigned, hv);
                }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                if (mz < p.M) *(u32x4*)(z + (long)mz * z_ld + (wn2 * TN2 + a) * 32 + 16 * j + 8 * hi) = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
            }
        }
        __syncthreads();                                              // Y and X tiles are free again
    }
}
