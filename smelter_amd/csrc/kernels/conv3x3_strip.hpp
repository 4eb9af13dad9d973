// conv3x3_strip.hpp — 3x3 / stride 1 / pad 1 / dilation 1 convolution with Cin % 64 == 0 (13 of ResNet-50's 16 3x3 layers, every
// U-Net layer): the implicit-GEMM kernel of conv_igemm64.hpp re-fetches the activation tile for each of the 9 taps, although the
// three taps of one kernel row read the SAME input pixels shifted by one.  Included by conv_igemm.hip (same ConvParams / launcher).
//
// With stride 1 and "same" padding the input pixel of output pixel m (linear index over n, oy, ox) under tap (kh, kw) is simply
// m + (kh-1)*W + (kw-1).  So for a tile of BM consecutive output pixels and one kernel row kh, ONE strip of BM+2 consecutive input
// pixels (x 64 channels = 128 B each) serves all three kw: tap kw reads strip rows r + kw.  A ring stage is therefore
//     [ strip(kh, cb): BM+2 rows ][ W(kh,0,cb): BN rows ][ W(kh,1,cb) ][ W(kh,2,cb) ]        (rows of 128 B, XOR-swizzled chunks)
// = 3 K tiles of MFMA work for (BM+2 + 3 BN) rows of LDS-DMA instead of 3 (BM + BN): the activation traffic into the LDS — the
// resource that bounds this kernel family — drops 3x, and there is one barrier per 3 K tiles.
//  * vertical padding: a strip pixel whose own image row is H-1 (strip kh = 0) or 0 (kh = 2) can only be read by taps that fall
//    outside the image, so it is fetched from the zero page; pixels beyond either end of the tensor likewise.
//  * horizontal padding cannot be decided per strip row (the same pixel is a valid tap of its neighbours): the A fragments of
//    kw = 0 / kw = 2 are zeroed per lane for output pixels with ox == 0 / ox == W-1 (v_cndmask next to the MFMAs).
//  * K order is (kh, cin block, kw) instead of (kh, kw, cin block): same products, different fp32 summation order.
//  * epilogue = the straight-line fast path of conv_igemm64.hpp (dense 16-byte output, ReLU family, bias + residual prefetched
//    with exact vmcnt accounting); the launcher only selects this kernel when those conditions hold.
#pragma once

// LEAN = 1: the strip is allocated BM+2 rows (the tail DMA runs with 16 active lanes) and the prologue's row table aliases the
// last ring stage (free until the first loop iteration): 128x64 / 2 stages = 81,280 B, i.e. TWO blocks per CU for the N = 64 layers.
template <int BM, int BN, int WGM, int WGN, int NSTAGE, int LEAN = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void conv3x3_strip_kernel(ConvParams p) {
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves");
    static_assert(NSTAGE >= 2 && NSTAGE <= 3, "2..3 stages");
    constexpr int NT = WGM * WGN * 64;
    constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
    constexpr int RPI = NT / 8;                                 // tile rows covered by one DMA instruction of the whole block
    static_assert(BM % RPI == 0, "strip geometry");
    constexpr int A_FULL = BM / RPI;                            // whole-block DMA instructions of the strip; + 1 tail (2 rows) by wave 0
    constexpr int B_IT = BN * 8 / NT;
    static_assert(TM >= 1 && TN >= 1 && B_IT >= 1, "tile too small for the wave layout");
    constexpr int LOADS = A_FULL + 3 * B_IT;                    // per wave per stage (wave 0: + 1)
    constexpr int A_ROWS = LEAN ? BM + 2 : BM + 8;              // strip rows allocated (the tail instruction writes 8 rows, or 2 with 16 active lanes)
    constexpr int A_BYTES = A_ROWS * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + 3 * B_BYTES, PIPE_BYTES = NSTAGE * STAGE_BYTES;
    constexpr int EP_LD = BN + 4, WROWS = 32 * TM;
    constexpr int EP_BUDGET = PIPE_BYTES > 69632 ? PIPE_BYTES : 69632;
    constexpr int WPP_MAX = EP_BUDGET / (WROWS * EP_LD * 4);
    constexpr int WPP = WPP_MAX >= WGM ? WGM : sm_pow2_floor(WPP_MAX < 1 ? 1 : WPP_MAX);
    constexpr int EP_ROWS = WROWS * WPP, EP_PASSES = WGM / WPP;
    static_assert(WGM % WPP == 0, "epilogue passes");
    constexpr int TABLE_BYTES = (BM + 8) * 8;                   // strip-row table of the prologue
    constexpr int LDS_BYTES = (PIPE_BYTES > EP_ROWS * EP_LD * 4 ? PIPE_BYTES : EP_ROWS * EP_LD * 4) + (LEAN ? 0 : TABLE_BYTES);
    static_assert(!LEAN || (TABLE_BYTES <= STAGE_BYTES && NSTAGE == 2), "lean: the table lives in ring stage 1");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    constexpr int CHUNKS = BN / 8, ITEMS = EP_ROWS * CHUNKS, IPT = (ITEMS + NT - 1) / NT;
    constexpr int EXTRA = 2 + EP_PASSES * IPT;                  // ordinary loads behind the prologue stages (bias + residual)
    static_assert(NT % CHUNKS == 0, "a thread keeps its channel chunk across items");
    static_assert((NSTAGE - 1) * (LOADS + 1) + EXTRA <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[LDS_BYTES];

    int mtile, ntile, ksplit;
    if (!map_block(p, mtile, ntile, ksplit)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int csrc = (tid & 7) ^ ((tid >> 4) & 7);             // source chunk for this thread's LDS slot (rows j*RPI + tid/8)
    const half_t* zero = (const half_t*)g_zero_page;
    const int cpt = p.Cin >> 6, nit = 3 * cpt;                  // ring stages = (kernel row, cin block)
    const int m0 = mtile * BM;

    // ---- strip-row table: row s <-> input pixel m0 + s - 1 of the kh = 1 strip; bit k = "fetch it for kernel row k" ------------
    {
        int2* tab = (int2*)(smem_raw + LDS_BYTES - TABLE_BYTES);
        for (int s = tid; s < BM + 8; s += NT) {
            const int HW = p.H * p.W, total = p.N * HW;
            const int p1 = m0 + s - 1;
            unsigned bits = 0;
            if (s < BM + 2) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int pk = p1 + (k - 1) * p.W;
                    if (pk >= 0 && pk < total) {
                        const int n = (int)sm_fastdiv((unsigned)pk, p.dhw_mul, p.dhw_sh), r = pk - n * HW;
                        const int y = (int)sm_fastdiv((unsigned)r, p.dw_mul, p.dw_sh);
                        const bool ok = k == 0 ? y != p.H - 1 : (k == 2 ? y != 0 : true);
                        bits |= (ok ? 1u : 0u) << k;
                    }
                }
            }
            tab[s] = int2{p1, (int)bits};
        }
        __syncthreads();
    }
    const half_t* arow[A_FULL + 1]; unsigned abits[A_FULL + 1];
    {
        const int2* tab = (const int2*)(smem_raw + LDS_BYTES - TABLE_BYTES);
#pragma unroll
        for (int j = 0; j <= A_FULL; j++) {
            const int s = j * RPI + (tid >> 3);                  // j == A_FULL: only wave 0 (rows BM .. BM+7) is used
            const int2 ri = tab[s < BM + 8 ? s : 0];
            arow[j] = p.x + (long)ri.x * p.x_ld + csrc * 8; abits[j] = (unsigned)ri.y;
        }
    }
    const half_t* brow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
        int n = ntile * BN + j * RPI + (tid >> 3); if (n >= p.Cout) n = p.Cout - 1;
        brow[j] = p.w + (long)n * p.K + csrc * 8;
    }
    // horizontal-padding masks of this lane's output pixels: kw = 0 is outside for ox == 0, kw = 2 for ox == W-1
    bool xl[TM], xr[TM];
#pragma unroll
    for (int b = 0; b < TM; b++) {
        const int m = m0 + wm * (TM * 32) + b * 32 + (lane & 31);
        const int q = (int)sm_fastdiv((unsigned)m, p.dw_mul, p.dw_sh), ox = m - q * p.W;
        xl[b] = ox == 0; xr[b] = ox == p.W - 1;
    }

    // ---- cursor of the next stage to issue (wave-uniform) -----------------------------------------------------------------
    int kh_i = 0, cb_i = 0;
    long a_off = -(long)p.W * p.x_ld;                           // (kh - 1) * W pixels + cb * 64 channels, relative to the kh = 1 strip
    long b_off = 0;                                             // (kh*3 + 0) * Cin + cb * 64
    const long tapstep = p.Cin;                                 // kw -> kw + 1 inside a weight row
    auto issue = [&](int slot) {
        unsigned char* st = smem_raw + slot * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < A_FULL; j++) {
            const half_t* src = ((abits[j] >> kh_i) & 1u) ? arow[j] + a_off : zero;
            SM_GLDS16(src, st + wave * 1024 + j * NT * 16);
        }
        if (wave == 0) {                                        // tail: strip rows BM .. BM+7 (BM, BM+1 are real)
            const half_t* src = ((abits[A_FULL] >> kh_i) & 1u) ? arow[A_FULL] + a_off : zero;
            if (!LEAN || lane < 16) SM_GLDS16(src, st + A_FULL * NT * 16);   // lean: only the 2 real rows are written
        }
#pragma unroll
        for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int j = 0; j < B_IT; j++) SM_GLDS16(brow[j] + b_off + kw * tapstep, st + A_BYTES + kw * B_BYTES + wave * 1024 + j * NT * 16);
        if (++cb_i == cpt) { cb_i = 0; ++kh_i; a_off += (long)p.W * p.x_ld - (long)(cpt - 1) * 64; b_off += 3 * tapstep - (long)(cpt - 1) * 64; }
        else { a_off += 64; b_off += 64; }
    };

    int issued = 0;
    const int pre = nit <= NSTAGE ? nit : NSTAGE - 1;
#pragma unroll
    for (int s = 0; s < NSTAGE; s++) if (s < pre) { issue(s); ++issued; }
    int slot = 0, islot = pre == NSTAGE ? 0 : pre;
    __builtin_amdgcn_sched_barrier(0);

    // ---- EXTRA ordinary loads (always issued; zero page when there is nothing to fetch): epilogue bias + residual ------------
    floatx4 eb0, eb1;
    {
        const float* bp = p.bias ? p.bias + min(ntile * BN + (tid % CHUNKS) * 8, p.Cout - 8) : (const float*)g_zero_page;
        eb0 = *(const floatx4*)bp; eb1 = *(const floatx4*)(bp + 4);
    }
    half8_t rres[EP_PASSES][IPT];
#pragma unroll
    for (int pw = 0; pw < EP_PASSES; pw++)
#pragma unroll
        for (int k2 = 0; k2 < IPT; k2++) {
            int it = tid + k2 * NT; if (ITEMS % NT != 0 && it >= ITEMS) it = ITEMS - 1;
            const int row = it / CHUNKS, ch = it - row * CHUNKS;
            const int m = min(m0 + pw * EP_ROWS + row, p.M - 1), co = min(ntile * BN + ch * 8, p.Cout - 8);
            rres[pw][k2] = *(const half8_t*)(p.res ? p.res + (long)m * p.res_ld + co : zero);
        }
    __builtin_amdgcn_sched_barrier(0);

    floatx16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // fragment addresses.  Weights: row (lane & 31), chunk (2 ks + lane/32) ^ swz(row).  Strip: row (lane & 31) + kw, swizzled by ITS row.
    unsigned voffB[4], voffA[3][4];
    {
        const unsigned swb = ((lane & 31) >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) voffB[ks] = (lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ swb) << 4);
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            const unsigned row = (lane & 31) + kw, swa = (row >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) voffA[kw][ks] = row * 128 + (((ks * 2 + (lane >> 5)) ^ swa) << 4);
        }
    }
    const half8_t hz = {0, 0, 0, 0, 0, 0, 0, 0};
    auto compute = [&](int slot) {
        const unsigned sA = slot * STAGE_BYTES + wm * (TM * 4096), sB = slot * STAGE_BYTES + A_BYTES + wn * (TN * 4096);
        half8_t fx[2][TM], fw[2][TN];
        auto fetch = [&](int step, int buf) {                   // step = kw * 4 + ks
            const int kw = step >> 2, ks = step & 3;
#pragma unroll
            for (int b = 0; b < TM; b++) fx[buf][b] = *(const half8_t*)(smem_raw + sA + voffA[kw][ks] + b * 4096);
#pragma unroll
            for (int a = 0; a < TN; a++) fw[buf][a] = *(const half8_t*)(smem_raw + sB + kw * B_BYTES + voffB[ks] + a * 4096);
        };
        fetch(0, 0);
#pragma unroll
        for (int step = 0; step < 12; step++) {
            if (step + 1 < 12) fetch(step + 1, (step + 1) & 1);
            const int kw = step >> 2;
#pragma unroll
            for (int b = 0; b < TM; b++) {
                half8_t xv = fx[step & 1][b];
                if (kw == 0) xv = xl[b] ? hz : xv;
                if (kw == 2) xv = xr[b] ? hz : xv;
#pragma unroll
                for (int a = 0; a < TN; a++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[step & 1][a], xv, acc[a][b], 0, 0, 0);
            }
        }
    };

    // ---- main loop: ONE barrier per stage (= 3 K tiles); counted vmcnt (wave 0 has one more DMA per stage) -----------------
    auto wait_for = [&](int newer, bool early) {
        // allowed in flight: `newer` younger stages (+ the EXTRA ordinary loads if this stage was issued in the prologue)
#define SM_STRIP_WAIT(N) do { if (wave == 0) { if (early) sm_wait_vmcnt<(N) * (LOADS + 1) + EXTRA>(); else sm_wait_vmcnt<(N) * (LOADS + 1)>(); } \
                               else { if (early) sm_wait_vmcnt<(N) * LOADS + EXTRA>(); else sm_wait_vmcnt<(N) * LOADS>(); } } while (0)
        if (NSTAGE >= 3 && newer >= 2) SM_STRIP_WAIT(NSTAGE >= 3 ? 2 : 0);
        else if (newer == 1) SM_STRIP_WAIT(1);
        else SM_STRIP_WAIT(0);
#undef SM_STRIP_WAIT
    };
    for (int i = 0; i < nit; i++) {
        wait_for(issued - i - 1, i < pre);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (issued < nit) { issue(islot); ++issued; if (++islot == NSTAGE) islot = 0; }
        compute(slot);
        if (++slot == NSTAGE) slot = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // ---- epilogue (fast path only) ---------------------------------------------------------------------------------------------
    float* stg = (float*)smem_raw;
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
#pragma unroll
    for (int pw = 0; pw < EP_PASSES; pw++) {
        if (wm / WPP == pw) {
            const int wrow = (wm % WPP) * WROWS;
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
        floatx4 s0[IPT], s1[IPT]; int rowi[IPT], chi[IPT];
#pragma unroll
        for (int k2 = 0; k2 < IPT; k2++) {
            int it = tid + k2 * NT; if (ITEMS % NT != 0 && it >= ITEMS) it = ITEMS - 1;
            rowi[k2] = it / CHUNKS; chi[k2] = it - rowi[k2] * CHUNKS;
            s0[k2] = *(const floatx4*)(stg + rowi[k2] * EP_LD + chi[k2] * 8); s1[k2] = *(const floatx4*)(stg + rowi[k2] * EP_LD + chi[k2] * 8 + 4);
        }
#pragma unroll
        for (int k2 = 0; k2 < IPT; k2++) {
            const int m = m0 + pw * EP_ROWS + rowi[k2], co = ntile * BN + chi[k2] * 8;
            float v[8] = {s0[k2][0] + eb0[0], s0[k2][1] + eb0[1], s0[k2][2] + eb0[2], s0[k2][3] + eb0[3],
                          s1[k2][0] + eb1[0], s1[k2][1] + eb1[1], s1[k2][2] + eb1[2], s1[k2][3] + eb1[3]};
            if (p.res) { const half8_t rr = rres[pw][k2];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += (float)rr[j]; }
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (half_t)(v[j] < 0.f ? v[j] * sl : v[j]);
            const bool live = (ITEMS % NT == 0 || tid + k2 * NT < ITEMS) && m < p.M && co < p.Cout;
            if (live) *(half8_t*)(p.y + (long)m * p.y_ld + co) = o;
        }
        if (pw + 1 < EP_PASSES) __syncthreads();
    }
}
