// conv3x3_p9.hpp — 3x3 / stride 1 / pad 1 convolution, Cin % 64 == 0, any Cout: ONE activation patch in LDS serves all NINE taps.
// Included by conv_igemm.hip (same ConvParams / launcher).  The MFMA-heavy 3x3 layers (ResNet-50 stages 2-4: K = 1152..4608; the
// U-Net below full resolution) are not HBM- but LDS-bound in the implicit-GEMM kernels: per 256x128x64 K tile the LDS takes 48 KB of
// DMA writes (64 B/clk) + 128 KB of fragment reads (256 B/clk) = 1280 clk against 1024 clk of MFMAs, and the two do not overlap
// (DESIGN.md §4.1b).  The strip kernel shares one activation strip between the 3 taps of a kernel row; this kernel goes all the way:
//  * K order (cin block, kh, kw).  Per 64-channel block the block's input PATCH — every input row its BM output pixels touch, PW =
//    TW + 2 pixels wide, zero page outside the image — is copied to LDS once (double-buffered, one LDS-DMA instruction per thread
//    per K tile, issued while the previous block's nine taps are multiplied) and read by all nine taps: tap (kh,kw) of a pixel with
//    patch slot q is slot q + kh*PW + kw.  DMA bytes per K tile: BN*128 (weights) + patch/9 = 21 KB instead of 48 KB at 256x128.
//  * tiles are BM CONSECUTIVE pixels of a column band in (image, row, column) order — TW = W for small images (a tile then spans
//    several rows, even several 7x7 images: each image contributes its own two halo rows to the patch), TW = 32 for wide ones —
//    so every tile is full whatever H and W are.
//  * weights: a ring of NSB tiles [BN][64] exactly as in conv_igemm64.hpp (source-side XOR swizzle, one barrier per K tile, counted
//    vmcnt).  Every K tile issues the SAME number of DMA instructions (B_IT + 1; out-of-range pieces go to a dummy sink), so the
//    counted waits are exact without a branch in the loop.
//  * fragment reads and DMAs are asm statements with hand-counted lgkmcnt / vmcnt (see conv3x3_ws.hpp for why).
//  * epilogue in registers: bias + activation, one rounding, v_permlane32_swap -> 16-byte stores; no LDS staging, no barrier.
//  * optional fused nearest 2x upsampling of the INPUT (p.ups): the patch is gathered from the half-resolution tensor, the
//    upsampled image never exists (MPSCNNUpsamplingNearestNode in front of a convolution, Converters.swift:524).
#pragma once

// POOL: the 2x2 / stride-2 max-pool of the output is a second output (p.y2) of the same launch — a wave's two fragments are image rows 2k, 2k + 1 of one 32-column band (TW = 32, tiles
// of 8 whole rows: the launcher checks), neighbouring columns are neighbouring lanes: two packed maxima per 32-bit word, the even lanes store.
template <int BM, int BN, int WGM, int WGN, int NSB = 3, bool POOL = false>
__global__ __launch_bounds__(WGM * WGN * 64) void conv3x3_p9_kernel(ConvParams p) {
    constexpr int NW = WGM * WGN, NT = NW * 64, TM = BM / (WGM * 32), TN = BN / (WGN * 32);
    constexpr int B_IT = BN * 8 / NT, B_BYTES = BN * 128;
    constexpr int NIC = 11 - NSB;                                     // patch pieces per thread: taps 0 .. 9-(NSB-1) carry one each (see the wait accounting below)
    static_assert(NW == 8 && TM >= 1 && TN >= 1 && B_IT >= 1 && NSB >= 2 && NSB <= 4, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char p9_smem[];
    const unsigned lds0 = ws_lds_addr(p9_smem);
    const unsigned dummy_a = lds0, patch_a = lds0 + WS_DUMMY, bring_a = patch_a + 2 * p.p9_patch_bytes;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const half_t* zero = (const half_t*)g_zero_page;
    const int H = p.H, W = p.W, TW = p.p9_TW, PW = TW + 2;
    // block -> (column band, pixel tile of the band, cout tile); cout tiles of one pixel tile are neighbours (they share the patch in L2)
    const int nt = p.nt, mtile_all = blockIdx.x / nt, ntile = blockIdx.x - mtile_all * nt;
    const int band = mtile_all / p.p9_tiles_per_band, mtile = mtile_all - band * p.p9_tiles_per_band;
    const int x0 = band * TW;
    const long f0 = (long)mtile * BM, ftotal = (long)p.N * H * TW;    // band-flat pixel index f = R * TW + xx, R = image * H + row
    const int R_first = (int)(f0 / TW), n_first = R_first / H, oy_first = R_first - n_first * H;

    // ---- per-thread DMA descriptors ---------------------------------------------------------------------------------------------
    // patch piece i: LDS slot s = i*NT + tid -> patch pixel ps = s/8 (row prow, column pc), chunk s%8; prow -> (image, input row)
    const half_t* psrc[NIC]; bool pin[NIC];
    const int Hs = p.ups ? H >> 1 : H, Ws = p.ups ? W >> 1 : W;
    const int c0 = H - oy_first + 2;                                  // patch rows of the first image: input rows oy_first-1 .. H
#pragma unroll
    for (int i = 0; i < NIC; i++) {
        const int s = i * NT + tid, ps = s >> 3, c = s & 7;
        const int prow = (int)sm_fastdiv((unsigned)ps, p.p9_pw_mul, p.p9_pw_sh), pc = ps - prow * PW;
        int n, iy;
        if (prow < c0) { n = n_first; iy = oy_first - 1 + prow; }
        else { const int q = (int)sm_fastdiv((unsigned)(prow - c0), p.p9_h2_mul, p.p9_h2_sh); n = n_first + 1 + q; iy = prow - c0 - q * (H + 2) - 1; }
        const int ix = x0 - 1 + pc;
        pin[i] = ps < p.p9_patch_px && n < p.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int sy = p.ups ? iy >> 1 : iy, sx = p.ups ? ix >> 1 : ix;
        psrc[i] = p.x + (((long)n * Hs + sy) * Ws + sx) * p.x_ld + ((c ^ ws_swz<8>(ps)) << 3);
    }
    const half_t* brow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
        int n = ntile * BN + j * (NT / 8) + (tid >> 3); if (n >= p.Cout) n = p.Cout - 1;
        brow[j] = p.w + (long)n * p.K + (((tid & 7) ^ ((tid >> 4) & 7)) << 3);
    }
    const int cblocks = p.Cin >> 6, NK = 9 * cblocks;
    auto issue_b = [&](int kt) {                                      // weight tile kt = (cin block, tap) -> ring stage kt % NSB; beyond the end: dummy
        const bool real = kt < NK;
        const int cb = kt / 9, tap = kt - cb * 9;
        const int koff = tap * p.Cin + cb * 64;
        const unsigned st = (unsigned)__builtin_amdgcn_readfirstlane((int)(real ? bring_a + (kt % NSB) * B_BYTES + wave * 1024 : dummy_a));
#pragma unroll
        for (int j = 0; j < B_IT; j++) ws_glds16(real ? brow[j] + koff : zero, real ? st + j * NT * 16 : dummy_a);
    };
    auto issue_piece = [&](int i, int cb) {                           // patch piece i of cin block cb (cb >= cblocks: dummy)
        const bool real = cb < cblocks;
        const bool mine = real && i * NT + wave * 64 < p.p9_patch_px * 8;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? patch_a + (cb & 1) * p.p9_patch_bytes + i * NT * 16 + wave * 1024 : dummy_a));
        ws_glds16((mine && pin[i]) ? psrc[i] + cb * 64 : zero, dst);
    };

    // ---- DMA order.  Iteration j of the main loop issues [patch piece of the NEXT cin block][weight tile j+NSB-1]; LDS-DMAs complete
    // in order, so when the wait of iteration i has retired weight tile i (issued in iteration i-NSB+1) everything issued before it
    // has landed too: a patch piece issued at tap t is in LDS from tap t+NSB-1 on, i.e. pieces may ride on taps 0 .. 9-(NSB-1) and the
    // patch is complete at tap 0 of the next cin block.  Later taps issue a dummy piece: every iteration issues EXACTLY B_IT + 1
    // instructions and `vmcnt((NSB-2)*(B_IT+1))` is exact.  The prologue mimics the pattern.
#pragma unroll
    for (int i = 0; i < NIC; i++) issue_piece(i, 0);
#pragma unroll
    for (int j = 0; j < NSB - 1; j++) { ws_glds16(zero, dummy_a); issue_b(j); }

    // ---- per-lane fragment slots (tile-invariant part of the A addresses) and output pixels ---------------------------------------
    // (TM = 4: 144 precomputed fragment addresses do not fit next to 128 accumulator registers — the big tile derives them per tap)
    constexpr bool A_RT = TM > 2;
    unsigned aoff[A_RT ? 1 : TM][A_RT ? 1 : 9][4]; int slotq[TM]; long opix[TM]; bool ovalid[TM];
    long ppix = 0;                                                    // (POOL) the pooled pixel this lane's column pair / this wave's row pair lands on
#pragma unroll
    for (int b = 0; b < TM; b++) {
        const long f = f0 + (wm * TM + b) * 32 + l31;
        const bool in = f < ftotal;
        const long fc = in ? f : ftotal - 1;
        const int R = (int)(fc / TW), xx = (int)(fc - (long)R * TW);
        const int prow0 = (R - R_first) + 2 * (R / H - n_first);
        const int ox = x0 + xx;
        ovalid[b] = in && ox < W;
        opix[b] = (long)R * W + ox;
        if constexpr (POOL) { if (b == 0) { const int n = R / H, row = R - n * H; ppix = ((long)n * (H >> 1) + (row >> 1)) * (W >> 1) + (ox >> 1); } }
        slotq[b] = prow0 * PW + xx;
        if constexpr (!A_RT) {
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int q = slotq[b] + (tap / 3) * PW + tap % 3;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) aoff[b][tap][ks] = patch_a + (unsigned)(q * 128 + (((ks * 2 + hi) ^ ws_swz<8>(q)) << 4));
            }
        }
    }
    unsigned boffv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) boffv[ks] = bring_a + (unsigned)((wn * TN * 32 + l31) * 128 + (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4));

    floatx16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // ---- main loop: cin blocks x 9 taps; ONE barrier per K tile ---------------------------------------------------------------------
    constexpr int VM_AHEAD = (NSB - 2) * (B_IT + 1);                  // DMA instructions younger than weight tile kt when its wait is reached
    constexpr int RD = TM + TN, PF = 2;                               // fragment reads per k step; k steps of prefetch inside a K tile
    static_assert(PF * RD <= 15, "lgkmcnt field");
#pragma unroll 1
    for (int cb = 0; cb < cblocks; cb++) {
        const unsigned pbuf = (unsigned)((cb & 1) * p.p9_patch_bytes);
        sm_static_for<0, 9>([&](auto tc) {
            constexpr int TAP = decltype(tc)::value;
            const int kt = cb * 9 + TAP;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AHEAD) : "memory");
            __builtin_amdgcn_s_barrier();                             // weight tile kt (and, at tap 0, this cin block's patch) landed everywhere; stage (kt-1) % NSB is free
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TAP < NIC) issue_piece(TAP, cb + 1); else ws_glds16(zero, dummy_a);
            issue_b(kt + NSB - 1);
            const unsigned sb = (unsigned)((kt % NSB) * B_BYTES);
            half8_t fx[PF + 1][TM], fw[PF + 1][TN];
            unsigned abase[A_RT ? TM : 1]; int asw[A_RT ? TM : 1];
            if constexpr (A_RT) {
#pragma unroll
                for (int b = 0; b < TM; b++) { const int q = slotq[b] + (TAP / 3) * PW + TAP % 3; abase[b] = patch_a + pbuf + (unsigned)(q * 128); asw[b] = ws_swz<8>(q); }
            }
            auto fetch = [&](auto kc) {
                constexpr int KSI = decltype(kc)::value, SLOT = KSI % (PF + 1);
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    half8_t& d = fx[SLOT][b];
                    unsigned ad;
                    if constexpr (A_RT) ad = abase[b] + (unsigned)(((KSI * 2 + hi) ^ asw[b]) << 4); else ad = aoff[b][TAP][KSI] + pbuf;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad));
                }
#pragma unroll
                for (int a = 0; a < TN; a++) { half8_t& d = fw[SLOT][a]; const unsigned ad = boffv[KSI] + sb; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(a * 4096)); }
            };
            sm_static_for<0, PF>([&](auto kc) { fetch(kc); });
            sm_static_for<0, 4>([&](auto kc) {
                constexpr int KSI = decltype(kc)::value;
                if constexpr (KSI + PF < 4) fetch(std::integral_constant<int, KSI + PF>());
                constexpr int AHEAD = (3 - KSI < PF ? 3 - KSI : PF) * RD;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[KSI % (PF + 1)][a], fx[KSI % (PF + 1)][b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }

    // ---- epilogue in registers: register 4q+r of acc[a][b] <-> cout (wn*TN+a)*32 + 8q + 4*hi + r of pixel (wm*TM+b)*32 + (lane & 31) ----
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    const bool relu0 = p.act != SM_ACT_NONE && sl == 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int co0 = ntile * BN + (wn * TN + a) * 32;
        floatx4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = co0 + 8 * q + 4 * hi;
            bq[q] = (p.bias && co + 3 < p.Cout) ? *(const floatx4*)(p.bias + co) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
        unsigned pm[4][2];                                            // (POOL) row-pair maximum of this lane's column, packed halves
#pragma unroll
        for (int b = 0; b < TM; b++) {
            unsigned d[4][2];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    float v0 = acc[a][b][4 * q + 2 * h2] + bq[q][2 * h2], v1 = acc[a][b][4 * q + 2 * h2 + 1] + bq[q][2 * h2 + 1];
                    if (!relu0) { v0 = v0 < 0.f ? v0 * sl : v0; v1 = v1 < 0.f ? v1 * sl : v1; }
                    half2_t hv = {(half_t)v0, (half_t)v1};
                    if (relu0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});   // plain ReLU: round, then one packed maximum (the same values: rounding is monotone)
                    d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    if constexpr (POOL) {
                        if (b == 0) pm[q][h2] = d[q][h2];
                        else pm[q][h2] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(half2_t, pm[q][h2]), hv));
                    }
                }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                const int co = co0 + 16 * j + 8 * hi;                 // this lane's 8 consecutive couts
                if (ovalid[b] && co < p.Cout) *(u32x4*)(p.y + opix[b] * p.y_ld + co) = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
            }
        }
        if constexpr (POOL) {
            static_assert(!POOL || TM == 2, "a wave's two fragments are the two rows of a pooling window");
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {                      // the neighbouring column = the neighbouring lane (quad_perm 1, 0, 3, 2)
                    const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)pm[q][h2], (int)pm[q][h2], 0xB1, 0xf, 0xf, false);
                    pm[q][h2] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(half2_t, pm[q][h2]), __builtin_bit_cast(half2_t, nb)));
                }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                sm_permlane32_swap(pm[2 * j][0], pm[2 * j + 1][0]);
                sm_permlane32_swap(pm[2 * j][1], pm[2 * j + 1][1]);
                const int co = co0 + 16 * j + 8 * hi;
                if (ovalid[0] && ovalid[1] && !(l31 & 1) && co < p.Cout) *(u32x4*)(p.y2 + ppix * p.y2_ld + co) = u32x4{pm[2 * j][0], pm[2 * j][1], pm[2 * j + 1][0], pm[2 * j + 1][1]};
            }
        }
    }
}
