// bneck128.hpp — the bottleneck block with a 128-channel middle (ResNet-50's 28 x 28 stage: 512 -> 128 -> 128 -> 512) as ONE launch; included by bneck.hip.
//
//     T1 = act1(w1 . x  + b1)            1x1 "reduce",  C0 -> 128           (MPSCNNConvolutionNode, /root/reference/Sources/Smelter/Converters.swift:187-338)
//     T2 = act2(w2 * T1 + b2)            3x3 / stride 1 / pad 1, 128 -> 128 (padding rule ONNXConvolutionPadding.swift:73-79: zeros outside the image)
//     y  = act3(w3 . T2 + b3 + res)      1x1 "expand",  128 -> n3, + residual (MPSNNAdditionNode, Converters.swift:430-451), ReLU family (:342-359)
//
// Layer by layer such a block moves 205 MB at bs 64 (x in, T1 out and back, T2 out and back, x again as residual, y out) in three launches of 15 + 23 + 22 us, each at its
// own memory or fill roofline; as one pass: x once (+ two halo rows per tile), the residual, y.  The weights (128 + 288 + 128 KB) cannot live in the LDS as the 64-channel
// block's do (bneck_kernel above) — so the tile is as large as the chip allows, ONE per CU (a band of TH rows x the whole width: 7 x 28 = 196 pixels; 64 images x 4 bands =
// 256 blocks), every weight is read exactly once per block, straight from global memory into MFMA A fragments (v_mfma_f32_16x16x32_f16: four lanes x 16 bytes = 64
// contiguous bytes of a weight row per instruction), and the three layers are three PHASES of the block's four waves (one per SIMD: 512 registers each):
//   1. reduce: x passes the LDS in 64-channel stages of (TH + 2) x W pixels (LDS-DMA, two stages: the copy of the next runs under the MFMAs of this one — every wave needs
//      every pixel, so it is staged once instead of loaded four times); wave w owns couts [32 w, 32 w + 32) of all <= 256 patch pixels (128 accumulator registers).  The
//      result — + b1, activation, one rounding — is written into the patch of the 3x3 in conv3x3_kq.hpp's slot format (rows RS = W + 1 .. slots apart: the zero columns
//      between rows are the left / right halo; rows outside the image stay the zeros the patch was cleared to: the 3x3's padding, not act1(b1)).
//   2. 3x3: conv3x3_kq.hpp's main loop without its DMA: wave w owns couts [32 w, 32 w + 32) of the tile's 13 pixel fragments, weights two units ahead in a register ring,
//      pixel fragments one unit ahead from the LDS; no barrier.
//   3. expand: T2 (+ b2, activation, one rounding) goes to LDS [pixel][128 channels] (over the dead x stages); wave w owns couts [n3 / 4 * w, ..) in passes of 32: the
//      weight rows are dealt to the two fragments so that a lane ends up with EIGHT consecutive couts of its pixel (conv3x3_kq.hpp's XCH order) — residual loads and stores
//      are 16 bytes per lane, 64 contiguous bytes per pixel and instruction.
// Numerics: every layer accumulates in fp32 over k ascending (MFMA steps of 32) from zero, + bias in fp32, activation, ONE rounding to fp16 of both hidden tensors (they stay
// in the parity oracle's rounding list).
#pragma once
// (output stores: sc1 — write through; same-box A/B of the four policies: 56.0 / 56.4 us against 61 (default), 57 (sc0 sc1), 58 (nt))
#ifndef BN2_ST_AUX
#define BN2_ST_AUX 16
#endif
#ifndef BN2_LDX_AUX                                                    // cache policy of the x stream (LDS-DMA: an assembler suffix, "" | " nt" | " sc1") and of the residual rows (aux operand: 0 | 2 = nt | 16 = sc1)
#define BN2_LDX_AUX ""
#endif
#ifndef BN2_LDR_AUX
#define BN2_LDR_AUX 0
#endif

struct Bn2Params {
    const half_t* x; const half_t* w1; const float* b1; const half_t* w2; const float* b2; const half_t* w3; const float* b3; const half_t* res; half_t* y;
    int N, H, W, C0, n3, x_ld, y_ld, res_ld;
    float sl1, sl2, sl3;             // f(v) = v < 0 ? v * sl : v   (1 = no activation, 0 = ReLU)
    int TH, bands, RS, PR, NP1;      // tile rows, bands per image, slots per patch row, patch rows (TH + 2), patch pixels inside image columns (PR * W)
    int slice_bytes, xc_bytes;       // one T1 slice (64 channels of the patch, 2 KB aligned), one x stage (NP1 rounded up to 16 pixels x 128 bytes)
    int cap;                         // 1: the residual IS the block input (res == x, same stride, n3 == C0): its first passes are kept from the x stages of phase 1 instead of re-read
    int ka, kb;                      // swizzle of the T1 patch: chunk c of slot (row r, column cc) at position c ^ ((ka (cc / 2) + kb r) & 7)   (conv3x3_kq.hpp)
    unsigned w_mul, w_sh, bands_mul, bands_sh;
};
constexpr int BN2_NCAP = 2;                                           // expand passes whose residual rows are kept from phase 1
constexpr int BN2_XRING = 4 * 16384;                                  // the x ring of phase 1: four stages of 256 pixels x 64 bytes
constexpr int BN2_CM = 128, BN2_NF1 = 16, BN2_NF2 = 13;                // middle channels; pixel fragments (16 pixels each) of the patch / of the tile

__device__ __forceinline__ void bn2_bdma16(unsigned voff, u32x4 rsrc, unsigned lds_dst, unsigned soff) {   // LDS-DMA through a buffer descriptor (conv3x3_kq.hpp kq_bdma16): dead lanes -> zeros
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen" BN2_LDX_AUX " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

#ifdef SM_BN_TL
#define BN2_STAMP(slot) do { if (lane == 0 && blockIdx.x < 256 && (slot) < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_bn_tl[(blockIdx.x * 2 + (wave & 1)) * 64 + (slot) + (wave >> 1) * 32] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define BN2_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(256) void bneck128_kernel(Bn2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bn2_smem[];
    const unsigned lds0 = (ws_lds_addr(bn2_smem) + 2047u) & ~2047u;    // slices start 2 KB aligned (the launcher adds the slack)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, RS = p.RS, TH = p.TH;
    const int n0 = (int)sm_fastdiv((unsigned)blockIdx.x, p.bands_mul, p.bands_sh), y0 = ((int)blockIdx.x - n0 * p.bands) * TH;
    const int PTe = min(TH, H - y0) * W;                              // output pixels of this tile
    const long pix0 = ((long)n0 * H + y0) * W;
    const unsigned t1_0 = lds0, xr_0 = lds0 + 2u * (unsigned)p.slice_bytes;
    const bn_lds_cptr ldsb = bn_lds_base(0u);                           // (addresses below are absolute LDS byte addresses)

    BN2_STAMP(0);
#ifdef BN2_STAGGER
    {   // experiment: the blocks of an XCD (b % 8 equal) start BN2_STAGGER clk apart in eight groups, so that they do not ask the L2 for the same weight rows at the same time
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), wait = (unsigned long long)((blockIdx.x >> 3) & 7) * BN2_STAGGER;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    // ---- phase 0: the patch is cleared (halo columns, rows outside the image) -------------------------------------------------------------------------------------
    {
        const u32x4 z = {0, 0, 0, 0};
        for (unsigned o = (unsigned)tid * 16u; o < 2u * (unsigned)p.slice_bytes; o += 256u * 16u) *(__attribute__((address_space(3))) u32x4*)(size_t)(t1_0 + o) = z;
    }

    // ---- phase 1: T1 = act1(w1 . x + b1) on the patch's image pixels i = r W + xx (r < PR, xx < W) -------------------------------------------------------------------
    // x passes the LDS in stages of 32 channels (64 bytes per pixel, <= 256 pixels: 16 KB), a ring of FOUR with THREE in flight: the phase is bound by the fabric (every CU
    // pulls its 270 KB at once), and what a CU gets is (bytes in flight) / latency — with one 32 KB stage in flight the first form of this loop took 53 k clk for 16 k of
    // traffic.  Per stage a wave issues exactly SIX vector-memory operations (four LDS-DMA pieces, padded with dead ones; the two A fragments of w1 for that k step), all as
    // asm statements, so the wait in front of a stage is a plain count: 6 (D - 1) younger operations may still be in flight.
    constexpr int XS = 4, XD = 3;                                      // ring stages, stages in flight ahead
    const int nst = p.C0 >> 5;                                         // 32-channel stages of x = k steps of the reduce
    const unsigned ld2 = (unsigned)p.x_ld * 2u;
    const u32x4 rs_x = {(unsigned)(uintptr_t)p.x, (unsigned)((uintptr_t)p.x >> 32) & 0xffffu, (unsigned)p.N * (unsigned)(H * W) * ld2, 0x00020000u};
    // DMA: piece q = 4 i + j (j = 16-byte piece of pixel i's 64 bytes) lands lane-linear; instruction I = q / 64 covers 16 pixels, wave w issues I = w, w + 4, w + 8, w + 12.
    // The source piece is j ^ swz(i), swz(i) = (i / 4) & 3 (the fragment reads below undo it: 16 consecutive pixels x one piece = 16 distinct 16-byte slots).
    // Offsets without the channel base; pixels outside the image / beyond the patch: dead (zeros, no memory access).
    unsigned dmo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = (wave + 4 * k) * 64 + lane, i = q >> 2, j = q & 3;
        const int r = (int)sm_fastdiv((unsigned)i, p.w_mul, p.w_sh), xx = i - r * W, iy = y0 - 1 + r;
        const bool ok = i < p.NP1 && (unsigned)iy < (unsigned)H;
        dmo[k] = ok ? (unsigned)(((n0 * H + iy) * W + xx)) * ld2 + (unsigned)((j ^ ((i >> 2) & 3)) << 4) : 0x80000000u;
    }
    // A fragments of w1: fragment j = couts 32 wave + 16 j + l15, channels 32 s + 8 g .. (asm loads: counted with the DMA pieces)
    const unsigned w1o0 = (unsigned)(((32 * wave + l15) * p.C0 + 8 * g) * 2), w1o1 = w1o0 + (unsigned)(16 * p.C0 * 2);
    half8_t wa[XS][2];
    auto issue_stage = [&](auto slotc, int s) __attribute__((always_inline)) {   // stage s -> ring slot; beyond the last stage: six dead operations (the counts stay exact)
        constexpr int SL = decltype(slotc)::value;
        const bool real = s < nst;
        const unsigned dst = xr_0 + (unsigned)SL * 16384u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            bn2_bdma16(real ? dmo[k] : 0x80000000u, rs_x, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)(wave + 4 * k) * 1024u)), (unsigned)(real ? s : 0) * 64u);
        const half_t* wbse = p.w1 + (real ? s : 0) * 32;
        // (asm operands are copied to locals first: inside generic lambdas hipcc does not capture a variable that only an asm operand names)
        { half8_t& d0 = wa[SL][0]; const unsigned o0 = w1o0; asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d0) : "v"(o0), "s"(wbse) : "memory"); }
        { half8_t& d1 = wa[SL][1]; const unsigned o1 = w1o1; asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d1) : "v"(o1), "s"(wbse) : "memory"); }
    };
    // B fragment (16 pixels x 32 channels) f of a stage: lane (pixel l15, quarter g) reads piece g of pixel 16 f + l15
    unsigned xba[BN2_NF1];
#pragma unroll
    for (int f = 0; f < BN2_NF1; f++) { const int i = f * 16 + l15; xba[f] = xr_0 + (unsigned)(i * 64 + ((g ^ ((i >> 2) & 3)) << 4)); }
    u32x4 rrc[BN2_NCAP][BN2_NF2];                                     // residual rows kept from phase 1 (p.cap)
    floatx4 acc1[BN2_NF1][2];
#pragma unroll
    for (int f = 0; f < BN2_NF1; f++) { acc1[f][0] = floatx4{0.f, 0.f, 0.f, 0.f}; acc1[f][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
    BN2_STAMP(1);
    issue_stage(std::integral_constant<int, 0>(), 0); issue_stage(std::integral_constant<int, 1>(), 1); issue_stage(std::integral_constant<int, 2>(), 2);
    // (for the epilogue of this phase, computed here where the wave would otherwise wait for its first stages: the patch slot of my pixel of fragment f — 0xffffffff outside the
    //  image — and the bias rows of my couts)
    unsigned t1w[BN2_NF1];
#pragma unroll
    for (int f = 0; f < BN2_NF1; f++) {
        const int i = f * 16 + l15, r = (int)sm_fastdiv((unsigned)i, p.w_mul, p.w_sh), xx = i - r * W, iy = y0 - 1 + r, cc = xx + 1;
        t1w[f] = (i < p.NP1 && (unsigned)iy < (unsigned)H) ? (unsigned)((r * RS + cc) * 16) : 0xffffffffu;      // (outside the image: the epilogue sends the lane's 8 bytes to a sink in the dead x ring — no branch per store)
    }
    floatx4 bq1v[2];
#pragma unroll
    for (int j = 0; j < 2; j++) bq1v[j] = p.b1 ? *(const floatx4*)(p.b1 + 32 * wave + 16 * j + 4 * g) : floatx4{0.f, 0.f, 0.f, 0.f};
    const int nf1 = (p.NP1 + 15) >> 4;                                 // fragments that hold patch pixels (wave-uniform)
    auto stage = [&](auto slotc, int s) __attribute__((always_inline)) {
        constexpr int SL = decltype(slotc)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (XD - 1)) : "memory");    // my six operations of stage s (the younger ones: stages s + 1, s + 2)
        __builtin_amdgcn_s_barrier();                                 // everybody's; everybody has left the slot stage s + 3 goes to (it held stage s - 1)
        __builtin_amdgcn_sched_barrier(0);
        issue_stage(std::integral_constant<int, (SL + XD) % XS>(), s + XD);
        if constexpr (SL < BN2_NCAP) {
            // the residual of my expand pass SL — couts [n3 / 4 * wave + 32 SL, + 32) of the tile's pixels — is channels 32 s .. of this very stage when s = (n3 / 128) wave + SL
            // (identity block: n3 = C0, n3 % 512 == 0 so that the stage sits in ring slot SL): 13 fragment reads keep it in registers; the fabric carries those rows once instead of twice
            if (p.cap && s == (p.n3 >> 7) * wave + SL) {
#pragma unroll
                for (int f = 0; f < BN2_NF2; f++) {
                    const int t = f * 16 + l15, i = (t < PTe ? t : 0) + W;                 // tile pixel t = patch pixel t + W (one row down)
                    rrc[SL][f] = __builtin_bit_cast(u32x4, bn_lds_read<half8_t>(ldsb, (int)(xr_0 + (unsigned)SL * 16384u + (unsigned)(i * 64 + ((g ^ ((i >> 2) & 3)) << 4)))));
                }
            }
        }
#pragma unroll
        for (int f = 0; f < BN2_NF1; f++) {
            if (f < nf1) {
                const half8_t bf = bn_lds_read<half8_t>(ldsb, (int)(xba[f] + (unsigned)SL * 16384u));
                acc1[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[SL][0], bf, acc1[f][0], 0, 0, 0);
                acc1[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[SL][1], bf, acc1[f][1], 0, 0, 0);
            }
        }
    };
#pragma unroll 1
    for (int s0 = 0; s0 < nst; s0 += 4) {                             // (C0 % 64 == 0: stages come in pairs; the ring position is the stage number mod 4)
        stage(std::integral_constant<int, 0>(), s0);
        stage(std::integral_constant<int, 1>(), s0 + 1);
        if (s0 + 2 < nst) { stage(std::integral_constant<int, 2>(), s0 + 2); stage(std::integral_constant<int, 3>(), s0 + 3); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the dead operations behind the last stage
    // (to the register allocator an asm output is written when the statement executes: keep the ring allocated up to here, behind the wait)
#pragma unroll
    for (int q = 0; q < XS; q++) { asm volatile("" ::"v"(wa[q][0])); asm volatile("" ::"v"(wa[q][1])); }
    BN2_STAMP(2);
    // epilogue 1: lane (pixel l15 of fragment f, g) holds couts 32 wave + 16 j + 4 g .. + 3 -> 8 bytes of the pixel's slot in slice (couts / 64), chunk (couts % 64) / 8
    {
        const unsigned plane_b = (unsigned)p.slice_bytes >> 3, sink = xr_0 + (unsigned)tid * 8u;      // (the x ring is dead: every stage has been multiplied, the dead operations behind the last one have landed)
        const bool relu1 = p.sl1 == 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const floatx4 bq = bq1v[j];
            const unsigned t1p = t1_0 + (unsigned)(4 * wave + 2 * j + (g >> 1)) * plane_b + (unsigned)((g & 1) * 8);      // my couts 32 wave + 16 j + 4 g .. + 3: half of plane 4 wave + 2 j + g / 2
#pragma unroll
            for (int f = 0; f < BN2_NF1; f++) {
                {
                    half4_t o;
                    if (relu1) {                                       // plain ReLU (wave-uniform): round, then one packed maximum per pair (the same values: rounding is monotone)
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = (half_t)(acc1[f][j][e] + bq[e]);
                        o = __builtin_elementwise_max(o, half4_t{0, 0, 0, 0});
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) { const float v = acc1[f][j][e] + bq[e]; o[e] = (half_t)(v < 0.f ? v * p.sl1 : v); }
                    }
                    *(__attribute__((address_space(3))) half4_t*)(size_t)(t1w[f] != 0xffffffffu ? t1p + t1w[f] : sink) = o;
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BN2_STAMP(3);
    __builtin_amdgcn_s_barrier();                                     // the patch is complete
    __builtin_amdgcn_sched_barrier(0);
    BN2_STAMP(4);

    // ---- phase 2: the 3x3 over the patch.  Unit U = (tap U / 2, 32-channel half U % 2) of super-block sb (64 channels = one slice): 13 x 2 MFMAs ------------------------
    // T1 lies in LDS as sixteen PLANES [8 channels][patch slot] x 16 bytes (slot = row * RS + column; planes 256-byte aligned): the fragment of unit (tap, half hh) of slice
    // sb is plane 8 sb + 4 hh + g at slot (fragment's slot) + kh RS + kw — "lane base + one scalar per unit", and sixteen consecutive pixels of one plane are sixteen
    // consecutive 16-byte slots whatever the row length (conv3x3_kq.hpp's swizzled 128-byte slots measured ~12 clk per read here: 22 k clk of LDS time in this loop)
    const unsigned plane_b2 = (unsigned)p.slice_bytes >> 3;
    unsigned pfa[BN2_NF2];                                            // address of my pixel's slot of fragment f under tap (0, 0), plane g
#pragma unroll
    for (int f = 0; f < BN2_NF2; f++) {
        const int t = f * 16 + l15, tc = t < PTe ? t : 0, vrow = (int)sm_fastdiv((unsigned)tc, p.w_mul, p.w_sh), xx = tc - vrow * W;
        pfa[f] = t1_0 + (unsigned)g * plane_b2 + (unsigned)((vrow * RS + xx) * 16);
    }
    const half_t* w2p[2];
#pragma unroll
    for (int j = 0; j < 2; j++) w2p[j] = p.w2 + (long)(32 * wave + 16 * j + l15) * (9 * BN2_CM) + 8 * g;      // + tap * 128 + sb * 64 + hh * 32
    floatx4 acc2[BN2_NF2][2];
#pragma unroll
    for (int f = 0; f < BN2_NF2; f++) { acc2[f][0] = floatx4{0.f, 0.f, 0.f, 0.f}; acc2[f][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
    constexpr int LA = 5;                                              // weight units in flight ahead (ring of 6; 18 % 6 == 0): every CU asks the L2 for the same rows at the same time — two units
                                                                       // ahead measured 600 clk per unit of waiting for them (with neither MFMAs nor fragment reads in the loop), as conv3x3_kq.hpp found
    half8_t wb[LA + 1][2];
    half8_t af[2][BN2_NF2];
    auto load_w2 = [&](auto slotc, int sb, int u) __attribute__((always_inline)) {     // unit u (tap u / 2, half u % 2) of super-block sb
        constexpr int SL = decltype(slotc)::value;
        const int off = (u >> 1) * BN2_CM + sb * 64 + (u & 1) * 32;
        wb[SL][0] = *(const half8_t*)(w2p[0] + off); wb[SL][1] = *(const half8_t*)(w2p[1] + off);
    };
    // fragment f under unit u of slice sb = pfa[f] + T with one per-unit scalar, opaque to the optimiser (18 x 13 addresses are NOT loop invariants to keep in registers)
    auto unit_T = [&](int sb, int u) -> unsigned {
        const int tap = u >> 1, kh = tap / 3, kw = tap - 3 * kh;
        unsigned T = (unsigned)(8 * sb + 4 * (u & 1)) * plane_b2 + (unsigned)((kh * RS + kw) * 16);
        asm volatile("" : "+s"(T));
        return T;
    };
    floatx4 bq2v[2];                                                  // (for the T2 epilogue: requested here, a whole loop ahead)
#pragma unroll
    for (int j = 0; j < 2; j++) bq2v[j] = p.b2 ? *(const floatx4*)(p.b2 + 32 * wave + 16 * j + 4 * g) : floatx4{0.f, 0.f, 0.f, 0.f};
    sm_static_for<0, LA>([&](auto uc) { load_w2(uc, 0, decltype(uc)::value); });
    // Fragment reads are asm statements with a HAND-COUNTED wait (conv3x3_kq.hpp's order): written as plain loads "one unit ahead", hipcc sank every read next to its MFMAs and
    // waited for it there — one exposed LDS latency per fragment, 32 k clk for 15 k of MFMAs.  Pinned by sched_barriers: fragment F of unit U + 1 is requested right behind the
    // MFMAs of fragment F of unit U; in front of those MFMAs the younger reads are the 12 - F left of this unit and the F already requested for the next: lgkmcnt(12) throughout
    // (the last unit of the layer requests nothing).
    {
        const unsigned T = unit_T(0, 0);
#pragma unroll
        for (int f = 0; f < BN2_NF2; f++) { half8_t& d = af[0][f]; const unsigned ad = pfa[f] + T; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad)); }
    }
#pragma unroll 1
    for (int sb = 0; sb < 2; sb++) {
        const bool last_sb = sb == 1;
        sm_static_for<0, 18>([&](auto uc) {
            constexpr int U = decltype(uc)::value;
            {   // weights LA units ahead (beyond the last unit: a harmless reload)
                constexpr int U2 = (U + LA) % 18;
                const int sb2 = U + LA >= 18 ? (sb + 1 < 2 ? sb + 1 : sb) : sb;
                load_w2(std::integral_constant<int, (U + LA) % (LA + 1)>(), sb2, U2);
            }
            constexpr int U1 = (U + 1) % 18;
            const int sb1 = U + 1 >= 18 ? (sb + 1 < 2 ? sb + 1 : sb) : sb;
            const unsigned T = unit_T(sb1, U1);
            const bool fetch = !(U == 17 && last_sb);                  // (wave-uniform)
            __builtin_amdgcn_sched_barrier(0);
            sm_static_for<0, BN2_NF2>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
#ifndef BN2_ABL_NO_READS
                if (fetch) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(12 - F) : "memory");
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef BN2_ABL_NO_MFMA
                acc2[F][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[U % (LA + 1)][0], af[U & 1][F], acc2[F][0], 0, 0, 0);
                acc2[F][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[U % (LA + 1)][1], af[U & 1][F], acc2[F][1], 0, 0, 0);
#else
                asm volatile("" :: "v"(wb[U % (LA + 1)][0]), "v"(wb[U % (LA + 1)][1]), "v"(af[U & 1][F]));
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef BN2_ABL_NO_READS
                if (fetch) { half8_t& d = af[(U + 1) & 1][F]; const unsigned ad = pfa[F] + T; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad)); }
#else
                if (fetch) { asm volatile("s_nop 0" ::: "memory"); }
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BN2_STAMP(5);
    // ---- phase 3: T2 = act2(acc2 + b2), rounded, -> LDS [pixel t][128 channels] (256-byte rows; chunk c16 of pixel t at position c16 ^ (t & 15)); then the expand --------
    const unsigned t2_0 = xr_0;                                        // (over the x stages: dead since the barrier that closed phase 1)
    const bool relu2 = p.sl2 == 0.f;
    {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const floatx4 bq = bq2v[j];
            const int c16 = 4 * wave + 2 * j + (g >> 1);
#pragma unroll
            for (int f = 0; f < BN2_NF2; f++) {
                const int t = f * 16 + l15;
                half4_t o;
                if (relu2) {
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = (half_t)(acc2[f][j][e] + bq[e]);
                    o = __builtin_elementwise_max(o, half4_t{0, 0, 0, 0});
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float v = acc2[f][j][e] + bq[e]; o[e] = (half_t)(v < 0.f ? v * p.sl2 : v); }
                }
                *(__attribute__((address_space(3))) half4_t*)(size_t)(t2_0 + (unsigned)(t * 256) + (unsigned)((c16 ^ (t & 15)) << 4) + (unsigned)((g & 1) * 8)) = o;
            }
        }
    }
    BN2_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    BN2_STAMP(7);
    {
        const int cpw = p.n3 >> 2, npass = cpw >> 5;                  // couts per wave, passes of 32
        const unsigned pix_all = (unsigned)p.N * (unsigned)(H * W);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(pix_all * (unsigned)p.y_ld * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res ? (int)(pix_all * (unsigned)p.res_ld * 2u) : 0, 0x00020000);
        unsigned t2a[BN2_NF2];                                        // B fragment of k step ks: chunk 4 ks + g of pixel t = 16 f + l15
#pragma unroll
        for (int f = 0; f < BN2_NF2; f++) t2a[f] = t2_0 + (unsigned)((f * 16 + l15) * 256);
        const int tsw = l15;                                           // (t & 15 = l15)
        // weight row l15 of fragment j is cout (l15 / 4) 8 + 4 j + l15 % 4 of the pass: an accumulator lane (rows 4 g .. 4 g + 3) then holds couts 8 g .. 8 g + 3 in fragment 0
        // and 8 g + 4 .. 8 g + 7 in fragment 1 — eight consecutive couts
        const int wrow = (l15 >> 2) * 8 + (l15 & 3);
        // a pass's weight fragments and residual rows are requested during the pass before it, each INTO THE REGISTERS IT REPLACES as soon as they are consumed (a k step's two
        // fragments behind that k step's MFMAs, a pixel fragment's residual row behind its conversion): one pass of flight time, no second register set
        half8_t w3a[4][2];
        u32x4 rr[BN2_NF2];                                            // residual: couts cb + 8 g .. + 7 of pixel t
        auto w3_ptr = [&](int ps, int ks, int j) -> const half8_t* { return (const half8_t*)(p.w3 + (long)(cpw * wave + 32 * (ps < npass ? ps : 0) + wrow + 4 * j) * BN2_CM + ks * 32 + 8 * g); };
        // per-fragment byte offsets of this lane's pixel (+ its 8 g couts) in the residual and in the output, without the pass's cout base (a dead lane stays beyond num_records
        // whatever is added)
        unsigned rb[BN2_NF2], yb[BN2_NF2];
#pragma unroll
        for (int f = 0; f < BN2_NF2; f++) {
            const int t = f * 16 + l15;
            rb[f] = t < PTe ? (unsigned)(pix0 + t) * (unsigned)p.res_ld * 2u + (unsigned)(cpw * wave + 8 * g) * 2u : BN_DEAD;
            yb[f] = t < PTe ? (unsigned)(pix0 + t) * (unsigned)p.y_ld * 2u + (unsigned)(cpw * wave + 8 * g) * 2u : BN_DEAD;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { w3a[ks][0] = *w3_ptr(0, ks, 0); w3a[ks][1] = *w3_ptr(0, ks, 1); }
        const bool cap = p.cap != 0, relu3 = p.sl3 == 0.f;
        floatx4 bqn0 = {0.f, 0.f, 0.f, 0.f}, bqn1 = {0.f, 0.f, 0.f, 0.f};
        if (p.b3) { bqn0 = *(const floatx4*)(p.b3 + cpw * wave + 8 * g); bqn1 = *(const floatx4*)(p.b3 + cpw * wave + 8 * g + 4); }
#pragma unroll
        for (int f = 0; f < BN2_NF2; f++) rr[f] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(cap ? BN_DEAD : rb[f]), 0, BN2_LDR_AUX);
#pragma unroll 1
        for (int ps = 0; ps < npass; ps++) {
            const int cb = cpw * wave + 32 * ps;                       // first cout of the pass
            // (kept rows: passes 0 .. BN2_NCAP - 1 take theirs from phase 1's registers; the rolling re-fill asks memory only for passes that have none)
            if (cap && ps < BN2_NCAP) {
                if (ps == 0) {
#pragma unroll
                    for (int f = 0; f < BN2_NF2; f++) rr[f] = rrc[0][f];
                } else {
#pragma unroll
                    for (int f = 0; f < BN2_NF2; f++) rr[f] = rrc[BN2_NCAP - 1][f];
                }
            }
            const bool nload = ps + 1 < npass && !(cap && ps + 1 < BN2_NCAP);
            const unsigned pso = (unsigned)(64 * ps), psn = nload ? (unsigned)(64 * (ps + 1)) : 0x80000000u;      // byte offset of this / the next pass's couts
            const floatx4 bq0 = bqn0, bq1 = bqn1;
            { const int cn = cpw * wave + 32 * (ps + 1 < npass ? ps + 1 : ps) + 8 * g; if (p.b3) { bqn0 = *(const floatx4*)(p.b3 + cn); bqn1 = *(const floatx4*)(p.b3 + cn + 4); } }      // (the next pass's bias rows: a pass of flight time)
            // the 13 fragments in two halves (7 + 6): 56 accumulator registers live instead of 104 — the epilogue's conversions want them in the vector file
            sm_static_for<0, 2>([&](auto hc) {
                constexpr int HF = decltype(hc)::value, F0 = HF ? 7 : 0, F1 = HF ? BN2_NF2 : 7;
                floatx4 acc3[7][2];
#pragma unroll
                for (int f = F0; f < F1; f++) { acc3[f - F0][0] = floatx4{0.f, 0.f, 0.f, 0.f}; acc3[f - F0][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
                // fragment reads TWO ahead of their MFMAs, the order pinned (left to itself hipcc requests all of a k step's fragments at once — 28 registers more than the
                // vector file has room for here: the first build spilled to scratch inside this loop — or none ahead at all)
                constexpr int NFH = F1 - F0, NST = 4 * NFH;           // steps (ks, f) of this half
                constexpr int RA = 5;                                  // read-ahead in steps (a step = 2 MFMAs = 32 clk; an LDS read takes > 100)
                half8_t bf[RA + 1];
                auto rd = [&](int st) -> half8_t { const int ks = st / NFH, f = F0 + st % NFH; return bn_lds_read<half8_t>(ldsb, (int)(t2a[f] + (unsigned)(((4 * ks + g) ^ tsw) << 4))); };
#pragma unroll
                for (int q = 0; q < RA; q++) bf[q] = rd(q);
                sm_static_for<0, NST>([&](auto sc) {
                    constexpr int ST = decltype(sc)::value, KS = ST / NFH, FI = ST % NFH;
                    if constexpr (ST + RA < NST) bf[(ST + RA) % (RA + 1)] = rd(ST + RA);
                    __builtin_amdgcn_sched_barrier(0);
                    acc3[FI][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3a[KS][0], bf[ST % (RA + 1)], acc3[FI][0], 0, 0, 0);
                    acc3[FI][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3a[KS][1], bf[ST % (RA + 1)], acc3[FI][1], 0, 0, 0);
                    if constexpr (HF == 1 && FI == NFH - 1) { w3a[KS][0] = *w3_ptr(ps + 1, KS, 0); w3a[KS][1] = *w3_ptr(ps + 1, KS, 1); }     // (consumed by both halves: the next pass's)
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (ps == 1) BN2_STAMP(16 + 2 * HF);
#pragma unroll
                for (int f = F0; f < F1; f++) {
                    const half8_t rh = __builtin_bit_cast(half8_t, rr[f]);
                    half8_t ov;
                    if (relu3) {
#pragma unroll
                        for (int e = 0; e < 4; e++) { ov[e] = (half_t)(acc3[f - F0][0][e] + bq0[e] + (float)rh[e]); ov[4 + e] = (half_t)(acc3[f - F0][1][e] + bq1[e] + (float)rh[4 + e]); }
                        ov = __builtin_elementwise_max(ov, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float v0 = acc3[f - F0][0][e] + bq0[e] + (float)rh[e], v1 = acc3[f - F0][1][e] + bq1[e] + (float)rh[4 + e];
                            ov[e] = (half_t)(v0 < 0.f ? v0 * p.sl3 : v0); ov[4 + e] = (half_t)(v1 < 0.f ? v1 * p.sl3 : v1);
                        }
                    }
                    rr[f] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(rb[f] + psn), 0, BN2_LDR_AUX);              // the next pass's row of this fragment
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), rs_y, (int)(yb[f] + pso), 0, BN2_ST_AUX);
                    if (((f - F0) & 1) == 1) __builtin_amdgcn_sched_barrier(0);        // (two fragments per scheduling region: their chains interleave; more and the conversions of all seven are in flight at once)
                }
                if (ps == 1) BN2_STAMP(17 + 2 * HF);
            });
            BN2_STAMP(8 + ps);
        }
    }
}

static bool bn2_geometry(const sm_bneck_desc* d, Bn2Params* p) {
    if (d->Cm != BN2_CM || (d->C0 & 63) || d->C0 < 64 || d->C0 > 2048 || (d->C3 & 127) || d->C3 < 128 || d->C3 > 2048) return false;
    const int W = d->W, H = d->H;
    if (W < 4 || W > 62) return false;
    int TH = std::min(H, std::min(208 / W, 256 / W - 2));
    if (TH < 1) return false;
    const int RS = (W + 1 + 7) / 8 * 8, PR = TH + 2;
    // (+ 1 slot: when a row has no spare zero column — RS == W + 1 — the bottom-right tap of the band's last pixel is slot 0 of one more row; it must be a zero of THIS plane,
    //  not the first slot of the next one / of the x ring: found by a shape fuzz against the separate launches at W = 31, 39, 55)
    const int plane = ((PR * RS + 1) * 16 + 255) / 256 * 256, slice = 8 * plane, np1 = PR * W, xc = ((np1 + 15) & ~15) * 128;      // (T1: sixteen 8-channel PLANES [slot] x 16 bytes, 256-byte aligned: see the kernel)
    const int t2 = 13 * 16 * 256;
    if ((size_t)2 * slice + (size_t)std::max(BN2_XRING, t2) + 2048 > 160 * 1024) return false;
    if (p) {
        p->TH = TH; p->bands = sm_ceil_div(H, TH); p->RS = RS; p->PR = PR; p->NP1 = np1; p->slice_bytes = slice; p->xc_bytes = xc;
        p->ka = 1; p->kb = (RS / 2) & 7;
        if (RS == 32 && W == 28) { p->ka = 2; p->kb = 4; }             // conflict-free pair for 28-pixel rows in 32-slot patch rows (tools/lds_swizzle_search.py, as conv_kq.hip)
        sm_fastdiv_make((unsigned)W, &p->w_mul, &p->w_sh); sm_fastdiv_make((unsigned)p->bands, &p->bands_mul, &p->bands_sh);
    }
    return true;
}
static size_t bn2_lds_bytes(const Bn2Params& p) { return (size_t)2 * p.slice_bytes + (size_t)std::max(BN2_XRING, 13 * 16 * 256) + 2048; }
