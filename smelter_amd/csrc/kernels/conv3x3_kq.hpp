// conv3x3_kq.hpp — 3x3 / stride 1 / pad 1 convolution for layers with FEW, SMALL images and MANY channels (ResNet-50 stages 2-4 at
// batch 64: 28x28x128, 14x14x256, 7x7x512): the K reduction is split over the four waves of a block, every wave owns the WHOLE
// output tile of the block.  Included by conv_igemm.hip (same ConvParams / launcher; tile-config kind 9).
//
// Why: inside a network a CU fills its LDS / registers at ≈27 B/clk, and that rate — not the MFMA pipe — bounds the
// tiled kernels: a 128x128 tile of the 14x14x256 layer pulls 590 KB of weights + 80 KB of patch through one CU's L1 and only 196 such
// tiles exist for 256 CUs.  A batch-64 layer of this family is exactly 256 tiles of 196 output pixels x BN couts:
//   14x14: one image     x 64 couts  (64 x 4 blocks)      fill per CU: 131 KB patch + 295 KB weights = 426 KB
//   28x28: 7 image rows  x 128 couts (256 x 1 blocks)                   74 KB       + 295 KB         = 369 KB
//    7x7 : four images   x 32 couts  (16 x 16 blocks)                  271 KB       + 295 KB         = 566 KB
// * tile = 196 pixels = 13 fragments of 16 pixels (v_mfma_f32_16x16x32_f16: no row of the 16-wide fragment grid is wasted but the
//   last 12 pixels: 94 % of the issued MFMAs are work, against 77 % for seven 32-pixel fragments);
// * wave (kg, ng), KG x NG = 4: input channels [64 kg, 64 kg + 64) of every 64 KG-channel super-block, couts [16 NF ng, 16 NF (ng+1)) of
//   the block's BN = 16 NF NG: 13 x NF accumulator fragments (208 registers at NF = 4), one wave per SIMD.  Nothing is loaded twice by
//   one CU: the four waves read disjoint weight slices, and a pixel fragment read from the LDS feeds NF MFMAs (64 B/clk/CU of LDS reads);
// * the input PATCH of the tile lies in LDS as KG slices [slot][64 channels] (128-byte rows, 16-byte chunks XOR-swizzled by slot/2).
//   Slots: image rows are RS = W + 1 apart and images H + 1 rows apart — the zero column between two rows serves as right halo of one
//   and left halo of the next, the zero row between two images as bottom / top halo: tap (kh, kw) of an output pixel with slot q is
//   slot q + (kh - 1) RS + (kw - 1), no masks anywhere; slots outside the image are filled from the zero page by the same LDS-DMA;
// * the WEIGHTS never touch the LDS: fragment (tap, 32-channel half, 16 couts) is 16 rows x 64 bytes of the OHWI tensor, loaded straight
//   into registers (global_load_dwordx4 with an SGPR base per unit), two units (2 x NF KB per wave) ahead of its use; the waits are
//   counted (loads only: vmcnt is exact);
// * per unit = (tap, half) a wave reads its 13 pixel fragments (issued one unit ahead, lgkmcnt counted) and issues 13 NF MFMAs; the
//   main loop has NO barrier (one after the patch has landed, per super-block);
// * NB = 2 (layers with more than one super-block whose patch fits the LDS twice): the patch of super-block sb + 1 is copied while sb is
//   multiplied — ONE LDS-DMA instruction per unit between the MFMAs (a burst would stall the wave on the address queue), from a per-lane
//   offset table built once; only the first super-block's patch is waited for.  A CU fills from the Infinity Cache at ≈13 B/clk (its L1's
//   outstanding requests against that latency): 66 KB = 5 k clk exposed instead of 132 KB = 10 k;
// * the KG partial sums meet in LDS (fp32, two rounds of 7 + 6 fragments over the patch area), are added in the fixed order kg = 0..KG-1,
//   get bias + activation, one rounding, and leave as 16-byte stores of 8 consecutive couts.
// Numerics: fp32 MFMA accumulation inside a channel slice over (tap, channel) ascending, then the slices in ascending order.
#pragma once

#ifdef SM_KQ_TL   // diagnostic build of conv_kq.hip only (tools/kq_tl.py): per block and wave, slot 0 = wall clock (100 MHz) at entry, slots 1..13 = shader clock at KQ_TL(0..12), slot 15 = wall clock at exit
__device__ unsigned long long g_kq_tl[1024 * 4 * 16];
#define KQ_TL(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) { unsigned long long* e_ = g_kq_tl + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16; \
    if ((i) == 0) e_[0] = wall_clock64(); e_[1 + (i)] = clock64(); } } while (0)
#define KQ_TL_END() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_kq_tl[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 15] = wall_clock64(); } while (0)
#else
#define KQ_TL(i)
#define KQ_TL_END()
#endif

// What the kernel reads, packed: the launcher's ConvParams is > 600 bytes with these fields scattered all over it, and the kernel entry then pays one scalar-cache
// round trip after the other (3 k clk before the first DMA by the timeline); 144 bytes load in three instructions.
struct KqParams {
    const half_t* x; const half_t* w; const float* bias; half_t* y;
    int N, H, W, Cin, Cout, K, x_ld, y_ld, act; float slope;
    int Ho, Wo;                                                        // output size (the stride-2 instantiations; H, W else)
    int nt, RS, NI, slice_bytes, TH, TI, mt, bands, mpx, nsb;
    unsigned nt_mul, nt_sh, dhw_mul, dhw_sh, dw_mul, dw_sh, h1_mul, h1_sh, bands_mul, bands_sh, rs_mul, rs_sh;
    int ka, kb;                                                        // LDS swizzle: chunk c of patch slot (row r, column cc) sits at position c ^ ((ka (cc / 2) + kb r) & 7), kb in {0, 4};
                                                                       // (1, (RS / 2) & 7) = the usual (slot / 2) & 7; the launcher picks a conflict-free pair where one exists (tools/lds_swizzle_search.py)
};
typedef unsigned kq_u32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA through a buffer descriptor: 16 bytes per lane from byte offset `voff` of the buffer to LDS address lds_dst + 16 lane; a lane whose offset is beyond
// num_records gets zeros WITHOUT a memory access — the zero page of the other kernels, minus the 64-bit pointer select per lane.
__device__ __forceinline__ void kq_bdma16(unsigned voff, kq_u32x4 rsrc, unsigned lds_dst, unsigned soff) {
    unsigned keep;
#ifndef KQ_DMA_AUX      // cache-policy experiment (tools/kq_tl.sh): "" (default) | " nt" | " sc1" | " sc0 sc1"
#define KQ_DMA_AUX ""
#endif
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen" KQ_DMA_AUX " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

// LDS writes done + workgroup barrier as ONE asm statement with a memory clobber: `__builtin_amdgcn_s_barrier()` is no fence to the optimiser, which hoisted the
// reads of the partner's sums above it (stale 8-byte pieces in a few lanes).  Stores to global memory stay in flight across it (no vmcnt wait).
constexpr int kq_dma_units(int U, int LA) { int n = 0; for (int d = 1; d <= LA; d++) n += ((U - d + 18) % 18) != 17; return n; }
__device__ __forceinline__ void kq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// S2 — 3x3 / STRIDE 2 / pad 1 (ONNXConvolutionPadding.swift:91-113: the first 3x3 of ResNet-50's stages 2-4), same kernel: the stride lives in the patch layout.  A patch
// row holds its columns DE-INTERLEAVED — column cc = input column + 1 sits in slot cc / 2 when cc is even, in slot RS / 2 + cc / 2 when it is odd — so that the taps of
// output pixel ox are the slots ox (kw = 0), RS / 2 + ox (kw = 1) and ox + 1 (kw = 2): the 16 lanes of a fragment read CONSECUTIVE slots as at stride 1 (a stride of two
// slots would put them on half the banks), the tap offsets stay wave-uniform constants, and the rows of output row oy are the patch rows 2 oy + kh.  Output tiles of
// NFR fragments (13 | 7): a band of 16 NFR / Wo output rows or 16 NFR / (Ho Wo) images.
template <int KG, int NG, int NF, int NB, int NFR = 13, bool S2 = false>
__global__ __launch_bounds__(256) void conv3x3_kq_kernel(KqParams p) {
    static_assert(18 % ((NF == 4 ? 2 : 5) + 1) == 0 && KG * NG == 4 && (NF == 1 || NF == 2 || NF == 4) && (NB == 1 || NB == 2) && (NFR == 13 || NFR == 7), "four waves");
    constexpr int WN = NF * 16, BN = NG * WN;                         // couts per wave / per block (NFR pixel fragments per tile)
    constexpr int HF = (NFR + 1) / 2;                                 // fragments the first of two K slices keeps in the exchange epilogue
    constexpr int ROUNDS = NF == 4 && NFR > 7 ? 2 : 1, RFR = ROUNDS == 2 ? 7 : NFR;   // the partial sums pass the LDS in ROUNDS rounds of <= RFR fragments
    constexpr int RLD = WN + 4, RROWS = RFR * 16;                     // reduce region: floats per pixel row, pixel rows per round
    constexpr int OCT = WN / 8, PPS = 64 / OCT;                       // 8-cout groups per pixel, pixels per wave step of the summing pass
    constexpr bool TAB = NF <= 2;                                     // fragment addresses from a per-lane table (13 x 3 x 2 registers) instead of 4 VALU per read: with <= 2 MFMAs per read the wave's issue slots are the limit
    constexpr bool XCH = KG <= 2 && NF >= 2;                          // epilogue by pairwise exchange in accumulator layout (below); else: every partial sum through the LDS
    constexpr int LA = NF == 4 ? 2 : 5;                               // weight units in flight ahead of the one being multiplied (ring of LA + 1; 18 % (LA + 1) == 0).  In the network a layer's
                                                                      // weights come from HBM (the step's 282 MB do not stay in the Infinity Cache): 2 units of 450 clk do not cover that
    constexpr int DMX = 18;                                           // NB = 2: DMA instructions a wave issues per super-block (one per unit, none in the last)
    extern __shared__ __attribute__((aligned(16))) unsigned char kq_smem[];
    const unsigned lds0 = (ws_lds_addr(kq_smem) + 2047u) & ~2047u;    // slices start 2 KB aligned (the launcher adds the slack)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = wave / NG, ng = wave % NG;
    KQ_TL(0);

    // ---- block -> (pixel tile, cout tile): the cout tiles of a pixel tile are neighbours on one XCD (they share the patch in its L2) ----
    const int bid = blockIdx.x, xcd = bid & (SM_NUM_XCD - 1), bslot = bid >> 3;
    const int mq = (int)sm_fastdiv((unsigned)bslot, p.nt_mul, p.nt_sh), ntile = bslot - mq * p.nt, mtile = xcd * p.mpx + mq;
    if (mq >= p.mpx || mtile >= p.mt) return;
    const int H = p.H, W = p.W, RS = p.RS, HW = H * W, HR = RS >> 1;
    const int Ho = S2 ? p.Ho : H, Wo = S2 ? p.Wo : W, HWo = Ho * Wo;  // output image
    const bool band = p.TH > 0;
    int n0, y0 = 0, imgs = 1, PTe;                                    // (y0: first OUTPUT row of a band)
    if (band) {
        n0 = (int)sm_fastdiv((unsigned)mtile, p.bands_mul, p.bands_sh);
        y0 = (mtile - n0 * p.bands) * p.TH;
        PTe = min(p.TH, Ho - y0) * Wo;
    } else {
        n0 = mtile * p.TI; imgs = min(p.TI, p.N - n0); PTe = imgs * HWo;
    }
    const long pix0 = ((long)n0 * Ho + y0) * Wo;                      // the tile's output pixels are pix0 .. pix0 + PTe - 1

    // ---- weight fragment offsets: lane (l15, g) of fragment j holds cout co0 + 16 j + l15, channels 8 g .. 8 g + 7 of the unit's 32 ----
    const int co0 = ntile * BN + ng * WN;
    unsigned boff[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
        // XCH (KG <= 2, NF >= 2): weight row l15 of fragment j is cout (j/2) 32 + (l15/4) 8 + (j%2) 4 + l15%4 of the wave's slice, so that an accumulator lane (rows 4 g .. 4 g + 3
        // of the fragment) holds couts 8 g .. 8 g + 3 in fragment 2 jp and 8 g + 4 .. 8 g + 7 in fragment 2 jp + 1: eight consecutive couts = one 16-byte store, no LDS transpose
        int co = XCH ? co0 + (j >> 1) * 32 + (l15 >> 2) * 8 + (j & 1) * 4 + (l15 & 3) : co0 + j * 16 + l15;
        if (co >= p.Cout) co = p.Cout - 1;
        boff[j] = (unsigned)(co * p.K + g * 8) * 2u;
    }
    const int nsb = p.nsb, Cin = p.Cin;
    half8_t wb[LA + 1][NF];
    auto load_b = [&](auto slotc, int sb, int tap, int hh) {         // weight fragments of unit (sb, tap, hh) -> ring slot
        constexpr int SLOT = decltype(slotc)::value;
        const half_t* ub = p.w + tap * Cin + (sb * KG + kg) * 64 + hh * 32;
#pragma unroll
        for (int j = 0; j < NF; j++) { half8_t& d = wb[SLOT][j]; const unsigned bo = boff[j]; asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(bo), "s"(ub) : "memory"); }
    };
    sm_static_for<0, LA>([&](auto uc) { constexpr int U = decltype(uc)::value; load_b(uc, 0, U / 2, U % 2); });   // (taps 0..2: kh = 0, no plane flip)

    // ---- patch DMA: instruction (r, k) copies slots 8 k .. 8 k + 7 of patch row r (RS % 8 == 0: an instruction never leaves its row, so the row's image, image
    // row and validity are wave-uniform); wave (kg, ng) issues every NG-th instruction of slice kg.  The byte offset of a lane's 16 bytes does not depend on the
    // super-block but for the channel base, which is the instruction's scalar offset. ----
    const unsigned buf_bytes = (unsigned)(KG * p.slice_bytes);     // one patch buffer (NB = 2: the second one follows it, then 1 KB of sink for idle DMA slots)
    const unsigned slice_o = (unsigned)(kg * p.slice_bytes);
    const unsigned ld2 = (unsigned)p.x_ld * 2u;
    const kq_u32x4 rs_x = {(unsigned)(uintptr_t)p.x, (unsigned)((uintptr_t)p.x >> 32) & 0xffffu, (unsigned)p.N * (unsigned)HW * ld2, 0x00020000u};
    const int KPR = RS >> 3, c8 = lane & 7, l8 = lane >> 3;
    auto dma_off = [&](int r, int k) -> unsigned {                    // byte offset (channel base excluded) of this lane's piece of instruction (r, k); dead lanes: beyond the buffer
        // row r: band mode = image row y0 - 1 + r of image n0; whole-image mode = row (r - 1) % (H + 1) of image n0 + (r - 1) / (H + 1), row H being the zero row between images
        int n = n0, y = (S2 ? 2 * y0 : y0) - 1 + r; bool rok;
        if (band) rok = (unsigned)y < (unsigned)H;
        else {
            const int vr = r - 1, im = vr < 0 ? 0 : (int)sm_fastdiv((unsigned)vr, p.h1_mul, p.h1_sh);
            y = vr - im * (H + 1); n = n0 + im; rok = vr >= 0 && y < H && im < imgs;
        }
        const unsigned rowbase = (unsigned)(((n * H + y) * W - 1)) * ld2;                  // + cc * ld2: pixel (y, cc - 1)
        const int sl = 8 * k + l8, cc = S2 ? (sl < HR ? 2 * sl : 2 * (sl - HR) + 1) : sl;   // slot of the row -> column
        const unsigned key = (unsigned)((p.ka * (sl >> 1) + p.kb * r) & 7);                 // swizzle key of slot (r, sl)
        const bool ok = rok && (unsigned)(cc - 1) < (unsigned)W;
        return ok ? rowbase + (unsigned)cc * ld2 + (((unsigned)c8 ^ key) << 4) : 0x80000000u;
    };
    unsigned dmo[NB == 2 ? DMX : 1];
    if constexpr (NB == 2) {                                          // table of this wave's <= 17 instructions (the launcher checked the count); first super-block: issue as they are made
        int r = 0, k = ng;
        while (k >= KPR) { k -= KPR; r++; }
        sm_static_for<0, DMX>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const int idx = ng + I * NG;
            dmo[I] = idx < p.NI ? dma_off(r, k) : 0x80000000u;
            if (idx < p.NI) kq_bdma16(dmo[I], rs_x, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + slice_o + (unsigned)idx * 1024u)), (unsigned)(kg * 128));
            k += NG;
            while (k >= KPR) { k -= KPR; r++; }
        });
    }
    unsigned pf[NFR];
    unsigned atab[TAB ? NFR : 1][3][2];                               // TAB: address of fragment f's slot under tap (0, kw), channel plane hp, in buffer 0: + kh RS 128 (+ the buffer) per read
    // (a row step of RS slots adds RS / 2 to slot / 2, i.e. flips bit 2 of the swizzle key when RS / 8 is odd: under an odd kh the two 64-byte planes of a slot trade
    // places — the unit then multiplies plane hp with the weights of channel half hp ^ 1)
    const int kflip = (p.kb >> 2) & 1;                                // (kb = 4: one patch row down flips bit 2 of the key)
    const unsigned G0 = (unsigned)g << 4;                             // chunk of half 0 this lane reads (half 1: chunk + 4 = address ^ 64)
    floatx4 acc[NFR][NF];
    half8_t af[2][NFR];
    unsigned ad[NFR];                                                 // fragment addresses of the unit being fetched
    auto tap_off = [&](int tap) {                                     // byte offset of a tap's slots (opaque to the optimiser: 18 x 13 addresses are NOT loop invariants to keep in registers)
        unsigned t = (unsigned)(((tap / 3) * RS + (S2 ? (tap % 3 == 1 ? HR : tap % 3 >> 1) : tap % 3)) << 7);
        asm volatile("" : "+s"(t));
        return t;
    };

#pragma unroll 1
    for (int sb = 0; sb < nsb; sb++) {
        if (sb == 0) KQ_TL(1);
        const unsigned pbuf = NB == 2 ? (unsigned)(sb & 1) * buf_bytes : 0u;             // this super-block's patch buffer
        if (NB == 1) {                                                // one buffer: every super-block's patch is copied (and waited for) here
            const unsigned ch0b = (unsigned)((sb * KG + kg) * 64) * 2u;
            int r = 0, k = ng;
            while (k >= KPR) { k -= KPR; r++; }
#pragma unroll 1
            for (int i = ng; i < p.NI; i += NG) {
                kq_bdma16(dma_off(r, k), rs_x, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + slice_o + (unsigned)i * 1024u)), ch0b);
                k += NG;
                while (k >= KPR) { k -= KPR; r++; }
            }
        }
        if (sb == 0) {
            KQ_TL(2);
            // (while the patch is in flight) pixel fragments: LDS byte offset of tap (0, 0) of this lane's pixel of fragment f inside a buffer, before the swizzle
            // (slices start 2 KB aligned, so bits 8..10 of an address are bits 1..3 of its slot: the swizzle key can be taken from the address); zero accumulators
#pragma unroll
            for (int f = 0; f < NFR; f++) {
                const int t = f * 16 + l15, tc = t < PTe ? t : 0;
                int vrow, xx;                                        // patch row of tap row 0, output column
                if (band) { vrow = (int)sm_fastdiv((unsigned)tc, p.dw_mul, p.dw_sh); xx = tc - vrow * Wo; if (S2) vrow *= 2; }
                else {
                    const int im = (int)sm_fastdiv((unsigned)tc, p.dhw_mul, p.dhw_sh), rr = tc - im * HWo, yy = (int)sm_fastdiv((unsigned)rr, p.dw_mul, p.dw_sh);
                    xx = rr - yy * Wo; vrow = im * (H + 1) + (S2 ? 2 * yy : yy);
                }
                pf[f] = lds0 + slice_o + (unsigned)((vrow * RS + xx) << 7);   // centre = slot (vrow + 1) RS + xx + 1; tap (kh, kw) = this slot + kh RS + kw
                if constexpr (TAB) {
#pragma unroll
                    for (int kw = 0; kw < 3; kw++) {
                        const int ko = S2 ? (kw == 1 ? HR : kw >> 1) : kw;
                        const int slot = vrow * RS + xx + ko, r = (int)sm_fastdiv((unsigned)slot, p.rs_mul, p.rs_sh), cc = slot - r * RS;
                        const unsigned a0 = pf[f] + (unsigned)(ko << 7) + (((unsigned)g ^ (unsigned)((p.ka * (cc >> 1) + p.kb * r) & 7)) << 4);
                        atab[f][kw][0] = a0; atab[f][kw][1] = a0 ^ 64u;
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < NFR; f++)
#pragma unroll
                for (int j = 0; j < NF; j++) acc[f][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        // NB = 1 (or the first super-block): everything this wave has issued.  NB = 2, later super-blocks: the last patch instruction was issued in unit 16 of
        // the previous super-block, one weight unit (NF loads) after it.
        if (NB == 1 || sb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF) : "memory");
        if (sb == 0) KQ_TL(3);
        __builtin_amdgcn_s_barrier();                                 // every wave's share of this patch has landed (and everybody has left the buffer the next one goes to)
        if (sb == 0) KQ_TL(4);
        __builtin_amdgcn_sched_barrier(0);
        {   // fragments of unit 0
            const unsigned T = tap_off(0) + pbuf;
#pragma unroll
            for (int f = 0; f < NFR; f++) {
                if constexpr (TAB) ad[f] = atab[f][0][0] + pbuf;
                else { const unsigned pt = pf[f] + T; ad[f] = pt ^ G0 ^ ((pt >> 4) & 0x70u); }
                half8_t& d = af[0][f];
                asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad[f]));
            }
        }
        const bool more = sb + 1 < nsb;
        const unsigned nbuf = lds0 + (NB == 2 ? (unsigned)((sb + 1) & 1) * buf_bytes : 0u) + slice_o, sink = lds0 + 2u * buf_bytes;
        const unsigned nch0b = (unsigned)(((sb + 1) * KG + kg) * 64) * 2u;

        // Unit U = (tap U/2, channel half U%2): 13 x NF MFMAs.  Between them, in program order pinned by sched_barriers: the address of fragment f of unit
        // U + 1 (two pieces) and its read.  In front of fragment f's MFMAs the younger LDS operations are the 12 - f reads left of this unit and the f reads
        // already issued for the next one: lgkmcnt(12) throughout.  NB = 2: one patch instruction of the next super-block per unit (units 0..16), behind
        // fragment 6 — an idle slot copies zeros into the sink, so that the counted weight wait is the same everywhere: younger than unit U's weights are the
        // two later units and the patch instructions of units U-2 and U-1.
        sm_static_for<0, 18>([&](auto uc) {
            constexpr int U = decltype(uc)::value, V = (U + 1) % 18;
            {   // weights two units ahead (beyond the last unit of the layer: a harmless reload keeps the counted wait exact)
                constexpr int U2 = (U + LA) % 18;
                const int sb2 = U + LA >= 18 ? (more ? sb + 1 : sb) : sb;
                load_b(std::integral_constant<int, (U + LA) % (LA + 1)>(), sb2, U2 / 2, TAB ? (U2 % 2) ^ ((((U2 / 2) / 3) & 1) & kflip) : U2 % 2);
            }
            unsigned T = 0u;
            if constexpr (TAB) { if (U < 17) { T = (unsigned)((((V / 2) / 3) * RS) << 7) + pbuf; asm volatile("" : "+s"(T)); } }
            else if (V % 2 == 0 && U < 17) T = tap_off(V / 2) + pbuf;
            if (sb == 0 && (U == 1 || U == 2 || U == 9)) KQ_TL(4 + (U == 1 ? 1 : U == 2 ? 2 : 3));
            constexpr int DYOUNG = NB == 2 ? kq_dma_units(U, LA) : 0;  // patch instructions issued since unit U's weights were: units U - LA .. U - 1 (of the previous super-block below 0), unit 17 has none
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA * NF + DYOUNG) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            sm_static_for<0, NFR>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(U == 17 ? NFR - 1 - F : NFR - 1) : "memory");   // (the last unit fetches nothing)
                __builtin_amdgcn_sched_barrier(0);
                unsigned pt = 0, ky = 0;
#pragma unroll
                for (int j = 0; j < NF; j++) {
                    acc[F][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[U % (LA + 1)][j], af[U % 2][F], acc[F][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (U < 17 && TAB) {
                        if (j == 0) { ad[F] = atab[F][(V / 2) % 3][V % 2] + T; __builtin_amdgcn_sched_barrier(0); }
                    }
                    if constexpr (U < 17 && !TAB) {
                        if (j == 0) {
                            if constexpr (V % 2 == 0) { pt = pf[F] + T; ky = pt >> 4; }
                            else ad[F] ^= 64u;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (j == (NF > 1 ? 1 : 0)) {
                            if constexpr (V % 2 == 0) ad[F] = pt ^ G0 ^ (ky & 0x70u);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (U < 17) {
                        if (j == NF - 1) {
                            half8_t& d = af[(U + 1) % 2][F];
                            asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad[F]));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if constexpr (NB == 2 && U < 17 && F == NFR / 2) {
                    const bool real = more && ng + U * NG < p.NI;
                    kq_bdma16(real ? dmo[U < DMX ? U : 0] : 0x80000000u, rs_x,
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(real ? nbuf + (unsigned)(ng + U * NG) * 1024u : sink)), nch0b);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (NB == 1 && more) __builtin_amdgcn_s_barrier();            // one buffer: everybody has finished reading this patch before the next one is copied over it
    }

    // ---- the KG partial sums meet in LDS; bias + activation + one rounding; 16-byte stores ----
    KQ_TL(8);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // The last two weight units issued are never multiplied.  To the register allocator an asm output is written when the statement executes, so a
    // destination whose value is dead would be handed to the epilogue's address arithmetic while the load is still in flight: keep the ring allocated
    // up to this point, behind the wait.
#pragma unroll
    for (int q = 0; q < LA + 1; q++)
#pragma unroll
        for (int j = 0; j < NF; j++) asm volatile("" ::"v"(wb[q][j]));
    kq_lds_barrier();                                                 // everybody has left the patch: its LDS becomes the exchange area
    KQ_TL(9);
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    if constexpr (XCH) {
        // KG = 2: wave kg keeps fragments [F0, F1) = kg ? [7, 13) : [0, 7) and hands the others to its partner (same ng, other kg) through the LDS in accumulator
        // layout (lane-linear 1 KB per fragment: conflict-free, half the bytes of the all-to-all pass); sum = slice 0 + slice 1 on both sides.  KG = 1: nothing to add.
        const int F0 = KG == 2 && kg ? HF : 0, F1 = KG == 2 && !kg ? HF : NFR;
        if constexpr (KG == 2) {
            float* const xb = (float*)(kq_smem + (lds0 - ws_lds_addr(kq_smem)));
            const int pw = (kg ^ 1) * NG + ng;                        // partner wave
            sm_static_for<0, NFR>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                if ((F < HF) != (kg == 0)) {                          // (wave-uniform) a fragment I give away: slot F % HF of MY region
#pragma unroll
                    for (int j = 0; j < NF; j++) *(floatx4*)&xb[((wave * HF + F % HF) * NF + j) * 256 + lane * 4] = acc[F][j];
                }
            });
            kq_lds_barrier();
            sm_static_for<0, NFR>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                if ((F < HF) == (kg == 0)) {
#pragma unroll
                    for (int j = 0; j < NF; j++) {
                        const floatx4 o4 = *(const floatx4*)&xb[((pw * HF + F % HF) * NF + j) * 256 + lane * 4];
                        acc[F][j] = kg == 0 ? acc[F][j] + o4 : o4 + acc[F][j];
                    }
                }
                if constexpr (F % 2 == 1) __builtin_amdgcn_sched_barrier(0);   // (at most two fragments of the partner's sums in flight: 208 accumulators leave no room for more)
            });
        }
        KQ_TL(10);
        // stores: descriptor on y, scalar offset = the fragment's first pixel, per-lane offset = (pixel in fragment, cout group) — no address arithmetic per store;
        // lanes without a cout group (partial last cout tile) or, in the tile's last fragments, without a pixel sit beyond num_records
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((unsigned)p.N * (unsigned)HWo * (unsigned)p.y_ld * 2u), 0x00020000);
        const unsigned ybase = (unsigned)pix0 * (unsigned)p.y_ld * 2u, yfrag = 16u * (unsigned)p.y_ld * 2u;
        unsigned yoff[NF / 2];
#pragma unroll
        for (int jp = 0; jp < NF / 2; jp++) { const int co = co0 + jp * 32 + g * 8; yoff[jp] = co < p.Cout ? (unsigned)l15 * (unsigned)p.y_ld * 2u + (unsigned)co * 2u : 0x80000000u; }
        floatx4 bq[NF];
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int co = co0 + (j >> 1) * 32 + g * 8 + (j & 1) * 4;
            bq[j] = (p.bias && co + 3 < p.Cout) ? *(const floatx4*)(p.bias + co) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
        sm_static_for<0, NFR>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            if (F >= F0 && F < F1) {
                const int t = F * 16 + l15;
#pragma unroll
                for (int jp = 0; jp < NF / 2; jp++) {
                    const floatx4 a0 = acc[F][2 * jp] + bq[2 * jp], a1 = acc[F][2 * jp + 1] + bq[2 * jp + 1];
                    half8_t ov;
                    if (sl == 0.f) {                                  // ReLU: one packed fp16 max per pair, after the rounding (max(x, 0) commutes with it)
#pragma unroll
                        for (int e = 0; e < 4; e++) { ov[e] = (half_t)a0[e]; ov[4 + e] = (half_t)a1[e]; }
                        ov = __builtin_elementwise_max(ov, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float v0 = a0[e] < 0.f ? a0[e] * sl : a0[e], v1 = a1[e] < 0.f ? a1[e] * sl : a1[e];
                            ov[e] = (half_t)v0; ov[4 + e] = (half_t)v1;
                        }
                    }
                    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
// (the fragment's pixel base goes into the VECTOR offset, the scalar offset stays the constant 0: with a 16-byte store whose soffset is an SGPR, hipcc
                    // lets the next VALU instruction overwrite the store's data registers at once — LLVM's hazard recogniser exempts that form — and on gfx950 a few lanes then
                    // stored the NEW contents of one data dword: wrong values in lanes 12..15 of every 16, found by tests/test_gpu_ops.py::test_conv3x3_wave_split)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, ov), rs_y, (int)((t < PTe && yoff[jp] != 0x80000000u) ? yoff[jp] + (ybase + (unsigned)F * yfrag) : 0x80000000u), 0, 0);   // (the select covers the WHOLE sum: a dead lane's marker plus a base near 2 GB would wrap into range — ADVICE r4)
                }
            }
        });
    } else {
    float* const red = (float*)(kq_smem + (lds0 - ws_lds_addr(kq_smem)));
    const int o = lane % OCT, co = co0 + o * 8;
    floatx4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && co + 7 < p.Cout) { bq0 = *(const floatx4*)(p.bias + co); bq1 = *(const floatx4*)(p.bias + co + 4); }
    sm_static_for<0, ROUNDS>([&](auto rc) {
        constexpr int RND = decltype(rc)::value, F0 = RND * RFR, NFRR = RND ? NFR - RFR : RFR;
#pragma unroll
        for (int fi = 0; fi < NFRR; fi++)
#pragma unroll
            for (int j = 0; j < NF; j++) *(floatx4*)&red[(wave * RROWS + fi * 16 + l15) * RLD + j * 16 + g * 4] = acc[F0 + fi][j];
        kq_lds_barrier();                                             // (LDS writes only: the stores of the previous round stay in flight across the barrier)
        constexpr int ITS = (NFRR * 16 + KG * PPS - 1) / (KG * PPS);
        floatx4 s0[ITS][KG], s1[ITS][KG];
#pragma unroll
        for (int it = 0; it < ITS; it++) {                            // every read of the pass in flight before the first sum
            int pp = kg * PPS + lane / OCT + it * KG * PPS; if (pp >= NFRR * 16) pp = 0;
#pragma unroll
            for (int k = 0; k < KG; k++) {
                s0[it][k] = *(const floatx4*)&red[((k * NG + ng) * RROWS + pp) * RLD + o * 8];
                s1[it][k] = *(const floatx4*)&red[((k * NG + ng) * RROWS + pp) * RLD + o * 8 + 4];
            }
        }
#pragma unroll
        for (int it = 0; it < ITS; it++) {
            const int pp = kg * PPS + lane / OCT + it * KG * PPS, t = F0 * 16 + pp;
            floatx4 a0 = s0[it][0], a1 = s1[it][0];
#pragma unroll
            for (int k = 1; k < KG; k++) { a0 += s0[it][k]; a1 += s1[it][k]; }
            a0 += bq0; a1 += bq1;
            half8_t ov;
            if (sl == 0.f) {                                          // ReLU: one packed fp16 max per pair, after the rounding (max(x, 0) commutes with it)
#pragma unroll
                for (int e = 0; e < 4; e++) { ov[e] = (half_t)a0[e]; ov[4 + e] = (half_t)a1[e]; }
                ov = __builtin_elementwise_max(ov, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v0 = a0[e] < 0.f ? a0[e] * sl : a0[e], v1 = a1[e] < 0.f ? a1[e] * sl : a1[e];
                    ov[e] = (half_t)v0; ov[4 + e] = (half_t)v1;
                }
            }
            if (pp < NFRR * 16 && t < PTe && co < p.Cout) *(half8_t*)(p.y + (pix0 + t) * p.y_ld + co) = ov;
        }
        if (RND + 1 < ROUNDS) { kq_lds_barrier(); KQ_TL(10); }
    });
    }
    KQ_TL(11);
    KQ_TL_END();
}
