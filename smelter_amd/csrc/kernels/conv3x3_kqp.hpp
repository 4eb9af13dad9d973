// conv3x3_kqp.hpp — PERSISTENT wave-split 3x3 / stride 1 / pad 1 convolution with the layer's WHOLE weight tensor in registers: the large-image, few-channel
// layers of an encoder / decoder (U-Net at 256 x 256: 128 -> 64, K = 1152; at 128 x 128: 64 -> 128, K = 576 — 9 Cin Cout <= 73 728 weights = 144 registers per
// lane).  Sibling of conv3x3_kq.hpp (same translation unit conv_kq.hip, tile-config kind 10 of conv_igemm.hip's launcher).
//
// Why: the tiled kernels re-stream 64 couts x K of weights per 256-pixel tile through the LDS and read one weight fragment per MFMA from it — the N = 64 decoder
// layers run at 716 TFLOP/s where the N = 256 ones reach 1150.  Here nothing but the input patch moves after the prologue:
//  * one block per CU lives for the whole launch and walks a contiguous range of 16-row x 14-column output tiles (224 pixels = 7 fragments of
//    v_mfma_f32_32x32x16_f16, none of them padded; 16 x 256 x 256 pixels = 256 blocks x 19 tiles exactly); ragged tiles at the right / bottom edge store what exists;
//  * wave (kg, ng), KG x NG = 4: input channels [64 kg, 64 kg + 64), couts [32 ng, 32 ng + 32) — its 9 taps x 4 k16 steps of weights are loaded ONCE, straight from
//    the OHWI tensor into 144 registers (112 of them AGPRs, next to the 112 accumulators); every wave holds the whole pixel tile (7 accumulator fragments);
//  * the tile's 18 x 16-slot input patch (one zero-padded halo ring; 128-byte slots, XOR-swizzled) lies in LDS per 64-channel slice, double-buffered: the patch
//    of tile t + 1 is copied while tile t is multiplied, one LDS-DMA instruction per unit between the MFMAs.  Lanes outside the image read beyond num_records of
//    the buffer descriptor (zeros, no access).  With the fused nearest 2x upsampling of the INPUT (p.ups) the patch is gathered from the half-resolution tensor;
//  * the first version used 16x16x32 MFMAs as conv3x3_kq.hpp does: with 2 MFMAs (32 clk) per fragment read the wave's issue slots were the limit (47 clk per step,
//    timeline).  A 32x32x16 MFMA occupies the pipe for 32 clk and hides its read, its wait and an address XOR: one read per MFMA;
//  * LDS swizzle: chunk c of patch slot (row r, column cc) sits at chunk position c ^ ((cc / 2 - r) & 7).  The usual key (slot / 2) & 7 is conflict-free for 32
//    CONSECUTIVE slots; a fragment's 32 pixels are 14-pixel rows inside 16-slot patch rows, and every ds_read_b128 then hit each bank twice (8 instead of 4 LDS
//    cycles: with four waves reading once per MFMA the LDS, not the matrix pipe, set the pace — 44 clk per 32-clk MFMA, timeline).  Subtracting the row makes every
//    16-lane service group of every (fragment, tap) hit 16 distinct 16-byte bank slots (searched exhaustively, tools/lds_swizzle_search.py);
//  * fragment addresses are a register table (7 x 9 taps, built once: the tile shape is fixed); k16 step = an XOR, the buffer toggles by one add per table entry
//    and tile;
//  * reads run 8 fragments ahead in a ring of 9 (lgkmcnt counted) instead of a whole unit: 36 registers, which is what makes room for the weights;
//  * epilogue: weight rows are assigned to couts so that 8 consecutive accumulator registers of a lane are 8 consecutive couts of its pixel and the other lane half
//    holds the next 8: two 16-byte stores per lane, each writing 32 contiguous bytes per pixel, no transpose.
//    KG = 2: the two channel halves trade half of their fragments through the patch buffer the tile has just released (accumulator layout, lane-linear).
// Numerics: fp32 MFMA accumulation inside a channel slice over (tap, channel) ascending, then slice 0 + slice 1.
#pragma once

struct KqpParams {
    const half_t* x; const half_t* w; const float* bias; half_t* y;
    int N, H, W, Cin, Cout, K, x_ld, y_ld, act; float slope;
    int ups;                                                           // 1: x is [N, H/2, W/2, Cin], read through a nearest 2x upsampling
    int tiles_x, tiles_y, ntiles, tpb;                                 // 16 x 14 tiles per image row / column, in all, per block
    unsigned tx_mul, tx_sh, txy_mul, txy_sh;                           // sm_fastdiv by tiles_x, tiles_x * tiles_y
};
constexpr int KQP_TH = 16, KQP_TW = 14;

template <int KG, int NG>
__global__ __launch_bounds__(256) void conv3x3_kqp_kernel(KqpParams p) {
    static_assert(KG * NG == 4 && (KG == 1 || KG == 2), "four waves");
    constexpr int NFR = 7, TH = KQP_TH, TW = KQP_TW, RS = 16, PR = TH + 2, PT = TH * TW;   // fragments of 32 pixels, tile, slots per patch row, patch rows, pixels per tile
    constexpr int SLICE = PR * RS * 128, BUF = KG * SLICE;            // one 64-channel patch slice (36 KB), one patch buffer
    constexpr int NDMA = PR * 2, DPW = NDMA / (4 / KG);               // patch DMA instructions per slice (2 per patch row) and per wave and tile
    constexpr int NU = 36, RING = 9, AHEAD = RING - 1, STEPS = NU * NFR;   // units = 9 taps x 4 k16 steps; fragment reads run AHEAD (unit, fragment) steps in front of their MFMAs
    static_assert(STEPS % RING == 0 && DPW <= NU && PT == NFR * 32, "ring / DMA schedule / tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char kqp_smem[];
    const unsigned lds0 = (ws_lds_addr(kqp_smem) + 2047u) & ~2047u;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5, l8 = lane >> 3, c8 = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = wave / NG, ng = wave % NG;
    const int H = p.H, W = p.W, Hs = p.ups ? H >> 1 : H, Ws = p.ups ? W >> 1 : W;
    const unsigned ld2 = (unsigned)p.x_ld * 2u, yld2 = (unsigned)p.y_ld * 2u;
    const int tile0 = blockIdx.x * p.tpb, tile1 = min(tile0 + p.tpb, p.ntiles);
    if (tile0 >= tile1) return;
    KQ_TL(0);

    // ---- weights -> registers, once.  A operand of tap, k16 step ks: lane (row rho = l31, channel group hi) holds channels 64 kg + 16 ks + 8 hi .. + 7 of the cout
    // that accumulator row rho stands for.  The 32x32 result layout gives lane half h the rows 8 q + 4 h + r (q = reg / 4, r = reg % 4): row rho is assigned cout
    // co0 + 16 (q / 2) + 8 h + 4 (q % 2) + r, so that registers 8 hf .. 8 hf + 7 of a lane are 8 consecutive couts and the two lane halves of a pixel are neighbours:
    // store hf writes 32 contiguous bytes per pixel ----
    const int co0 = ng * 32;
    half8_t wreg[9][4];
    {
        int co = co0 + 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3); if (co >= p.Cout) co = p.Cout - 1;
        const unsigned wo = (unsigned)(co * p.K + kg * 64 + hi * 8) * 2u;
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) wreg[tap][ks] = *(const half8_t*)((const char*)p.w + wo + (unsigned)(tap * p.Cin + ks * 16) * 2u);
        // The weight registers and the 112 accumulators share the AGPR half of the register file (an MFMA takes its A operand from there as well); the architectural
        // VGPRs stay free for the address table, the fragment ring and the epilogue's arithmetic.  Left to itself hipcc kept the weights in VGPRs and spilled the
        // epilogue to scratch memory (12 k clk per tile, timeline); with ALL of them pinned the allocator's accumulator renaming ran out: taps 0..6.
#pragma unroll
        for (int tap = 0; tap < 7; tap++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) asm volatile("" : "+a"(wreg[tap][ks]));
    }

    const unsigned slice_o = (unsigned)(kg * SLICE);
    // patch DMA: this wave's instruction d copies slots 8 k .. 8 k + 7 of patch row r, (r, k) = divmod(w4 + (4 / KG) d, 2), w4 = the wave's rank among the waves
    // that share the slice; the lane's piece = chunk c8 of slot 8 k + l8
    constexpr int WPS = 4 / KG;                                       // waves per slice
    const int w4 = KG == 2 ? ng : wave;
    unsigned dcol[2];                                                 // byte offset of the lane's pixel inside its patch row, for k = 0 | 1: column part + channel slice (the chunk depends on the row: swizzle)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int cc = 8 * k + l8;
        dcol[k] = (unsigned)(p.ups ? (cc - 1) >> 1 : cc) * ld2 + (unsigned)(kg * 128);
    }
    const kq_u32x4 rs_x = {(unsigned)(uintptr_t)p.x, (unsigned)((uintptr_t)p.x >> 32) & 0xffffu, (unsigned)p.N * (unsigned)(Hs * Ws) * ld2, 0x00020000u};
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((unsigned)p.N * (unsigned)(H * W) * yld2), 0x00020000);
    floatx4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int co = co0 + 16 * (q >> 1) + 8 * hi + 4 * (q & 1); bq[q] = (p.bias && co + 3 < p.Cout) ? *(const floatx4*)(p.bias + co) : floatx4{0.f, 0.f, 0.f, 0.f}; }
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    const unsigned yco = (co0 + 8 * hi < p.Cout) ? (unsigned)(co0 + 8 * hi) * 2u : 0x80000000u;

    auto tile_origin = [&](int tile, int& n, int& y0, int& x0) {
        n = (int)sm_fastdiv((unsigned)tile, p.txy_mul, p.txy_sh);
        const int r = tile - n * p.tiles_x * p.tiles_y, ty = (int)sm_fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        y0 = ty * TH; x0 = (r - ty * p.tiles_x) * TW;
    };
    auto issue_dma = [&](auto dc, int n, int y0, int x0, unsigned buf) {   // instruction DC of the patch of tile (n, y0, x0) -> buffer `buf`
        constexpr int D = decltype(dc)::value;
        const int idx = w4 + WPS * D, r = idx >> 1, cc = 8 * (idx & 1) + l8;
        const bool ok = (unsigned)(y0 - 1 + r) < (unsigned)H && (unsigned)(x0 - 1 + cc) < (unsigned)W;
        // patch row r of the tile, first column x0 - 1 (may be negative: only lanes with a pixel use it)
        const int org = p.ups ? (n * Hs + (y0 >> 1) + ((r - 1) >> 1)) * Ws + (x0 >> 1) : (n * H + y0 - 1 + r) * W + x0 - 1;
        const unsigned chunk = (unsigned)c8 ^ (unsigned)(((cc >> 1) - r) & 7);       // LDS position c8 of slot (r, cc) holds this source chunk
        kq_bdma16(ok ? (unsigned)org * ld2 + dcol[idx & 1] + (chunk << 4) : 0x80000000u, rs_x, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + buf + slice_o + (unsigned)idx * 1024u)), 0u);
    };

    int n, y0, x0;
    tile_origin(tile0, n, y0, x0);
    sm_static_for<0, DPW>([&](auto dc) { issue_dma(dc, n, y0, x0, 0u); });
    // (while the first patch is in flight) LDS address of fragment f's pixel under tap (kh, kw), k16 step 0, in buffer 0: slot (ty + kh, tx + kw), this lane-half's
    // 16-byte chunk at its swizzled position; k16 step ks: ^ (ks << 5) (chunk 2 ks + hi: the XOR commutes with the swizzle's)
    unsigned atab[NFR][9];
#pragma unroll
    for (int f = 0; f < NFR; f++) {
        const int t = f * 32 + l31, ty = t / TW, tx = t - ty * TW;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int r = ty + tap / 3, cc = tx + tap % 3;
            atab[f][tap] = lds0 + slice_o + (unsigned)((r * RS + cc) << 7) + (((unsigned)hi ^ (unsigned)(((cc >> 1) - r) & 7)) << 4);
        }
    }
    KQ_TL(1);
    floatx16 acc[NFR];
    half8_t af[RING];
    unsigned cur = 0;                                                 // byte offset of the buffer holding this tile's patch (the table addresses already include it)
#pragma unroll 1
    for (int tile = tile0; tile < tile1; tile++) {
        int nn, ny0, nx0;
        const bool more = tile + 1 < tile1;
        tile_origin(more ? tile + 1 : tile, nn, ny0, nx0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this tile's patch (issued a tile ago) — and the previous tile's stores
        if (tile == tile0) KQ_TL(2);
        kq_lds_barrier();                                             // everybody's share has landed; everybody has left the other buffer (exchange of the previous tile)
        if (tile == tile0) KQ_TL(3);
        if (tile == tile0 + 1) KQ_TL(7);
#pragma unroll
        for (int f = 0; f < NFR; f++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[f][r] = 0.f;
        sm_static_for<0, AHEAD>([&](auto qc) {                        // first AHEAD fragment reads of the tile
            constexpr int Q = decltype(qc)::value, U = Q / NFR, F = Q % NFR, TAP = U / 4, KS = U % 4;
            half8_t& d = af[Q % RING]; const unsigned ad = atab[F][TAP] ^ (unsigned)(KS << 5);
            asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad));
        });
        const unsigned nbuf = cur ? 0u : (unsigned)BUF;
        sm_static_for<0, STEPS>([&](auto qc) {
            constexpr int Q = decltype(qc)::value, U = Q / NFR, F = Q % NFR, TAP = U / 4, KS = U % 4;
            constexpr int Q2 = Q + AHEAD;
            if constexpr (Q2 < STEPS) {
                constexpr int U2 = Q2 / NFR, F2 = Q2 % NFR, TAP2 = U2 / 4, KS2 = U2 % 4;
                half8_t& d = af[Q2 % RING]; const unsigned ad = atab[F2][TAP2] ^ (unsigned)(KS2 << 5);
                asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad));
            }
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(Q2 < STEPS ? AHEAD : STEPS - 1 - Q) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            acc[F] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[TAP][KS], af[Q % RING], acc[F], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // one patch instruction of the next tile per unit (the last tile of the range copies nothing).  It costs the issuing wave ~150 clk (timeline): the copy is
            // 72 KB per tile at the ~27 B/clk a CU fills at, and a wave waits in the address queue while its instruction is served — with one wave per SIMD that time
            // is NOT overlapped with MFMAs.  Staggering the four waves' issue points over the unit made it worse (12.5 k instead of 11.2 k clk per tile).
            if constexpr (F == 3 && U < DPW) {
                if (more) issue_dma(std::integral_constant<int, U>(), nn, ny0, nx0, nbuf);
                __builtin_amdgcn_sched_barrier(0);
            }
        });

        // ---- epilogue ----
        if (tile == tile0) KQ_TL(4);
        if (tile == tile0 + 1) KQ_TL(8);
        const int rows = min(TH, H - y0), cols = min(TW, W - x0);
        int lpx = l31;
        asm volatile("" : "+v"(lpx));                                 // (opaque per tile: the store offsets are NOT tile invariants to hoist out of the loop and spill to scratch memory)
        const unsigned yorg = (unsigned)((n * H + y0) * W + x0) * yld2;
        int F0 = 0, F1 = NFR;
        if constexpr (KG == 2) {
            F0 = kg ? 4 : 0; F1 = kg ? NFR : 4;                       // slice 0 keeps fragments 0..3, slice 1 keeps 4..6
            kq_lds_barrier();                                         // everybody has read the last fragment of this patch: its buffer is the exchange area now
            float* const xb = (float*)(kqp_smem + (lds0 - ws_lds_addr(kqp_smem)) + cur);
            const int pw = (kg ^ 1) * NG + ng;
            sm_static_for<0, NFR>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                if ((F < 4) != (kg == 0)) {                           // a fragment I give away: slot F % 4 of MY region (4 KB per fragment: 4 x 16 bytes per lane)
#pragma unroll
                    for (int q = 0; q < 4; q++) *(floatx4*)&xb[((wave * 4 + F % 4) * 4 + q) * 256 + lane * 4] = floatx4{acc[F][4 * q], acc[F][4 * q + 1], acc[F][4 * q + 2], acc[F][4 * q + 3]};
                }
            });
            kq_lds_barrier();
            sm_static_for<0, NFR>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                if ((F < 4) == (kg == 0)) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const floatx4 o4 = *(const floatx4*)&xb[((pw * 4 + F % 4) * 4 + q) * 256 + lane * 4];
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[F][4 * q + r] = kg == 0 ? acc[F][4 * q + r] + o4[r] : o4[r] + acc[F][4 * q + r];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                    // (one fragment of the partner's sums in flight at a time: the registers are full)
            });
        }
        sm_static_for<0, NFR>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            if (F >= F0 && F < F1) {
                const int t = F * 32 + lpx, ty = t / TW, tx = t - ty * TW;                  // (constant divisor: a multiply)
                const bool ok = ty < rows && tx < cols;
                const unsigned off = ok ? yorg + (unsigned)(ty * W + tx) * yld2 + yco : 0x80000000u;
                typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {                      // couts co0 + 16 hf + 8 hi .. + 7 = registers 8 hf .. 8 hf + 7
                    half8_t ov;
                    const floatx4 a0 = floatx4{acc[F][8 * hf], acc[F][8 * hf + 1], acc[F][8 * hf + 2], acc[F][8 * hf + 3]} + bq[2 * hf];
                    const floatx4 a1 = floatx4{acc[F][8 * hf + 4], acc[F][8 * hf + 5], acc[F][8 * hf + 6], acc[F][8 * hf + 7]} + bq[2 * hf + 1];
                    if (sl == 0.f) {                                  // ReLU (wave-uniform branch): one packed fp16 max per pair, after the rounding
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
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, ov), rs_y, (int)(off + 32u * hf), 0, SM_ST_AUX);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (tile == tile0) KQ_TL(5);
        if (tile == tile0 + 1) KQ_TL(9);
        // next tile: its patch is in the other buffer
        const unsigned delta = nbuf - cur;                            // (+ BUF / - BUF, wrapping)
        n = nn; y0 = ny0; x0 = nx0; cur = nbuf;
#pragma unroll
        for (int f = 0; f < NFR; f++)
#pragma unroll
            for (int kw = 0; kw < 9; kw++) atab[f][kw] += delta;
        if (tile == tile0) KQ_TL(6);
    }
    KQ_TL(11);
    KQ_TL_END();
}
