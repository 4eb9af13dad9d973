// conv3x3_ws.hpp — WEIGHT-STATIONARY persistent 3x3 / stride 1 / pad 1 convolution for layers whose whole weight tensor fits the
// LDS next to two activation patches: Cin in {32, 64}, Cout in {32, 64} (ResNet-50's three 56x56 64->64 layers; the seven
// full- and half-resolution layers of the U-Net, M = 1..4.2 M pixels).  Included by conv_igemm.hip (same ConvParams / launcher).
//
// Why: in the tiled implicit-GEMM kernels a 128x64 tile of such a layer re-streams (128 + 64) x 128 B per K tile through the
// per-CU L1 -> LDS path (≈27 B/clk/CU inside a network) for 256 clk of MFMA work — L1-fill bound at ≈1/4 of the
// MFMA rate, plus a block prologue + epilogue per tile.  Here
//  * ONE block per CU lives for the whole launch; the weights [9 taps][Cout][Cin] (18..72 KB) and the bias go to LDS once;
//  * a block walks a contiguous range of 2-D output tiles (TH x TW pixels of one image, TH*TW <= 32*NW*TM); per tile the
//    (TH+2) x (TW+2) input PATCH is copied to LDS by the LDS-DMA (lane-linear image, XOR swizzle on the source side, zero page
//    outside the image — so padding costs no masks anywhere) and serves all nine taps: tap (kh,kw) of output pixel (ty,tx) is patch
//    pixel (ty+kh, tx+kw).  L1 traffic per output pixel: (TH+2)(TW+2)/(TH*TW) ≈ 1.3 input pixels instead of 9 (+ no weights);
//  * two patch buffers: the DMA of tile t+1 is issued before the MFMAs of tile t; ONE barrier per tile (≥ 4.6 k clk of MFMAs);
//  * no LDS epilogue: the accumulators get bias + activation in registers, are rounded once to fp16, and v_permlane32_swap
//    pairs the two half-waves' 4-channel groups into 8 consecutive channels per lane -> 16-byte stores.  The stores of tile t are
//    issued at the START of iteration t+1 (after its barrier), so the `s_waitcnt vmcnt(0)` in front of every barrier only ever
//    waits for operations that had a whole tile of MFMAs to complete (loads and stores retire out of order relative to each
//    other on gfx9-family counters: a counted wait across both kinds is not safe, a late vmcnt(0) is);
//  * numerics = the tiled kernels': fp32 MFMA accumulation over k = (kh, kw, cin) ascending from zero, + bias in fp32, activation,
//    one rounding.
#pragma once

template <int CP> __device__ __forceinline__ int ws_swz(int idx) {   // XOR key of the 16-byte chunk of LDS row `idx` (row = CP chunks)
    static_assert(CP == 2 || CP == 4 || CP == 8 || CP == 16, "2..16 chunks per row");
    return (idx / (16 / CP)) & (CP - 1);                              // 16 consecutive rows x one chunk -> 16 distinct 16-byte bank slots
}

constexpr int ws_pstride(int fixed_bytes) { const int half = ((160 * 1024 - fixed_bytes) / 2) & ~1023; return half < 60 * 1024 ? half : 60 * 1024; }
constexpr int WS_DUMMY = 1024;                                        // LDS sink of the DMA instructions a wave has no patch slots for

// NIC = LDS-DMA instructions per thread per patch (the launcher picks a tile geometry that needs no more)
// ABL (tools/conv_bench.py, SM_ABLATION builds only): bit 0 = no DMA after the first patch, bit 1 = no MFMAs, bit 2 = no fragment reads, bit 3 = no stores
//
// STEM > 0 (with CIN = 64): the image-stem form.  The input is the ROW-PAIR image (layout.hip: pixel = [row y | row y+1] x 4 channels =
// 16 bytes, zero halo instead of padding) and the layer is a Cin = 64, kH' = STEM, kW' = 1, row-dilation-2, stride s convolution: "input
// pixel" (y, x) = the 8 consecutive row-pair pixels (y, x .. x+7), so neighbouring pixels OVERLAP (x_ld = 8).  The patch holds plain
// 16-byte row-pair pixels — ((TH-1)s + 2(STEM-1) + 1) x ((TW-1)s + 8) of them — and the A fragment of k step ks of tap j is row-pair pixel
// (ty*s + 2j, tx*s + 2ks + hi): lanes of a half-wave read 16 (s = 1) or 32 (s = 2) bytes apart, no swizzle needed or possible.
// PSB = distance between the two patch buffers (0: as large as the weights allow) — small patches leave room for a second block per CU.
// HEAD: instantiated with the fused 1x1 head (p.y3) — a separate instantiation because the extra MFMA epilogue costs the plain kernels 30-40
// registers (one resident block instead of two on the 64 -> 32 layers)
// HX: the EXPAND head (ResNet bottleneck tail): y3 = act3(w3 . y + b3 + res3) with n3 = 32 .. 256 channels.  w3 [n3][NOUT] and b3 live in LDS
// beside the 3x3 weights (the launcher shrinks the tiles accordingly), the residual rows of a tile are requested when its k loop starts and
// consumed with the deferred stores one tile later (TM = 1: 64 registers), so neither the 3x3 output nor any operand latency is exposed.
constexpr int WS_HX_N3 = 256;                                         // LDS is reserved for this many head channels
constexpr int ws_hx_bytes(int nout) { return WS_HX_N3 * nout * 2 + WS_HX_N3 * 4; }
// NLW > 0 (stem form only): LOADER WAVES.  The block has NLW more waves that build the row-pair patches straight from the fp32 NCHW input
// (p.xf: <= 4 planes per image, zero outside the image = the convolution's padding) — convert, pack [row y | row y+1] x 4 channels, one
// ds_write_b128 per patch pixel — into the buffer the MFMA waves will read next, under the same one barrier per tile; the MFMA waves issue
// no DMA at all.  The separate input-conversion launch (nchw_to_rowpair_kernel: the batch read once and written twice over) disappears.
template <int CIN, int NOUT, int NW, int TM, int NIC = (CIN == 64 ? 12 : 6), int ABL = 0, int STEM = 0, int PSB = 0, bool HEAD = false, bool HX = false, int NLW = 0>
__global__ __launch_bounds__((NW + NLW) * 64) void conv3x3_ws_kernel(ConvParams p) {
    constexpr bool ST = STEM > 0;
    static_assert(NLW == 0 || (ST && !HEAD && !HX), "loader waves: stem form");
    constexpr int NTAP = ST ? STEM : 9;
    static_assert(!ST || (CIN == 64 && STEM <= 4), "stem form: 64 = 8 row-pair pixels x 8 halfs");
    static_assert(!HX || (!ST && !HEAD && TM == 1 && NOUT == 64), "expand head: plain form, one fragment per wave");
    constexpr int NT = NW * 64, CP = CIN / 8, TN = NOUT / 32, KS = CIN / 16;
    constexpr int ROWB = CP * 16;                                     // bytes per pixel / per weight row of one tap
    constexpr int W_BYTES = NTAP * NOUT * ROWB, BIAS_BYTES = NOUT * 4;
    constexpr int HX_BYTES = HX ? ws_hx_bytes(NOUT) : 0, CP3 = NOUT / 8, ROWB3 = NOUT * 2, HXC = WS_HX_N3 / 32;
    static_assert(TN >= 1 && TN <= 2 && TM >= 1 && TM <= 4, "wave tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    unsigned char* const wl = ws_smem;                                // weights [tap][cout][CP chunks], chunk c of row n stored at c ^ swz(n)
    float* const bl = (float*)(ws_smem + W_BYTES);
    unsigned char* const w3l = ws_smem + W_BYTES + BIAS_BYTES;       // HX: head weights [n3][CP3 chunks], same swizzle; then the head bias
    float* const b3l = (float*)(w3l + WS_HX_N3 * ROWB3);
    unsigned char* const dummy = ws_smem + W_BYTES + BIAS_BYTES + HX_BYTES;
    unsigned char* const patch0 = dummy + WS_DUMMY;
    // the second patch buffer sits a COMPILE-TIME distance behind the first, so that the buffer choice is an instruction offset of the
    // fragment reads (ds_read offset field: 16 bits), not an address computation
    constexpr int WS_PSTRIDE = PSB ? PSB : ws_pstride(W_BYTES + BIAS_BYTES + HX_BYTES + WS_DUMMY);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const half_t* zero = (const half_t*)g_zero_page;
    const int TH = p.ws_TH, TW = p.ws_TW, SS = ST ? p.sH : 1;
    const int PW = ST ? (TW - 1) * SS + 8 : TW + 2, PH = ST ? (TH - 1) * SS + 2 * (NTAP - 1) + 1 : TH + 2;
    const int OH = ST ? p.Ho : p.H, OW = ST ? p.Wo : p.W;             // output image (the plain form is "same" size)
    const int patch_slots = PH * PW * (ST ? 1 : CP), NI = (patch_slots + NT - 1) / NT;

    // ---- weights + bias -> LDS (once per block) -------------------------------------------------------------------------
    if (NLW == 0 || wave < NW) {
        constexpr int WSLOTS = NTAP * NOUT * CP;
        static_assert((NOUT * CP) % 64 == 0, "a wave instruction stays inside the weight image");
#pragma unroll 1
        for (int s0 = wave * 64; s0 < WSLOTS; s0 += NT) {
            const int s = s0 + lane, row = s / CP, c = s & (CP - 1);  // row = tap * NOUT + n
            const int tap = row / NOUT, n = row - tap * NOUT;
            SM_GLDS16(p.w + (long)n * (NTAP * CIN) + tap * CIN + ((c ^ ws_swz<CP>(n)) << 3), wl + s0 * 16);
        }
        for (int i = tid; i < NOUT; i += NT) bl[i] = p.bias ? p.bias[i] : 0.f;
        if constexpr (HX) {
#pragma unroll 1
            for (int s0 = wave * 64; s0 < p.n3 * CP3; s0 += NT) {     // (n3 % 32 == 0: whole wave instructions)
                const int s = s0 + lane, row = s / CP3, c = s & (CP3 - 1);
                SM_GLDS16(p.w3 + (long)row * NOUT + ((c ^ ws_swz<CP3>(row)) << 3), w3l + s0 * 16);
            }
            for (int i = tid; i < p.n3; i += NT) b3l[i] = p.b3 ? p.b3[i] : 0.f;
        }
    }
    // ---- per-thread patch DMA descriptors: slot s = i*NT + tid -> (patch row, patch column, source chunk) ---------------
    int rel[NIC]; unsigned prpc[NIC];
#pragma unroll
    for (int i = 0; i < NIC; i++) {
        const int s = i * NT + tid, ps = ST ? s : s / CP, c = ST ? 0 : s & (CP - 1);
        const int pr = (int)sm_fastdiv((unsigned)ps, p.ws_pw_mul, p.ws_pw_sh), pc = ps - pr * PW;
        // (p.ups: the patch is gathered from the half-resolution tensor; tile origins are even, so input pixel (y0-1+pr, x0-1+pc) of
        //  the upsampled image is low-resolution pixel (y0/2 + ((pr-1)>>1), x0/2 + ((pc-1)>>1)) — tile-invariant like the plain case)
        if constexpr (ST) rel[i] = (pr * p.W + pc) * 8;
        else rel[i] = (p.ups ? ((pr - 1) >> 1) * (p.W >> 1) + ((pc - 1) >> 1) : pr * p.W + pc) * p.x_ld + ((c ^ ws_swz<CP>(ps)) << 3);
        prpc[i] = s < patch_slots ? ((unsigned)pr << 16) | (unsigned)pc : 0x7fff0000u;        // beyond the patch: never inside the image -> zero page
    }
    // ---- per-lane fragment rows: tile pixel i = (wave*TM + b)*32 + (lane & 31) -> (ty, tx) ------------------------------
    int slot0[TM]; unsigned tyx[TM];
#pragma unroll
    for (int b = 0; b < TM; b++) {
        const int i = (wave * TM + b) * 32 + l31;
        const int ty = (int)sm_fastdiv((unsigned)i, p.ws_tw_mul, p.ws_tw_sh), tx = i - ty * TW;
        const bool in = i < TH * TW;
        slot0[b] = in ? (ty * PW + tx) * SS : 0;
        tyx[b] = in ? ((unsigned)ty << 16) | (unsigned)tx : 0x7fff7fffu;                        // outside the tile: fails every bounds check
    }
    int swn[TN];
#pragma unroll
    for (int a = 0; a < TN; a++) swn[a] = ws_swz<CP>(a * 32 + l31);

    // ---- this block's contiguous tile range; tile id -> (image, column band, row band), row band fastest ---------------
    const int T = p.ws_ntiles, G = gridDim.x;
    const int t_begin = (int)((long)blockIdx.x * T / G), t_end = (int)((long)(blockIdx.x + 1) * T / G);
    auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
        n = (int)sm_fastdiv((unsigned)t, p.ws_txy_mul, p.ws_txy_sh);
        const int r = t - n * (p.ws_tiles_x * p.ws_tiles_y);
        const int bx = (int)sm_fastdiv((unsigned)r, p.ws_ty_mul, p.ws_ty_sh), by = r - bx * p.ws_tiles_y;
        y0 = by * TH; x0 = bx * TW;
    };
    if constexpr (NLW > 0) {
        if (wave >= NW) {
            // =========================== LOADER WAVES ===========================
            constexpr int NLT = NLW * 64, LNI = (NIC * NT + NLT - 1) / NLT;      // patch pixels per loader thread (launcher: patch_slots <= NIC * NT)
            constexpr unsigned DEAD = 0x80000000u;
            const int ltid = tid - NT;
            const int C = p.in_C, IH = p.in_H, IW = p.in_W;
            const unsigned plane = (unsigned)IH * (unsigned)IW;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xf, 0, (int)((unsigned)p.N * (unsigned)C * plane * 4u), 0x00020000);
            int lpr[LNI], lpc[LNI], lrel[LNI];
#pragma unroll
            for (int i = 0; i < LNI; i++) {
                const int sl = i * NLT + ltid;
                const int pr = (int)sm_fastdiv((unsigned)sl, p.ws_pw_mul, p.ws_pw_sh);
                lpr[i] = sl < patch_slots ? pr : -0x40000000; lpc[i] = sl - pr * PW;   // beyond the patch: a row far outside the image (nothing read, nothing written)
                lrel[i] = sl < patch_slots ? pr * IW + lpc[i] : 0;                       // dword offset from the tile's first input pixel (tile-invariant)
            }
            // the loads of a patch are issued a whole tile before they are converted and written (two register sets, alternating): what a loader
            // wave waits for at the top of a tile was requested one tile ago, and the next request is already out before it writes.
            // The VALU is scarce (a loader and an MFMA wave share a SIMD, ~4 clk per vector instruction): per load only the select of a dead offset is a
            // vector instruction — the vector offset is the offset of the ROW that is read (the descriptor's range check sees it alone), the plane offset
            // rides in the instruction's scalar offset operand.
            // ccv = the number of planes as a compile-time constant (3: RGB, the loads of the fourth plane vanish), or 0 = runtime (dead offset by select)
            auto request = [&](float (&v)[LNI][2][4], int t, auto ccv) __attribute__((always_inline)) {
                constexpr int CC = decltype(ccv)::value;
                int n, y0, x0; tile_origin(t, n, y0, x0);
                const int r0 = y0 * SS - p.in_pt, c0 = x0 * SS - p.in_pl;               // input pixel of patch pixel (0, 0), top row
                const int base = (int)((unsigned)n * (unsigned)C * plane) + r0 * IW + c0;
                const bool tile_on = t < t_end;
#pragma unroll
                for (int i = 0; i < LNI; i++) {
                    const int r = r0 + lpr[i], c = c0 + lpc[i];
                    const bool okc = (unsigned)c < (unsigned)IW && tile_on;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const unsigned voff = (okc && (unsigned)(r + h) < (unsigned)IH) ? (unsigned)(base + lrel[i] + h * IW) << 2 : DEAD;
#pragma unroll
                        for (int ch = 0; ch < 4; ch++) {
                            if constexpr (CC > 0) {
                                if (ch < CC) v[i][h][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)((unsigned)ch * plane * 4u), 0));   // (folds: the loop is unrolled)
                                else v[i][h][ch] = 0.f;
                            } else v[i][h][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ch < C ? voff : DEAD), (int)((unsigned)ch * plane * 4u), 0));
                        }
                    }
                }
            };
            auto deposit = [&](const float (&v)[LNI][2][4], int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < LNI; i++) {
                    const half8_t o = {(half_t)v[i][0][0], (half_t)v[i][0][1], (half_t)v[i][0][2], (half_t)v[i][0][3], (half_t)v[i][1][0], (half_t)v[i][1][1], (half_t)v[i][1][2], (half_t)v[i][1][3]};
                    if (lpr[i] >= 0) *(half8_t*)(patch0 + buf * WS_PSTRIDE + (i * NLT + ltid) * 16) = o;
                }
            };
            auto lrun = [&](auto ccv) __attribute__((always_inline)) {
                float va[LNI][2][4], vb[LNI][2][4];
                request(va, t_begin, ccv); deposit(va, 0);
                request(va, t_begin + 1, ccv);                        // (beyond the range: dead offsets, nothing is read)
#pragma unroll 1
                for (int t = t_begin; t < t_end; t += 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my share of patch t is in LDS
                    __builtin_amdgcn_s_barrier();                         // ... everybody's; the MFMA waves are done with the other buffer
                    request(vb, t + 2, ccv);
                    if (t + 1 < t_end) deposit(va, 1);
                    if (t + 1 >= t_end) break;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    request(va, t + 3, ccv);
                    if (t + 2 < t_end) deposit(vb, 0);
                }
            };
            if (C == 3) lrun(std::integral_constant<int, 3>()); else lrun(std::integral_constant<int, 0>());
            return;
        }
    }
    auto issue_patch = [&](int buf, int t) {
        if constexpr (NLW > 0) return;
        int n, y0, x0; tile_origin(t, n, y0, x0);
        const int iy0 = ST ? y0 * SS : y0 - 1, ix0 = ST ? x0 * SS : x0 - 1;      // input pixel of patch pixel (0, 0)
        const half_t* xb = p.ups ? p.x + (((long)n * (p.H >> 1) + (y0 >> 1)) * (p.W >> 1) + (x0 >> 1)) * p.x_ld
                                 : p.x + (((long)n * p.H + iy0) * p.W + ix0) * p.x_ld;             // dereferenced only inside the image
        unsigned char* dst = patch0 + buf * WS_PSTRIDE + wave * 1024;
#pragma unroll
        for (int i = 0; i < NIC; i++) {
            if (i < NI && i * NT + wave * 64 < patch_slots) {         // wave-uniform
                const int pr = (int)(prpc[i] >> 16), pc = (int)(prpc[i] & 0xffff);
                const bool in = (unsigned)(iy0 + pr) < (unsigned)p.H && (unsigned)(ix0 + pc) < (unsigned)p.W;
                SM_GLDS16_AUX(in ? xb + rel[i] : zero, dst + i * NT * 16, SM_WS_X_NT);
            }
        }
    };

    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;            // f(v) = v < 0 ? v*sl : v  (ReLU family only: launcher)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 outv[TN][TM][2];                                            // packed results of the PREVIOUS tile: [a][b][j] = couts a*32 + 16j + 8*hi .. +7 of pixel b
    int pn = 0, py0 = 0, px0 = 0; bool pending = false;
    u32x4 rres[HX ? TM : 1][HX ? HXC : 1][2];                         // HX: residual rows of the PREVIOUS tile, [b][c][j] = head channels 32c + 16j + 8hi .. +7 of pixel b
    const unsigned char* const aw3 = w3l + l31 * ROWB3;               // HX: A fragment of head row 32c + l31 = aw3 + c*32*ROWB3 + (((2ks + hi) ^ sw3) << 4)
    const int sw3 = ws_swz<CP3>(l31);                                 // (rows 32c + l31: the key does not depend on c)
    // HX addresses its two streams through buffer descriptors: a dead lane (pixel outside the image / the tile) carries an offset beyond
    // num_records, so its loads return 0 and its stores are dropped — no branch anywhere in the head epilogue (one scheduling region
    // for all chunks: the MFMAs of one chunk run beside the VALU work of its neighbours).  Launcher: both tensors < 2 GB.
    constexpr unsigned HX_DEAD = 0x80000000u;
    __amdgpu_buffer_rsrc_t rs_y3, rs_r3;
    if constexpr (HX) {
        const unsigned pix_all = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W;
        rs_y3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y3, 0, (int)(pix_all * (unsigned)p.y3_ld * 2u), 0x00020000);
        rs_r3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.res3, 0, p.res3 ? (int)(pix_all * (unsigned)p.res3_ld * 2u) : 0, 0x00020000);
    }
    // one-hot A fragments: E_j[r][k] = (r == 16j + k), so that mfma(E_j, R_j, acc) adds the residual rows R_j (which arrive in the B layout:
    // 8 consecutive channels per lane) to the accumulator tile — exact (1.0 x fp16, one nonzero product per element)
    half8_t onehot[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) onehot[j][i] = (half_t)((HX && l31 == 16 * j + 8 * hi + i) ? 1.f : 0.f);
    auto load_res = [&](int n, int y0, int x0) __attribute__((always_inline)) {
        if constexpr (HX) {
#pragma unroll
            for (int b = 0; b < TM; b++) {
                const int oy = y0 + (int)(tyx[b] >> 16), ox = x0 + (int)(tyx[b] & 0xffff);
                const bool live = oy < p.H && ox < p.W && (int)(tyx[b] >> 16) < TH;
                const unsigned off = (live && !(p.hx_dbg & 2)) ? (unsigned)(((n * p.H + oy) * p.W + ox) * p.res3_ld + 8 * hi) * 2u : HX_DEAD;
#pragma unroll
                for (int c = 0; c < HXC; c++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        rres[b][c][j] = c * 32 < p.n3 ? __builtin_amdgcn_raw_buffer_load_b128(rs_r3, (int)(off + (unsigned)(c * 64 + 32 * j)), 0, 0) : u32x4{0, 0, 0, 0};   // (wave-uniform select)
            }
        }
    };
    auto hx_store = [&](auto relu0c) __attribute__((always_inline)) {
        // expand head: outv[a][b][j] is the MFMA B fragment of k step 2a + j (see HEAD below); per 32 head channels 4 MFMAs, + the
        // residual (two one-hot MFMAs), + b3 in fp32, activation, one rounding, lane swaps, 16-byte stores.
        constexpr bool RELU0 = decltype(relu0c)::value;               // slope 0: ReLU as a packed fp16 max after the rounding (max commutes with it)
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int oy = py0 + (int)(tyx[b] >> 16), ox = px0 + (int)(tyx[b] & 0xffff);
            const bool live = oy < p.H && ox < p.W && (int)(tyx[b] >> 16) < TH;
            const unsigned off = (live && !(p.hx_dbg & 1)) ? (unsigned)(((pn * p.H + oy) * p.W + ox) * p.y3_ld + 8 * hi) * 2u : HX_DEAD;
            // four chunks (128 head channels) at a time: their MFMA chains are independent, so the matrix pipe never waits for its own result
            // and the VALU work of one chunk runs under the MFMAs of the others (one wave per SIMD: nothing else would fill the gaps)
#pragma unroll
            for (int g = 0; g < HXC / 4; g++) {
                if (g * 128 < p.n3 && !(p.hx_dbg & 4)) {              // wave-uniform (n3 = 256: both groups)
                    floatx16 a3[4];
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
#pragma unroll
                        for (int r = 0; r < 16; r++) a3[cc][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < NOUT / 16; ks++)
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) {
                            const half8_t wf = *(const half8_t*)(aw3 + (g * 4 + cc) * 32 * ROWB3 + (((ks * 2 + hi) ^ sw3) << 4));
                            a3[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(half8_t, outv[ks / 2][b][ks % 2]), a3[cc], 0, 0, 0);
                        }
                    // (no residual: the rows were loaded through an empty descriptor and are 0 — adding them is exact, and cheaper than a branch here.
                    //  The bias is added AFTER the residual, on the way out of the accumulators: between the two MFMA groups it would cost a round
                    //  trip of all 64 accumulator registers through the VALU; without a residual the result is still the tiled kernels' acc + bias.)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) a3[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(onehot[j], __builtin_bit_cast(half8_t, rres[b][g * 4 + cc][j]), a3[cc], 0, 0, 0);
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        const int c = g * 4 + cc;
                        unsigned d[4][2];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const floatx4 bq = *(const floatx4*)(b3l + c * 32 + 8 * q + 4 * hi);
#pragma unroll
                            for (int h2 = 0; h2 < 2; h2++) {
                                float v0 = a3[cc][4 * q + 2 * h2] + bq[2 * h2], v1 = a3[cc][4 * q + 2 * h2 + 1] + bq[2 * h2 + 1];
                                if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * p.sl3 : v0; v1 = v1 < 0.f ? v1 * p.sl3 : v1; }
                                half2_t hv = {(half_t)v0, (half_t)v1};
                                if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                                d[q][h2] = __builtin_bit_cast(unsigned, hv);
                            }
                        }
                        const unsigned offc = c * 32 < p.n3 ? off : HX_DEAD;      // (a group's chunks beyond n3 are computed on whatever the LDS holds and dropped)
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                            sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]}, rs_y3, (int)(offc + (unsigned)(c * 64 + 32 * j)), 0, SM_ST_AUX);
                        }
                    }
                }
            }
        }
    };
    auto store_prev = [&]() __attribute__((always_inline)) {          // (three call sites: left out of line, its captured arrays would live in scratch)
        if constexpr (HX) {
            if (p.sl3 == 0.f) hx_store(std::true_type()); else hx_store(std::false_type());
        }
        if constexpr (!ST && HEAD) {
            // 1x1 head: outv[a][b][j] (couts a*32 + 16j + 8hi .. +7 of pixel lane & 31, already activated and rounded) IS the MFMA B fragment of
            // k step 2a + j, so y3 = w3 . y costs NOUT/16 MFMAs per fragment and no memory; register 4q + r <-> head channel 8q + 4hi + r
            if (p.y3) {
                // A fragments of w3 [n3][NOUT] (row = lane & 31, zero beyond n3), one per 16 input channels: re-read per tile (L1-resident, <= 2 KB) so
                // that layers without a head carry no registers for it
                half8_t w3f[NOUT / 16];
                const half8_t hz8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < NOUT / 16; ks++) w3f[ks] = l31 < p.n3 ? *(const half8_t*)(p.w3 + l31 * NOUT + ks * 16 + hi * 8) : hz8;
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    floatx16 a3;
#pragma unroll
                    for (int r = 0; r < 16; r++) a3[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < NOUT / 16; ks++) a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[ks], __builtin_bit_cast(half8_t, outv[ks / 2][b][ks % 2]), a3, 0, 0, 0);
                    const int oy = py0 + (int)(tyx[b] >> 16), ox = px0 + (int)(tyx[b] & 0xffff);
                    if (oy < p.H && ox < p.W && (int)(tyx[b] >> 16) < TH) {
                        half_t* dst = p.y3 + (((long)pn * p.H + oy) * p.W + ox) * p.y3_ld + 4 * hi;
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (8 * q < p.n3) {
                                half4_t o;
#pragma unroll
                                for (int r = 0; r < 4; r++) { float v = a3[4 * q + r] + (p.b3 ? p.b3[8 * q + 4 * hi + r] : 0.f); o[r] = (half_t)(v < 0.f ? v * p.sl3 : v); }
                                *(half4_t*)(dst + 8 * q) = o;
                            }
                    }
                }
            }
        }
        if constexpr (TM == 2 && !ST) {
            // fused 2x2 / stride-2 max-pool of the output (p.y2; launcher: TW == 32, even H and W, so a wave's two fragments are image rows
            // 2k and 2k+1 of the SAME columns and lanes 2i, 2i+1 are neighbouring columns): max over the two fragments, then over the lane
            // pair (DPP quad_perm [1,0,3,2]); even lanes store.  Max of fp16 values is exact and commutes with the ReLU-family activation.
            if (p.y2) {
                const int oy = py0 + (int)(tyx[0] >> 16), ox = px0 + (int)(tyx[0] & 0xffff);
                const bool live = oy < p.H && ox < p.W && (int)(tyx[0] >> 16) < TH && !(l31 & 1);
                half_t* dst = p.y2 + (((long)pn * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.y2_ld + 8 * hi;
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        half8_t m = __builtin_elementwise_max(__builtin_bit_cast(half8_t, outv[a][0][j]), __builtin_bit_cast(half8_t, outv[a][1][j]));
                        u32x4 mu = __builtin_bit_cast(u32x4, m), nb;
#pragma unroll
                        for (int k = 0; k < 4; k++) nb[k] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mu[k], 0xB1, 0xf, 0xf, false);
                        m = __builtin_elementwise_max(m, __builtin_bit_cast(half8_t, nb));
                        if (live) *(half8_t*)(dst + a * 32 + 16 * j) = m;
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int oy = py0 + (int)(tyx[b] >> 16), ox = px0 + (int)(tyx[b] & 0xffff);
            if (p.y && oy < OH && ox < OW && (int)(tyx[b] >> 16) < TH) {
                half_t* dst = p.y + (((long)pn * OH + oy) * OW + ox) * p.y_ld + 8 * hi;
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int j = 0; j < 2; j++) { if constexpr (!(ABL & 8)) *(u32x4*)(dst + a * 32 + 16 * j) = outv[a][b][j]; else asm volatile("" ::"v"(outv[a][b][j]), "v"(dst)); }
            }
        }
    };

    // ---- fragment addresses: tile-invariant, computed once.  A: byte offset inside a patch of (fragment b, tap, k step) for this lane;
    // B: the weight image never moves — one register per k step, the tap is an instruction offset.
    constexpr int NS = NTAP * KS;                                     // k steps of 16 per tile
    unsigned aoff[TM][NS];
#pragma unroll
    for (int b = 0; b < TM; b++)
#pragma unroll
        for (int tap = 0; tap < NTAP; tap++) {
            const int q = ST ? slot0[b] + 2 * tap * PW : slot0[b] + (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
                aoff[b][tap * KS + ks] = ws_lds_addr(patch0) + (ST ? (unsigned)((q + ks * 2 + hi) << 4) : (unsigned)(q * ROWB + (((ks * 2 + hi) ^ ws_swz<CP>(q)) << 4)));
        }
    // (ds_read offsets are 16 bits: taps 0..4 are addressed from the first base, taps 5..8 from a second one)
    constexpr int TAPB = NOUT * ROWB;
    static_assert(4 * TAPB < 65536, "tap offset field");
    unsigned boff[2][TN][KS];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            boff[0][a][ks] = ws_lds_addr(wl) + (unsigned)(((a * 32 + l31) * CP + ((ks * 2 + hi) ^ swn[a])) * 16);
            boff[1][a][ks] = boff[0][a][ks] + (NTAP > 5 ? 5 : 0) * TAPB;
        }

    // one tile: DMA pieces of the NEXT patch spread between the k steps (a VMEM instruction stalls its wave while the address queue
    // is full: issued as one burst the 11 pieces cost ~2 k clk of dead issue time per tile), fragments requested PF k steps ahead of
    // the MFMAs that consume them (the compiler's own schedule waited lgkmcnt(0) before every MFMA pair: <= 50 % MFMA duty)
    constexpr int PF = (TM + TN) * 4 <= 15 ? 4 : 3;                    // k steps of fragment prefetch (lgkmcnt field: PF * (TM + TN) <= 15)
    auto do_tile = [&](auto bufc, auto relu0c, int t) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool RELU0 = decltype(relu0c)::value;               // slope 0: ReLU as a packed fp16 max AFTER the rounding (max commutes with it): 1 instead of 6 VALU per pair
        int n, y0, x0; tile_origin(t, n, y0, x0);
        // the NEXT patch (the last tile of the range re-fetches itself: no branch anywhere in the k loop — behind a join hipcc waits
        // lgkmcnt(0) for fragment reads it has just issued).  A wave without slots in piece i sends it to the dummy sink.
        int nn, ny0, nx0; tile_origin(t + 1 < t_end ? t + 1 : t, nn, ny0, nx0);
        const int iy0 = ST ? ny0 * SS : ny0 - 1, ix0 = ST ? nx0 * SS : nx0 - 1;
        const half_t* xb = p.ups ? p.x + (((long)nn * (p.H >> 1) + (ny0 >> 1)) * (p.W >> 1) + (nx0 >> 1)) * p.x_ld
                                 : p.x + (((long)nn * p.H + iy0) * p.W + ix0) * p.x_ld;
        const unsigned dst_a = ws_lds_addr(patch0) + (BUF ^ 1) * WS_PSTRIDE + wave * 1024, dummy_a = ws_lds_addr(dummy);
        auto piece = [&](int i) {
            const bool mine = i * NT + wave * 64 < patch_slots;       // wave-uniform
            const int pr = (int)(prpc[i] >> 16), pc = (int)(prpc[i] & 0xffff);
            const bool in = (unsigned)(iy0 + pr) < (unsigned)p.H && (unsigned)(ix0 + pc) < (unsigned)p.W;
            ws_glds16(in ? xb + rel[i] : zero, (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? dst_a + i * NT * 16 : dummy_a)));
        };
        floatx16 acc[TN][TM];                                         // (never zeroed: the first k step accumulates onto a literal 0, an inline constant of the MFMA)
        // Fragment reads are asm statements with HAND-COUNTED waits: whatever sits between two compiler-visible LDS reads (a DMA builtin,
        // an asm statement) makes hipcc wait lgkmcnt(0) before the next MFMA, i.e. for reads it has only just issued.  DS operations
        // return in order, so before the MFMAs of k step s exactly the fetches of steps s+1 .. s+PF may still be in flight.
        half8_t fx[PF + 1][TM], fw[PF + 1][TN];
        auto fetch = [&](auto sc) {
            constexpr int S2 = decltype(sc)::value, TAP = S2 / KS, KSI = S2 % KS, SLOT = S2 % (PF + 1);
#pragma unroll
            for (int b = 0; b < TM; b++) { half8_t& d = fx[SLOT][b]; const unsigned ad = aoff[b][S2]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(BUF * WS_PSTRIDE)); }
#pragma unroll
            for (int a = 0; a < TN; a++) { half8_t& d = fw[SLOT][a]; const unsigned ad = boff[TAP >= 5][a][KSI]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"((TAP >= 5 ? TAP - 5 : TAP) * TAPB)); }
        };
        constexpr int RD = TM + TN;                                   // reads per fetch
        static_assert(PF * RD <= 15, "lgkmcnt field");
        if constexpr (ABL & 4) {
#pragma unroll
            for (int i = 0; i <= PF; i++) {
#pragma unroll
                for (int b = 0; b < TM; b++) fx[i][b] = half8_t{1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
                for (int a = 0; a < TN; a++) fw[i][a] = half8_t{1, 2, 3, 4, 5, 6, 7, 8};
            }
        }
        if constexpr (!(ABL & 4)) sm_static_for<0, PF>([&](auto sc) { fetch(sc); });
        if (pending) store_prev();
        load_res(n, y0, x0);                                          // HX: this tile's residual rows, a whole k loop ahead of their use
        sm_static_for<0, NS>([&](auto sc) {
            constexpr int S2 = decltype(sc)::value;
#pragma unroll
            for (int i = 0; i < NIC; i++) if (i * NS / (3 * NIC) == S2 && !(ABL & 1) && NLW == 0) piece(i);   // one piece per k step over the FIRST THIRD of the tile: the rest of the tile is their flight time
            if constexpr (S2 + PF < NS && !(ABL & 4)) fetch(std::integral_constant<int, S2 + PF>());
            constexpr int AHEAD = (NS - 1 - S2 < PF ? NS - 1 - S2 : PF) * RD;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((ABL & 4) ? 15 : AHEAD) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    if constexpr ((ABL & 2) != 0) { asm volatile("" ::"v"(fw[S2 % (PF + 1)][a]), "v"(fx[S2 % (PF + 1)][b])); if constexpr (S2 == 0) { for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f; } }
                    else if constexpr (S2 == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)][b], (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)][b], acc[a][b], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- epilogue in registers: register 4q+r of acc[a][b] <-> cout a*32 + 8q + 4*hi + r of pixel b*32 + (lane & 31) ----
#pragma unroll
        for (int a = 0; a < TN; a++) {
            floatx4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const floatx4*)(bl + a * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int b = 0; b < TM; b++) {
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        float v0 = acc[a][b][4 * q + 2 * h2] + bq[q][2 * h2], v1 = acc[a][b][4 * q + 2 * h2 + 1] + bq[q][2 * h2 + 1];
                        if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * sl : v0; v1 = v1 < 0.f ? v1 * sl : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                        d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    }
                // (q, q+1): low half-wave keeps couts 8q..8q+3 and receives 8q+4..8q+7 from its partner lane; the high half-wave ends
                // up with 8q+8..8q+15 — 8 consecutive channels = 16 bytes per lane
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                    sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                    outv[a][b][j] = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
                }
            }
        }
        pn = n; py0 = y0; px0 = x0; pending = true;
    };

    if (t_begin < t_end) issue_patch(0, t_begin);
    auto run = [&](auto relu0c) __attribute__((always_inline)) {
#pragma unroll 1
        for (int t = t_begin; t < t_end; t += 2) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my share of patch t (issued during tile t-1), my stores of tile t-2, [first tile: the weights]
            __builtin_amdgcn_s_barrier();                                 // everybody's share landed; everybody is done reading the other buffer
            __builtin_amdgcn_sched_barrier(0);
            do_tile(std::integral_constant<int, 0>(), relu0c, t);
            if (t + 1 >= t_end) break;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            do_tile(std::integral_constant<int, 1>(), relu0c, t + 1);
        }
    };
    if (p.act != SM_ACT_NONE && p.slope == 0.f) run(std::true_type()); else run(std::false_type());
    if (pending) store_prev();
}
