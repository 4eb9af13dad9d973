// stem_pool.hpp — image stem + the 3x3 / stride-2 max-pool behind it in ONE launch (ResNet-50: Conv 7x7/2 + ReLU -> MaxPool 3x3/2; MPSCNNConvolutionNode
// Converters.swift:253 followed by MPSCNNPoolingMaxNode :662 with PyTorchPoolPadding's window start, PyTorchPoolPadding.swift:60-75).  Included by conv_igemm.hip.
//
// Why: the stem's output is the largest tensor of the network (bs 64: 64 x 112 x 112 x 64 halfs = 103 MB), written once by the stem and read once by the
// pool (19 us of pure traffic behind a 46 us convolution).  Here it never exists: the convolution is conv3x3_ws.hpp's stem form with loader waves (row-pair
// patches built from the fp32 NCHW batch inside the launch, weights resident in LDS, MFMA waves that touch no global memory at all), its output tile goes to
// an LDS hand-over buffer instead of HBM, and the loader waves — which have time to spare — reduce the PREVIOUS tile's 3x3 windows from there and store the
// pooled pixels (a quarter of the bytes).
//   * a convolution tile is TH x TW output pixels with TH = 2 PH + 1, TW = 2 PW + 1 ODD and tiles advance by TH - 1, TW - 1: the windows of PH x PW pooled
//     pixels (rows 2p .. 2p+2: the reference's window start is 0 for the shapes taken here) lie inside one tile, at the price of one recomputed row and column
//     per tile (15 x 17 tiles: +14 % MFMA work on a launch that is nowhere near MFMA-bound);
//   * hand-over: [pixel][NOUT halfs], 16-byte chunk c of pixel i stored at c ^ ((i >> 1) & 7) (conflict-free for the MFMA waves' 16-byte writes), two buffers
//     alternating per tile under the kernel's ONE barrier per tile;
//   * window taps outside the image (the last pooled row / column of an even-sized map) are skipped — max over the taps that exist, as the reference's
//     pooling with edge clamping does; max of fp16 values is exact, so the result is bit-identical to the two separate launches.
#pragma once

// CX = the number of input planes when it is known at compile time (3: RGB), 0 = read it from the parameters (1 .. 4)
template <int NOUT, int TAPS, int PSB, int CX>
__global__ __launch_bounds__(512) void conv_stem_pool_kernel(ConvParams p) {
    constexpr int NW = 4, TM = 2, NLW = 4, NT = NW * 64, NLT = NLW * 64, CP = 8, TN = NOUT / 32, KS = 4, NIC = 6, CPO = NOUT / 8;
    constexpr int ROWB = 128, W_BYTES = TAPS * NOUT * ROWB, BIAS_BYTES = NOUT * 4, HO_BYTES = NW * TM * 32 * NOUT * 2;
    static_assert(NOUT == 64 || NOUT == 32, "one or two 32-channel fragments");
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned char* const wl = sp_smem;                                // weights [tap][cout][8 chunks], chunk c of row n at c ^ swz(n)
    float* const bl = (float*)(sp_smem + W_BYTES);
    unsigned char* const ho = sp_smem + W_BYTES + BIAS_BYTES;        // hand-over [buffer][pixel][NOUT halfs]
    unsigned char* const patch0 = ho + 2 * HO_BYTES;                 // row-pair patches, PSB bytes apart

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = p.ws_TH, TW = p.ws_TW, SS = p.sH;
    const int PW = (TW - 1) * SS + 8, PH = (TH - 1) * SS + 2 * (TAPS - 1) + 1;      // patch of 16-byte row-pair pixels
    const int patch_slots = PH * PW;
    const int QH = (TH - 1) >> 1, QW = (TW - 1) >> 1;                 // pooled pixels per tile
    const int OH = p.Ho, OW = p.Wo, POH = p.pq_OH, POW = p.pq_OW;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int T = p.ws_ntiles, G = gridDim.x;
    const int t_begin = (int)((long)blockIdx.x * T / G), t_end = (int)((long)(blockIdx.x + 1) * T / G);
    auto tile_origin = [&](int t, int& n, int& by, int& bx) {
        n = (int)sm_fastdiv((unsigned)t, p.ws_txy_mul, p.ws_txy_sh);
        const int r = t - n * (p.ws_tiles_x * p.ws_tiles_y);
        bx = (int)sm_fastdiv((unsigned)r, p.ws_ty_mul, p.ws_ty_sh); by = r - bx * p.ws_tiles_y;
    };

    if (wave >= NW) {
        // =========================== LOADER + POOL WAVES ===========================
        constexpr int LNI = (NIC * NT + NLT - 1) / NLT;
        constexpr unsigned DEAD = 0x80000000u;
        const int ltid = tid - NT;
        const int C = CX ? CX : p.in_C, IH = p.in_H, IW = p.in_W;
        const unsigned plane = (unsigned)IH * (unsigned)IW;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xf, 0, (int)((unsigned)p.N * (unsigned)C * plane * 4u), 0x00020000);
        // patch slot i of this thread: patch pixel (lpr, lpc), tile-invariant; lrel = its dword offset from the tile's first input pixel
        int lpr[LNI], lpc[LNI], lrel[LNI];
#pragma unroll
        for (int i = 0; i < LNI; i++) {
            const int sl = i * NLT + ltid;
            const int pr = (int)sm_fastdiv((unsigned)sl, p.ws_pw_mul, p.ws_pw_sh);
            lpr[i] = sl < patch_slots ? pr : -0x40000000; lpc[i] = sl - pr * PW;
            lrel[i] = sl < patch_slots ? pr * IW + lpc[i] : 0;
        }
        // The VALU is this kernel's scarce resource (one loader and one MFMA wave share a SIMD: ~4 clk per vector instruction): per load only the select
        // of a dead offset is a vector instruction — row (h) and plane (ch) offsets ride in the instruction's scalar offset operand.
        auto request = [&](float (&v)[LNI][2][4], int t) __attribute__((always_inline)) {
            int n, by, bx; tile_origin(t, n, by, bx);
            const int r0 = by * (TH - 1) * SS - p.in_pt, c0 = bx * (TW - 1) * SS - p.in_pl;       // input pixel of patch pixel (0, 0), top row
            const int base = (int)((unsigned)n * (unsigned)C * plane) + r0 * IW + c0;               // (dwords)
            const bool tile_on = t < t_end;
#pragma unroll
            for (int i = 0; i < LNI; i++) {
                const int r = r0 + lpr[i], c = c0 + lpc[i];
                const bool okc = (unsigned)c < (unsigned)IW && tile_on;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    // (the vector offset is the offset of the row that is read — the range check of the descriptor sees it alone, not the scalar part —
                    //  and the plane offset rides in the instruction's scalar offset operand)
                    const unsigned voff = (okc && (unsigned)(r + h) < (unsigned)IH) ? (unsigned)(base + lrel[i] + h * IW) << 2 : DEAD;
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) {
                        if (ch < C) v[i][h][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)((unsigned)ch * plane * 4u), SM_STEM_IN_AUX));   // (wave-uniform branch)
                        else v[i][h][ch] = 0.f;
                    }
                }
            }
        };
        auto deposit = [&](const float (&v)[LNI][2][4], int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < LNI; i++) {
                const half8_t o = {(half_t)v[i][0][0], (half_t)v[i][0][1], (half_t)v[i][0][2], (half_t)v[i][0][3], (half_t)v[i][1][0], (half_t)v[i][1][1], (half_t)v[i][1][2], (half_t)v[i][1][3]};
                if (lpr[i] >= 0) *(half8_t*)(patch0 + buf * PSB + (i * NLT + ltid) * 16) = o;
            }
        };
        // ---- pooling: item = (pooled pixel of the tile, 8-channel vector); tile-invariant part computed once ----
        constexpr int NPI = 2;                                        // QH * QW * CPO <= 64 * 8 = 2 * NLT items per tile
        int q_py[NPI], q_px[NPI]; unsigned q_cv16[NPI], q_off[NPI][9];
#pragma unroll
        for (int k = 0; k < NPI; k++) {
            const int it = ltid + k * NLT, cv = it % CPO, pp = it / CPO;
            const int py = pp / QW, px = pp - py * QW;
            q_py[k] = py < QH ? py : -0x40000000; q_px[k] = px; q_cv16[k] = (unsigned)(cv * 16);
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {                          // byte offset of tap (tp / 3, tp % 3) inside a hand-over buffer
                const unsigned i = py < QH ? (unsigned)((2 * py + tp / 3) * TW + 2 * px + tp % 3) : 0u;
                q_off[k][tp] = i * (NOUT * 2) + ((((unsigned)cv ^ (i >> 1)) & (CPO - 1)) << 4);
            }
        }
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y2, 0, (int)((unsigned)p.N * (unsigned)POH * (unsigned)POW * (unsigned)p.y2_ld * 2u), 0x00020000);
        // Only the taps of window row 2 / column 2 can lie outside the image (window start 2p, launcher); such a tap reads tap (0, 0) again — max is idempotent —
        // so the validity costs one select on the ADDRESS per edge tap and nothing on the data (selects on the value became exec-masked branches).
        auto pool = [&](int t, int buf) __attribute__((always_inline)) {
            int n, by, bx; tile_origin(t, n, by, bx);
            const unsigned hb = ws_lds_addr(ho) + (unsigned)(buf * HO_BYTES);
            const int cy0 = by * (TH - 1), cx0 = bx * (TW - 1);       // convolution pixel of tile pixel (0, 0)
#pragma unroll
            for (int k = 0; k < NPI; k++) {
                const int gy = by * QH + q_py[k], gx = bx * QW + q_px[k];
                const bool live = q_py[k] >= 0 && gy < POH && gx < POW;
                const bool oky = cy0 + 2 * q_py[k] + 2 < OH, okx = cx0 + 2 * q_px[k] + 2 < OW;
                half8_t m;
#pragma unroll
                for (int tp = 0; tp < 9; tp++) {
                    const bool edge_y = tp / 3 == 2, edge_x = tp % 3 == 2;
                    unsigned ad = q_off[k][tp];
                    if (edge_y && edge_x) ad = (oky && okx) ? ad : q_off[k][0];
                    else if (edge_y) ad = oky ? ad : q_off[k][0];
                    else if (edge_x) ad = okx ? ad : q_off[k][0];
                    const half8_t v = *(const __attribute__((address_space(3))) half8_t*)(size_t)(hb + ad);
                    m = tp == 0 ? v : __builtin_elementwise_max(m, v);
                }
                const unsigned off = live ? (unsigned)((((n * POH + gy) * POW + gx) * p.y2_ld) * 2) + q_cv16[k] : DEAD;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), rs_y, (int)off, 0, SM_ST_AUX);
            }
        };
        float va[LNI][2][4], vb[LNI][2][4];
        request(va, t_begin); deposit(va, 0);
        request(va, t_begin + 1);
#pragma unroll 1
        for (int t = t_begin; t < t_end; t += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my share of patch t is in LDS
            __builtin_amdgcn_s_barrier();                             // ... everybody's; the MFMA waves are done with the other patch buffer and wrote tile t-1's hand-over
            // These waves load AND store: with both kinds pending hipcc can only wait vmcnt(0) for a loaded value (the two kinds retire out of order), which
            // would also wait for the request issued a line below.  A compiler-visible vmcnt(0) HERE — where everything pending had a whole tile to complete —
            // empties its scoreboard: deposit() then needs no wait at all and the new request stays in flight across the tile.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // vmcnt(0), expcnt / lgkmcnt untouched
            request(vb, t + 2);
            if (t + 1 < t_end) deposit(va, 1);
            if (t > t_begin) pool(t - 1, 1);
            if (t + 1 >= t_end) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            request(va, t + 3);
            if (t + 2 < t_end) deposit(vb, 0);
            pool(t, 0);
        }
        __builtin_amdgcn_s_barrier();                                 // the last tile's hand-over
        if (t_begin < t_end) pool(t_end - 1, (t_end - 1 - t_begin) & 1);
        return;
    }

    // =========================== MFMA WAVES ===========================
    {   // weights + bias -> LDS (once per block)
        constexpr int WSLOTS = TAPS * NOUT * CP;
#pragma unroll 1
        for (int s0 = wave * 64; s0 < WSLOTS; s0 += NT) {
            const int s = s0 + lane, row = s / CP, c = s & (CP - 1);
            const int tap = row / NOUT, n = row - tap * NOUT;
            SM_GLDS16(p.w + (long)n * (TAPS * 64) + tap * 64 + ((c ^ ws_swz<CP>(n)) << 3), wl + s0 * 16);
        }
        for (int i = tid; i < NOUT; i += NT) bl[i] = p.bias ? p.bias[i] : 0.f;
    }
    int slot0[TM]; unsigned hoff[TM];
#pragma unroll
    for (int b = 0; b < TM; b++) {
        const int i = (wave * TM + b) * 32 + l31;
        const int ty = (int)sm_fastdiv((unsigned)i, p.ws_tw_mul, p.ws_tw_sh), tx = i - ty * TW;
        slot0[b] = i < TH * TW ? (ty * PW + tx) * SS : 0;
        hoff[b] = (unsigned)(i * (NOUT * 2));                         // hand-over row of this lane's pixel (rows beyond the tile are written and never read)
    }
    const int hkey = ((wave * TM * 32 + l31) >> 1) & 7;               // (+ 32 b: the key does not depend on b)
    int swn[TN];
#pragma unroll
    for (int a = 0; a < TN; a++) swn[a] = ws_swz<CP>(a * 32 + l31);
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    constexpr int NS = TAPS * KS;
    unsigned aoff[TM][NS];
#pragma unroll
    for (int b = 0; b < TM; b++)
#pragma unroll
        for (int tap = 0; tap < TAPS; tap++) {
            const int q = slot0[b] + 2 * tap * PW;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) aoff[b][tap * KS + ks] = ws_lds_addr(patch0) + (unsigned)((q + ks * 2 + hi) << 4);
        }
    constexpr int TAPB = NOUT * ROWB;
    unsigned boff[TN][KS];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) boff[a][ks] = ws_lds_addr(wl) + (unsigned)(((a * 32 + l31) * CP + ((ks * 2 + hi) ^ swn[a])) * 16);
    static_assert((TAPS - 1) * TAPB < 65536, "tap offset field");
    constexpr int PF = (TM + TN) * 4 <= 15 ? 4 : 3, RD = TM + TN;
    static_assert(PF * RD <= 15, "lgkmcnt field");

    auto do_tile = [&](auto bufc, auto relu0c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool relu0 = decltype(relu0c)::value;               // slope 0: ReLU as a packed fp16 max after the rounding
        floatx16 acc[TN][TM];
        half8_t fx[PF + 1][TM], fw[PF + 1][TN];
        auto fetch = [&](auto sc) __attribute__((always_inline)) {
            constexpr int S2 = decltype(sc)::value, TAP = S2 / KS, KSI = S2 % KS, SLOT = S2 % (PF + 1);
#pragma unroll
            for (int b = 0; b < TM; b++) { half8_t& d = fx[SLOT][b]; const unsigned ad = aoff[b][S2]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(BUF * PSB)); }
#pragma unroll
            for (int a = 0; a < TN; a++) { half8_t& d = fw[SLOT][a]; const unsigned ad = boff[a][KSI]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(TAP * TAPB)); }
        };
        sm_static_for<0, PF>([&](auto sc) { fetch(sc); });
        sm_static_for<0, NS>([&](auto sc) {
            constexpr int S2 = decltype(sc)::value;
            if constexpr (S2 + PF < NS) fetch(std::integral_constant<int, S2 + PF>());
            constexpr int AHEAD = (NS - 1 - S2 < PF ? NS - 1 - S2 : PF) * RD;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    // (the first k step accumulates onto a literal zero — an inline constant of the instruction: no register zeroing)
                    if constexpr (S2 == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)][b], (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)][b], acc[a][b], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        });
        // epilogue in registers (conv3x3_ws.hpp): bias, activation, one rounding, lane swaps -> 8 consecutive channels per lane -> hand-over buffer BUF
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
                        if constexpr (!relu0) { v0 = v0 < 0.f ? v0 * sl : v0; v1 = v1 < 0.f ? v1 * sl : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if constexpr (relu0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});      // (max commutes with the rounding)
                        d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                    sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                    const int chunk = a * 4 + 2 * j + hi;             // channels 8 chunk .. + 7
                    *(u32x4*)(ho + BUF * HO_BYTES + hoff[b] + ((chunk ^ hkey) & (CPO - 1)) * 16) = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
                }
            }
        }
    };
    auto run = [&](auto relu0c) __attribute__((always_inline)) {
#pragma unroll 1
        for (int t = t_begin; t < t_end; t += 2) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my hand-over writes of tile t-1; [first tile: the weights]
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            do_tile(std::integral_constant<int, 0>(), relu0c);
            if (t + 1 >= t_end) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            do_tile(std::integral_constant<int, 1>(), relu0c);
        }
    };
    if (p.act != SM_ACT_NONE && p.slope == 0.f) run(std::true_type()); else run(std::false_type());
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
