// conv3x3_wsx.hpp — the EXPAND head of conv3x3_ws.hpp (3x3 64 -> 64 / stride 1 / pad 1, then 1x1 to n3 <= 256 channels + residual +
// activation: a ResNet bottleneck's last two layers) with the two layers on DIFFERENT WAVES of one persistent block.
//
// Why: in the single-role kernel (conv3x3_ws.hpp HX) a wave runs the head of tile t-1 and then the k loop of tile t.  The LDS (72 KB of
// 3x3 weights + two patches) leaves one block of four waves per CU, i.e. ONE wave per SIMD: while that wave works through the head's
// ~900 VALU instructions per tile the matrix pipe idles, and while it sits in the k loop's counted waits the VALU idles (measured: the
// head costs 16 of 55 us, additive).  Here a block is 8 waves, two per SIMD:
//   * waves 0..3, PRODUCERS: conv3x3_ws.hpp's tile loop unchanged (weights resident, patch DMA one tile ahead, hand-counted fragment
//     reads) for one 32-pixel fragment each; the finished fragment — bias, activation, rounded to fp16, eight consecutive channels per
//     lane, which IS the MFMA B layout of the head — goes to a 16 KB hand-over buffer in LDS (two of them, alternating per tile);
//   * waves 4..7, CONSUMERS: consumer j owns head channels [64j, 64j + 64) for all four fragments of the tile: its slice of w3 lives in
//     32 registers for the whole launch; per fragment 8 + 4 MFMAs (w3 . y, then the residual rows through two one-hot MFMAs), + b3,
//     activation, one rounding, lane swaps, 16-byte stores — a pixel's 128 contiguous bytes come from ONE wave.  The residual registers
//     of a (fragment, chunk) are refilled with the NEXT tile's rows as soon as they are consumed: one tile of flight time, no double
//     buffer.  Both streams go through buffer descriptors (dead lanes: offset beyond num_records), so the head has no branch.
//   * ONE s_barrier per tile for both roles: after barrier k the producers compute tile k into hand-over buffer k & 1 while the
//     consumers work on tile k-1 from buffer (k-1) & 1.
// Numerics = conv3x3_ws.hpp HX: (acc + residual) + b3 in fp32, one rounding; without a residual bit-identical to the two separate launches.
#pragma once

// SIDE: the head has a second GEMM instead of a residual tensor — y3 = act3(w3 . y + ws . xs + b3), xs = a 64-channel tensor of the same
// spatial size (the block input) and ws [n3][64] (the projection shortcut of a bottleneck's first block: three layers, one launch).  The
// "residual" registers then hold xs rows: a 64-channel row in the store layout IS the B operand of the four k steps, so the loads, their
// refill and the register budget are the residual's; ws sits in 32 more registers per consumer.
template <int NIC = 6, bool SIDE = false>
__global__ __launch_bounds__(512) void conv3x3_wsx_kernel(ConvParams p) {
    constexpr int CIN = 64, NOUT = 64, NWP = 4, NTP = NWP * 64, NT = 512, CP = 8, TN = 2, KS = 4, NTAP = 9;
    constexpr int ROWB = CP * 16, W_BYTES = NTAP * NOUT * ROWB, BIAS_BYTES = NOUT * 4, B3_BYTES = WS_HX_N3 * 4, HO_BYTES = NWP * 32 * NOUT * 2;
    static_assert(BIAS_BYTES + B3_BYTES + 2 * HO_BYTES + W_BYTES + WS_DUMMY == W_BYTES + BIAS_BYTES + ws_hx_bytes(NOUT) + WS_DUMMY, "the launcher sizes the tiles with the HX layout's fixed bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    unsigned char* const wl = ws_smem;
    float* const bl = (float*)(ws_smem + W_BYTES);
    float* const b3l = (float*)(ws_smem + W_BYTES + BIAS_BYTES);
    unsigned char* const ho = ws_smem + W_BYTES + BIAS_BYTES + B3_BYTES;          // hand-over: [buffer][fragment][k step][lane] x 16 bytes
    unsigned char* const dummy = ho + 2 * HO_BYTES;
    unsigned char* const patch0 = dummy + WS_DUMMY;
    constexpr int WS_PSTRIDE = ws_pstride(W_BYTES + BIAS_BYTES + B3_BYTES + 2 * HO_BYTES + WS_DUMMY);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const half_t* zero = (const half_t*)g_zero_page;
    const int TH = p.ws_TH, TW = p.ws_TW, PW = TW + 2, PH = TH + 2;
    const int patch_slots = PH * PW * CP, NI = (patch_slots + NTP - 1) / NTP;

    // ---- 3x3 weights, both biases -> LDS (all eight waves) ----------------------------------------------------------------------
    {
        constexpr int WSLOTS = NTAP * NOUT * CP;
        static_assert(WSLOTS % NT == 0, "whole block passes");
#pragma unroll 1
        for (int s0 = wave * 64; s0 < WSLOTS; s0 += NT) {
            const int s = s0 + lane, row = s / CP, c = s & (CP - 1);
            const int tap = row / NOUT, n = row - tap * NOUT;
            SM_GLDS16(p.w + (long)n * (NTAP * CIN) + tap * CIN + ((c ^ ws_swz<CP>(n)) << 3), wl + s0 * 16);
        }
        for (int i = tid; i < NOUT; i += NT) bl[i] = p.bias ? p.bias[i] : 0.f;
        for (int i = tid; i < WS_HX_N3; i += NT) b3l[i] = (p.b3 && i < p.n3) ? p.b3[i] : 0.f;
    }
    const int T = p.ws_ntiles, G = gridDim.x;
    const int t_begin = (int)((long)blockIdx.x * T / G), t_end = (int)((long)(blockIdx.x + 1) * T / G);
    auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
        n = (int)sm_fastdiv((unsigned)t, p.ws_txy_mul, p.ws_txy_sh);
        const int r = t - n * (p.ws_tiles_x * p.ws_tiles_y);
        const int bx = (int)sm_fastdiv((unsigned)r, p.ws_ty_mul, p.ws_ty_sh), by = r - bx * p.ws_tiles_y;
        y0 = by * TH; x0 = bx * TW;
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    if (wave < NWP) {
        // =========================== PRODUCERS: the 3x3 convolution, one 32-pixel fragment per wave ===========================
        int rel[NIC]; unsigned prpc[NIC];
#pragma unroll
        for (int i = 0; i < NIC; i++) {
            const int s = i * NTP + tid, ps = s / CP, c = s & (CP - 1);
            const int pr = (int)sm_fastdiv((unsigned)ps, p.ws_pw_mul, p.ws_pw_sh), pc = ps - pr * PW;
            rel[i] = (pr * p.W + pc) * p.x_ld + ((c ^ ws_swz<CP>(ps)) << 3);
            prpc[i] = s < patch_slots ? ((unsigned)pr << 16) | (unsigned)pc : 0x7fff0000u;
        }
        int slot0;
        {
            const int i = wave * 32 + l31;
            const int ty = (int)sm_fastdiv((unsigned)i, p.ws_tw_mul, p.ws_tw_sh), tx = i - ty * TW;
            slot0 = i < TH * TW ? ty * PW + tx : 0;
        }
        int swn[TN];
#pragma unroll
        for (int a = 0; a < TN; a++) swn[a] = ws_swz<CP>(a * 32 + l31);
        auto issue_patch = [&](int buf, int t) {
            int n, y0, x0; tile_origin(t, n, y0, x0);
            const int iy0 = y0 - 1, ix0 = x0 - 1;
            const half_t* xb = p.x + (((long)n * p.H + iy0) * p.W + ix0) * p.x_ld;
            unsigned char* dst = patch0 + buf * WS_PSTRIDE + wave * 1024;
#pragma unroll
            for (int i = 0; i < NIC; i++) {
                if (i < NI && i * NTP + wave * 64 < patch_slots) {
                    const int pr = (int)(prpc[i] >> 16), pc = (int)(prpc[i] & 0xffff);
                    const bool in = (unsigned)(iy0 + pr) < (unsigned)p.H && (unsigned)(ix0 + pc) < (unsigned)p.W;
                    SM_GLDS16_AUX(in ? xb + rel[i] : zero, dst + i * NTP * 16, SM_WS_X_NT);
                }
            }
        };
        const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
        constexpr int NS = NTAP * KS;
        unsigned aoff[NS];
#pragma unroll
        for (int tap = 0; tap < NTAP; tap++) {
            const int q = slot0 + (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) aoff[tap * KS + ks] = ws_lds_addr(patch0) + (unsigned)(q * ROWB + (((ks * 2 + hi) ^ ws_swz<CP>(q)) << 4));
        }
        constexpr int TAPB = NOUT * ROWB;
        unsigned boff[2][TN][KS];
#pragma unroll
        for (int a = 0; a < TN; a++)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                boff[0][a][ks] = ws_lds_addr(wl) + (unsigned)(((a * 32 + l31) * CP + ((ks * 2 + hi) ^ swn[a])) * 16);
                boff[1][a][ks] = boff[0][a][ks] + 5 * TAPB;
            }
        constexpr int PF = 4, RD = 1 + TN;
        static_assert(PF * RD <= 15, "lgkmcnt field");
        auto do_tile = [&](auto bufc, auto relu0c, int t) {
            constexpr int BUF = decltype(bufc)::value;
            constexpr bool PRELU0 = decltype(relu0c)::value;           // the 3x3's own activation is a plain ReLU: packed fp16 max after the rounding
            int nn, ny0, nx0; tile_origin(t + 1 < t_end ? t + 1 : t, nn, ny0, nx0);
            const int iy0 = ny0 - 1, ix0 = nx0 - 1;
            const half_t* xb = p.x + (((long)nn * p.H + iy0) * p.W + ix0) * p.x_ld;
            const unsigned dst_a = ws_lds_addr(patch0) + (BUF ^ 1) * WS_PSTRIDE + wave * 1024, dummy_a = ws_lds_addr(dummy);
            auto piece = [&](int i) {
                const bool mine = i * NTP + wave * 64 < patch_slots;
                const int pr = (int)(prpc[i] >> 16), pc = (int)(prpc[i] & 0xffff);
                const bool in = (unsigned)(iy0 + pr) < (unsigned)p.H && (unsigned)(ix0 + pc) < (unsigned)p.W;
                ws_glds16(in ? xb + rel[i] : zero, (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? dst_a + i * NTP * 16 : dummy_a)));
            };
            floatx16 acc[TN];                                          // (the first k step accumulates onto a literal 0)
            half8_t fx[PF + 1], fw[PF + 1][TN];
            auto fetch = [&](auto sc) {
                constexpr int S2 = decltype(sc)::value, TAP = S2 / KS, KSI = S2 % KS, SLOT = S2 % (PF + 1);
                { half8_t& d = fx[SLOT]; const unsigned ad = aoff[S2]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(BUF * WS_PSTRIDE)); }
#pragma unroll
                for (int a = 0; a < TN; a++) { half8_t& d = fw[SLOT][a]; const unsigned ad = boff[TAP >= 5][a][KSI]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"((TAP >= 5 ? TAP - 5 : TAP) * TAPB)); }
            };
            sm_static_for<0, PF>([&](auto sc) { fetch(sc); });
            sm_static_for<0, NS>([&](auto sc) {
                constexpr int S2 = decltype(sc)::value;
#pragma unroll
                for (int i = 0; i < NIC; i++) if (i * NS / (3 * NIC) == S2) piece(i);
                if constexpr (S2 + PF < NS) fetch(std::integral_constant<int, S2 + PF>());
                constexpr int AHEAD = (NS - 1 - S2 < PF ? NS - 1 - S2 : PF) * RD;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TN; a++) {
                    if constexpr (S2 == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)], (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S2 % (PF + 1)][a], fx[S2 % (PF + 1)], acc[a], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // epilogue: bias, activation, one rounding, lane swaps -> k step 2a + j of the head's B operand -> hand-over buffer BUF
#pragma unroll
            for (int a = 0; a < TN; a++) {
                floatx4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; q++) bq[q] = *(const floatx4*)(bl + a * 32 + 8 * q + 4 * hi);
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        float v0 = acc[a][4 * q + 2 * h2] + bq[q][2 * h2], v1 = acc[a][4 * q + 2 * h2 + 1] + bq[q][2 * h2 + 1];
                        if constexpr (!PRELU0) { v0 = v0 < 0.f ? v0 * sl : v0; v1 = v1 < 0.f ? v1 * sl : v1; }
                        half2_t hv = {(half_t)v0, (half_t)v1};
                        if constexpr (PRELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                        d[q][h2] = __builtin_bit_cast(unsigned, hv);
                    }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                    sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                    *(u32x4*)(ho + BUF * HO_BYTES + ((wave * 4 + 2 * a + j) * 64 + lane) * 16) = u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]};
                }
            }
        };
        issue_patch(0, t_begin);
        auto prun = [&](auto relu0c) __attribute__((always_inline)) {
#pragma unroll 1
            for (int t = t_begin; t < t_end; t += 2) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my share of patch t; my hand-over writes of tile t-1; [first tile: the weights]
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                do_tile(std::integral_constant<int, 0>(), relu0c, t);
                if (t + 1 >= t_end) break;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                do_tile(std::integral_constant<int, 1>(), relu0c, t + 1);
            }
        };
        if (p.act != SM_ACT_NONE && p.slope == 0.f) prun(std::true_type()); else prun(std::false_type());
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the last tile's hand-over
        __builtin_amdgcn_s_barrier();
    } else {
        // =========================== CONSUMERS: the 1x1 head, 64 head channels per wave ==========================================
        const int jw = wave - NWP;                                        // head channels [64 jw, 64 jw + 64) = chunks 2 jw, 2 jw + 1
        constexpr unsigned DEAD = 0x80000000u;
        const unsigned pix_all = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W;
        const __amdgpu_buffer_rsrc_t rs_y3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y3, 0, (int)(pix_all * (unsigned)p.y3_ld * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.res3, 0, p.res3 ? (int)(pix_all * (unsigned)p.res3_ld * 2u) : 0, 0x00020000);
        half8_t w3r[2][NOUT / 16];                                        // A fragments of my 64 rows of w3 (row = lane & 31 of chunk cc), resident
        const half8_t hz8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int cc = 0; cc < 2; cc++)
#pragma unroll
            for (int ks = 0; ks < NOUT / 16; ks++) {
                const int row = (2 * jw + cc) * 32 + l31;
                w3r[cc][ks] = row < p.n3 ? *(const half8_t*)(p.w3 + (long)row * NOUT + ks * 16 + hi * 8) : hz8;
            }
        half8_t onehot[2];                                                // E_j[r][k] = (r == 16j + k): mfma(E_j, R_j, acc) adds residual rows that arrive in the B layout
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) onehot[j][i] = (half_t)((!SIDE && l31 == 16 * j + 8 * hi + i) ? 1.f : 0.f);
        half8_t wsr[SIDE ? 2 : 1][NOUT / 16];                             // SIDE: my 64 rows of the second GEMM's weights
        if constexpr (SIDE) {
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
                for (int ks = 0; ks < NOUT / 16; ks++) {
                    const int row = (2 * jw + cc) * 32 + l31;
                    wsr[cc][ks] = row < p.n3 ? *(const half8_t*)(p.w3s + (long)row * NOUT + ks * 16 + hi * 8) : hz8;
                }
        }
        unsigned tyx[NWP];                                                // tile pixel of (fragment f, lane & 31)
#pragma unroll
        for (int f = 0; f < NWP; f++) {
            const int i = f * 32 + l31;
            const int ty = (int)sm_fastdiv((unsigned)i, p.ws_tw_mul, p.ws_tw_sh), tx = i - ty * TW;
            tyx[f] = i < TH * TW ? ((unsigned)ty << 16) | (unsigned)tx : 0x7fff7fffu;
        }
        const bool chunk_on[2] = {(2 * jw) * 32 < p.n3, (2 * jw + 1) * 32 < p.n3};     // wave-uniform
        auto pixel_off = [&](int t, int f, int ld, bool on) -> unsigned {              // byte offset of (tile t, fragment f, lane)'s 64-channel slice, or DEAD
            int n, y0, x0; tile_origin(t, n, y0, x0);
            const int ty = (int)(tyx[f] >> 16), oy = y0 + ty, ox = x0 + (int)(tyx[f] & 0xffff);
            const bool live = on && t < t_end && oy < p.H && ox < p.W && ty < TH;
            return live ? (unsigned)(((n * p.H + oy) * p.W + ox) * ld + 64 * jw + 8 * hi) * 2u : DEAD;
        };
        // (SIDE: every consumer reads the SAME 64 channels of xs — no 64 jw)
        const unsigned r_sub = SIDE ? (unsigned)(64 * jw) * 2u : 0u;
        u32x4 rres[NWP][2][2];                                            // residual rows [fragment][chunk][j] of the tile whose head runs next
        const bool res_on = p.res3 != nullptr && !(p.hx_dbg & 2);
#pragma unroll
        for (int f = 0; f < NWP; f++) {
            unsigned off = pixel_off(t_begin, f, p.res3_ld, res_on); if (off != DEAD) off -= r_sub;
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
                for (int j = 0; j < 2; j++) rres[f][cc][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r3, (int)(off + (unsigned)(cc * 64 + 32 * j)), 0, 0);
        }
        auto run = [&](auto relu0c) __attribute__((always_inline)) {
            constexpr bool RELU0 = decltype(relu0c)::value;               // slope 0: ReLU as a packed fp16 max after the rounding
            __builtin_amdgcn_s_barrier();                                 // barrier 0: nothing to consume yet
#pragma unroll 1
            for (int t = t_begin; t < t_end; t++) {
                __builtin_amdgcn_s_barrier();                             // barrier t - t_begin + 1: tile t's fragments are in hand-over buffer (t - t_begin) & 1
                if (p.hx_dbg & 4) continue;
                const unsigned char* hb = ho + ((t - t_begin) & 1) * HO_BYTES;
#pragma unroll
                for (int f = 0; f < NWP; f++) {
                    half8_t h2[NOUT / 16];
#pragma unroll
                    for (int ks = 0; ks < NOUT / 16; ks++) h2[ks] = *(const half8_t*)(hb + ((f * 4 + ks) * 64 + lane) * 16);
                    const unsigned offy = pixel_off(t, f, p.y3_ld, !(p.hx_dbg & 1));
                    unsigned offr = pixel_off(t + 1, f, p.res3_ld, res_on); if (offr != DEAD) offr -= r_sub;
                    floatx16 a3[2];
#pragma unroll
                    for (int cc = 0; cc < 2; cc++) {
#pragma unroll
                        for (int r = 0; r < 16; r++) a3[cc][r] = 0.f;
#pragma unroll
                        for (int ks = 0; ks < NOUT / 16; ks++) a3[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3r[cc][ks], h2[ks], a3[cc], 0, 0, 0);
                        if constexpr (SIDE) {
#pragma unroll
                            for (int ks = 0; ks < NOUT / 16; ks++) a3[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wsr[cc][ks], __builtin_bit_cast(half8_t, rres[f][ks / 2][ks % 2]), a3[cc], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int j = 0; j < 2; j++) a3[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(onehot[j], __builtin_bit_cast(half8_t, rres[f][cc][j]), a3[cc], 0, 0, 0);
                        }
                    }
                    // these registers are free again: the NEXT tile's rows of the same fragment — a whole tile of flight time
#pragma unroll
                    for (int cc = 0; cc < 2; cc++)
#pragma unroll
                        for (int j = 0; j < 2; j++) rres[f][cc][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r3, (int)(offr + (unsigned)(cc * 64 + 32 * j)), 0, 0);
#pragma unroll
                    for (int cc = 0; cc < 2; cc++) {
                        const int c = 2 * jw + cc;
                        unsigned d[4][2];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const floatx4 bq = *(const floatx4*)(b3l + c * 32 + 8 * q + 4 * hi);
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                float v0 = a3[cc][4 * q + 2 * h] + bq[2 * h], v1 = a3[cc][4 * q + 2 * h + 1] + bq[2 * h + 1];
                                if constexpr (!RELU0) { v0 = v0 < 0.f ? v0 * p.sl3 : v0; v1 = v1 < 0.f ? v1 * p.sl3 : v1; }
                                half2_t hv = {(half_t)v0, (half_t)v1};
                                if constexpr (RELU0) hv = __builtin_elementwise_max(hv, half2_t{0, 0});
                                d[q][h] = __builtin_bit_cast(unsigned, hv);
                            }
                        }
                        const unsigned offc = chunk_on[cc] ? offy : DEAD;
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            sm_permlane32_swap(d[2 * j][0], d[2 * j + 1][0]);
                            sm_permlane32_swap(d[2 * j][1], d[2 * j + 1][1]);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{d[2 * j][0], d[2 * j][1], d[2 * j + 1][0], d[2 * j + 1][1]}, rs_y3, (int)(offc + (unsigned)(cc * 64 + 32 * j)), 0, SM_ST_AUX);
                        }
                    }
                }
            }
        };
        if (p.sl3 == 0.f) run(std::true_type()); else run(std::false_type());
    }
}
