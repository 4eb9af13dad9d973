// bottleneck.hip — a whole ResNet bottleneck block in ONE launch for gfx950:
//     h1 = act1(W1 . x + b1)            1x1 reduce  C -> Cm            (MPSCNNConvolutionNode, Converters.swift:253; ReLU :351)
//     h2 = act2(W2 * h1 + b2)           3x3 / stride 1 / pad 1, Cm -> Cm
//     y  = act3(W3 . h2 + b3 + x)       1x1 expand  Cm -> C = 4 Cm, + the block input (MPSNNAdditionNode :440), ReLU
// Why.  Stages 1 and 2 of ResNet-50 at bs 64 are memory-bound layer by layer: per block the three launches move x three times (reduce input,
// residual, and — written by the previous block — once more) plus h1 and h2 twice each: 204 MB per stage-2 block (37 us at the practical
// 5.5 TB/s) around 19 us of MFMA work, measured 62 us in three launches; the block as a whole only HAS to read x and write y (102 MB).
// Here a block of eight waves owns a tile of 14 x 14 output pixels (56 x 56 images: 16 tiles, 28 x 28: 4 — 256 tiles per bs-64 batch at 28 x 28,
// one per CU) and runs the three layers back to back on it, h1 and h2 living in LDS only (rounded to fp16 exactly where the separate launches
// store them):
//   A  reduce on the 16 x 16 PATCH (the 3x3's halo ring is recomputed: 256 / 196): wave w owns patch fragment w (32 pixels); its X fragments
//      come straight from global memory into registers (16 bytes per lane and k step, a chunk of k steps ahead), W1's A fragments from an LDS
//      ring; patch pixels outside the image get zero X fragments AND a zero in the bias column, so h1 there is act1(0) = 0 — the 3x3's padding;
//   B  3x3 on the patch: wave w < 7 owns output fragment w; B operand = h1 fragment of tap (kh, kw) (LDS, 16-byte slots XOR-swizzled by pixel),
//      A operand = W2 fragments from the ring; k order (tap, channel group) = W2's OHWI rows;
//   C  expand: the wave's h2 fragments (all of K = Cm) are read into registers ONCE and serve every cout chunk; per chunk of 128 couts the W3
//      fragments come from the ring, the residual rows (x at the output pixels, requested a chunk ahead) are added in fp32 after the lane swap,
//      16-byte stores.
// All weight fragments arrive by LDS-DMA in MFMA-fragment order (1 KB per fragment = one wave instruction, lane-linear in LDS: conflict-free
// reads, no registers), a chunk ahead of their use, two buffers per phase; the rings of the three phases reuse the LDS of whichever of h1 / h2 is
// not live (A: h2's, C: h1's).  Numerics = the three launches': fp32 accumulation over k ascending, (acc + bias) [+ residual], activation, one rounding.
#include <algorithm>

#include "common.hpp"

namespace {

struct BtlParams {
    const half_t *x, *w1, *w2, *w3; const float *b1, *b2, *b3; half_t* y;
    int N, H, W, x_ld, y_ld;
    int tiles_x, tiles_y, ntiles;
    unsigned tx_mul, tx_sh, txy_mul, txy_sh;
    float sl1, sl2, sl3;             // activation slopes: f(v) = v < 0 ? v * sl : v  (1 = none, 0 = ReLU)
};
#ifdef SM_TIMELINE
__device__ unsigned long long g_btl_tl[8];
#define BTL_TL(i) do { if (blockIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tl[i] += now - tl_t0; tl_t0 = now; } } while (0)
#else
#define BTL_TL(i)
#endif

inline void btl_fastdiv_make(unsigned d, unsigned* mul, unsigned* sh) { unsigned s = 0; while ((1ull << s) < d) s++; *mul = (unsigned)(((1ull << (31 + s)) / d) + 1); *sh = 31 + s; }
__device__ __forceinline__ unsigned btl_fastdiv(unsigned n, unsigned mul, unsigned sh) { return (unsigned)(((unsigned long long)n * mul) >> sh); }
__device__ __forceinline__ float btl_act(float v, float sl) { return v < 0.f ? v * sl : v; }

#define BTL_GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void btl_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __attribute__((aligned(128))) unsigned int g_btl_zero[32];    // 128 zero bytes: the row of a patch pixel outside the image

// Operand tiles in LDS are SLABS: [rows][64 halfs] = 128 bytes per row, the row's eight 16-byte slots XOR-swizzled by the row (slot s at s ^ (row & 7)), copied
// by the LDS-DMA with the swizzle on the SOURCE side — a wave instruction = 8 rows x 8 slots, i.e. eight whole 128-byte lines (a fragment-order gather, one
// 32-byte piece of 32 different lines per instruction, moved every line four times: 13.8 k clk for W1 alone).  Fragment of rows r0 .. r0+31, k step ks of the
// slab: lane (l31, hi) reads slot (2 ks + hi) ^ (row & 7) of row r0 + l31: conflict-free.
// LDS is ONE pool whose layout follows the phases (h1 / h2 are written in a phase's EPILOGUE, so while a phase multiplies, everything but its input is free):
//   A  ring of NSA stages, stage = X slab [256 patch pixels] + W1 slab [Cm rows] of one 64-channel block of K; counted vmcnt (the phase issues nothing but
//      these DMAs);                                                                                                       epilogue: h1 -> [0, R1)
//   B  [0, R1) h1 | [R1, ..) W2 ring of NB slabs [Cm rows] (k order (tap, channel) = W2's OHWI rows: a slab = 64 consecutive k), NB - 1 in flight;
//                                                                                                                          epilogue: h2 -> [R1, R1 + R2)
//   C  h2 read into registers, barrier, then [0, W3 bytes) all of W3 as slabs [C rows] per 64-channel block of Cm, requested at once
template <int CM>
__global__ __launch_bounds__(512) void btl_kernel(BtlParams p) {
    constexpr int C = 4 * CM, NF1 = CM / 32, KS2 = 9 * CM / 16, KS3 = CM / 16, NF3 = C / 32, TO = 14, PW = 16, OPX = TO * TO;
    constexpr int HS = CM * 2;                                          // bytes per pixel row of h1 / h2 (CM / 8 slots of 16 bytes)
    constexpr int R1 = 256 * HS, R2 = 224 * HS;                         // h1 patch; h2 (rows >= 196: never written, read by idle lanes only)
    constexpr int POOL = 156 * 1024;
    constexpr int XSL = 256 * 128, W1SL = CM * 128, STA = XSL + W1SL, NKA = C / 64;          // phase A: slabs, stage bytes, stages of K
    constexpr int NSA = POOL / STA < NKA ? POOL / STA : NKA;            // ring slots
    constexpr int LA = (256 + CM) / 64;                                 // DMA instructions per wave and stage (64 rows per block-wide round)
    constexpr int W2SL = CM * 128, NKB = KS2 / 4;                       // phase B: slab bytes, slabs
    constexpr int NB = (POOL - R1) / W2SL < NKB ? (POOL - R1) / W2SL : NKB;
    constexpr int LB = CM / 64;
    constexpr int W3SL = C * 128, NK3 = CM / 64;                        // phase C: slab [C rows] per 64-channel block of K
    static_assert(NSA >= 2 && NSA <= 9 && NB >= 2 && NB <= 9 && NK3 * W3SL <= POOL && R1 + R2 <= POOL, "LDS plan");
    constexpr int NCC = 4, NCHC = NF3 / NCC;                            // phase C: cout fragments per chunk, chunks
    extern __shared__ __attribute__((aligned(16))) unsigned char btl_smem[];
    unsigned char* const h1 = btl_smem;
    unsigned char* const h2 = btl_smem + R1;
    float* const b1l = (float*)(btl_smem + POOL);                       // [CM] [CM] [C]
    float* const b2l = b1l + CM;
    float* const b3l = b2l + CM;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const half8_t hz = {0, 0, 0, 0, 0, 0, 0, 0};
    const half_t* const zero = (const half_t*)g_btl_zero;
    auto swz = [](int px) { return CM == 128 ? (px & 15) : ((px >> 1) & 7); };      // h1 / h2: slot s of pixel px is stored at s ^ swz(px)

    for (int i = tid; i < CM; i += 512) { b1l[i] = p.b1 ? p.b1[i] : 0.f; b2l[i] = p.b2 ? p.b2[i] : 0.f; }
    for (int i = tid; i < C; i += 512) b3l[i] = p.b3 ? p.b3[i] : 0.f;

    // slab of `rows` rows (a multiple of 64): row r <- 64 halfs at w + r * ld + k0; thread tid of round i covers row 64 i + tid / 8, LDS slot tid % 8 <- source slot ^ (row & 7)
    const int drow = tid >> 3, dslot = tid & 7;
    auto dma_w = [&](const half_t* w, long ld, int k0, int rows, unsigned char* dst) __attribute__((always_inline)) {
        for (int i = 0; i < rows / 64; i++) {
            const int r = i * 64 + drow;
            BTL_GLDS16(w + (long)r * ld + k0 + ((dslot ^ (r & 7)) << 3), dst + (i * 512 + wave * 64) * 16);
        }
    };
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        n = (int)btl_fastdiv((unsigned)t, p.txy_mul, p.txy_sh);
        const int r = t - n * (p.tiles_x * p.tiles_y), by = (int)btl_fastdiv((unsigned)r, p.tx_mul, p.tx_sh);
        oy0 = by * TO; ox0 = (r - by * p.tiles_x) * TO;
    };
    // lane constants: my patch pixel (phase A), my output pixel (phases B, C)
    const int pp = wave * 32 + l31;
    const int q = wave * 32 + l31, qy = q / TO, qx = q - qy * TO;       // (wave 7: q >= 224, idle in B and C)
    const bool act_q = wave < 7;
    const int qpatch = (qy < TO ? qy : 0) * PW + qx;                    // patch pixel of tap (0, 0) of my output pixel
    const unsigned foff = (unsigned)(pp * 128), fkey = (unsigned)(pp & 7);                 // phase A: my B fragment row inside an X slab
    const unsigned woff = (unsigned)(l31 * 128), wkey = (unsigned)(l31 & 7);               // A fragments: row a*32 + l31 of a weight slab ((a*32 + l31) & 7 = l31 & 7)
#ifdef SM_TIMELINE
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_t0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();

#pragma unroll 1
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
        BTL_TL(0);
        // =============================== phase A: h1 = act1(W1 . x + b1) on the 16 x 16 patch =======================================
        {
            // this thread's four X rows of a stage (round i: patch pixel 64 i + tid / 8): source pointer (slot swizzle included) or the zero page
            const half_t* xrow[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = i * 64 + drow, iy = oy0 - 1 + (r >> 4), ix = ox0 - 1 + (r & 15);
                const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                xrow[i] = in ? p.x + (((long)n * p.H + iy) * p.W + ix) * p.x_ld + ((dslot ^ (r & 7)) << 3) : nullptr;
            }
            auto issue_a = [&](int kb) __attribute__((always_inline)) {          // K block kb -> slot kb % NSA
                unsigned char* const st = btl_smem + (kb % NSA) * STA;
#pragma unroll
                for (int i = 0; i < 4; i++) BTL_GLDS16(xrow[i] ? xrow[i] + kb * 64 : zero + (dslot << 3), st + (i * 512 + wave * 64) * 16);
                dma_w(p.w1, C, kb * 64, CM, st + XSL);
            };
            const int iyq = oy0 - 1 + (pp >> 4), ixq = ox0 - 1 + (pp & 15);
            const bool inimg = (unsigned)iyq < (unsigned)p.H && (unsigned)ixq < (unsigned)p.W;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the previous tile's stores: loads and stores retire out of order with each other, the counted waits below see loads only)
            int issued = 0;
#pragma unroll
            for (int s2 = 0; s2 < NSA; s2++) { issue_a(issued); ++issued; }
            floatx16 acc[NF1];
            {   // bias as a k step of its own: A = {hi(b), lo(b)} in k = 0, 1, B = {1, 1} inside the image (0 outside: h1 = act1(0) = 0 there)
                half8_t fone = hz; if (inimg && !hi) { fone[0] = (half_t)1.f; fone[1] = (half_t)1.f; }
#pragma unroll
                for (int a = 0; a < NF1; a++) {
                    half8_t fb = hz; const float bv = b1l[a * 32 + l31];
                    if (!hi) { const half_t bh = (half_t)bv; fb[0] = bh; fb[1] = (half_t)(bv - (float)bh); }
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fone, (floatx16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
            }
            BTL_TL(1);
#pragma unroll 1
            for (int kb = 0; kb < NKA; kb++) {
                const int newer = issued - 1 - kb;                         // stages behind stage kb that may stay in flight (the count is exact: nothing else is issued)
                if (newer >= 8) btl_wait_vmcnt<8 * LA>(); else if (newer == 7) btl_wait_vmcnt<7 * LA>(); else if (newer == 6) btl_wait_vmcnt<6 * LA>();
                else if (newer == 5) btl_wait_vmcnt<5 * LA>(); else if (newer == 4) btl_wait_vmcnt<4 * LA>(); else if (newer == 3) btl_wait_vmcnt<3 * LA>();
                else if (newer == 2) btl_wait_vmcnt<2 * LA>(); else if (newer == 1) btl_wait_vmcnt<LA>(); else btl_wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                             // everybody's share of stage kb; the slot of stage kb-1 is free
                asm volatile("" ::: "memory");
                if (kb >= 1 && issued < NKA) { issue_a(issued); ++issued; }
                const unsigned char* const st = btl_smem + (kb % NSA) * STA;
                half8_t fxv[2], fw[2][NF1];                                // fragments of k step ks + 1 are requested before the MFMAs of k step ks
                fxv[0] = *(const half8_t*)(st + foff + (((0 + hi) ^ fkey) << 4));
#pragma unroll
                for (int a = 0; a < NF1; a++) fw[0][a] = *(const half8_t*)(st + XSL + a * 4096 + woff + (((0 + hi) ^ wkey) << 4));
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    if (ks + 1 < 4) {
                        fxv[(ks + 1) & 1] = *(const half8_t*)(st + foff + (((2 * (ks + 1) + hi) ^ fkey) << 4));
#pragma unroll
                        for (int a = 0; a < NF1; a++) fw[(ks + 1) & 1][a] = *(const half8_t*)(st + XSL + a * 4096 + woff + (((2 * (ks + 1) + hi) ^ wkey) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < NF1; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks & 1][a], fxv[ks & 1], acc[a], 0, 0, 0);
                }
            }
            BTL_TL(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                 // everybody is done reading the ring: its LDS takes h1 (and W2's ring)
            asm volatile("" ::: "memory");
            // phase B's first slabs go out before the epilogue
#pragma unroll
            for (int j = 0; j < NB; j++) dma_w(p.w2, 9 * CM, j * 64, CM, btl_smem + R1 + j * W2SL);
            // epilogue: register 4q + r of acc[a] <-> channel a*32 + 8q + 4hi + r of patch pixel pp -> 8 bytes into slot (ch / 8) ^ swz(pp)
            const int sw = swz(pp);
#pragma unroll
            for (int a = 0; a < NF1; a++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    half4_t o;
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)btl_act(acc[a][4 * qq + r2], p.sl1);
                    const int ch = a * 32 + 8 * qq + 4 * hi;
                    *(half4_t*)(h1 + pp * HS + (((ch >> 3) ^ sw) << 4) + (ch & 7) * 2) = o;
                }
        }
        // =============================== phase B: h2 = act2(W2 * h1 + b2), 3x3 on the patch =======================================
        {
            constexpr int KPT = CM / 16;                                   // k steps per tap
            unsigned char* const ring = btl_smem + R1;
            floatx16 acc[NF1];
#pragma unroll
            for (int a = 0; a < NF1; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
            BTL_TL(3);
            int issued = NB;                                               // slabs requested so far (phase A's tail requested the first NB)
#pragma unroll 1
            for (int j = 0; j < NKB; j++) {
                const int newer = issued - 1 - j;
                if (newer >= 8) btl_wait_vmcnt<8 * LB>(); else if (newer == 7) btl_wait_vmcnt<7 * LB>(); else if (newer == 6) btl_wait_vmcnt<6 * LB>();
                else if (newer == 5) btl_wait_vmcnt<5 * LB>(); else if (newer == 4) btl_wait_vmcnt<4 * LB>(); else if (newer == 3) btl_wait_vmcnt<3 * LB>();
                else if (newer == 2) btl_wait_vmcnt<2 * LB>(); else if (newer == 1) btl_wait_vmcnt<LB>(); else btl_wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (j = 0: my h1 writes)
                __builtin_amdgcn_s_barrier();                             // everybody's share of slab j; the slot of slab j-1 is free
                asm volatile("" ::: "memory");
                if (j >= 1 && issued < NKB) { dma_w(p.w2, 9 * CM, issued * 64, CM, ring + ((j - 1) % NB) * W2SL); ++issued; }
                if (act_q) {
                    const unsigned char* const wb = ring + (j % NB) * W2SL;
                    half8_t fh[2], fw[2][NF1];
                    auto fetch = [&](int ks, int slot) __attribute__((always_inline)) {
                        const int kk = j * 4 + ks, tap = kk / KPT, cg = kk - tap * KPT;        // (wave-uniform)
                        const int px = qpatch + (tap / 3) * PW + (tap % 3);
                        fh[slot] = *(const half8_t*)(h1 + px * HS + (((2 * cg + hi) ^ swz(px)) << 4));
#pragma unroll
                        for (int a = 0; a < NF1; a++) fw[slot][a] = *(const half8_t*)(wb + a * 4096 + woff + (((2 * ks + hi) ^ wkey) << 4));
                    };
                    fetch(0, 0);
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        if (ks + 1 < 4) fetch(ks + 1, (ks + 1) & 1);
#pragma unroll
                        for (int a = 0; a < NF1; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks & 1][a], fh[ks & 1], acc[a], 0, 0, 0);
                    }
                }
            }
            BTL_TL(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                 // the ring is dead: its LDS takes h2
            asm volatile("" ::: "memory");
            if (act_q) {
                const int sw = swz(q);
#pragma unroll
                for (int a = 0; a < NF1; a++)
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const floatx4 bq = *(const floatx4*)(b2l + a * 32 + 8 * qq + 4 * hi);
                        half4_t o;
#pragma unroll
                        for (int r2 = 0; r2 < 4; r2++) o[r2] = (half_t)btl_act(acc[a][4 * qq + r2] + bq[r2], p.sl2);
                        const int ch = a * 32 + 8 * qq + 4 * hi;
                        *(half4_t*)(h2 + q * HS + (((ch >> 3) ^ sw) << 4) + (ch & 7) * 2) = o;
                    }
            }
        }
        // =============================== phase C: y = act3(W3 . h2 + b3 + x) ===============================================================
        {
            const int oy = oy0 + qy, ox = ox0 + qx;
            const bool valid = act_q && q < OPX && oy < p.H && ox < p.W;
            const long opix = ((long)n * p.H + oy) * p.W + ox;
            half8_t rres[2][NCC][2];                                      // residual rows of my pixel: [buffer][cout fragment][j] = channels 32 a + 16 j + 8 hi ..
            auto load_res = [&](int j, half8_t (&d)[NCC][2]) __attribute__((always_inline)) {
#pragma unroll
                for (int a = 0; a < NCC; a++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) d[a][jj] = valid ? *(const half8_t*)(p.x + opix * p.x_ld + (j * NCC + a) * 32 + 16 * jj + 8 * hi) : hz;
            };
            load_res(0, rres[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                 // h2 complete
            asm volatile("" ::: "memory");
            half8_t fh[KS3];
            if (act_q) {
                const int sw = swz(q);
#pragma unroll
                for (int ks = 0; ks < KS3; ks++) fh[ks] = *(const half8_t*)(h2 + q * HS + (((2 * ks + hi) ^ sw) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                 // h2 is in registers: the whole pool takes W3
            asm volatile("" ::: "memory");
            BTL_TL(5);
#pragma unroll
            for (int kb = 0; kb < NK3; kb++) dma_w(p.w3, CM, kb * 64, C, btl_smem + kb * W3SL);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            BTL_TL(6);
#pragma unroll 1
            for (int j = 0; j < NCHC; j += 2) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int jj = j + u;
                    if (jj < NCHC) {
                        if (jj + 1 < NCHC) load_res(jj + 1, rres[u ^ 1]);
                        if (act_q) {
                            floatx16 acc[NCC];                             // the chunk's four MFMA chains interleaved
#pragma unroll
                            for (int a = 0; a < NCC; a++)
#pragma unroll
                                for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
#pragma unroll
                            for (int ks = 0; ks < KS3; ks++) {
                                half8_t fw[NCC];
#pragma unroll
                                for (int a = 0; a < NCC; a++) fw[a] = *(const half8_t*)(btl_smem + (ks >> 2) * W3SL + (jj * NCC + a) * 4096 + woff + (((2 * (ks & 3) + hi) ^ wkey) << 4));
#pragma unroll
                                for (int a = 0; a < NCC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a], fh[ks], acc[a], 0, 0, 0);
                            }
#pragma unroll
                            for (int a = 0; a < NCC; a++) {
                                const int co0 = (jj * NCC + a) * 32;
#pragma unroll
                                for (int j2 = 0; j2 < 2; j2++) {
                                    const floatx4 ba = *(const floatx4*)(b3l + co0 + 8 * (2 * j2) + 4 * hi), bb = *(const floatx4*)(b3l + co0 + 8 * (2 * j2 + 1) + 4 * hi);
                                    float o[8];
#pragma unroll
                                    for (int r2 = 0; r2 < 4; r2++) {
                                        unsigned ua = __builtin_bit_cast(unsigned, acc[a][4 * (2 * j2) + r2] + ba[r2]);
                                        unsigned ub = __builtin_bit_cast(unsigned, acc[a][4 * (2 * j2 + 1) + r2] + bb[r2]);
                                        sm_permlane32_swap(ua, ub);
                                        o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                                    }
                                    half8_t ov;
#pragma unroll
                                    for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)btl_act(o[r2] + (float)rres[u][a][j2][r2], p.sl3);
                                    if (valid) *(half8_t*)(p.y + opix * p.y_ld + co0 + 16 * j2 + 8 * hi) = ov;
                                }
                            }
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                 // W3's LDS is free for the next tile's first stages
            asm volatile("" ::: "memory");
            BTL_TL(7);
        }
    }
#ifdef SM_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 8; i++) g_btl_tl[i] = tl[i];
#endif
}

}  // namespace

extern "C" {

#ifdef SM_TIMELINE
int sm_bottleneck_debug_timeline(unsigned long long* out8) {
    SM_HIP_CHECK(hipDeviceSynchronize());
    SM_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_btl_tl), 8 * sizeof(unsigned long long)));
    return SM_OK;
}
#endif

// 1x1 (C -> Cm, act1) -> 3x3 / 1 / pad 1 (Cm -> Cm, act2) -> 1x1 (Cm -> C) + x, act3 in one launch; C = 4 Cm, Cm = 64 | 128; weights fp16:
// w1 [Cm][C], w2 [Cm][3][3][Cm] (OHWI), w3 [C][Cm]; biases fp32 or NULL; activations of the ReLU family (SM_ACT_NONE | RELU | PRELU with one slope each)
int sm_bottleneck_supported(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm) {
    return N > 0 && H > 0 && W > 0 && (Cm == 64 || Cm == 128) && C == 4 * Cm && (long)N * H * W < (1L << 31) / 1024;
}
int sm_bottleneck_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, const void* x, int32_t x_ld, const void* w1, const float* b1, int32_t act1, float slope1,
                      const void* w2, const float* b2, int32_t act2, float slope2, const void* w3, const float* b3, int32_t act3, float slope3, void* y, int32_t y_ld, sm_stream_t stream) {
    SM_REQUIRE(sm_bottleneck_supported(N, H, W, C, Cm) && x && w1 && w2 && w3 && y, "sm_bottleneck_fwd: unsupported geometry (C = 4 Cm, Cm = 64 | 128)");
    SM_REQUIRE(act1 <= SM_ACT_PRELU && act2 <= SM_ACT_PRELU && act3 <= SM_ACT_PRELU, "sm_bottleneck_fwd: ReLU-family activations only");
    BtlParams p{};
    p.x = (const half_t*)x; p.w1 = (const half_t*)w1; p.w2 = (const half_t*)w2; p.w3 = (const half_t*)w3; p.b1 = b1; p.b2 = b2; p.b3 = b3; p.y = (half_t*)y;
    p.N = N; p.H = H; p.W = W; p.x_ld = x_ld ? x_ld : C; p.y_ld = y_ld ? y_ld : C;
    SM_REQUIRE((p.x_ld & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w1) | ((uintptr_t)w2) | ((uintptr_t)w3)) & 15) == 0,
               "sm_bottleneck_fwd: operands must be 16-byte aligned, pixel strides % 8 == 0");
    p.sl1 = act1 == SM_ACT_NONE ? 1.f : act1 == SM_ACT_RELU ? 0.f : slope1;
    p.sl2 = act2 == SM_ACT_NONE ? 1.f : act2 == SM_ACT_RELU ? 0.f : slope2;
    p.sl3 = act3 == SM_ACT_NONE ? 1.f : act3 == SM_ACT_RELU ? 0.f : slope3;
    p.tiles_x = sm_ceil_div(W, 14); p.tiles_y = sm_ceil_div(H, 14);
    const long nt = (long)N * p.tiles_x * p.tiles_y;
    p.ntiles = (int)nt;
    btl_fastdiv_make((unsigned)p.tiles_x, &p.tx_mul, &p.tx_sh); btl_fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh);
    const int grid = (int)std::min<long>(nt, SM_NUM_CU);
    const size_t lds = (size_t)156 * 1024 + (size_t)6 * Cm * 4;      // the pool + the three bias vectors
    if (Cm == 128) hipLaunchKernelGGL(btl_kernel<128>, dim3(grid), dim3(512), lds, sm_s(stream), p);
    else hipLaunchKernelGGL(btl_kernel<64>, dim3(grid), dim3(512), lds, sm_s(stream), p);
    SM_LAUNCH_CHECK(); return SM_OK;
}

}  // extern "C"
