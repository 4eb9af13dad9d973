// conv_igemm64.hpp — the fast path of the implicit-GEMM convolution: Cin % 64 == 0 (every ResNet-50 layer but the
// stem), BK = 64, operands streamed HBM/L2 -> LDS by the LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane) with no VGPR
// round trip, epilogue transposed through LDS so that every global access is a full 16-byte lane / 128-byte line.
// Included by conv_igemm.hip (same ConvParams, same launcher).
//
//  * K tiling: with Cin % 64 == 0 a 64-wide K tile lies inside ONE (kh,kw) tap, so the tap and its channel base are
//    wave-uniform per tile; a row of the activation tile is 128 contiguous bytes of one input pixel (or zeros).
//  * LDS-DMA writes lane-linear (wave base + lane*16 B), so the bank swizzle is applied on the SOURCE side: the lane
//    that fills 16-byte slot c' of row r fetches chunk c = c' ^ ((r >> 1) & 7); readers apply the same XOR
//    (cdna_hip_programming.md §5.4 rule 21).  With 128-byte rows this makes every 16-lane service group of
//    ds_read_b128 hit 16 distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md §LDS).
//  * Zero padding / ragged tiles: out-of-image taps, rows >= M and couts >= Cout fetch from a zero page.
//  * Pipeline: NSTAGE-deep LDS ring.  NSTAGE == 2: tile t+1 in flight while tile t feeds the MFMAs, __syncthreads()
//    carries the vmcnt(0) that publishes the DMA'd tile.  NSTAGE >= 3: the layers here are short, latency-bound GEMMs
//    (M = 3k..200k pixels, 1..72 K tiles per block), so NSTAGE-1 tiles are kept in flight across raw s_barriers with a
//    COUNTED s_waitcnt vmcnt(N) (cdna_hip_programming.md §5 "Pipelining across barriers", T3+T4): the only wait is for
//    the oldest tile.
//  * Epilogue: per wave-row pass, accumulators -> LDS as fp32 [32*TM rows][BN+4] (conflict-free ds_write_b128),
//    then 256 threads each take 8 consecutive couts of one pixel: + bias + residual (16-byte coalesced load),
//    activation, ONE rounding to fp16, 16-byte coalesced store.
#pragma once
#include <type_traits>

__device__ __attribute__((aligned(64))) unsigned int g_zero_page[64];   // zero-initialised at module load

#define SM_GLDS16(gptr, ldsptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void sm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL: ablation switch for tools/conv_bench.py (0 = product; 1 = no MFMA; 2 = no global->LDS traffic after the prologue;
// 3 = no LDS fragment reads)
template <int BM, int BN, int WGM, int WGN, int NSTAGE = 2, int ABL = 0, int ILV = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_igemm64_kernel(ConvParams p) {
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves");
    constexpr int NT = WGM * WGN * 64;                          // threads per block
    static_assert(NSTAGE >= 1 && NSTAGE <= 4, "1..4 stages");
    constexpr int BK64 = 64;
    constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
    constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;      // 16-byte pieces per thread per tile
    constexpr int RPI = NT / 8;                                 // tile rows covered by one DMA instruction of the whole block
    static_assert(A_IT >= 1 && B_IT >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave layout");
    constexpr int LOADS = A_IT + B_IT;                          // LDS-DMA instructions per wave per tile
    constexpr int STAGE = (BM + BN) * BK64;                     // halfs per stage
    constexpr int EP_ROWS = 32 * TM, EP_LD = BN + 4;            // epilogue staging (floats)
    constexpr int LDS_BYTES = (NSTAGE * STAGE * 2 > EP_ROWS * EP_LD * 4) ? NSTAGE * STAGE * 2 : EP_ROWS * EP_LD * 4;
    static_assert(NSTAGE < 3 || (NSTAGE - 2) * LOADS <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[LDS_BYTES];
    half_t* smem = (half_t*)smem_raw;

    int mtile, ntile, ksplit;
    if (!map_block(p, mtile, ntile, ksplit)) return;
    const int ktiles = p.K / BK64;
    const int kt_begin = ksplit * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split; if (kt_end > ktiles) kt_end = ktiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int csrc = (tid & 7) ^ ((tid >> 4) & 7);             // source chunk for this thread's LDS slot (rows j*32 + tid/8)
    const half_t* zero = (const half_t*)g_zero_page;

    // ---- per-thread row state -----------------------------------------------------------------------
    const half_t* arow[A_IT]; int aiy[A_IT], aix[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; j++) {
        const int m = mtile * BM + j * RPI + (tid >> 3);
        if (m < p.M) {
            int n = m / (p.Ho * p.Wo), r = m - n * (p.Ho * p.Wo), oy = r / p.Wo, ox = r - oy * p.Wo;
            aiy[j] = oy * p.sH - p.pt; aix[j] = ox * p.sW - p.pl;
            arow[j] = p.x + (((long)n * p.H + aiy[j]) * p.W + aix[j]) * p.x_ld + csrc * 8;
        } else { aiy[j] = -(1 << 28); aix[j] = -(1 << 28); arow[j] = zero; }     // never valid
    }
    const half_t* brow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
        const int n = ntile * BN + j * RPI + (tid >> 3);
        brow[j] = n < p.Cout ? p.w + (long)n * p.K + csrc * 8 : nullptr;
    }
    // Residual operand (fused MPSNNAdditionNode): its loads do not depend on the GEMM, so they are put in flight FIRST
    // (16-byte coalesced, same (row, chunk) ownership as the epilogue stores) and consumed ~a whole K loop later.
    constexpr int CHUNKS = BN / 8, ITEMS = EP_ROWS * CHUNKS, IPT = (ITEMS + NT - 1) / NT;
    const bool vec_ok = ((p.y_ld & 7) == 0) && ((((uintptr_t)p.y) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) && ((p.Cout & 7) == 0) &&
                        (p.res == nullptr || ((p.res_ld & 7) == 0 && (((uintptr_t)p.res) & 15) == 0));
    half8_t rres[WGM][IPT];
    if (p.res && vec_ok && p.splitk == 1) {
#pragma unroll
        for (int pw = 0; pw < WGM; pw++)
#pragma unroll
            for (int k2 = 0; k2 < IPT; k2++) {
                const int it = tid + k2 * NT, row = it / CHUNKS, ch = it - row * CHUNKS;
                const int m = mtile * BM + pw * EP_ROWS + row, co = ntile * BN + ch * 8;
                half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (it < ITEMS && m < p.M && co < p.Cout) v = *(const half8_t*)(p.res + (long)m * p.res_ld + co);
                rres[pw][k2] = v;
            }
    }
    const int cpt = p.Cin / BK64;                               // K tiles per tap
    // K rotation: the blocks that run concurrently on one XCD share weight tiles.  If they all walk K in the same order,
    // every weight tile is a first touch (HBM latency) for all of them at once, once per K tile; starting each block at a
    // different K tile makes a tile a miss for ONE block and an L2 hit for the others.
    const int nk = kt_end - kt_begin;
    const int rot = p.krot ? (mtile * 5 + ntile * 3) % nk : 0;
    auto kt_of = [&](int i) { int r = i + rot; if (r >= nk) r -= nk; return kt_begin + r; };   // i-th tile this block processes

    // LDS-DMA of one K tile = LOADS instructions per wave.  ILV = 1 spreads them behind the MFMAs of each k-step of the
    // tile being computed; measured on MI355X this is SLOWER than issuing the whole burst first (14x14 3x3 256->256:
    // 35.3 vs 28.8 us; the tile then lands later and the barrier wait grows), so ILV = 0 is the default.
    struct TileSrc { half_t* sa; half_t* sb; int dy, dx; long toff; long koff; };
    auto tile_src = [&](int buf, int kt) {
        TileSrc t;
        t.sa = smem + buf * STAGE + (wave * 64) * 8;            // wave-uniform LDS base; lane*16 B added by the DMA
        t.sb = smem + buf * STAGE + BM * BK64 + (wave * 64) * 8;
        const int tap = kt / cpt, cblk = kt - tap * cpt, kh = tap / p.kW, kw = tap - kh * p.kW;     // wave-uniform (SALU)
        t.dy = kh * p.dH; t.dx = kw * p.dW;
        t.toff = ((long)t.dy * p.W + t.dx) * p.x_ld + cblk * BK64;
        t.koff = (long)kt * BK64;
        return t;
    };
    auto piece = [&](const TileSrc& t, int i) {                 // i-th DMA instruction of the tile, 0 <= i < LOADS
        if (i < A_IT) {
            const bool ok = (unsigned)(aiy[i] + t.dy) < (unsigned)p.H && (unsigned)(aix[i] + t.dx) < (unsigned)p.W;
            const half_t* src = ok ? arow[i] + t.toff : zero;
            SM_GLDS16(src, t.sa + i * NT * 8);
        } else {
            const int j = i - A_IT;
            const half_t* src = brow[j] ? brow[j] + t.koff : zero;
            SM_GLDS16(src, t.sb + j * NT * 8);
        }
    };
    auto issue = [&](int buf, int kt) {
        const TileSrc t = tile_src(buf, kt);
#pragma unroll
        for (int i = 0; i < LOADS; i++) piece(t, i);
    };

    floatx16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // MFMAs of the tile in `buf`; when `nxt >= 0` the DMA of tile `kt_next` into slot `nxt` is interleaved
    auto compute = [&](int buf, bool first, int nxt, int kt_next) {
        const half_t* sa = smem + buf * STAGE; const half_t* sb = sa + BM * BK64;
        TileSrc t{};
        const bool dma = nxt >= 0 && ABL != 2;
        if (dma) { t = tile_src(nxt, kt_next); if (!ILV) {
#pragma unroll
            for (int i = 0; i < LOADS; i++) piece(t, i); } }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int c = ks * 2 + (lane >> 5);
            half8_t fx[TM], fw[TN];
#pragma unroll
            for (int b = 0; b < TM; b++) { const int row = (wm * TM + b) * 32 + (lane & 31); if (ABL != 3 || first) fx[b] = *(const half8_t*)(sa + row * BK64 + ((c ^ ((row >> 1) & 7)) << 3)); }
#pragma unroll
            for (int a = 0; a < TN; a++) { const int row = (wn * TN + a) * 32 + (lane & 31); if (ABL != 3 || first) fw[a] = *(const half8_t*)(sb + row * BK64 + ((c ^ ((row >> 1) & 7)) << 3)); }
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    if (ABL != 1) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a], fx[b], acc[a][b], 0, 0, 0);
                    else { asm volatile("" ::"v"(fw[a]), "v"(fx[b])); }
                }
            if (ILV && dma) {
#pragma unroll
                for (int i = (ks * LOADS) / 4; i < ((ks + 1) * LOADS) / 4; i++) piece(t, i);
            }
        }
    };

    if (nk <= NSTAGE) {
        // short K (the 1x1 convs with Cin <= 64*NSTAGE are HBM-streaming layers): everything this block will ever read
        // is put in flight at once — no serialised round trips, maximum bytes in flight per CU — then one barrier.
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) if (i < nk) issue(i, kt_of(i));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) if (i < nk) compute(i, i == 0, -1, 0);
        __syncthreads();
    } else if constexpr (NSTAGE == 1) {
        return;                                                   // the launcher never selects a 1-stage config for K > 64
    } else if constexpr (NSTAGE == 2) {
        issue(0, kt_of(0));
        __syncthreads();
        int cur = 0;
        for (int i = 0; i < nk; i++) {
            compute(cur, i == 0, i + 1 < nk ? (cur ^ 1) : -1, kt_of(i + 1 < nk ? i + 1 : i));
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // tile t lives in ring slot (t - kt_begin) % NSTAGE; NSTAGE-1 tiles in flight across raw barriers, counted vmcnt
        int issued = 0;
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; s++) if (issued < nk) { issue(s, kt_of(issued)); ++issued; }
        int slot = 0, islot = NSTAGE - 1;
        for (int i = 0; i < nk; i++) {
            const int newer = issued - i - 1;                   // tiles issued after tile i, allowed to stay in flight
            if (newer >= 2) sm_wait_vmcnt<(NSTAGE >= 4 ? 2 : 0) * LOADS>();
            else if (newer == 1) sm_wait_vmcnt<(NSTAGE >= 3 ? 1 : 0) * LOADS>();
            else sm_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                       // everybody's share of tile kt landed; slot of tile kt-1 is free
            __builtin_amdgcn_sched_barrier(0);
            const bool more = issued < nk;
            compute(slot, i == 0, more ? islot : -1, kt_of(more ? issued : i));
            if (more) ++issued;
            if (++islot == NSTAGE) islot = 0;
            if (++slot == NSTAGE) slot = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's fragment reads are done before it passes the next barrier
        }
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: LDS transpose, coalesced 16-byte global accesses ---------------------------------------------
    float* stg = (float*)smem_raw;
#pragma unroll
    for (int pw = 0; pw < WGM; pw++) {
        if (wm == pw) {
#pragma unroll
            for (int b = 0; b < TM; b++)
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int row = b * 32 + (lane & 31), col = (wn * TN + a) * 32 + 8 * q + 4 * (lane >> 5);
                        *(floatx4*)(stg + row * EP_LD + col) = floatx4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                    }
        }
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < IPT; k2++) {
            const int it = tid + k2 * NT;
            if (it >= ITEMS) continue;
            const int row = it / CHUNKS, ch = it - row * CHUNKS;
            const int m = mtile * BM + pw * EP_ROWS + row, co = ntile * BN + ch * 8;
            if (m >= p.M || co >= p.Cout) continue;
            floatx4 v0 = *(const floatx4*)(stg + row * EP_LD + ch * 8), v1 = *(const floatx4*)(stg + row * EP_LD + ch * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.splitk > 1) {
                float* dst = p.ws + ((long)ksplit * p.M + m) * p.Cout + co;
                if ((p.Cout & 3) == 0 && co + 7 < p.Cout) { *(floatx4*)dst = v0; *(floatx4*)(dst + 4) = v1; }
                else for (int j = 0; j < 8 && co + j < p.Cout; j++) dst[j] = v[j];
                continue;
            }
            if (vec_ok) {
                if (p.bias) { floatx4 b0 = *(const floatx4*)(p.bias + co), b1 = *(const floatx4*)(p.bias + co + 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) { v[j] += b0[j]; v[4 + j] += b1[j]; } }
                if (p.res) { const half8_t rr = rres[pw][k2];     // prefetched before the main loop
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] += (float)rr[j]; }
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = (half_t)sm_act(v[j], p.act, p.slope);
                *(half8_t*)(p.y + (long)m * p.y_ld + co) = o;
            } else {
                for (int j = 0; j < 8 && co + j < p.Cout; j++) {
                    float t = v[j];
                    if (p.bias) t += p.bias[co + j];
                    if (p.res) t += (float)p.res[(long)m * p.res_ld + co + j];
                    p.y[(long)m * p.y_ld + co + j] = (half_t)sm_act(t, p.act, p.slope);
                }
            }
        }
        __syncthreads();
    }
}
