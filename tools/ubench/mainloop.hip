// micro-benchmark: ceiling of the LDS -> MFMA part of the implicit-GEMM main loop (no global traffic at all).
// A [BM][64] and B [BN][64] fp16 tiles live in LDS (2 stages, XOR-swizzled 16-B chunks like conv_igemm64.hpp); every
// block runs nk K-tiles: fragment reads (ds_read_b128) + v_mfma_f32_32x32x16_f16, one barrier per tile (VAR-dependent).
//   hipcc --offload-arch=gfx950 -O3 mainloop.hip -o mainloop && ./mainloop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// VAR 0: product structure (per k-step: read fragments, MFMAs; __syncthreads per tile)
// VAR 1: VAR 0 without the barrier
// VAR 2: all fragments of the tile read up front, then all MFMAs; barrier per tile
// VAR 3: k-step software pipeline (fragments of step s+1 requested before the MFMAs of step s), barrier per tile
// VAR 4: VAR 3 without barrier
// VAR 5: MFMA only (fragments read once)
#define SM_GLDS16(gptr, ldsptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DMA variants: the block streams (BM+BN) x 128 B per K tile from an L2-resident buffer (rows of 128 B at stride RS bytes,
// like the im2col gather) into an NST-deep LDS ring while it computes; one s_barrier per tile, counted vmcnt.
template <int BM, int BN, int WGM, int WGN, int NST, int MFMA, int ILV = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void kd(float* out, const char* __restrict__ src, unsigned src_mask, int RS, int nk) {
    constexpr int NT = WGM * WGN * 64, TM = BM / (WGM * 32), TN = BN / (WGN * 32), STAGE_BYTES = (BM + BN) * 128, LOADS = (BM + BN) * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WGN, wn = wave % WGN;
    for (int i = tid; i < NST * STAGE_BYTES / 2; i += NT) ((_Float16*)smem_raw)[i] = (_Float16)(((i * 2654435761u) >> 20 & 255) * (1.f / 256.f) - 0.5f);
    __syncthreads();
    floatx16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    unsigned voff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) voff[ks] = (lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    // per-thread source row offsets (row = piece*NT/8 + tid/8, 16-B chunk tid&7)
    unsigned roff[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; i++) roff[i] = (unsigned)(blockIdx.x * 977 + i * (NT / 8) + (tid >> 3)) * (unsigned)RS + (tid & 7) * 16;
    unsigned koff = 0;
    auto issue = [&](int slot) {
        unsigned char* sa = smem_raw + slot * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < LOADS; i++) SM_GLDS16(src + ((roff[i] + koff) & src_mask), sa + i * NT * 16);
        koff += 128;
    };
    auto piece = [&](int slot, int i) {
        unsigned char* sa = smem_raw + slot * STAGE_BYTES + wave * 1024;
        SM_GLDS16(src + ((roff[i] + koff) & src_mask), sa + i * NT * 16);
    };
    // ILV: the LOADS DMA instructions of the next tile are spread between the MFMAs (one every MPL MFMAs)
    auto compute_ilv = [&](int slot, int islot, bool more) {
        const unsigned sA = slot * STAGE_BYTES + wm * (TM * 4096), sB = slot * STAGE_BYTES + BM * 128 + wn * (TN * 4096);
        constexpr int NM = 4 * TM * TN, MPL = NM / LOADS > 0 ? NM / LOADS : 1;
        half8_t fx[TM], fw[TN];
        int cnt = 0, pi = 0;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int b = 0; b < TM; b++) fx[b] = *(const half8_t*)(smem_raw + sA + voff[ks] + b * 4096);
#pragma unroll
            for (int a = 0; a < TN; a++) fw[a] = *(const half8_t*)(smem_raw + sB + voff[ks] + a * 4096);
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a], fx[b], acc[a][b], 0, 0, 0);
                    if (++cnt % MPL == 0 && pi < LOADS) { if (more) piece(islot, pi); pi++; }
                }
        }
#pragma unroll
        for (; pi < LOADS; pi++) if (more) piece(islot, pi);
        if (more) koff += 128;
    };
    auto compute = [&](int slot) {
        const unsigned sA = slot * STAGE_BYTES + wm * (TM * 4096), sB = slot * STAGE_BYTES + BM * 128 + wn * (TN * 4096);
        half8_t fx[TM], fw[TN];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int b = 0; b < TM; b++) fx[b] = *(const half8_t*)(smem_raw + sA + voff[ks] + b * 4096);
#pragma unroll
            for (int a = 0; a < TN; a++) fw[a] = *(const half8_t*)(smem_raw + sB + voff[ks] + a * 4096);
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    if (MFMA) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a], fx[b], acc[a][b], 0, 0, 0);
                    else asm volatile("" ::"v"(fw[a]), "v"(fx[b]));
                }
        }
    };
    int issued = 0;
#pragma unroll
    for (int s2 = 0; s2 < NST - 1; s2++) { issue(s2); ++issued; }
    int slot = 0, islot = NST - 1;
    for (int i = 0; i < nk; i++) {
        const int newer = issued - i - 1;
        if (NST >= 4 && newer >= 3) wait_vmcnt<(NST >= 4 ? 3 : 0) * LOADS>();
        else if (NST >= 3 && newer == 2) wait_vmcnt<(NST >= 3 ? 2 : 0) * LOADS>();
        else if (NST >= 2 && newer == 1) wait_vmcnt<(NST >= 2 ? 1 : 0) * LOADS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (ILV) { const bool more = issued < nk; compute_ilv(slot, islot, more); if (more) { ++issued; if (++islot == NST) islot = 0; } }
        else { if (issued < nk) { issue(islot); ++issued; if (++islot == NST) islot = 0; } compute(slot); }
        if (++slot == NST) slot = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) s += acc[a][b][r];
    out[(size_t)blockIdx.x * NT + tid] = s;
}

template <int BM, int BN, int WGM, int WGN, int NST, int MFMA, int ILV = 0> void rund(float* out, const char* src, size_t src_bytes, int RS, int bpc, int active_cus = 256) {
    constexpr int NT = WGM * WGN * 64;
    const int grid = active_cus * bpc, nk = 512;
    const size_t lds = (size_t)NST * (BM + BN) * 128;
    auto fn = kd<BM, BN, WGM, WGN, NST, MFMA, ILV>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), lds, 0, out, src, (unsigned)(src_bytes - 1), RS, 16);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), lds, 0, out, src, (unsigned)(src_bytes - 1), RS, nk);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double fl = 2.0 * BM * BN * 64 * (double)nk * grid, by = (double)(BM + BN) * 128 * nk * grid;
    printf("DMA ilv %d %3dx%-3d waves %dx%d stages %d mfma %d blocks/CU %d src %5.1f MB RS %5d: %7.1f TFLOP/s-equiv  %6.2f TB/s L2->LDS  (%5.0f cycles/tile @2.1GHz)\n", ILV, BM, BN, WGM, WGN, NST, MFMA, bpc,
           src_bytes / 1e6, RS, fl / ms / 1e9, by / ms / 1e9, ms * 1e-3 / nk / ((grid + active_cus * bpc - 1) / (active_cus * bpc)) * 2.1e9);
}

template <int BM, int BN, int WGM, int WGN, int VAR, int BK = 64>
__global__ __launch_bounds__(WGM * WGN * 64) void k(float* out, int nk) {
    constexpr int NT = WGM * WGN * 64, TM = BM / (WGM * 32), TN = BN / (WGN * 32), STAGE = (BM + BN) * BK, KS = BK / 16, CH = BK / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* smem = (half_t*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
    for (int i = tid; i < 2 * STAGE; i += NT) smem[i] = (half_t)(((i * 2654435761u) >> 20 & 255) * (1.f / 256.f) - 0.5f);
    __syncthreads();
    floatx16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    auto fragA = [&](const half_t* sa, int ks, int b) {
        const int c = ks * 2 + (lane >> 5), row = (wm * TM + b) * 32 + (lane & 31);
        return *(const half8_t*)(sa + row * BK + ((c ^ ((row >> 1) & (CH - 1))) << 3));
    };
    auto fragB = [&](const half_t* sb, int ks, int a) {
        const int c = ks * 2 + (lane >> 5), row = (wn * TN + a) * 32 + (lane & 31);
        return *(const half8_t*)(sb + row * BK + ((c ^ ((row >> 1) & (CH - 1))) << 3));
    };
    half8_t fx[2][TM], fw[2][TN];
    if (VAR == 5) {
#pragma unroll
        for (int b = 0; b < TM; b++) fx[0][b] = fragA(smem, 0, b);
#pragma unroll
        for (int a = 0; a < TN; a++) fw[0][a] = fragB(smem + BM * BK, 0, a);
    }
    for (int i = 0; i < nk; i++) {
        const half_t* sa = smem + (i & 1) * STAGE; const half_t* sb = sa + BM * BK;
        if (VAR == 0 || VAR == 1) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                for (int b = 0; b < TM; b++) fx[0][b] = fragA(sa, ks, b);
#pragma unroll
                for (int a = 0; a < TN; a++) fw[0][a] = fragB(sb, ks, a);
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[0][a], fx[0][b], acc[a][b], 0, 0, 0);
            }
            if (VAR == 0) __syncthreads();
        } else if (VAR == 2) {
            half8_t ax[KS][TM], aw[KS][TN];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                for (int b = 0; b < TM; b++) ax[ks][b] = fragA(sa, ks, b);
#pragma unroll
                for (int a = 0; a < TN; a++) aw[ks][a] = fragB(sb, ks, a);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[ks][a], ax[ks][b], acc[a][b], 0, 0, 0);
            __syncthreads();
        } else if (VAR == 3 || VAR == 4) {
#pragma unroll
            for (int b = 0; b < TM; b++) fx[0][b] = fragA(sa, 0, b);
#pragma unroll
            for (int a = 0; a < TN; a++) fw[0][a] = fragB(sb, 0, a);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int b = 0; b < TM; b++) fx[(ks + 1) & 1][b] = fragA(sa, ks + 1, b);
#pragma unroll
                    for (int a = 0; a < TN; a++) fw[(ks + 1) & 1][a] = fragB(sb, ks + 1, a);
                }
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks & 1][a], fx[ks & 1][b], acc[a][b], 0, 0, 0);
            }
            if (VAR == 3) __syncthreads();
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[0][a], fx[0][b], acc[a][b], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) s += acc[a][b][r];
    out[(size_t)blockIdx.x * NT + tid] = s;
}

template <int BM, int BN, int WGM, int WGN, int VAR, int BK = 64> void run(float* out, int bpc) {
    constexpr int NT = WGM * WGN * 64;
    const int grid = 256 * bpc, nk = 512 * 64 / BK;
    const size_t lds = 2 * (BM + BN) * BK * 2;
    auto fn = k<BM, BN, WGM, WGN, VAR, BK>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), lds, 0, out, 16);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), lds, 0, out, nk);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double fl = 2.0 * BM * BN * BK * (double)nk * grid;
    printf("%3dx%-3d BK=%3d waves %dx%d var %d blocks/CU %d lds %3zu KB: %7.1f TFLOP/s (%4.1f%% of 2.5 PF)\n", BM, BN, BK, WGM, WGN, VAR, bpc, lds >> 10, fl / ms / 1e9,
           fl / ms / 1e9 / 25.0);
}

int main(int argc, char** argv) {
    float* out; CK(hipMalloc(&out, 256 * 4 * 512 * 4));
    if (argc > 1) {   // DMA study
        char* src; const size_t SB = 64u << 20; CK(hipMalloc(&src, SB)); CK(hipMemset(src, 0x3c, SB));
        const size_t sb = (size_t)1 << 20; const int RS = 512;
        rund<128, 128, 2, 2, 2, 1, 0>(out, src, sb, RS, 1); rund<128, 128, 2, 2, 2, 1, 1>(out, src, sb, RS, 1);
        rund<128, 128, 2, 2, 3, 1, 0>(out, src, sb, RS, 1); rund<128, 128, 2, 2, 3, 1, 1>(out, src, sb, RS, 1);
        rund<128, 128, 2, 2, 4, 1, 0>(out, src, sb, RS, 1); rund<128, 128, 2, 2, 4, 1, 1>(out, src, sb, RS, 1);
        rund<128, 128, 2, 4, 2, 1, 0>(out, src, sb, RS, 1); rund<128, 128, 2, 4, 2, 1, 1>(out, src, sb, RS, 1);
        rund<128, 128, 2, 4, 3, 1, 0>(out, src, sb, RS, 1); rund<128, 128, 2, 4, 3, 1, 1>(out, src, sb, RS, 1);
        rund<128, 128, 2, 4, 4, 1, 0>(out, src, sb, RS, 1); rund<128, 128, 2, 4, 4, 1, 1>(out, src, sb, RS, 1);
        rund<128, 128, 2, 2, 2, 1, 0>(out, src, sb, RS, 2); rund<128, 128, 2, 2, 2, 1, 1>(out, src, sb, RS, 2);
        rund<256, 128, 4, 2, 2, 1, 0>(out, src, sb, RS, 1); rund<256, 128, 4, 2, 2, 1, 1>(out, src, sb, RS, 1);
        rund<256, 128, 4, 2, 3, 1, 0>(out, src, sb, RS, 1); rund<256, 128, 4, 2, 3, 1, 1>(out, src, sb, RS, 1);
        rund<128, 64, 2, 2, 2, 1, 0>(out, src, sb, RS, 2); rund<128, 64, 2, 2, 2, 1, 1>(out, src, sb, RS, 2);
        rund<128, 64, 2, 2, 4, 1, 0>(out, src, sb, RS, 2); rund<128, 64, 2, 2, 4, 1, 1>(out, src, sb, RS, 2);
        rund<64, 64, 2, 2, 2, 1, 0>(out, src, sb, RS, 4); rund<64, 64, 2, 2, 2, 1, 1>(out, src, sb, RS, 4);
        rund<64, 64, 2, 2, 4, 1, 0>(out, src, sb, RS, 4); rund<64, 64, 2, 2, 4, 1, 1>(out, src, sb, RS, 4);
        return 0;
    }
#define ALLV(BM, BN, WGM, WGN, bpc) run<BM, BN, WGM, WGN, 0>(out, bpc); run<BM, BN, WGM, WGN, 1>(out, bpc); run<BM, BN, WGM, WGN, 2>(out, bpc); \
    run<BM, BN, WGM, WGN, 3>(out, bpc); run<BM, BN, WGM, WGN, 4>(out, bpc); run<BM, BN, WGM, WGN, 5>(out, bpc);
    ALLV(128, 128, 2, 2, 1) ALLV(128, 128, 2, 2, 2) ALLV(128, 128, 2, 4, 1) ALLV(128, 128, 2, 4, 2)
    ALLV(256, 128, 4, 2, 1) ALLV(256, 128, 2, 2, 1) ALLV(128, 64, 2, 2, 2) ALLV(64, 64, 2, 2, 2) ALLV(64, 64, 2, 2, 4)
    run<128, 128, 2, 2, 0, 128>(out, 1); run<128, 128, 2, 2, 3, 128>(out, 1); run<256, 128, 2, 2, 3, 128>(out, 1);
    return 0;
}
