// micro-benchmark: do ds_read_b128 and v_mfma_f32_32x32x16_f16 overlap?  One block per CU, NW waves, every wave runs `iters` steps of
// R conflict-free ds_read_b128 (asm, counted lgkmcnt, PF steps of prefetch, data consumed by the MFMAs when DEP) + M MFMAs.
//   hipcc --offload-arch=gfx950 -O3 ldsmfma.hip -o ldsmfma && ./ldsmfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>()); static_for<I + 1, N>(f); } }

// R reads + M mfma per step; PF prefetch depth; DEP: MFMA operands come from the reads
template <int NW, int R, int M, int PF, int DEP>
__global__ __launch_bounds__(NW * 64) void k(float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 2; i += NW * 64) ((_Float16*)smem)[i] = (_Float16)(((i * 2654435761u) >> 20 & 255) * (1.f / 256.f) - 0.5f);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + (wave & 3) * 8192 + (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    floatx16 acc[4];
    for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    constexpr int RING = PF + 1, RR = R > 0 ? R : 1;
    half8_t f[RING][RR];
    for (int i = 0; i < RING; i++) for (int j = 0; j < RR; j++) f[i][j] = half8_t{1, 2, 3, 4, 5, 6, 7, 8};
    long long t0 = clock64();
    for (int it = 0; it < iters; it += RING) {
        static_for<0, RING>([&](auto sc) {
            constexpr int S = decltype(sc)::value, SLOT = (S + PF) % RING;
            if constexpr (R > 0) {
#pragma unroll
                for (int j = 0; j < R; j++) { half8_t& d = f[SLOT][j]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(base), "n"(j * 4096 + S * 1024)); }
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PF * R > 15 ? 15 : PF * R) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < M; m++) {
                const half8_t a = DEP ? f[S % RING][m % RR] : half8_t{1, 2, 3, 4, 5, 6, 7, 8}, b = DEP ? f[S % RING][(m + 1) % RR] : half8_t{8, 7, 6, 5, 4, 3, 2, 1};
                acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 4], 0, 0, 0);
            }
            if constexpr (!DEP && R > 0) { for (int j = 0; j < R; j++) asm volatile("" ::"v"(f[S % RING][j])); }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = clock64();
    float s = 0; for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * NW * 64 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int NW, int R, int M, int PF, int DEP> void run(const char* name) {
    float* out; long long* clk; CK(hipMalloc(&out, 256 * NW * 64 * 4)); CK(hipMalloc(&clk, 8));
    const int iters = 6000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<NW, R, M, PF, DEP>), dim3(256), dim3(NW * 64), 65536, 0, out, iters, clk);
    CK(hipEventRecord(a)); hipLaunchKernelGGL((k<NW, R, M, PF, DEP>), dim3(256), dim3(NW * 64), 65536, 0, out, iters, clk); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b)); long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double steps = iters;
    printf("%-44s NW=%d R=%d M=%d PF=%d dep=%d : %7.1f clk/step (s_memtime)  %6.1f us  -> %5.0f TFLOP/s, LDS %5.0f B/clk/CU\n", name, NW, R, M, PF, DEP, c / steps, ms * 1e3,
           256.0 * NW * M * 32768.0 * steps / (ms * 1e-3) / 1e12, NW * R * 1024.0 * steps / (double)c);
    CK(hipFree(out)); CK(hipFree(clk));
}
int main() {
    run<4, 0, 2, 2, 0>("MFMA only, 1 wave/SIMD");
    run<8, 0, 2, 2, 0>("MFMA only, 2 waves/SIMD");
    run<4, 3, 0, 4, 0>("reads only (3/step), 1 wave/SIMD");
    run<8, 3, 0, 4, 0>("reads only (3/step), 2 waves/SIMD");
    run<4, 3, 2, 4, 0>("3 reads + 2 MFMA independent, 1 w/SIMD");
    run<4, 3, 2, 4, 1>("3 reads + 2 MFMA dependent, 1 w/SIMD");
    run<8, 3, 2, 4, 1>("3 reads + 2 MFMA dependent, 2 w/SIMD");
    run<4, 2, 2, 4, 1>("2 reads + 2 MFMA dependent, 1 w/SIMD");
    run<8, 2, 2, 4, 1>("2 reads + 2 MFMA dependent, 2 w/SIMD");
    run<4, 1, 2, 4, 1>("1 read + 2 MFMA dependent, 1 w/SIMD");
    run<4, 4, 4, 3, 1>("4 reads + 4 MFMA dependent, 1 w/SIMD");
    run<8, 4, 4, 3, 1>("4 reads + 4 MFMA dependent, 2 w/SIMD");
    run<4, 2, 4, 4, 1>("2 reads + 4 MFMA dependent, 1 w/SIMD");
    run<8, 2, 4, 4, 1>("2 reads + 4 MFMA dependent, 2 w/SIMD");
    return 0;
}
