// micro-benchmark: L2-resident LDS-DMA bandwidth when the 128-byte lines of a tile are `stride` bytes apart
// (the A-operand gather of a 1x1 conv with Cin = stride/2), all blocks walking K tiles in lockstep vs staggered.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, int stride, int rows_total, int ktiles, int iters, int stagger, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char smem[4][8192];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = (blockIdx.x * 128) % rows_total;          // this block's 128 rows (pixels)
    for (int it = 0; it < iters; it++) {
        int kt = (it + (stagger ? blockIdx.x : 0)) % ktiles;  // K tile = 128-byte column block
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int row = row0 + j * 32 + (tid >> 3);
            const char* src = buf + (size_t)row * stride + (size_t)kt * 128 + (tid & 7) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem[wave] + j * 1024), 16, 0, 0);
        }
        __syncthreads();
    }
    if (((unsigned*)smem[wave])[lane] == 0x12345678u) sink[0] = 1;
}

int main() {
    char* buf; unsigned* sink;
    size_t bytes = 64u << 20;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&sink, 64));
    for (int stride : {128, 256, 512, 1024, 2048, 4096}) {
        int ktiles = stride / 128;
        int rows_total = (2 << 20) / stride;                   // 2 MB working set -> L2 resident
        for (int stagger : {0, 1}) {
            int grid = 256 * 4, iters = 400;
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, stride, rows_total, ktiles, 10, stagger, sink);
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, stride, rows_total, ktiles, iters, stagger, sink);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("stride %5d B (Cin=%4d) stagger=%d : %7.2f TB/s\n", stride, stride / 2, stagger, (double)grid * iters * 16384.0 / ms / 1e9);
        }
    }
    return 0;
}
