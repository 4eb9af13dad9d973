// micro-benchmark: L2 -> CU bandwidth for (a) LDS-DMA (global_load_lds_dwordx4), (b) plain global_load_dwordx4 to VGPRs,
// on an L2-resident working set.  hipcc --offload-arch=gfx950 -O3 l2bw.hip -o l2bw && ./l2bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int ROWB>   // ROWB: bytes of contiguous run per 8 lanes (128 = full line per 8 lanes)
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, size_t ws_bytes, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char smem[4][16384];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // each block walks a 64 KB window (256 threads x 16 B x 16 instr) starting at a block-dependent offset
    size_t base = ((size_t)blockIdx.x * 65536) % ws_bytes;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        size_t off = (base + (size_t)it * 65536 * 7) % ws_bytes;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // lane -> address: rows of ROWB bytes; 8 lanes cover one 128-B line when ROWB == 128
            const char* src = buf + off + (size_t)j * 4096 + (size_t)tid * 16;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem[wave] + j * 1024), 16, 0, 0);
            } else {
                uint4 v = *(const uint4*)src;
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
        if (MODE == 0) __syncthreads();
    }
    if (MODE == 0) acc.x = ((unsigned*)smem[wave])[lane];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int MODE> void run(const char* name, const char* buf, size_t ws, int blocks_per_cu, unsigned* sink) {
    int grid = 256 * blocks_per_cu, iters = 200;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, 128>), dim3(grid), dim3(256), 0, 0, buf, ws, 10, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k<MODE, 128>), dim3(grid), dim3(256), 0, 0, buf, ws, iters, sink);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)grid * iters * 65536.0;
    printf("%-10s ws=%6.1f MB blocks/CU=%d : %7.2f TB/s\n", name, ws / 1e6, blocks_per_cu, bytes / ms / 1e9);
}

int main() {
    char* buf; unsigned* sink;
    size_t maxws = 512u << 20;
    CK(hipMalloc(&buf, maxws + (1 << 20))); CK(hipMemset(buf, 1, maxws + (1 << 20))); CK(hipMalloc(&sink, 64));
    for (size_t ws : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)512 << 20})
        for (int bpc : {1, 2, 4}) { run<0>("lds-dma", buf, ws, bpc, sink); run<1>("vgpr-load", buf, ws, bpc, sink); }
    return 0;
}
