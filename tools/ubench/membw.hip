// micro-benchmark: HBM read / write / copy bandwidth with 16-byte lanes, grid-stride, for several footprints
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE> __global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, unsigned* sink) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        else if (MODE == 1) dst[i] = uint4{(unsigned)i, 1, 2, 3};
        else dst[i] = src[i];
    }
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
int main() {
    size_t maxb = 1024u << 20; uint4 *a, *b; unsigned* sink;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&sink, 64)); CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 2, maxb));
    const char* names[] = {"read", "write", "copy"};
    for (size_t mb : {26, 103, 206, 512, 1024}) {
        size_t n = mb * (1u << 20) / 16;
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, a, b, n, sink);
                                  else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, a, b, n, sink);
                                  else hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, a, b, n, sink); };
            launch(); CK(hipEventRecord(e0, 0)); for (int i = 0; i < 10; i++) launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            double bytes = (double)n * 16 * (mode == 2 ? 2 : 1);
            printf("%5zu MB %-5s : %7.1f us  %6.2f TB/s\n", mb, names[mode], ms * 1e3, bytes / ms / 1e9);
        }
    }
    return 0;
}
