// Per-XCD workgroup start skew: is the spread of block start times (s_memrealtime) across XCDs real or a clock-offset artifact?
// Each block records its start/end wall clock and XCC id; the kernel spins `spin` wall-clock ticks (100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -o xcdskew xcdskew.hip ; run: ./xcdskew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
template <int LDS> __global__ __launch_bounds__(512) void probe(unsigned long long* out, int spin) {
    __shared__ unsigned char smem[LDS];
    unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t0; out[blockIdx.x * 4 + 1] = wall_clock64(); out[blockIdx.x * 4 + 2] = xcc & 0xf; out[blockIdx.x * 4 + 3] = smem[0]; }
}
template <int LDS> void run(const char* name, int grid, int threads, int spin, bool cold, char* scratch, size_t sb, unsigned long long* d) {
    std::vector<unsigned long long> h(grid * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        if (cold) hipMemsetAsync(scratch, rep, sb, 0);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe<LDS>, dim3(grid), dim3(threads), 0, 0, d, spin);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost);
        unsigned long long w0 = ~0ull, w1 = 0; double mn[8], mx[8]; for (int i = 0; i < 8; i++) { mn[i] = 1e30; mx[i] = -1; }
        for (int i = 0; i < grid; i++) { w0 = std::min(w0, h[i * 4]); w1 = std::max(w1, h[i * 4 + 1]); }
        for (int i = 0; i < grid; i++) { int x = (int)h[i * 4 + 2] & 7; double s = (h[i * 4] - w0) / 100.0; mn[x] = std::min(mn[x], s); mx[x] = std::max(mx[x], s); }
        printf("%-28s %s rep %d: event %.2f us, span %.2f us | first start per XCC:", name, cold ? "cold" : "warm", rep, ms * 1e3, (w1 - w0) / 100.0);
        for (int i = 0; i < 8; i++) printf(" %.2f", mn[i]); printf(" | last start:"); for (int i = 0; i < 8; i++) printf(" %.2f", mx[i]); printf("\n");
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 32);
    char* scratch; size_t sb = (size_t)768 << 20; hipMalloc(&scratch, sb);
    for (int cold = 0; cold < 2; cold++) {
        run<1024>("256 blocks x64thr 1KB", 256, 64, 500, cold, scratch, sb, d);
        run<1024>("196 blocks x512thr 1KB", 196, 512, 1000, cold, scratch, sb, d);
        run<65536>("196 blocks x512thr 64KB(st)", 196, 512, 1000, cold, scratch, sb, d);
        run<65536>("392 blocks x256thr 64KB(st)", 392, 256, 1000, cold, scratch, sb, d);
        run<1024>("2048 blocks x256thr 1KB", 2048, 256, 200, cold, scratch, sb, d);
    }
    return 0;
}
