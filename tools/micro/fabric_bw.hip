// fabric_bw.hip — how many bytes per second reach the CUs from each level of the memory system when EVERY CU streams at once (the condition of a convolution launch:
// each CU pulls its own copy of the weight tile and its pixel tile).  256-thread blocks, B blocks per CU, every wave issues 1 KB requests (global_load_dwordx4 per lane, consecutive
// lanes = consecutive 16 bytes), U requests in flight per wave.  Footprint and window decide the level (see main): a shared 2 MB (every XCD's L2), private 512 KB windows re-read
// (Infinity Cache), private 4 MB windows read once (HBM).  A window is larger than the CU's 32 KB L1 and is walked front to back, 4 KB (one KB per wave) per step.
//   hipcc --offload-arch=gfx950 -O3 -o build/fabric_bw tools/micro/fabric_bw.hip && build/fabric_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
template <int U, bool DMA>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ buf, size_t footprint, size_t window, int iters, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the block's window: spread over the footprint; wave w takes KB w, w + 4, ... of it (the block reads 4 KB consecutive per step); addresses = buffer descriptor of the window +
    // one constant lane offset + a scalar running offset: no vector address arithmetic between the requests
    const size_t nwin = footprint / window;
    const char* base = buf + (size_t)(blockIdx.x % nwin) * window;      // (private windows when the footprint holds one per block; else shared round-robin)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)window, 0x00020000);
    const u32x4s rsv = {(unsigned)(uintptr_t)base, (unsigned)((uintptr_t)base >> 32) & 0xffffu, (unsigned)window, 0x00020000u};
    const unsigned voff = (unsigned)lane * 16u;
    unsigned soff = (unsigned)wave * 1024u;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (DMA) {
                unsigned keep;
                const unsigned dst = lds0 + (unsigned)((wave * U + u) * 1024);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(rsv), "s"(__builtin_amdgcn_readfirstlane((int)dst)), "s"(soff) : "memory");
            } else {
                v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
            }
            soff += 4096u; if (soff >= (unsigned)window) soff -= (unsigned)window;
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!DMA) {
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= v[u];
        }
    }
    if (acc.x == 0x12345678u && acc.y == 7u) sink[0] = acc.z ^ acc.w;
}

template <int U, bool DMA>
double run(const char* buf, size_t footprint, size_t window, int blocks_per_cu, unsigned* sink, int cus, size_t per_block) {
    const int grid = cus * blocks_per_cu;
    const int iters = (int)(per_block / (4 * U * 1024));              // per_block: bytes every block streams per launch (window re-read per_block / window times)
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t lds = DMA ? (size_t)4 * U * 1024 : 0;
    for (int i = 0; i < 2; i++) stream_kernel<U, DMA><<<grid, 256, lds>>>(buf, footprint, window, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; i++) stream_kernel<U, DMA><<<grid, 256, lds>>>(buf, footprint, window, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return (double)grid * (double)iters * 4 * U * 1024 * reps / (ms * 1e-3) / 1e12;   // TB/s
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs\n", pr.gcnArchName, cus);
    const size_t big = (size_t)4 << 30;
    char* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    // level            footprint   window per block   bytes per block and launch
    // L2:              2 MB shared by all (8 windows of 256 KB; every XCD's L2 holds all of it), each block re-reads its window 16 times
    // Infinity Cache:  one private 512 KB window per block slot (32 blocks x 512 KB = 16 MB per XCD, beyond its 4 MB L2; <= 512 MB... total 128 - 512 MB), re-read 8 times
    // HBM:             one private 4 MB window per block slot, read ONCE per launch (1 - 4 GB per launch, beyond the 256 MB Infinity Cache)
    struct { const char* name; size_t fp, window, per_block; } lv[] = {
        {"L2 (2 MB shared)", (size_t)2 << 20, (size_t)256 << 10, (size_t)4 << 20},
        {"Infinity Cache (512 KB/block)", (size_t)128 << 20, (size_t)512 << 10, (size_t)4 << 20},
        {"HBM (4 MB/block, once)", big, (size_t)4 << 20, (size_t)4 << 20}};
    printf("%-32s %-7s %8s %8s %8s %8s   (TB/s over all CUs; per CU GB/s in brackets)\n", "level", "blk/CU", "U=2", "U=4", "U=8", "U=16");
    for (auto& l : lv)
        for (int b : {1, 2, 4}) {
            if ((size_t)cus * b * l.window > l.fp && l.fp > ((size_t)2 << 20)) continue;       // private windows must fit the footprint
            double r2 = run<2, false>(buf, l.fp, l.window, b, sink, cus, l.per_block), r4 = run<4, false>(buf, l.fp, l.window, b, sink, cus, l.per_block),
                   r8 = run<8, false>(buf, l.fp, l.window, b, sink, cus, l.per_block), r16 = run<16, false>(buf, l.fp, l.window, b, sink, cus, l.per_block);
            printf("%-32s %-7d %8.2f %8.2f %8.2f %8.2f   [%.0f %.0f %.0f %.0f]\n", l.name, b, r2, r4, r8, r16, r2 * 1e3 / cus, r4 * 1e3 / cus, r8 * 1e3 / cus, r16 * 1e3 / cus);
        }
    printf("LDS-DMA (buffer_load_dwordx4 ... lds)\n");
    for (auto& l : lv)
        for (int b : {1, 2}) {
            if ((size_t)cus * b * l.window > l.fp && l.fp > ((size_t)2 << 20)) continue;
            double r4 = run<4, true>(buf, l.fp, l.window, b, sink, cus, l.per_block), r8 = run<8, true>(buf, l.fp, l.window, b, sink, cus, l.per_block), r16 = run<16, true>(buf, l.fp, l.window, b, sink, cus, l.per_block);
            printf("%-32s %-7d %8s %8.2f %8.2f %8.2f   [%.0f %.0f %.0f]\n", l.name, b, "", r4, r8, r16, r4 * 1e3 / cus, r8 * 1e3 / cus, r16 * 1e3 / cus);
        }
    return 0;
}
