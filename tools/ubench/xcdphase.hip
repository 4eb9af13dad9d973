// What does a PHASE BOUNDARY cost inside one persistent launch when producer and consumer sit on the SAME XCD?  (DESIGN.md 4.2 priced the per-stage
// persistent launch for ResNet-50 from MI355X_MICROARCH.md's table; this measures the hand-off in the form such a kernel would use.)
// 256 blocks (one per CU), each learns its XCD from HW_REG_XCC_ID and takes a slot from that XCD's ticket.  Phase p: a block writes CHUNK bytes of a
// pattern f(p, xcd, slot) into buffer p & 1, drains its stores (s_waitcnt vmcnt(0)), meets the other blocks of ITS XCD at a counter (one relaxed
// agent-scope atomic add per block, one lane polling with sc1 loads), then reads the chunk its neighbour slot wrote — by LDS-DMA, as a conv kernel's
// loader would — and checks every word.  Variants of the consumer side:
//   0  plain LDS-DMA loads, no fence                      (expected stale: the L1 may hold last round's lines of the same addresses)
//   1  LDS-DMA with sc1 (L1 bypass), no fence             (producer's plain stores sit in the SAME XCD's L2: L2-served)
//   2  agent acquire (buffer_inv sc1) by every wave, plain loads
//   3  release (buffer_wbl2 sc1) before the arrive + acquire after the poll, plain loads: the architecturally sanctioned form
//   4  variant 1 with a CROSS-XCD neighbour (control: must be stale / wrong without the release)
// Baseline: the same write and read phases as separate kernel launches (hipGraph-free, same stream).
// build: hipcc --offload-arch=gfx950 -O3 -o xcdphase xcdphase.hip ; run: ./xcdphase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int CHUNK = 64 * 1024, NBLK = 256, NT = 256, LDSB = 96 * 1024;

__device__ __forceinline__ unsigned pattern(int p, int xcd, int slot, int i) { return (unsigned)(p * 2654435761u) ^ (unsigned)(xcd * 40503u + slot * 9973u + i * 7u + 12345u); }

template <int AUX> __device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

struct Ctl { unsigned ticket[8]; unsigned bar[8 * 32]; unsigned errors[8]; unsigned timeouts; unsigned long long clk[4]; };

template <int VAR>
__global__ __launch_bounds__(NT) void phases(unsigned char* buf0, unsigned char* buf1, Ctl* ctl, int nphase) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_xcd, s_slot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        s_xcd = (int)(xcc & 7);
        s_slot = (int)atomicAdd(&ctl->ticket[s_xcd], 1u);
    }
    __syncthreads();
    const int xcd = s_xcd, slot = s_slot;
    if (slot >= 32) { if (tid == 0) atomicAdd(&ctl->timeouts, 1000u); return; }        // (placement is not b % 8: give up loudly)
    unsigned errs = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t_bar = 0;
    for (int p = 0; p < nphase; p++) {
        unsigned char* wb = (p & 1) ? buf1 : buf0;
        // ---- produce: CHUNK bytes, 16-byte stores ----
        unsigned char* mine = wb + ((size_t)(xcd * 32 + slot)) * CHUNK;
        for (int i = tid; i < CHUNK / 16; i += NT) {
            uint4 v; v.x = pattern(p, xcd, slot, 4 * i); v.y = pattern(p, xcd, slot, 4 * i + 1); v.z = pattern(p, xcd, slot, 4 * i + 2); v.w = pattern(p, xcd, slot, 4 * i + 3);
            *(uint4*)(mine + (size_t)i * 16) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
        // ---- XCD-local barrier ----
        if (tid == 0) {
            if (VAR == 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(&ctl->bar[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = 32u * (unsigned)(p + 1);
            int spins = 0;
            while (__hip_atomic_load(&ctl->bar[xcd * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 2000000) { atomicAdd(&ctl->timeouts, 1u); break; }
            }
            if (VAR == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (VAR == 2 && lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        t_bar += __builtin_amdgcn_s_memtime() - tb0;
        // ---- consume the neighbour's chunk through the LDS-DMA, check every word ----
        const int nx = VAR == 4 ? (xcd + 1) & 7 : xcd, ns = (slot + 1) & 31;
        const unsigned char* theirs = wb + ((size_t)(nx * 32 + ns)) * CHUNK;
        for (int i0 = 0; i0 < CHUNK / 16; i0 += NT) {
            const int i = i0 + tid;
            if (VAR == 1 || VAR == 4) glds16<16>(theirs + (size_t)i * 16, smem + (size_t)(i0 + wave * 64) * 16);
            else glds16<0>(theirs + (size_t)i * 16, smem + (size_t)(i0 + wave * 64) * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < CHUNK / 16; i += NT) {
            const uint4 v = *(const uint4*)(smem + (size_t)i * 16);
            errs += (v.x != pattern(p, nx, ns, 4 * i)) + (v.y != pattern(p, nx, ns, 4 * i + 1)) + (v.z != pattern(p, nx, ns, 4 * i + 2)) + (v.w != pattern(p, nx, ns, 4 * i + 3));
        }
        __syncthreads();
    }
    if (errs) atomicAdd(&ctl->errors[xcd], errs);
    if (tid == 0 && blockIdx.x == 0) { ctl->clk[0] = __builtin_amdgcn_s_memtime() - t0; ctl->clk[1] = t_bar; }
}

// the same phases as separate launches
__global__ __launch_bounds__(NT) void k_write(unsigned char* wb, int p) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, tid = threadIdx.x;
    unsigned char* mine = wb + ((size_t)(xcd * 32 + slot)) * CHUNK;
    for (int i = tid; i < CHUNK / 16; i += NT) {
        uint4 v; v.x = pattern(p, xcd, slot, 4 * i); v.y = pattern(p, xcd, slot, 4 * i + 1); v.z = pattern(p, xcd, slot, 4 * i + 2); v.w = pattern(p, xcd, slot, 4 * i + 3);
        *(uint4*)(mine + (size_t)i * 16) = v;
    }
}
__global__ __launch_bounds__(NT) void k_read(const unsigned char* wb, int p, Ctl* ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, tid = threadIdx.x, wave = tid >> 6, ns = (slot + 1) & 31;
    const unsigned char* theirs = wb + ((size_t)(xcd * 32 + ns)) * CHUNK;
    for (int i0 = 0; i0 < CHUNK / 16; i0 += NT) glds16<0>(theirs + (size_t)(i0 + tid) * 16, smem + (size_t)(i0 + wave * 64) * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned errs = 0;
    for (int i = tid; i < CHUNK / 16; i += NT) {
        const uint4 v = *(const uint4*)(smem + (size_t)i * 16);
        errs += (v.x != pattern(p, xcd, ns, 4 * i)) + (v.y != pattern(p, xcd, ns, 4 * i + 1)) + (v.z != pattern(p, xcd, ns, 4 * i + 2)) + (v.w != pattern(p, xcd, ns, 4 * i + 3));
    }
    if (errs) atomicAdd(&ctl->errors[xcd], errs);
}

template <int VAR> void run(unsigned char* b0, unsigned char* b1, Ctl* ctl, int nphase) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(ctl, 0, sizeof(Ctl));
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(phases<VAR>, dim3(NBLK), dim3(NT), LDSB, 0, b0, b1, ctl, nphase);
        hipEventRecord(b, 0);
        if (hipEventSynchronize(b) != hipSuccess) { printf("variant %d: launch failed\n", VAR); return; }
        float ms; hipEventElapsedTime(&ms, a, b);
        Ctl h; hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long e = 0; for (int i = 0; i < 8; i++) e += h.errors[i];
        printf("variant %d rep %d: %8.2f us per phase (%d phases) | barrier part %.2f us per phase (block 0) | wrong words %llu of %llu | timeouts %u | tickets", VAR, rep, ms * 1e3 / nphase, nphase,
               (double)h.clk[1] / nphase / 100.0, e, (unsigned long long)nphase * NBLK * (CHUNK / 4), h.timeouts);
        for (int i = 0; i < 8; i++) printf(" %u", h.ticket[i]);
        printf("\n");
    }
}

int main() {
    unsigned char *b0, *b1; Ctl* ctl;
    hipMalloc(&b0, (size_t)NBLK * CHUNK); hipMalloc(&b1, (size_t)NBLK * CHUNK); hipMalloc(&ctl, sizeof(Ctl));
    hipFuncSetAttribute((const void*)phases<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    const int nphase = 40;
    run<0>(b0, b1, ctl, nphase); run<1>(b0, b1, ctl, nphase); run<2>(b0, b1, ctl, nphase); run<3>(b0, b1, ctl, nphase); run<4>(b0, b1, ctl, nphase);
    // separate launches
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(ctl, 0, sizeof(Ctl));
        hipEventRecord(a, 0);
        for (int p = 0; p < nphase; p++) {
            hipLaunchKernelGGL(k_write, dim3(NBLK), dim3(NT), 0, 0, (p & 1) ? b1 : b0, p);
            hipLaunchKernelGGL(k_read, dim3(NBLK), dim3(NT), LDSB, 0, (p & 1) ? b1 : b0, p, ctl);
        }
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        Ctl h; hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long e = 0; for (int i = 0; i < 8; i++) e += h.errors[i];
        printf("separate launches rep %d: %8.2f us per phase (write + read kernel) | wrong words %llu\n", rep, ms * 1e3 / nphase, e);
    }
    return 0;
}
