// smelter_amd/csrc/kernels/common.hpp — shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include "../../../include/smelter_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F> __device__ __forceinline__ void sm_static_for(F&& f) {       // f(integral_constant<int, I>) ... f(<N-1>)
    if constexpr (I < N) { f(std::integral_constant<int, I>()); sm_static_for<I + 1, N>(f); }
}

// Cache policy of the OUTPUT stores of the convolution kernels (aux operand of __builtin_amdgcn_raw_buffer_store_*; 0 = default, 16 = sc1, 17 = sc0 sc1, 2 = nt).
// A layer's output is never re-read through this launch's L2 — the next launch invalidates it — but with the default policy the last <= 32 MB of it sit DIRTY in the
// L2s when the kernel ends, and the kernel boundary then waits for their write-back (MI355X_MICROARCH.md, boundary row: + B / 6 TB/s for B dirty bytes).  sc1 stores
// write through and drop the line.  Build-time switch for the A/B (make EXTRA=-DSM_ST_AUX=16).
// Cache policy of the kernels that read the fp32 IMAGE BATCH (stem_pool.hpp, stem3.hip, mbconv5s_kernel): 2 = nt.  The batch is read once per step and is 38 - 154 MB; loaded with
// the default policy it takes that much of the 256 MB Infinity Cache away from the activation arena and the weights (round 6, ranked in the network: ResNet-50 1.000 -> 0.971 ms).
#ifndef SM_STEM_IN_AUX
#define SM_STEM_IN_AUX 2
#endif
#ifndef SM_ST_AUX
#define SM_ST_AUX 0
#endif
// Experiment / ablation switches by environment variable exist in DIAGNOSTIC builds only (make EXTRA=-DSM_ABLATION; tools/lib_ab.sh): the product library's kernel
// launchers consult no environment variable (smelter_hip.h: everything a caller can set is a field of smelter_config or an argument).
#ifdef SM_ABLATION
#define SM_ABL_ENV(name) getenv(name)
#else
#define SM_ABL_ENV(name) ((const char*)nullptr)
#endif
#define SM_WAVE 64
#define SM_NUM_XCD 8
#define SM_NUM_CU 256

void sm_set_error(const char* fmt, ...);

#define SM_HIP_CHECK(expr)                                                                            \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            sm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return SM_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define SM_LAUNCH_CHECK()                                                                             \
    do {                                                                                              \
        hipError_t _e = hipGetLastError();                                                            \
        if (_e != hipSuccess) {                                                                       \
            sm_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return SM_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define SM_REQUIRE(cond, ...)                                                                         \
    do {                                                                                              \
        if (!(cond)) { sm_set_error(__VA_ARGS__); return SM_ERR_INVALID_ARGUMENT; }                   \
    } while (0)

// the MPSCNNNeuron*Node family with default parameters (smelter_hip.h SM_ACT_*); out of line: the conv epilogues only pay a
// uniform branch for it
__device__ __noinline__ static float sm_act_other(float v, int act, float slope) {
    switch (act) {
    case SM_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SM_ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);
    case SM_ACT_TANH: return tanhf(v);
    case SM_ACT_ELU: return v >= 0.f ? v : slope * (expf(v) - 1.f);
    case SM_ACT_ABS: return fabsf(v);
    case SM_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));   // expf overflows past 88.7; log1p(e^v) - v < 2.1e-9 beyond 20
    case SM_ACT_SOFTSIGN: return v / (1.f + fabsf(v));
    case SM_ACT_EXP: return expf(v);
    case SM_ACT_LOG: return logf(v);
    case SM_ACT_CLIP: return fminf(fmaxf(v, 0.f), slope);
    }
    return v;
}
__device__ __forceinline__ float sm_act(float v, int act, float slope) {
    // MPSCNNNeuronReLUNode(a): f(x) = x >= 0 ? x : a*x   (Converters.swift:351,376)
    if (act <= SM_ACT_PRELU) return (act != SM_ACT_NONE && v < 0.f) ? v * slope : v;
    return sm_act_other(v, act, slope);
}

// v_permlane32_swap_b32 a, b: lanes 32..63 of `a` trade places with lanes 0..31 of `b` (a[32+i] <-> b[i]).  As an asm statement: the
// builtin's SECOND result was silently replaced by a copy of the first when its operands were bit-cast floats and the results were
// consumed inside a divergent branch (ROCm 7.2, conv_igemm64.hpp's register epilogue; every second 4-channel group came out as a
// duplicate).  The s_nop covers the VALU-write -> v_permlane* hazard the compiler would otherwise pad (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void sm_permlane32_swap(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

static inline int sm_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline hipStream_t sm_s(sm_stream_t s) { return (hipStream_t)s; }
// grid for grid-stride elementwise kernels: enough blocks to fill the chip, capped (guide G11)
static inline int sm_stream_grid(long work_items, int block) {
    long b = (work_items + block - 1) / block;
    long cap = (long)SM_NUM_CU * 8;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
