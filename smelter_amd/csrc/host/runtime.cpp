// runtime.cpp — error state, device/stream plumbing, HIP-event timers and the host-side weight preparation
// helpers of the C ABI (include/smelter_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../kernels/common.hpp"
#include "half.hpp"

static thread_local char g_err[512] = "";

void sm_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}

struct sm_timer { hipEvent_t a, b; };

extern "C" {

const char* smelter_last_error_string(void) { return g_err; }
const char* smelter_version(void) { return "smelter_hip 0.1 (gfx950)"; }

int sm_device_count(int32_t* n) { int c = 0; SM_HIP_CHECK(hipGetDeviceCount(&c)); *n = c; return SM_OK; }
int sm_set_device(int32_t dev) { SM_HIP_CHECK(hipSetDevice(dev)); return SM_OK; }
int sm_malloc(void** p, size_t bytes) { SM_HIP_CHECK(hipMalloc(p, bytes ? bytes : 1)); return SM_OK; }
int sm_free(void* p) { if (p) SM_HIP_CHECK(hipFree(p)); return SM_OK; }
int sm_memcpy_h2d(void* dst, const void* src, size_t bytes, sm_stream_t s) { SM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, sm_s(s))); return SM_OK; }
int sm_memcpy_d2h(void* dst, const void* src, size_t bytes, sm_stream_t s) { SM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, sm_s(s))); return SM_OK; }
int sm_memset(void* dst, int value, size_t bytes, sm_stream_t s) { SM_HIP_CHECK(hipMemsetAsync(dst, value, bytes, sm_s(s))); return SM_OK; }
int sm_stream_create(sm_stream_t* s) { hipStream_t h; SM_HIP_CHECK(hipStreamCreateWithFlags(&h, hipStreamNonBlocking)); *s = (sm_stream_t)h; return SM_OK; }
int sm_stream_destroy(sm_stream_t s) { SM_HIP_CHECK(hipStreamDestroy(sm_s(s))); return SM_OK; }
int sm_stream_synchronize(sm_stream_t s) { SM_HIP_CHECK(hipStreamSynchronize(sm_s(s))); return SM_OK; }

int sm_timer_create(sm_timer** t) {
    sm_timer* x = new sm_timer();
    SM_HIP_CHECK(hipEventCreate(&x->a)); SM_HIP_CHECK(hipEventCreate(&x->b)); *t = x; return SM_OK;
}
int sm_timer_start(sm_timer* t, sm_stream_t s) { SM_HIP_CHECK(hipEventRecord(t->a, sm_s(s))); return SM_OK; }
int sm_timer_stop(sm_timer* t, sm_stream_t s) { SM_HIP_CHECK(hipEventRecord(t->b, sm_s(s))); return SM_OK; }
int sm_timer_elapsed_ms(sm_timer* t, float* ms) { SM_HIP_CHECK(hipEventSynchronize(t->b)); SM_HIP_CHECK(hipEventElapsedTime(ms, t->a, t->b)); return SM_OK; }
int sm_timer_destroy(sm_timer* t) { if (t) { (void)hipEventDestroy(t->a); (void)hipEventDestroy(t->b); delete t; } return SM_OK; }

// ---- Float16.swift:17-77 ------------------------------------------------------------------------------
void sm_half_to_float(const uint16_t* src, float* dst, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = smelter::half_bits_to_float(src[i]); }
void sm_float_to_half(const float* src, uint16_t* dst, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = smelter::float_to_half_bits(src[i]); }

// ---- Array+Extensions.swift:52-93 ----------------------------------------------------------------------
// The reference walks (oc, ic, kh, kw) with a scalar 4-deep loop and strided reads.  Here the OHWI image is
// produced output-major so that writes are sequential and the ic gather stays inside one kh*kw-strided plane.
int sm_reformat_conv_weight(const void* src, void* dst, int32_t es, int32_t OC, int32_t IC, int32_t KH, int32_t KW, int32_t is_transpose) {
    SM_REQUIRE(src && dst && (es == 2 || es == 4) && OC > 0 && IC > 0 && KH > 0 && KW > 0, "sm_reformat_conv_weight: bad argument");
    const size_t taps = (size_t)KH * KW;
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* s = (const T*)src; T* d = (T*)dst;
        for (int oc = 0; oc < OC; oc++)
            for (size_t t = 0; t < taps; t++) {
                T* drow = d + ((size_t)oc * taps + t) * IC;
                if (!is_transpose) { const T* sp = s + (size_t)oc * IC * taps + t; for (int ic = 0; ic < IC; ic++) drow[ic] = sp[(size_t)ic * taps]; }
                else { size_t st = taps - 1 - t; /* flip H and W: (KH-1-kh)*KW + (KW-1-kw) == taps-1-t */
                       const T* sp = s + (size_t)oc * taps + st; for (int ic = 0; ic < IC; ic++) drow[ic] = sp[(size_t)ic * OC * taps]; }
            }
    };
    if (es == 2) run((uint16_t)0); else run((uint32_t)0);
    return SM_OK;
}

// ---- ONNXConvolutionPadding.swift:91-113 ---------------------------------------------------------------
void sm_conv_padded_size(int32_t in_w, int32_t in_h, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t pt, int32_t pl, int32_t pb,
                         int32_t pr, int32_t oph, int32_t opw, int32_t is_transpose, int32_t* out_w, int32_t* out_h) {
    if (is_transpose) { *out_h = (in_h - 1) * sh - pt - pb + kh + oph; *out_w = (in_w - 1) * sw - pl - pr + kw + opw; }
    else { *out_h = (in_h + pt + pb - kh) / sh + 1; *out_w = (in_w + pl + pr - kw) / sw + 1; }
}
// ---- PyTorchPoolPadding.swift:94-103 -------------------------------------------------------------------
void sm_pool_padded_size(int32_t in_w, int32_t in_h, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int32_t* out_w,
                         int32_t* out_h) {
    *out_w = (int32_t)((float)(in_w + 2 * pw - kw) / (float)sw + 1.0f);
    *out_h = (int32_t)((float)(in_h + 2 * ph - kh) / (float)sh + 1.0f);
}
// ---- PyTorchPoolPadding.swift:56-75 --------------------------------------------------------------------
int32_t sm_pool_window_start(int32_t in, int32_t out, int32_t k, int32_t s) { return -(((out - 1) * s + k - in) / 2); }

}  // extern "C"
