/* oracle/orc_core.c — ORACLE (test infrastructure).  The arithmetic Smelter performs itself on the CPU:
 * fp16 codec, typed tensor views, conv-weight re-layout, size/offset rules, readback packing.
 * Each function cites the reference lines it restates. */
#include "orc.h"
#include <math.h>
#include <string.h>

/* ---- Float16.swift:17-45 (vImageConvert_Planar16FtoPlanarF): exact widening, subnormals kept ---- */
float orc_half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal: normalise */
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

/* ---- Float16.swift:53-77 (vImageConvert_PlanarFtoPlanar16F): round-to-nearest-even, overflow -> inf,
 *      underflow -> subnormal/zero, NaN stays NaN (quiet bit forced) ---- */
uint16_t orc_float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (ax > 0x7F800000u ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);           /* >= 65520 rounds to inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                        /* <= 2^-25 rounds to zero */
    int e = (int)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                      /* subnormal: extra shift */
    uint32_t halfman = man >> shift, rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (halfman & 1u))) halfman++;
    uint32_t h = (e < -14) ? halfman : (((uint32_t)(e + 15) << 10) + (halfman - 0x400u)); /* carry rolls into exp */
    return (uint16_t)(sign | h);
}
void orc_half_to_float_n(const uint16_t* s, float* d, size_t n) { for (size_t i = 0; i < n; i++) d[i] = orc_half_to_float(s[i]); }
void orc_float_to_half_n(const float* s, uint16_t* d, size_t n) { for (size_t i = 0; i < n; i++) d[i] = orc_float_to_half(s[i]); }
void orc_round_to_half(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) x[i] = orc_half_to_float(orc_float_to_half(x[i]));
}

/* ---- Onnx_TensorProto+Extensions.swift ---- */
enum { DT_FLOAT = 1, DT_UINT8, DT_INT8, DT_UINT16, DT_INT16, DT_INT32, DT_INT64, DT_STRING, DT_BOOL, DT_FLOAT16,
       DT_DOUBLE, DT_UINT32, DT_UINT64 };

long orc_tensor_length(const orc_tensor* t) { /* :64-66 */
    long n = 1; for (int i = 0; i < t->ndims; i++) n *= (long)t->dims[i]; return n;
}

/* Data+Extensions.swift:4-14: count = bytes / stride (floor), little endian reinterpret */
#define EMIT(count, expr) do { long _n = (long)(count); if (dst) { if ((size_t)_n > cap) _n = (long)cap; \
    for (long i = 0; i < _n; i++) dst[i] = (expr); } return (long)(count); } while (0)

long orc_tensor_floats(const orc_tensor* t, float* dst, size_t cap) { /* :36-62 */
    switch (t->data_type) {
    case DT_INT32: case DT_INT16: case DT_INT8: case DT_UINT16: case DT_UINT8: case DT_BOOL:
        EMIT(t->n_int32, (float)t->int32_data[i]);
    case DT_INT64: EMIT(t->n_int64, (float)t->int64_data[i]);            /* :46-47: no raw fallback here */
    case DT_UINT32: case DT_UINT64: EMIT(t->n_uint64, (float)t->uint64_data[i]);
    case DT_FLOAT:
        if (t->n_float == 0) { const uint8_t* p = t->raw_data.p; float v;                     /* :51-53 */
            EMIT(t->raw_data.n / 4, (memcpy(&v, p + 4 * i, 4), v)); }
        EMIT(t->n_float, t->float_data[i]);
    case DT_DOUBLE: EMIT(t->n_double, (float)t->double_data[i]);
    case DT_FLOAT16: { const uint8_t* p = t->raw_data.p; uint16_t h;                          /* :57-58 */
        EMIT(t->raw_data.n / 2, (memcpy(&h, p + 2 * i, 2), orc_half_to_float(h))); }
    default: return ORC_ERR_UNSUPPORTED_DTYPE;                                                /* :60 fatalError */
    }
}

/* Int(Float) truncates toward zero and traps on NaN / out-of-range (Swift); here: saturate, NaN -> 0 (a C cast would be undefined) */
static int64_t trunc_i64(double v) { return v != v ? 0 : v >= 9.2233720368547748e18 ? INT64_MAX : v <= -9.2233720368547758e18 ? INT64_MIN : (int64_t)v; }
long orc_tensor_integers(const orc_tensor* t, int64_t* dst, size_t cap) { /* :2-34; Int(Float) truncates toward 0 */
    switch (t->data_type) {
    case DT_INT32: case DT_INT16: case DT_INT8: case DT_UINT16: case DT_UINT8: case DT_BOOL:
        EMIT(t->n_int32, (int64_t)t->int32_data[i]);
    case DT_INT64:
        if (t->n_int64 == 0) { const uint8_t* p = t->raw_data.p; int64_t v;                   /* :11-14 */
            EMIT(t->raw_data.n / 8, (memcpy(&v, p + 8 * i, 8), v)); }
        EMIT(t->n_int64, t->int64_data[i]);
    case DT_UINT32: case DT_UINT64: EMIT(t->n_uint64, (int64_t)t->uint64_data[i]);
    case DT_FLOAT:
        if (t->n_float == 0) { const uint8_t* p = t->raw_data.p; float v;                     /* :21-24 */
            EMIT(t->raw_data.n / 4, (memcpy(&v, p + 4 * i, 4), trunc_i64(v))); }
        EMIT(t->n_float, trunc_i64(t->float_data[i]));
    case DT_DOUBLE: EMIT(t->n_double, trunc_i64(t->double_data[i]));
    case DT_FLOAT16: { const uint8_t* p = t->raw_data.p; uint16_t h;                          /* :29-30 */
        EMIT(t->raw_data.n / 2, (memcpy(&h, p + 2 * i, 2), trunc_i64(orc_half_to_float(h)))); }
    default: return ORC_ERR_UNSUPPORTED_DTYPE;                                                /* :32 fatalError */
    }
}

/* ---- Array+Extensions.swift:52-93 ---- */
void orc_reformat_conv_weight(const void* src, void* dst, int es, int OC, int IC, int KH, int KW, int is_transpose) {
    const uint8_t* s = (const uint8_t*)src; uint8_t* d = (uint8_t*)dst;
    for (int oc = 0; oc < OC; oc++)
        for (int ic = 0; ic < IC; ic++)
            for (int kh = 0; kh < KH; kh++)
                for (int kw = 0; kw < KW; kw++) {
                    size_t in_idx, out_idx;
                    if (is_transpose) { /* :71-76 */
                        in_idx = (size_t)ic * OC * KH * KW + (size_t)oc * KH * KW + (size_t)kh * KW + kw;
                        out_idx = (size_t)oc * KH * KW * IC + (size_t)(KH - 1 - kh) * KW * IC + (size_t)(KW - 1 - kw) * IC + ic;
                    } else {            /* :78-83 */
                        in_idx = (size_t)oc * IC * KH * KW + (size_t)ic * KH * KW + (size_t)kh * KW + kw;
                        out_idx = (size_t)oc * IC * KH * KW + (size_t)kh * KW * IC + (size_t)kw * IC + ic;
                    }
                    memcpy(d + out_idx * es, s + in_idx * es, es);
                }
}

/* ---- ONNXConvolutionPadding.swift:91-113 (Swift Int division truncates toward zero) ---- */
void orc_conv_padded_size(int in_w, int in_h, int kh, int kw, int sh, int sw, int pt, int pl, int pb, int pr,
                          int oph, int opw, int is_transpose, int* out_w, int* out_h) {
    if (is_transpose) { /* :97-103 */
        *out_h = (in_h - 1) * sh - pt - pb + kh + oph;
        *out_w = (in_w - 1) * sw - pl - pr + kw + opw;
    } else {            /* :105-110 — dilation is NOT part of the rule */
        *out_h = (in_h + pt + pb - kh) / sh + 1;
        *out_w = (in_w + pl + pr - kw) / sw + 1;
    }
}

/* ---- PyTorchPoolPadding.swift:94-103: Int(Float(in + 2p - k) / Float(s) + 1.0) ---- */
void orc_pool_padded_size(int in_w, int in_h, int kh, int kw, int sh, int sw, int ph, int pw, int* out_w, int* out_h) {
    *out_w = (int)((float)(in_w + 2 * pw - kw) / (float)sw + 1.0f);
    *out_h = (int)((float)(in_h + 2 * ph - kh) / (float)sh + 1.0f);
}
/* ---- PyTorchPoolPadding.swift:56-75: offset = k/2 - ((out-1)*s + k - in)/2  => window start = -extra/2 ---- */
void orc_pool_window_start(int in, int out, int k, int s, int* start) {
    int read = (out - 1) * s + k, extra = read - in;
    *start = -(extra / 2); /* Swift Int division truncates toward zero */
}

/* ---- MPSImage+Extensions.swift:26-59 ---- */
size_t orc_float_array_count(int n, int c, int h, int w) {
    int slices = (c + 3) / 4, cpp = (c < 3) ? c : slices * 4;                /* :27-31 */
    return (size_t)w * h * cpp * n;                                           /* :40 */
}
void orc_pack_float_array(const float* x, int n, int c, int h, int w, float* dst) {
    int slices = (c + 3) / 4, comps = (c < 3) ? c : 4;                        /* :27,36 */
    memset(dst, 0, orc_float_array_count(n, c, h, w) * sizeof(float));        /* :41 zero-initialised */
    for (int i = 0; i < slices * n; i++) {                                    /* :48: slice index = image*slices + s */
        int img = i / slices, sl = i % slices;
        float* o = dst + (size_t)w * h * comps * i;                           /* :50 */
        for (int y = 0; y < h; y++)
            for (int xx = 0; xx < w; xx++)
                for (int k = 0; k < comps; k++) {
                    int ch = sl * 4 + k;
                    o[((size_t)y * w + xx) * comps + k] = ch < c ? x[(((size_t)img * c + ch) * h + y) * w + xx] : 0.f;
                }
    }
}
