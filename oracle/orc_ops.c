/* oracle/orc_ops.c — ORACLE (test infrastructure).  Per-node arithmetic in NCHW fp32.
 *
 * Smelter delegates these to Apple MPS node classes (closed source); the call sites are the
 * `MPS*Node(` constructors in /root/reference/Sources/Smelter/Converters.swift (cited per function).
 * The arithmetic restated here is the ONNX operator definition restricted to the attributes Smelter reads,
 * accumulating in fp32 in a fixed (ic, kh, kw) order.  OpenMP over (image x output channel / row) so the
 * same code doubles as bench.py's cpu_baseline ("port").
 */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Converters.swift:253-264 (MPSCNNConvolutionNode) + ONNXConvolutionPadding.swift:73-79 (offset k/2 - pad,
 * edge mode zero):  y[n,k,oy,ox] = b[k] + sum_{c,kh,kw} x[n, g*Cg+c, oy*sh-pt+kh*dh, ox*sw-pl+kw*dw] * W[k,kh,kw,c] */
static int g_acc64 = 0;   /* fp64 accumulation: spot-check mode that separates accumulation-order effects from fp16 storage effects */
void orc_set_acc64(int on) { g_acc64 = on; }

#define ORC_CONV_BODY(ACC_T)                                                                                   \
    const int Cg = d->C / d->groups, Kg = d->K / d->groups;                                                   \
    const long total = (long)d->N * d->K;                                                                     \
    _Pragma("omp parallel")                                                                                   \
    {                                                                                                         \
        ACC_T* acc = (ACC_T*)malloc(sizeof(ACC_T) * (size_t)d->Ho * d->Wo);                                  \
        _Pragma("omp for schedule(dynamic, 1)")                                                               \
        for (long nk = 0; nk < total; nk++) {                                                                 \
            int n = (int)(nk / d->K), k = (int)(nk % d->K), g = k / Kg;                                       \
            float b = bias ? bias[k] : 0.f;                                                                   \
            for (int i = 0; i < d->Ho * d->Wo; i++) acc[i] = 0;                                               \
            for (int c = 0; c < Cg; c++) {                                                                    \
                const float* xp = x + ((size_t)n * d->C + (size_t)g * Cg + c) * d->H * d->W;                  \
                for (int kh = 0; kh < d->kh; kh++)                                                            \
                    for (int kw = 0; kw < d->kw; kw++) {                                                      \
                        ACC_T wv = w[(((size_t)k * d->kh + kh) * d->kw + kw) * Cg + c];                       \
                        for (int oy = 0; oy < d->Ho; oy++) {                                                  \
                            int iy = oy * d->sh - d->pt + kh * d->dh;                                         \
                            if (iy < 0 || iy >= d->H) continue;                                               \
                            const float* xr = xp + (size_t)iy * d->W;                                         \
                            ACC_T* ar = acc + (size_t)oy * d->Wo;                                             \
                            int x0 = kw * d->dw - d->pl;                                                      \
                            for (int ox = 0; ox < d->Wo; ox++) {                                              \
                                int ix = ox * d->sw + x0;                                                     \
                                if (ix >= 0 && ix < d->W) ar[ox] += xr[ix] * wv;                              \
                            }                                                                                 \
                        }                                                                                     \
                    }                                                                                         \
            }                                                                                                 \
            float* yp = y + ((size_t)n * d->K + k) * d->Ho * d->Wo;                                           \
            for (int i = 0; i < d->Ho * d->Wo; i++) yp[i] = (float)(acc[i] + b);                              \
        }                                                                                                     \
        free(acc);                                                                                            \
    }

void orc_conv2d(const orc_conv_desc* d, const float* x, const float* w, const float* bias, float* y) {
    if (g_acc64) { ORC_CONV_BODY(double) } else { ORC_CONV_BODY(float) }
}

/* Converters.swift:265-287 (MPSCNNConvolutionTransposeNode).  The data source hands MPS W_t[oc][kh'][kw'][ic] =
 * W[ic][oc][kH-1-kh'][kW-1-kw'] (Array+Extensions.swift:71-76); in those terms the ONNX ConvTranspose sum is a gather:
 * input row iy feeds output row oy through tap kh' iff iy*sH == oy + pt - (kH-1-kh')*dH. */
#define ORC_CONVT_BODY(ACC)                                                                                         \
    const int N = d->N, C = d->C, H = d->H, W = d->W, OC = d->K, KH = d->kh, KW = d->kw, Ho = d->Ho, Wo = d->Wo;   \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                                        \
    for (int n = 0; n < N; n++)                                                                                     \
        for (int oc = 0; oc < OC; oc++)                                                                             \
            for (int oy = 0; oy < Ho; oy++)                                                                         \
                for (int ox = 0; ox < Wo; ox++) {                                                                   \
                    ACC acc = 0;                                                                                    \
                    for (int kh = 0; kh < KH; kh++) {                                                               \
                        int ty = oy + d->pt - (KH - 1 - kh) * d->dh;                                                \
                        if (ty < 0 || ty % d->sh) continue;                                                         \
                        int iy = ty / d->sh; if (iy >= H) continue;                                                 \
                        for (int kw = 0; kw < KW; kw++) {                                                           \
                            int tx = ox + d->pl - (KW - 1 - kw) * d->dw;                                            \
                            if (tx < 0 || tx % d->sw) continue;                                                     \
                            int ix = tx / d->sw; if (ix >= W) continue;                                             \
                            const float* wp = w + (((size_t)oc * KH + kh) * KW + kw) * C;                           \
                            for (int ic = 0; ic < C; ic++) acc += (ACC)x[(((size_t)n * C + ic) * H + iy) * W + ix] * (ACC)wp[ic]; \
                        }                                                                                           \
                    }                                                                                               \
                    y[(((size_t)n * OC + oc) * Ho + oy) * Wo + ox] = (float)acc + (bias ? bias[oc] : 0.f);          \
                }
void orc_conv_transpose2d(const orc_conv_desc* d, const float* x, const float* w, const float* bias, float* y) {
    if (g_acc64) { ORC_CONVT_BODY(double) } else { ORC_CONVT_BODY(float) }
}

/* Converters.swift:288-302 (MPSCNNFullyConnectedNode), weight read as [out,in] :35-39 ; alpha/beta/trans ignored */
void orc_gemm_fc(const float* x, const float* w, const float* bias, float* y, int n, int in_f, int out_f) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < n; i++)
        for (int o = 0; o < out_f; o++) {
            float acc = 0.f;
            const float* xr = x + (size_t)i * in_f; const float* wr = w + (size_t)o * in_f;
            for (int k = 0; k < in_f; k++) acc += xr[k] * wr[k];
            y[(size_t)i * out_f + o] = acc + (bias ? bias[o] : 0.f);
        }
}

/* Converters.swift:797-827 (MPSCNNBatchNormalizationNode): y = (x-mean)*gamma/sqrt(var+eps)+beta */
void orc_batchnorm(const float* x, float* y, int n, int c, int hw, const float* gamma, const float* beta,
                   const float* mean, const float* var, float eps) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < n; i++)
        for (int ch = 0; ch < c; ch++) {
            float s = gamma[ch] / sqrtf(var[ch] + eps), b = beta[ch], m = mean[ch];
            const float* xp = x + ((size_t)i * c + ch) * hw; float* yp = y + ((size_t)i * c + ch) * hw;
            for (int j = 0; j < hw; j++) yp[j] = (xp[j] - m) * s + b;
        }
}

/* Converters.swift:342-384 (MPSCNNNeuronReLUNode(a)): x >= 0 ? x : a*x */
void orc_relu(const float* x, float* y, size_t n, float slope) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) y[i] = x[i] >= 0.f ? x[i] : slope * x[i];
}

/* Converters.swift:430-451 (MPSNNAdditionNode) */
void orc_add(const float* a, const float* b, float* y, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) y[i] = a[i] + b[i];
}

void orc_binary(int op, const float* a, const float* b, float* y, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) y[i] = op == 0 ? a[i] + b[i] : op == 1 ? a[i] - b[i] : op == 2 ? a[i] * b[i] : a[i] / b[i];
}

void orc_neuron(int kind, const float* x, float* y, size_t n, float a) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        float v = x[i], r = v;
        switch (kind) {
        case 3: r = 1.f / (1.f + expf(-v)); break;
        case 4: r = fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f); break;     /* MPSCNNNeuronHardSigmoidNode(source:) defaults a=0.2 b=0.5 */
        case 5: r = tanhf(v); break;
        case 6: r = v >= 0.f ? v : a * (expf(v) - 1.f); break;
        case 7: r = fabsf(v); break;
        case 8: r = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))); break;   /* stable form: no overflow for v > 88 (fp16 activations reach 65504) */
        case 9: r = v / (1.f + fabsf(v)); break;
        case 10: r = expf(v); break;
        case 11: r = logf(v); break;
        case 12: r = fminf(fmaxf(v, 0.f), a); break;                       /* SUPERSET: ONNX Clip with lower bound 0 (ReLU6: a = 6) */
        }
        y[i] = r;
    }
}

void orc_softmax(const float* x, float* y, int n, int c, int hw, int logsm) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < n; i++)
        for (int p = 0; p < hw; p++) {
            const float* xp = x + (size_t)i * c * hw + p; float* yp = y + (size_t)i * c * hw + p;
            float mx = xp[0]; for (int k = 1; k < c; k++) if (xp[(size_t)k * hw] > mx) mx = xp[(size_t)k * hw];
            float sum = 0.f; for (int k = 0; k < c; k++) sum += expf(xp[(size_t)k * hw] - mx);
            for (int k = 0; k < c; k++) { float t = xp[(size_t)k * hw] - mx; yp[(size_t)k * hw] = logsm ? t - logf(sum) : expf(t) / sum; }
        }
}

void orc_group_norm(const float* x, float* y, int n, int c, int hw, int groups, const float* gamma, const float* beta, float eps) {
    const int cg = c / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < n; i++)
        for (int g = 0; g < groups; g++) {
            const float* xp = x + ((size_t)i * c + (size_t)g * cg) * hw; float* yp = y + ((size_t)i * c + (size_t)g * cg) * hw;
            const size_t cnt = (size_t)cg * hw;
            double s = 0; for (size_t k = 0; k < cnt; k++) s += xp[k];
            const float mean = (float)(s / (double)cnt);
            double q = 0; for (size_t k = 0; k < cnt; k++) { double d = (double)xp[k] - mean; q += d * d; }
            const float rstd = 1.f / sqrtf((float)(q / (double)cnt) + eps);
            for (int ch = 0; ch < cg; ch++) {
                const float ga = gamma ? gamma[g * cg + ch] : 1.f, be = beta ? beta[g * cg + ch] : 0.f;
                for (int k = 0; k < hw; k++) yp[(size_t)ch * hw + k] = (xp[(size_t)ch * hw + k] - mean) * rstd * ga + be;
            }
        }
}

void orc_pad2d(const float* x, float* y, int n, int c, int h, int w, int pt, int pl, int pb, int pr, int mode) {
    const int ho = h + pt + pb, wo = w + pl + pr;
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < n; i++)
        for (int ch = 0; ch < c; ch++)
            for (int oy = 0; oy < ho; oy++)
                for (int ox = 0; ox < wo; ox++) {
                    int iy = oy - pt, ix = ox - pl, zero = 0;
                    if (mode == 2) { iy = iy < 0 ? 0 : iy >= h ? h - 1 : iy; ix = ix < 0 ? 0 : ix >= w ? w - 1 : ix; }
                    else if (mode == 1) {
                        if (h > 1) { int per = 2 * (h - 1); iy = ((iy % per) + per) % per; if (iy >= h) iy = per - iy; } else iy = 0;
                        if (w > 1) { int per = 2 * (w - 1); ix = ((ix % per) + per) % per; if (ix >= w) ix = per - ix; } else ix = 0;
                    } else zero = iy < 0 || iy >= h || ix < 0 || ix >= w;
                    y[(((size_t)i * c + ch) * ho + oy) * wo + ox] = zero ? 0.f : x[(((size_t)i * c + ch) * h + iy) * w + ix];
                }
}

static float orc_lanczos3(float t) {
    t = fabsf(t);
    if (t < 1e-6f) return 1.f;
    if (t >= 3.f) return 0.f;
    const float pt = 3.14159265358979f * t;
    return 3.f * sinf(pt) * sinf(pt / 3.f) / (pt * pt);
}
void orc_resize_nchw(const float* x, float* y, int planes, int H, int W, int Ho, int Wo, int mode) {
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
#pragma omp parallel for schedule(static)
    for (int pl = 0; pl < planes; pl++) {
        const float* src = x + (size_t)pl * H * W; float* dst = y + (size_t)pl * Ho * Wo;
        for (int oy = 0; oy < Ho; oy++)
            for (int ox = 0; ox < Wo; ox++) {
                const float cy = ((float)oy + 0.5f) * sy - 0.5f, cx = ((float)ox + 0.5f) * sx - 0.5f;
                float r;
                if (mode == 2) {
                    const float fy = fminf(fmaxf(cy, 0.f), (float)(H - 1)), fx = fminf(fmaxf(cx, 0.f), (float)(W - 1));
                    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
                    const float wy = fy - (float)y0, wx = fx - (float)x0;
                    r = (src[(size_t)y0 * W + x0] * (1.f - wx) + src[(size_t)y0 * W + x1] * wx) * (1.f - wy) +
                        (src[(size_t)y1 * W + x0] * (1.f - wx) + src[(size_t)y1 * W + x1] * wx) * wy;
                } else {
                    const float fsy = fmaxf(sy, 1.f), fsx = fmaxf(sx, 1.f);
                    const int ry = (int)ceilf(3.f * fsy), rx = (int)ceilf(3.f * fsx), by = (int)floorf(cy), bx = (int)floorf(cx);
                    float acc = 0.f, wsum = 0.f;
                    for (int dy = -ry + 1; dy <= ry; dy++) {
                        const float wy = orc_lanczos3(((float)(by + dy) - cy) / fsy);
                        if (wy == 0.f) continue;
                        int iy = by + dy; iy = iy < 0 ? 0 : iy >= H ? H - 1 : iy;
                        float racc = 0.f, rw = 0.f;
                        for (int dx = -rx + 1; dx <= rx; dx++) {
                            const float wx = orc_lanczos3(((float)(bx + dx) - cx) / fsx);
                            if (wx == 0.f) continue;
                            int ix = bx + dx; ix = ix < 0 ? 0 : ix >= W ? W - 1 : ix;
                            racc += wx * src[(size_t)iy * W + ix]; rw += wx;
                        }
                        acc += wy * racc; wsum += wy * rw;
                    }
                    r = acc / wsum;
                }
                dst[(size_t)oy * Wo + ox] = r;
            }
    }
}

/* Converters.swift:652-695 (MPSCNNPoolingMaxNode) + PyTorchPoolPadding.swift:56-75 (window start) */
void orc_maxpool(const float* x, float* y, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int ph,
                 int pw, int ho, int wo) {
    int y0, x0; (void)ph; (void)pw;
    orc_pool_window_start(h, ho, kh, sh, &y0);
    orc_pool_window_start(w, wo, kw, sw, &x0);
#pragma omp parallel for schedule(static)
    for (long nc = 0; nc < (long)n * c; nc++) {
        const float* xp = x + (size_t)nc * h * w; float* yp = y + (size_t)nc * ho * wo;
        for (int oy = 0; oy < ho; oy++)
            for (int ox = 0; ox < wo; ox++) {
                float m = -FLT_MAX; /* out-of-image taps are ignored */
                for (int a = 0; a < kh; a++) {
                    int iy = oy * sh + y0 + a; if (iy < 0 || iy >= h) continue;
                    for (int b = 0; b < kw; b++) {
                        int ix = ox * sw + x0 + b; if (ix < 0 || ix >= w) continue;
                        float v = xp[(size_t)iy * w + ix]; if (v > m) m = v;
                    }
                }
                yp[(size_t)oy * wo + ox] = m;
            }
    }
}

/* Converters.swift:607-650 (MPSCNNPoolingAverageNode): divisor = full window (zero padded taps count) */
void orc_avgpool(const float* x, float* y, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int ph,
                 int pw, int ho, int wo) {
    int y0, x0; (void)ph; (void)pw;
    orc_pool_window_start(h, ho, kh, sh, &y0);
    orc_pool_window_start(w, wo, kw, sw, &x0);
#pragma omp parallel for schedule(static)
    for (long nc = 0; nc < (long)n * c; nc++) {
        const float* xp = x + (size_t)nc * h * w; float* yp = y + (size_t)nc * ho * wo;
        for (int oy = 0; oy < ho; oy++)
            for (int ox = 0; ox < wo; ox++) {
                float s = 0.f;
                for (int a = 0; a < kh; a++) {
                    int iy = oy * sh + y0 + a; if (iy < 0 || iy >= h) continue;
                    for (int b = 0; b < kw; b++) {
                        int ix = ox * sw + x0 + b; if (ix < 0 || ix >= w) continue;
                        s += xp[(size_t)iy * w + ix];
                    }
                }
                yp[(size_t)oy * wo + ox] = s / (float)(kh * kw);
            }
    }
}

/* Converters.swift:578-605 (MPSCNNPoolingAverageNode kernel = input W x H) + GlobalPoolPadding.swift:16-26 */
void orc_global_avgpool(const float* x, float* y, int n, int c, int hw) {
#pragma omp parallel for schedule(static)
    for (long nc = 0; nc < (long)n * c; nc++) {
        const float* xp = x + (size_t)nc * hw; float s = 0.f;
        for (int j = 0; j < hw; j++) s += xp[j];
        y[nc] = s / (float)hw;
    }
}

/* Converters.swift:478-552: MPSCNNUpsamplingNearestNode / MPSCNNUpsamplingBilinearNode(alignCorners),
 * integer scale factors.  nearest: src = floor(o / s).  bilinear, alignCorners: src = o*(in-1)/(out-1);
 * otherwise half-pixel centres src = (o+0.5)/s - 0.5 clamped at 0. */
void orc_upsample(const float* x, float* y, int n, int c, int h, int w, int sy, int sx, int bilinear,
                  int align_corners) {
    int ho = h * sy, wo = w * sx;
#pragma omp parallel for schedule(static)
    for (long nc = 0; nc < (long)n * c; nc++) {
        const float* xp = x + (size_t)nc * h * w; float* yp = y + (size_t)nc * ho * wo;
        for (int oy = 0; oy < ho; oy++)
            for (int ox = 0; ox < wo; ox++) {
                if (!bilinear) { yp[(size_t)oy * wo + ox] = xp[(size_t)(oy / sy) * w + ox / sx]; continue; }
                float fy, fx;
                if (align_corners) {
                    fy = ho > 1 ? (float)oy * (float)(h - 1) / (float)(ho - 1) : 0.f;
                    fx = wo > 1 ? (float)ox * (float)(w - 1) / (float)(wo - 1) : 0.f;
                } else {
                    fy = ((float)oy + 0.5f) / (float)sy - 0.5f; if (fy < 0.f) fy = 0.f;
                    fx = ((float)ox + 0.5f) / (float)sx - 0.5f; if (fx < 0.f) fx = 0.f;
                }
                int y0 = (int)fy, x0 = (int)fx;
                if (y0 > h - 1) y0 = h - 1;
                if (x0 > w - 1) x0 = w - 1;
                int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
                float ly = fy - (float)y0, lx = fx - (float)x0;
                float v00 = xp[(size_t)y0 * w + x0], v01 = xp[(size_t)y0 * w + x1];
                float v10 = xp[(size_t)y1 * w + x0], v11 = xp[(size_t)y1 * w + x1];
                float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
                yp[(size_t)oy * wo + ox] = top + (bot - top) * ly;
            }
    }
}

/* Converters.swift:554-574 (MPSNNConcatenationNode): feature-channel concatenation of k operands */
void orc_concat_c(const float* const* xs, const int* cs, int k, float* y, int n, int hw) {
    int ctot = 0; for (int i = 0; i < k; i++) ctot += cs[i];
    for (int img = 0; img < n; img++) {
        int off = 0;
        for (int i = 0; i < k; i++) {
            memcpy(y + ((size_t)img * ctot + off) * hw, xs[i] + (size_t)img * cs[i] * hw, sizeof(float) * (size_t)cs[i] * hw);
            off += cs[i];
        }
    }
}
