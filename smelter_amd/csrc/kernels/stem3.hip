// stem3.hip — 3x3 image stem for gfx950: fp32 NCHW image with C <= 3 planes -> 3x3 convolution (stride 1 | 2, symmetric padding 0 | 1) + bias + ReLU-family
// activation -> NHWC fp16, ONE launch, no converted copy of the image.  Replaces the first `MPSCNNConvolutionNode` (Converters.swift:253) of a graph whose input is an
// image (U-Net 512^2 bs 16: 1 -> 32 channels, 4.2 M pixels).
//
// The layer is 27 multiply-adds per output at most: its time is the OUTPUT write (64 bytes per pixel at 32 channels), so the kernel is built to do nothing else that
// shows.  A block of eight waves owns a block tile of 64 (stride 2: 32) output rows x 32 columns — large, because a block tile is one memory round trip + two barriers
// whatever its size; the threads copy the input region under it, element by element in row order, into the LDS as fp16 PIXEL SLOTS of 8 bytes (c0, c1, c2, -): the three taps of a kernel row are 24 contiguous bytes, and the B fragment of 32 output
// pixels of a row is ONE ds_read2_b64 per k step at a lane-constant address (the layout mbconv5s_kernel uses, mbconv.hip).  K = (kh, kw in 0..3, c in 0..3) = 48 =
// three k steps of v_mfma_f32_32x32x16_f16 with zeros in the weight fragment at the padding positions; the fourth "kw" slot of kernel row 1 is overwritten in registers
// with (1, 1): the two bias columns (hi(b), lo(b) as two halfs: the fp32 bias enters through the MFMA).  D = W x X^T has lane = pixel, registers = channels: the
// epilogue is the inverted-residual kernels' (v_permlane32_swap -> 8 consecutive channels per lane -> 16-byte stores).  Two blocks per CU: one stages while the other
// computes and stores.  Numerics: products of fp16 values accumulated in fp32, one rounding to fp16 — those of the generic kernels (the fp32 image is rounded to fp16
// first, as the converted copy was).
#include <algorithm>

#include "common.hpp"

namespace {

struct Stem3Params {
    const float* x; const half_t* w; const float* bias; half_t* y;
    int N, C, in_H, in_W, Ho, Wo, Cout, s, pad, y_ld; float slope, hi;   // slope: 0 = ReLU, 1 = none, else leaky; hi: upper bound (+inf but for SM_ACT_CLIP: ONNX Clip(0, hi), ReLU6)
    int tiles_x, tiles_y, ntiles, IR, IC, pitch, nelem;              // block tiles; input region (rows, columns), LDS row pitch in pixel slots, C * IR * IC
    unsigned tx_mul, tx_sh, txy_mul, txy_sh, ic_mul, ic_sh, pl_mul, pl_sh;   // fastdiv by tiles_x, tiles_x * tiles_y, IC, IR * IC
};
inline void s3_fastdiv_make(unsigned d, unsigned* mul, unsigned* sh) { unsigned s = 0; while ((1ull << s) < d) s++; *mul = (unsigned)(((1ull << (31 + s)) / d) + 1); *sh = 31 + s; }
__device__ __forceinline__ unsigned s3_fastdiv(unsigned n, unsigned mul, unsigned sh) { return (unsigned)(((unsigned long long)n * mul) >> sh); }

constexpr int S3_COLS = 32, S3_NW = 8;                               // block tile = (8 WR) x 32 output pixels: eight waves of WR rows each

template <int NF, int WR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem3_kernel(Stem3Params p) {   // (128 registers: two blocks per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char s3_smem[];
    constexpr int KP = 48, NT = S3_NW * 64;
    half_t* const wl = (half_t*)s3_smem;                                 // [NF * 32][48]: k = 16 kh + 4 kw + c; (kh 1, kw 3, c 0 | 1) = hi(b), lo(b)
    unsigned char* const reg = (unsigned char*)(wl + NF * 32 * KP);      // input region: [IR][pitch] pixel slots of 8 bytes
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pitch = p.pitch, IR = p.IR, IC = p.IC, st = p.s;

    for (int i = tid; i < NF * 32 * KP; i += NT) {
        const int n = i / KP, k = i - n * KP, kh = k >> 4, kw = (k >> 2) & 3, c = k & 3;
        half_t v = (half_t)0.f;
        if (n < p.Cout) {
            if (kw < 3 && c < p.C) v = p.w[((long)n * 9 + kh * 3 + kw) * p.C + c];
            else if (kh == 1 && kw == 3 && c < 2 && p.bias) { const float b = p.bias[n]; const half_t bh = (half_t)b; v = c == 0 ? bh : (half_t)(b - (float)bh); }
        }
        wl[i] = v;
    }
    for (int i = tid; i < IR * pitch * 2; i += NT) ((unsigned*)reg)[i] = 0u;              // (fourth halfs, planes >= C and the spare columns are never written again)
    __syncthreads();
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    half8_t fw[NF][3];                                                   // the weights stay in registers for the whole kernel
#pragma unroll
    for (int a = 0; a < NF; a++)
#pragma unroll
        for (int ks = 0; ks < 3; ks++) fw[a][ks] = *(const half8_t*)(wl + (a * 32 + l31) * KP + ks * 16 + hi * 8);
    unsigned boff[3];                                                    // byte offset in `reg` of the lane's two slots (kw = 2 hi, 2 hi + 1) of kernel row kh for the wave's first output row
#pragma unroll
    for (int kh = 0; kh < 3; kh++) boff[kh] = (unsigned)((((wave * WR) * st + kh) * pitch + l31 * st + 2 * hi) * 8);
    const unsigned rowstep = (unsigned)(st * pitch * 8);
    const unsigned plane = (unsigned)p.in_H * (unsigned)p.in_W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.N * (unsigned)p.C * plane * 4u), 0x00020000);

#pragma unroll 1
    for (int bt = blockIdx.x; bt < p.ntiles; bt += gridDim.x) {
        const int n = (int)s3_fastdiv((unsigned)bt, p.txy_mul, p.txy_sh);
        const int r0 = bt - n * (p.tiles_x * p.tiles_y), by = (int)s3_fastdiv((unsigned)r0, p.tx_mul, p.tx_sh), bx = r0 - by * p.tiles_x;
        const int iry0 = by * (S3_NW * WR) * st - p.pad, irx0 = bx * S3_COLS * st - p.pad;    // input pixel of the region's corner
        // ---- stage the input region: element e = (plane, row, column) in row order, thread t takes t, t + 512, ...; eight loads in flight per lane (sixteen spill at 128 registers: 82 -> 101 us) ---------------------
#pragma unroll 1
        for (int e0 = tid; e0 < p.nelem; e0 += 8 * NT) {
            float g[8]; unsigned dst[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned e = (unsigned)(e0 + i * NT), c = s3_fastdiv(e, p.pl_mul, p.pl_sh), rem = e - c * (unsigned)(IR * IC), rr = s3_fastdiv(rem, p.ic_mul, p.ic_sh), x = rem - rr * (unsigned)IC;
                const int gy = iry0 + (int)rr, gx = irx0 + (int)x;
                const bool ok = e < (unsigned)p.nelem && (unsigned)gy < (unsigned)p.in_H && (unsigned)gx < (unsigned)p.in_W;
                const unsigned off = ok ? ((((unsigned)n * (unsigned)p.C + c) * (unsigned)p.in_H + (unsigned)gy) * (unsigned)p.in_W + (unsigned)gx) * 4u : 0x80000000u;
                g[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, SM_STEM_IN_AUX));
                dst[i] = ((rr * (unsigned)pitch + x) * 4u + c) * 2u;
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (e0 + i * NT < p.nelem) *(half_t*)(reg + dst[i]) = (half_t)g[i];
        }
        __syncthreads();
        // ---- WR output rows of 32 pixels per wave: three MFMAs per cout block, epilogue, 16-byte stores ----------------------------------------------------------
        const int ox = bx * S3_COLS + l31;
#pragma unroll 2
        for (int r = 0; r < WR; r++) {
            const int oy = by * (S3_NW * WR) + wave * WR + r;
            floatx16 acc[NF];
#pragma unroll
            for (int a = 0; a < NF; a++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[a][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                const u32x2* const sl = (const u32x2*)(reg + boff[ks] + (unsigned)r * rowstep);                 // two 8-byte slots: one ds_read2_b64
                const u32x2 s0v = sl[0], s1v = sl[1];
                u32x4 d = {s0v[0], s0v[1], s1v[0], s1v[1]};
                if (hi) { d[2] = ks == 1 ? 0x3c003c00u : 0u; d[3] = 0u; }                  // the fourth "kw" slot: (1, 1) against the bias columns in kernel row 1, zeros elsewhere
#pragma unroll
                for (int a = 0; a < NF; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[a][ks], __builtin_bit_cast(half8_t, d), acc[a], 0, 0, 0);
            }
            const bool pin = oy < p.Ho && ox < p.Wo;
            const long opix = ((long)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
            for (int a = 0; a < NF; a++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float o[8];
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) {
                        const float fa = acc[a][4 * (2 * j) + r2], fb = acc[a][4 * (2 * j + 1) + r2];   // (a copy first: __builtin_bit_cast of a vector ELEMENT lvalue read element 0)
                        unsigned ua = __builtin_bit_cast(unsigned, fa), ub = __builtin_bit_cast(unsigned, fb);
                        sm_permlane32_swap(ua, ub);
                        o[r2] = __builtin_bit_cast(float, ua); o[4 + r2] = __builtin_bit_cast(float, ub);
                    }
                    const int co = a * 32 + 16 * j + 8 * hi;
                    if (pin && co < p.Cout) {
                        half8_t ov;
                        if (p.slope == 0.f) {
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)o[r2];
                            ov = __builtin_elementwise_max(ov, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
                            if (p.hi != INFINITY) { const half_t hh = (half_t)p.hi; ov = __builtin_elementwise_min(ov, half8_t{hh, hh, hh, hh, hh, hh, hh, hh}); }   // (min commutes with the rounding: it is monotone)
                        } else {
#pragma unroll
                            for (int r2 = 0; r2 < 8; r2++) ov[r2] = (half_t)(o[r2] < 0.f ? o[r2] * p.slope : o[r2]);
                        }
                        *(half8_t*)(p.y + opix * p.y_ld + co) = ov;
                    }
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // every wave has read its fragments: the region may be overwritten (stores stay in flight)
        asm volatile("" ::: "memory");
    }
}

int s3_wr(int s) { return s == 1 ? 8 : 4; }                          // output rows per wave: block tiles of 64 x 32 (stride 2: 32 x 32) output pixels
void s3_geometry(int s, int* IR, int* IC, int* pitch) {
    *IR = (S3_NW * s3_wr(s) - 1) * s + 3; *IC = (S3_COLS - 1) * s + 3; *pitch = (*IC + 3) & ~1;    // (one spare slot behind the last column: the fourth "kw" of its pixel)
}
size_t s3_lds(int Cout, int s) {
    int IR, IC, pitch; s3_geometry(s, &IR, &IC, &pitch);
    return (size_t)((Cout + 31) / 32) * 32 * 48 * 2 + (size_t)IR * pitch * 8;
}

}  // namespace

extern "C" {

int sm_stem3x3_f32_supported(int32_t N, int32_t C, int32_t in_H, int32_t in_W, int32_t Cout, int32_t stride, int32_t pad) {
    if (N <= 0 || C < 1 || C > 3 || in_H <= 0 || in_W <= 0 || Cout <= 0 || Cout > 64 || (Cout & 7) || stride < 1 || stride > 2 || pad < 0 || pad > 1) return 0;
    if (in_H + 2 * pad < 3 || in_W + 2 * pad < 3) return 0;
    if ((long)N * C * in_H * in_W * 4 >= 0x7fff0000L) return 0;          // (one buffer descriptor over the image batch)
    const long Ho = (in_H + 2 * pad - 3) / stride + 1, Wo = (in_W + 2 * pad - 3) / stride + 1;
    return (long)N * Ho * Wo < (1L << 31) && s3_lds(Cout, stride) <= 80 * 1024;
}

// y [N, Ho, Wo, Cout] fp16 (pixel stride y_ld halfs, 16-byte aligned, y_ld % 8 == 0) = act(conv3x3(x_nchw fp32 [N, C, in_H, in_W], w [Cout][3][3][C] fp16 (OHWI), stride,
// padding pad on all four sides) + bias); act in the ReLU family (SM_ACT_NONE | SM_ACT_RELU | leaky with `slope`) or SM_ACT_CLIP (min(max(x, 0), slope)).
int sm_stem3x3_f32_fwd(const float* x_nchw, int32_t N, int32_t C, int32_t in_H, int32_t in_W, const void* w_ohwi, const float* bias, int32_t Cout, int32_t stride, int32_t pad,
                       int32_t act, float slope, void* y, int32_t y_ld, sm_stream_t stream) {
    SM_REQUIRE(x_nchw && w_ohwi && y && sm_stem3x3_f32_supported(N, C, in_H, in_W, Cout, stride, pad) && (act <= SM_ACT_PRELU || act == SM_ACT_CLIP), "sm_stem3x3_f32_fwd: unsupported geometry");
    Stem3Params p{};
    p.x = x_nchw; p.w = (const half_t*)w_ohwi; p.bias = bias; p.y = (half_t*)y;
    p.N = N; p.C = C; p.in_H = in_H; p.in_W = in_W; p.Cout = Cout; p.s = stride; p.pad = pad; p.y_ld = y_ld ? y_ld : Cout;
    p.Ho = (in_H + 2 * pad - 3) / stride + 1; p.Wo = (in_W + 2 * pad - 3) / stride + 1;
    SM_REQUIRE((p.y_ld & 7) == 0 && (((uintptr_t)y) & 15) == 0, "sm_stem3x3_f32_fwd: output must be 16-byte aligned, pixel stride % 8 == 0");
    p.slope = act == SM_ACT_NONE ? 1.f : (act == SM_ACT_RELU || act == SM_ACT_CLIP) ? 0.f : slope; p.hi = act == SM_ACT_CLIP ? slope : INFINITY;
    s3_geometry(stride, &p.IR, &p.IC, &p.pitch); p.nelem = C * p.IR * p.IC;
    p.tiles_x = sm_ceil_div(p.Wo, S3_COLS); p.tiles_y = sm_ceil_div(p.Ho, S3_NW * s3_wr(stride));
    const long nbt = (long)N * p.tiles_x * p.tiles_y;
    SM_REQUIRE(nbt < (1L << 31), "sm_stem3x3_f32_fwd: too many tiles");
    p.ntiles = (int)nbt;
    s3_fastdiv_make((unsigned)p.tiles_x, &p.tx_mul, &p.tx_sh); s3_fastdiv_make((unsigned)(p.tiles_x * p.tiles_y), &p.txy_mul, &p.txy_sh);
    s3_fastdiv_make((unsigned)p.IC, &p.ic_mul, &p.ic_sh); s3_fastdiv_make((unsigned)(p.IR * p.IC), &p.pl_mul, &p.pl_sh);
    const unsigned grid = (unsigned)std::min<long>(nbt, 2L * SM_NUM_CU);
    void (*fn)(Stem3Params) = stride == 1 ? (Cout <= 32 ? stem3_kernel<1, 8> : stem3_kernel<2, 8>) : (Cout <= 32 ? stem3_kernel<1, 4> : stem3_kernel<2, 4>);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), s3_lds(Cout, stride), sm_s(stream), p);
    SM_LAUNCH_CHECK(); return SM_OK;
}

}  // extern "C"
