// layout.hip — boundary layout / dtype conversion.
//
// In the reference flow the application hands MPS an MPSImage made from a texture and reads the result back
// with MPSImage.toFloatArray() (/root/reference/README.md:33-51,
// Sources/Smelter/Extensions/Foundation/MPSImage+Extensions.swift:9-59).  Here the application hands over an
// NCHW fp32 device tensor; internally everything is NHWC fp16.
#include "common.hpp"

namespace {

// NCHW fp32 -> NHWC fp16, channels zero-padded to `ld`, zero spatial halo.  One thread per output pixel:
// reads are coalesced along W inside each channel plane, writes are `ld` contiguous halfs per thread
// (8 or 16 bytes for the usual C=3 -> ld=4/8 image input).
__global__ void nchw_to_nhwc_kernel(const float* x, half_t* y, int N, int C, int H, int W, int ld, int ht, int hl, int hb, int hr) {
    const int Hp = H + ht + hb, Wp = W + hl + hr;
    const long total = (long)N * Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n = (int)(i / ((long)Hp * Wp)), r = (int)(i - (long)n * Hp * Wp), py = r / Wp, px = r - py * Wp;
        int iy = py - ht, ix = px - hl;
        bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const float* src = x + ((long)n * C * H + iy) * W + ix;
        half_t* dst = y + i * ld;
        for (int c0 = 0; c0 < ld; c0 += 8) {
            half_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { int c = c0 + j; v[j] = (in && c < C) ? (half_t)src[(long)c * H * W] : (half_t)0.f; }
            if (c0 + 8 <= ld && (ld & 7) == 0) *(half8_t*)(dst + c0) = *(half8_t*)v;
            else if (c0 + 4 <= ld && (ld & 3) == 0) { *(half4_t*)(dst + c0) = *(half4_t*)v; if (c0 + 8 <= ld) *(half4_t*)(dst + c0 + 4) = *(half4_t*)(v + 4); }
            else for (int j = 0; j < 8 && c0 + j < ld; j++) dst[c0 + j] = v[j];
        }
    }
}

// NCHW fp32 (C <= 4) -> "row-pair image" fp16 [N][Hs][Ws][8]: pixel (y, x) holds image pixel (y - pt, x - pl) in halfs 0..3
// and the pixel BELOW it, (y + 1 - pt, x - pl), in halfs 4..7 (zeros outside the image / beyond C).  64 consecutive halfs
// starting at (y, x) are then 8 pixels of kernel row kh and 8 pixels of kernel row kh + 1: one BK = 64 tile of the
// first-layer convolution (executor.cpp "image stem").  One thread per pixel, one 16-byte store.
// Grid: x = column blocks, y = groups of RP output rows, z = image.  A thread owns one column of RP consecutive rows: the
// bottom half of row r is the top half of row r + 1, so RP + 1 input rows are read for RP output pixels (no 64-bit
// index arithmetic, no division: the old flat grid-stride version spent 35 us on 92 MB).
constexpr int RP_ROWS = 4;
__global__ __launch_bounds__(256) void nchw_to_rowpair_kernel(const float* x, half_t* y, int N, int C, int H, int W, int Hs, int Ws, int pt, int pl) {
    const int px = blockIdx.x * 256 + threadIdx.x, py0 = blockIdx.y * RP_ROWS, n = blockIdx.z;
    if (px >= Ws) return;
    const int ix = px - pl;
    const bool colok = (unsigned)ix < (unsigned)W;
    const float* plane = x + (long)n * C * H * W + (colok ? ix : 0);
    const long cs = (long)H * W;
    half_t top[4] = {0, 0, 0, 0};
    {
        const int iy = py0 - pt;
        if (colok && (unsigned)iy < (unsigned)H)
#pragma unroll
            for (int c = 0; c < 4; c++) if (c < C) top[c] = (half_t)plane[(long)iy * W + c * cs];
    }
#pragma unroll
    for (int r = 0; r < RP_ROWS; r++) {
        const int py = py0 + r; if (py >= Hs) break;
        half_t bot[4] = {0, 0, 0, 0};
        const int iy = py - pt + 1;
        if (colok && (unsigned)iy < (unsigned)H)
#pragma unroll
            for (int c = 0; c < 4; c++) if (c < C) bot[c] = (half_t)plane[(long)iy * W + c * cs];
        half8_t v = {top[0], top[1], top[2], top[3], bot[0], bot[1], bot[2], bot[3]};
        *(half8_t*)(y + (((long)n * Hs + py) * Ws + px) * 8) = v;
#pragma unroll
        for (int c = 0; c < 4; c++) top[c] = bot[c];
    }
}

// Configuration.inputConstraint .forceInputScale (ONNXGraph.swift:219-241): resample every fp32 plane to the model's input size.
__device__ __forceinline__ float lanczos3(float t) {
    t = fabsf(t);
    if (t < 1e-6f) return 1.f;
    if (t >= 3.f) return 0.f;
    const float pt = 3.14159265358979f * t;
    return 3.f * sinf(pt) * sinf(pt / 3.f) / (pt * pt);
}
__global__ void resize_plane_kernel(const float* x, float* y, int planes, int H, int W, int Ho, int Wo, int mode) {
    const long total = (long)planes * Ho * Wo;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo); long t = i / Wo; const int oy = (int)(t % Ho); const long pl = t / Ho;
        const float* src = x + pl * H * W;
        const float cy = ((float)oy + 0.5f) * sy - 0.5f, cx = ((float)ox + 0.5f) * sx - 0.5f;
        float r;
        if (mode == SM_INPUT_FORCE_SCALE_BILINEAR) {
            const float fy = fminf(fmaxf(cy, 0.f), (float)(H - 1)), fx = fminf(fmaxf(cx, 0.f), (float)(W - 1));
            const int y0 = (int)fy, x0 = (int)fx, y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
            const float wy = fy - (float)y0, wx = fx - (float)x0;
            r = (src[(long)y0 * W + x0] * (1.f - wx) + src[(long)y0 * W + x1] * wx) * (1.f - wy) +
                (src[(long)y1 * W + x0] * (1.f - wx) + src[(long)y1 * W + x1] * wx) * wy;
        } else {
            const float fsy = fmaxf(sy, 1.f), fsx = fmaxf(sx, 1.f);                 // filter scale: wider when shrinking
            const int ry = (int)ceilf(3.f * fsy), rx = (int)ceilf(3.f * fsx);
            const int by = (int)floorf(cy), bx = (int)floorf(cx);
            float acc = 0.f, wsum = 0.f;
            for (int dy = -ry + 1; dy <= ry; dy++) {
                const float wy = lanczos3(((float)(by + dy) - cy) / fsy);
                if (wy == 0.f) continue;
                const int iy = min(max(by + dy, 0), H - 1);
                float racc = 0.f, rw = 0.f;
                for (int dx = -rx + 1; dx <= rx; dx++) {
                    const float wx = lanczos3(((float)(bx + dx) - cx) / fsx);
                    if (wx == 0.f) continue;
                    racc += wx * src[(long)iy * W + min(max(bx + dx, 0), W - 1)]; rw += wx;
                }
                acc += wy * racc; wsum += wy * rw;
            }
            r = acc / wsum;
        }
        y[i] = r;
    }
}

// NHWC fp16 -> NCHW fp32.  32x32 (pixel x channel) tile transposed through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const half_t* x, float* y, int HW, int C, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? (float)x[((long)n * HW + p) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) y[((long)n * C + c) * HW + p] = tile[tx][r];
    }
}

// NHWC fp16 -> MPSImage.toFloatArray() order: [image][slice = c/4][H][W][comps], comps = C < 3 ? C : 4,
// channels beyond C zero (MPSImage+Extensions.swift:27-41).
__global__ void nhwc_to_float_array_kernel(const half_t* x, float* y, int N, int C, int HW, int ld) {
    const int slices = (C + 3) / 4, comps = C < 3 ? C : 4;
    const long total = (long)N * slices * HW * comps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int k = (int)(i % comps); long t = i / comps;
        int p = (int)(t % HW); t /= HW;
        int sl = (int)(t % slices), n = (int)(t / slices);
        int c = sl * 4 + k;
        y[i] = c < C ? (float)x[((long)n * HW + p) * ld + c] : 0.f;
    }
}

// MPSNNReshapeNode (Converters.swift:859-864, :898-903) when the element order really changes: the reference's images keep the
// linear (c, y, x) element order across a reshape; ours are NHWC with a pixel stride, so destination element (p, c) of the
// [Co][HWo] view is source element lin = c*HWo + p -> (lin / HWs, lin % HWs) of the [Cs][HWs] view.  HBM-bound gather, one
// element per thread (reshapes on the hot path are aliases; this is the correctness path for spatial <-> channel reshapes).
__global__ void reshape_nchw_order_kernel(const half_t* x, half_t* y, int N, int Cs, int HWs, int x_ld, int Co, int HWo, int y_ld) {
    const long per = (long)Co * HWo, total = (long)N * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Co); long t = i / Co;
        const int p = (int)(t % HWo), n = (int)(t / HWo);
        const long lin = (long)c * HWo + p;
        const int cs = (int)(lin / HWs), ps = (int)(lin - (long)cs * HWs);
        y[((long)n * HWo + p) * y_ld + c] = x[((long)n * HWs + ps) * x_ld + cs];
    }
}

}  // namespace

extern "C" {

int sm_reshape_nchw_order(const void* x, void* y, int32_t N, int32_t Cs, int32_t HWs, int32_t x_ld, int32_t Co, int32_t HWo, int32_t y_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && Cs > 0 && HWs > 0 && Co > 0 && HWo > 0, "sm_reshape_nchw_order: bad argument");
    SM_REQUIRE((long)Cs * HWs == (long)Co * HWo, "sm_reshape_nchw_order: element count changes");
    x_ld = x_ld ? x_ld : Cs; y_ld = y_ld ? y_ld : Co;
    SM_REQUIRE(x_ld >= Cs && y_ld >= Co, "sm_reshape_nchw_order: pixel stride smaller than the channel count");
    hipLaunchKernelGGL(reshape_nchw_order_kernel, dim3(sm_stream_grid((long)N * Co * HWo, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, (half_t*)y, N, Cs, HWs, x_ld, Co, HWo, y_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_nchw_f32_to_nhwc_f16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t y_ld, int32_t ht, int32_t hl,
                            int32_t hb, int32_t hr, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && ht >= 0 && hl >= 0 && hb >= 0 && hr >= 0, "sm_nchw_f32_to_nhwc_f16: bad argument");
    y_ld = y_ld ? y_ld : C;
    SM_REQUIRE(y_ld >= C, "sm_nchw_f32_to_nhwc_f16: y_ld < C");
    long pix = (long)N * (H + ht + hb) * (W + hl + hr);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(sm_stream_grid(pix, 256)), dim3(256), 0, sm_s(st), x, (half_t*)y, N, C, H, W, y_ld, ht, hl, hb, hr);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_nchw_f32_to_rowpair_f16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Hs, int32_t Ws, int32_t pt, int32_t pl,
                               sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && pt >= 0 && pl >= 0, "sm_nchw_f32_to_rowpair_f16: bad argument");
    SM_REQUIRE((((uintptr_t)y) & 15) == 0, "sm_nchw_f32_to_rowpair_f16: destination not 16-byte aligned");
    SM_REQUIRE(N <= 65535 && (Hs + RP_ROWS - 1) / RP_ROWS <= 65535, "sm_nchw_f32_to_rowpair_f16: grid too large");
    hipLaunchKernelGGL(nchw_to_rowpair_kernel, dim3((Ws + 255) / 256, (Hs + RP_ROWS - 1) / RP_ROWS, N), dim3(256), 0, sm_s(st), x, (half_t*)y, N, C, H, W, Hs, Ws, pt, pl);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_resize_nchw_f32(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t mode, sm_stream_t st) {
    SM_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && (mode == SM_INPUT_FORCE_SCALE_BILINEAR || mode == SM_INPUT_FORCE_SCALE_LANCZOS),
               "sm_resize_nchw_f32: bad argument");
    hipLaunchKernelGGL(resize_plane_kernel, dim3(sm_stream_grid((long)planes * Ho * Wo, 256)), dim3(256), 0, sm_s(st), x, y, planes, H, W, Ho, Wo, mode);
    SM_LAUNCH_CHECK(); return SM_OK;
}

int sm_nhwc_f16_to_nchw_f32(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t x_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "sm_nhwc_f16_to_nchw_f32: bad argument");
    x_ld = x_ld ? x_ld : C;
    SM_REQUIRE(N <= 65535 && (C + 31) / 32 <= 65535, "sm_nhwc_f16_to_nchw_f32: grid too large");
    dim3 g((H * W + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, g, dim3(256), 0, sm_s(st), (const half_t*)x, y, H * W, C, x_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

size_t sm_float_array_count(int32_t N, int32_t C, int32_t H, int32_t W) {
    int slices = (C + 3) / 4, cpp = C < 3 ? C : slices * 4;
    return (size_t)W * H * cpp * N;
}

int sm_nhwc_f16_to_float_array(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t x_ld, sm_stream_t st) {
    SM_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "sm_nhwc_f16_to_float_array: bad argument");
    x_ld = x_ld ? x_ld : C;
    long total = (long)sm_float_array_count(N, C, H, W);
    hipLaunchKernelGGL(nhwc_to_float_array_kernel, dim3(sm_stream_grid(total, 256)), dim3(256), 0, sm_s(st), (const half_t*)x, y, N, C, H * W, x_ld);
    SM_LAUNCH_CHECK(); return SM_OK;
}

}  // extern "C"
