// pw_thin.hpp — thin pointwise convolution: 1x1 / stride 1 / no padding with a SHORT reduction (Cin <= 192, Cin % 64 != 0)
// over a huge pixel count: the expand / project 1x1 layers of MobileNet-v2 (16->96, 24->144, 32->192, 96->24, 144->32 ...).
// Included by conv_igemm.hip.  These layers are pure HBM streaming (<= 150 FLOP per output byte is never the limit); in the
// tiled implicit-GEMM kernels every 64..128-pixel tile pays a whole block prologue + epilogue for ONE K tile (2.1..3.1 TB/s).
//
// Here nothing is tiled over K and there is no block-wide synchronisation after the weights are staged:
//  * the whole weight matrix [Cout][Cin] (<= 48 KB) is copied once per block into LDS, rows padded to a multiple of 16 halfs (+16 B);
//  * a WAVE owns 32 consecutive pixels at a time (grid-stride): its MFMA B-operand fragments are loaded straight from global
//    memory — lane (pixel = lane & 31, k half = lane >> 5) reads 16 bytes at x[pixel][16 ks + 8 half], i.e. for dense rows the
//    wave reads ONE contiguous run of 32 * Cin halfs; all K/16 fragments stay in registers;
//  * per pass of 64 output channels: 2 x K/16 v_mfma_f32_32x32x16_f16 with the weight fragments read from LDS, then the
//    accumulators go through a wave-private fp32 LDS tile [32 pixels][64+4] so that each lane leaves with 8 consecutive
//    channels of one pixel: + bias + residual, activation, one rounding, 16-byte store — 8 lanes per 128-byte run;
//  * same numerics as the tiled kernels: fp32 accumulation over k in ascending order, bias and residual added in fp32.
#pragma once

constexpr int PWT_WAVES = 4;                                    // waves per block (small blocks: several per CU hide each other's latency)
constexpr int PWT_STG_LD = 68;                                  // floats per staged pixel row (64 + 4: conflict-free b128 writes)
constexpr int PWT_STG_BYTES = 32 * PWT_STG_LD * 4;              // per wave
constexpr int PWT_KSMAX = 12;                                   // Cin <= 192

// KSN = k steps of 16 (compile time: the fragments live in registers); PF = prefetch the next pixel block's fragments while this one
// is multiplied and stored (costs KSN*4 more VGPRs)
template <int KSN, int PF>
__global__ __launch_bounds__(PWT_WAVES * 64) void pw_thin_kernel(ConvParams p, int kp, int wrow_bytes, int nblocks32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    unsigned char* wl = pw_smem + PWT_WAVES * PWT_STG_BYTES;    // weights [Cout rounded up to 64][wrow_bytes], then the bias [cout64] fp32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.Cin;                                        // kp = Cin rounded up to 16 = 16 * KSN
    const int cout64 = (p.Cout + 63) & ~63;

    // ---- weights -> LDS, zero-padded in k (to kp) and in cout (to a multiple of 64) -------------------------------------
    {
        const int chunks = kp >> 3;                             // 16-byte chunks per row
        for (int i = tid; i < cout64 * chunks; i += PWT_WAVES * 64) {
            const int n = i / chunks, c = i - n * chunks;
            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (n < p.Cout && c * 8 < K) v = *(const half8_t*)(p.w + (long)n * K + c * 8);   // Cin % 8 == 0: whole chunks
            *(half8_t*)(wl + n * wrow_bytes + c * 16) = v;
        }
    }
    // bias -> LDS as well: an ordinary load inside the pixel loop would make every pass wait (vmcnt is in order) for the stores
    // of the previous pass and for the prefetch of the next pixel block
    float* bl = (float*)(wl + cout64 * wrow_bytes);
    for (int i = tid; i < cout64; i += PWT_WAVES * 64) bl[i] = (p.bias && i < p.Cout) ? p.bias[i] : 0.f;
    __syncthreads();

    float* stg = (float*)(pw_smem + wave * PWT_STG_BYTES);
    const int px = lane & 31, kh = lane >> 5;
    const bool relu_family = p.act <= SM_ACT_PRELU;
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    const int passes = cout64 >> 6;
    const int gw = blockIdx.x * PWT_WAVES + wave, gstride = gridDim.x * PWT_WAVES;
    // activation fragments of 32 pixels: all k steps, straight from global (zero beyond Cin; rows beyond M are clamped, never stored)
    auto load_frags = [&](int blk, half8_t (&xf)[KSN]) {
        const int m = blk * 32 + px;
        const half_t* row = p.x + (long)(m < p.M ? m : p.M - 1) * p.x_ld;
#pragma unroll
        for (int ks = 0; ks < KSN; ks++) {
            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ks * 16 + kh * 8 < K) v = *(const half8_t*)(row + ks * 16 + kh * 8);
            xf[ks] = v;
        }
    };
    half8_t xa[KSN], xn[KSN];                                  // (xn is dead when PF == 0)
    if (gw < nblocks32) load_frags(gw, xa);
    for (int blk = gw; blk < nblocks32; blk += gstride) {
        const int m0 = blk * 32;
        if (PF && blk + gstride < nblocks32) load_frags(blk + gstride, xn);
        for (int ps = 0; ps < passes; ps++) {
            floatx16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const unsigned char* w0 = wl + (ps * 64 + px) * wrow_bytes + kh * 16;
            const unsigned char* w1 = w0 + 32 * wrow_bytes;
#pragma unroll
            for (int ks = 0; ks < KSN; ks++) {
                const half8_t f0 = *(const half8_t*)(w0 + ks * 32), f1 = *(const half8_t*)(w1 + ks * 32);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, xa[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, xa[ks], acc1, 0, 0, 0);
            }
            // accumulators (cout x pixel) -> wave-private LDS [pixel][64 couts]: register 4q+r <-> cout 8q + 4*kh + r of its block
#pragma unroll
            for (int q = 0; q < 4; q++) {
                *(floatx4*)(stg + px * PWT_STG_LD + 8 * q + 4 * kh) = floatx4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
                *(floatx4*)(stg + px * PWT_STG_LD + 32 + 8 * q + 4 * kh) = floatx4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private tile: the wave's own writes are ordered, no barrier
            const int ch = lane & 7, co = ps * 64 + ch * 8;
            if (co < p.Cout) {
                const floatx4 b0 = *(const floatx4*)(bl + co), b1 = *(const floatx4*)(bl + co + 4);
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int r = it * 8 + (lane >> 3), m = m0 + r;
                    const floatx4 s0 = *(const floatx4*)(stg + r * PWT_STG_LD + ch * 8), s1 = *(const floatx4*)(stg + r * PWT_STG_LD + ch * 8 + 4);
                    float v[8] = {s0[0] + b0[0], s0[1] + b0[1], s0[2] + b0[2], s0[3] + b0[3], s1[0] + b1[0], s1[1] + b1[1], s1[2] + b1[2], s1[3] + b1[3]};
                    if (m < p.M) {
                        if (p.res) { const half8_t rr = *(const half8_t*)(p.res + (long)m * p.res_ld + co);
#pragma unroll
                            for (int j = 0; j < 8; j++) v[j] += (float)rr[j]; }
                        half8_t o;
                        if (relu_family) {
#pragma unroll
                            for (int j = 0; j < 8; j++) o[j] = (half_t)(v[j] < 0.f ? v[j] * sl : v[j]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; j++) o[j] = (half_t)sm_act(v[j], p.act, p.slope);
                        }
                        *(half8_t*)(p.y + (long)m * p.y_ld + co) = o;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staging reads done before the next pass overwrites the tile
        }
        if (PF) {
#pragma unroll
            for (int ks = 0; ks < KSN; ks++) xa[ks] = xn[ks];
        } else if (blk + gstride < nblocks32) load_frags(blk + gstride, xa);
    }
}

// eligibility + launch (called from launch() before the tiled MFMA paths)
static int pw_thin_ksn(int cin) { const int k = (cin + 15) >> 4; return k <= 1 ? 1 : k <= 2 ? 2 : k <= 4 ? 4 : k <= 6 ? 6 : k <= 9 ? 9 : 12; }   // instantiated k-step counts
static bool pw_thin_eligible(const sm_conv_desc* d, const ConvParams& p) {
    static const bool no_thin = SM_ABL_ENV("SMELTER_NO_PW_THIN") != nullptr, thin_res = SM_ABL_ENV("SMELTER_PW_THIN_RES") != nullptr;   // experiment switches, read once
    if (no_thin) return false;
    if (d->groups != 1 || p.kH != 1 || p.kW != 1 || p.sH != 1 || p.sW != 1 || p.pt != 0 || p.pl != 0 || p.Ho != p.H || p.Wo != p.W) return false;
    static const bool k64 = SM_ABL_ENV("SMELTER_PW_THIN_K64") != nullptr;      // experiment: also take Cin == 64 (ResNet-50 layer-1 1x1 convs)
    if (p.yplace || (p.Cin & 7) || ((p.Cin % 64) == 0 && !(k64 && p.Cin == 64)) || p.Cin > 16 * PWT_KSMAX || (p.Cout & 7)) return false;
    if ((p.x_ld & 7) || (p.y_ld & 7) || (((uintptr_t)p.x) & 15) || (((uintptr_t)p.y) & 15) || (((uintptr_t)p.w) & 15) || (((uintptr_t)p.bias) & 15)) return false;
    if (p.res && ((p.res_ld & 7) || (((uintptr_t)p.res) & 15))) return false;
    // The kernel handles a residual operand (parity-tested), but its ordinary loads sit in vmcnt order between the stores of one pass
    // and the staging of the next: measured 95 vs 83 us on MobileNet-v2's 144->24 +res layer, so the graph path leaves those tiled.
    if (p.res && !thin_res) return false;
    const int kp = 16 * pw_thin_ksn(p.Cin), cout64 = (p.Cout + 63) & ~63;
    if ((size_t)cout64 * (kp * 2 + 16 + 4) > 40 * 1024) return false;        // weights + bias + 4 staging tiles <= 75 KB: at least two blocks per CU
    return p.M >= 8192;                                                        // a streaming kernel: small maps stay on the tiled path
}
static void pw_thin_launch(ConvParams& p, hipStream_t s) {
    const int ksn = pw_thin_ksn(p.Cin), kp = 16 * ksn, cout64 = (p.Cout + 63) & ~63, wrow = kp * 2 + 16;
    const size_t lds = (size_t)PWT_WAVES * PWT_STG_BYTES + (size_t)cout64 * (wrow + 4);
    const int nblocks32 = sm_ceil_div(p.M, 32);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(156 * 1024) / lds));
    const dim3 grid(std::min(sm_ceil_div(nblocks32, PWT_WAVES), SM_NUM_CU * per_cu)), blk(PWT_WAVES * 64);
    if (ksn == 1) hipLaunchKernelGGL((pw_thin_kernel<1, 1>), grid, blk, lds, s, p, kp, wrow, nblocks32);
    else if (ksn == 2) hipLaunchKernelGGL((pw_thin_kernel<2, 1>), grid, blk, lds, s, p, kp, wrow, nblocks32);
    else if (ksn == 4) hipLaunchKernelGGL((pw_thin_kernel<4, 1>), grid, blk, lds, s, p, kp, wrow, nblocks32);
    else if (ksn == 6) hipLaunchKernelGGL((pw_thin_kernel<6, 1>), grid, blk, lds, s, p, kp, wrow, nblocks32);
    else if (ksn == 9) hipLaunchKernelGGL((pw_thin_kernel<9, 0>), grid, blk, lds, s, p, kp, wrow, nblocks32);
    else hipLaunchKernelGGL((pw_thin_kernel<12, 0>), grid, blk, lds, s, p, kp, wrow, nblocks32);
}
