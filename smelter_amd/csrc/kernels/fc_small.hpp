// fc_small.hpp — fully connected layers on FEW rows (MPSCNNFullyConnectedNode, /root/reference/Sources/Smelter/Converters.swift:297; a classifier behind a global pool: M = batch
// rows, e.g. ResNet-50's 64 x 2048 -> 1000).  Included by conv_igemm.hip; takes the launch before the tiled MFMA paths (no tile configuration).
//
// The tiled kernels cut such a layer into <= 8 pixel tiles x N tiles and make up for the missing parallelism with split-K + a reduce launch (ResNet-50: 9.7 + 4.4 us for
// 0.26 GFLOP).  What bounds the layer is how fast a CU can pull its operands: here a block is 16 outputs x 16 rows, so the grid is (N / 16) x (M / 16) = 252 blocks for
// the ResNet-50 head and a block reads 64 KB of weights + 64 KB of activations.  K is split over the block's four waves; both MFMA operands (v_mfma_f32_16x16x32_f16: four lanes
// per matrix row = 64 contiguous bytes per row and instruction) come straight from global memory, every load of a wave is in flight before its first MFMA.  The four partial
// sums meet in LDS and are added in a FIXED order (wave 0 + 1 + 2 + 3: the same bits on every run), + bias in fp32, activation, one rounding.
// With more rows (MobileNet-v2: 256 x 1280 -> 1000) a block takes RG groups of 16 rows against the same 16 weight rows (RG = 2, 4: the grid stays near one block per CU and
// the weights are pulled M / (16 RG) times instead of M / 16).
#pragma once

template <int RG>
__global__ __launch_bounds__(256) void fc_small_kernel(ConvParams p) {
    __shared__ float part[4][RG][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const int ot = blockIdx.x % p.nt, mt = blockIdx.x / p.nt;       // neighbouring blocks share the activation rows (L2)
    const int o0 = ot * 16, m0 = mt * 16 * RG;
    const int ksteps = p.K >> 5, k_lo = (int)((long)wave * ksteps / 4), k_hi = (int)((long)(wave + 1) * ksteps / 4);
    const int orow = min(o0 + r16, p.Cout - 1);                     // rows beyond the edge: clamped (valid addresses), their results are dropped
    const half_t* wp = p.w + (long)orow * p.K + kq * 8;
    const half_t* xp[RG];
#pragma unroll
    for (int g = 0; g < RG; g++) xp[g] = p.x + (long)min(m0 + 16 * g + r16, p.M - 1) * p.x_ld + kq * 8;
    floatx4 acc[RG];
#pragma unroll
    for (int g = 0; g < RG; g++) acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};
    constexpr int UN = RG == 1 ? 8 : 4;                              // k steps per batch of loads (16 - 20 loads in flight per lane)
    for (int k0 = k_lo; k0 < k_hi; k0 += UN) {
        half8_t a[UN], b[UN][RG];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int ks = min(k0 + u, k_hi - 1);                    // (a ragged tail re-reads the last step; its product is not added)
            a[u] = *(const half8_t*)(wp + ks * 32);
#pragma unroll
            for (int g = 0; g < RG; g++) b[u][g] = *(const half8_t*)(xp[g] + ks * 32);
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (k0 + u < k_hi) {
#pragma unroll
                for (int g = 0; g < RG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u][g], acc[g], 0, 0, 0);
            }
    }
    // D[out row 4 (lane >> 4) + r][image column lane & 15]
#pragma unroll
    for (int g = 0; g < RG; g++) *(floatx4*)&part[wave][g][lane][0] = acc[g];
    __syncthreads();
    if (wave < RG) {                                                  // wave g adds group g's four partial sums (RG <= 4)
        const int g = wave;
        floatx4 s = *(const floatx4*)&part[0][g][lane][0];
#pragma unroll
        for (int w = 1; w < 4; w++) { const floatx4 t = *(const floatx4*)&part[w][g][lane][0]; s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3]; }
        const int o = o0 + 4 * kq, m = m0 + 16 * g + r16;
        if (m < p.M && o < p.Cout) {                                  // (Cout % 4 == 0: launcher)
            const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
            half4_t out;
#pragma unroll
            for (int r = 0; r < 4; r++) { float v = s[r] + (p.bias ? p.bias[o + r] : 0.f); out[r] = (half_t)(v < 0.f ? v * sl : v); }
            *(half4_t*)(p.y + (long)m * p.y_ld + o) = out;
        }
    }
}

static bool fc_small_eligible(const sm_conv_desc* d, const ConvParams& p) {
    if (d->groups != 1 || p.kH != 1 || p.kW != 1 || p.H != 1 || p.W != 1 || p.Ho != 1 || p.Wo != 1 || p.pt || p.pl || p.yplace || p.res || p.ups || p.x2) return false;
    if (p.M > 256 || p.K < 256 || (p.K & 31) || (p.Cout & 3) || p.act > SM_ACT_PRELU) return false;
    if ((p.x_ld & 7) || (p.y_ld & 3) || (((uintptr_t)p.x) & 15) || (((uintptr_t)p.w) & 15) || (((uintptr_t)p.y) & 7)) return false;
    return (long)sm_ceil_div(p.Cout, 16) * sm_ceil_div(p.M, 16) < (1L << 30);
}
static void fc_small_launch(ConvParams& p, hipStream_t s) {
    p.nt = sm_ceil_div(p.Cout, 16);
    const int mt16 = sm_ceil_div(p.M, 16);
    const int rg = ((long)p.nt * mt16 <= 384 || mt16 < 2) ? 1 : ((long)p.nt * mt16 <= 768 || mt16 < 4) ? 2 : 4;      // the largest grid that stays near one block per CU
    const dim3 grid((unsigned)(p.nt * sm_ceil_div(p.M, 16 * rg)));
    if (rg == 1) hipLaunchKernelGGL(fc_small_kernel<1>, grid, dim3(256), 0, s, p);
    else if (rg == 2) hipLaunchKernelGGL(fc_small_kernel<2>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(fc_small_kernel<4>, grid, dim3(256), 0, s, p);
}
