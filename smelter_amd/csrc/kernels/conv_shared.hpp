// conv_shared.hpp — what the convolution translation units share: the launch parameter block, division by launch constants, the LDS-DMA /
// LDS-address helpers and the compile-time loop.  conv_igemm.hip holds the launcher and most kernels; conv_kq.hip the wave-split 3x3 kernel
// (its own translation unit: a one-kernel change rebuilds in seconds instead of the minutes conv_igemm.hip takes).
#pragma once
#include <type_traits>
#include "common.hpp"

struct ConvParams {
    const half_t* x; const half_t* w; const float* bias; const half_t* res; half_t* y;
    float* ws;                       // split-K partials (fp32) or nullptr
    int N, H, W, Cin, Cout, kH, kW, sH, sW, dH, dW, pt, pl, Ho, Wo;
    int x_ld, y_ld, res_ld, act; float slope;
    int M, K, mt, nt, splitk, ktiles_per_split;
    int order, mt_per_xcd, cpx;      // block -> tile map (map_block)
    unsigned dhw_mul, dhw_sh, dw_mul, dw_sh;   // sm_fastdiv constants for / (Ho*Wo) and / Wo
    unsigned pm_mul, pm_sh, mt_mul, mt_sh, nt_mul, nt_sh;   // ... for / (nt*splitk), / mt, / nt (map_block)
    unsigned rowpat;                 // sum over kh < kH of 1 << (kh*kW): tap-mask pattern (conv_igemm64.hpp)
    int yplace, ysH, ysW, yoH, yoW, yH, yW;   // output placement (sm_conv_desc.y_sH ...): strided/offset output pixels
    // weight-stationary 3x3 kernel (conv3x3_ws.hpp): tile geometry chosen by ws_setup()
    int ws_TH, ws_TW, ws_tiles_x, ws_tiles_y, ws_ntiles, ws_patch_bytes;
    unsigned ws_pw_mul, ws_pw_sh, ws_tw_mul, ws_tw_sh, ws_txy_mul, ws_txy_sh, ws_ty_mul, ws_ty_sh;
    // nine-tap patch kernel (conv3x3_p9.hpp): geometry chosen by p9_setup(); ups = the input is read through a fused nearest 2x upsampling
    int p9_TW, p9_tiles_per_band, p9_patch_bytes, p9_patch_px, ups;
    half_t* y2; int y2_ld;           // conv3x3_ws only: second output = 2x2 / stride-2 max-pool of y (nullptr = none)
    // conv3x3_ws only: a 1x1 convolution applied to the output tile while it is still in registers ("head": w3 [n3][Cout] fp16, b3 fp32,
    // y3 = act3(w3 . y + b3), n3 <= 32); y itself is stored only if p.y != nullptr
    const half_t* w3; const float* b3; half_t* y3; int n3, y3_ld, act3; float sl3;
    // ... the EXPAND head (" xhead" instantiations): n3 = 32 .. 256 (multiple of 32), w3 / b3 in LDS, optional residual added to the head's output
    const float* xf; int in_C, in_H, in_W, in_pt, in_pl;   // stem form with loader waves (conv3x3_ws.hpp NLW): the fp32 NCHW input the patches are built from
    const half_t* w3s;               // conv3x3_wsx.hpp SIDE: weights [n3][64] of the head's second GEMM; res3 / res3_ld then name its 64-channel input tensor
    const half_t* res3; int res3_ld; int hx_dbg;   // hx_dbg: SM_HX_DEBUG ablation bits (1 = no stores, 2 = no residual loads, 4 = no head at all)
    unsigned p9_pw_mul, p9_pw_sh, p9_h2_mul, p9_h2_sh;
    int pq_OH, pq_OW;                // stem_pool.hpp: size of the pooled output (y2)
    // conv1x1_ps.hpp: SECOND SOURCE of a pointwise convolution whose reduction runs over two tensors, K = K1 + K2 (sm_conv2d_dual_fwd_cfg): K tiles
    // [0, kt_sw) read x as usual, K tiles [kt_sw, K/64) read pixel (n, oy*x2_s, ox*x2_s) of x2 [N, x2_H, x2_W, K2] (pixel stride x2_ld)
    const half_t* x2; int x2_ld, x2_H, x2_W, x2_s, kt_sw;
    // conv3x3_kq.hpp: tile geometry chosen by kq_setup() (TH > 0: a tile is TH rows of one image; else TI whole images)
    int kq_RS, kq_NI, kq_slice_bytes, kq_TH, kq_TI, kq_mt, kq_bands, kq_mpx, kq_nsb;
    unsigned kq_rs_mul, kq_rs_sh, kq_h1_mul, kq_h1_sh, kq_bands_mul, kq_bands_sh;
};

// Division of n < 2^31 by a launch-time constant d: q = (n * mul) >> sh with s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1
// (< 2^32), sh = 31 + s.  Exact: mul*d - 2^(31+s) = e in (0, d], so n*mul / 2^(31+s) = n/d + n*e / (d * 2^(31+s)) and the
// excess is < 2^-s <= 1/d, too small to carry the fraction of n/d (<= (d-1)/d) over the next integer.
inline void sm_fastdiv_make(unsigned d, unsigned* mul, unsigned* sh) {
    unsigned s = 0; while ((1ull << s) < d) s++;
    *mul = (unsigned)(((1ull << (31 + s)) / d) + 1); *sh = 31 + s;
}
__device__ __forceinline__ unsigned sm_fastdiv(unsigned n, unsigned mul, unsigned sh) { return (unsigned)(((unsigned long long)n * mul) >> sh); }

// LDS-DMA as an asm statement (cdna_hip_programming.md §5.7): behind a `global_load_lds` BUILTIN hipcc's waitcnt pass treats the LGKM
// counter as out of order and waits lgkmcnt(0) — for fragment reads it has only just issued — before the next MFMA (one exposed LDS
// latency per DMA piece, ~30 % of the k loop).  The asm form is invisible to that pass; its completion is waited for explicitly
// (counted s_waitcnt vmcnt + s_barrier).  M0 is saved and restored inside the statement.
__device__ __forceinline__ void ws_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned ws_lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }

template <int N> __device__ __forceinline__ void sm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// conv_kq.hip: launches instantiation `variant` (kq_variant(KG, NG, NF, NB, NFR, S2)) of conv3x3_kq_kernel
int kq_variant(int KG, int NG, int NF, int NB, int NFR, int S2);
void kq_launch(int variant, const ConvParams& p, unsigned grid, size_t lds, hipStream_t s);
// ... and conv3x3_kqp_kernel<KG, NG> (persistent, weights in registers); -1: no such instantiation
int kqp_launch(int KG, int NG, const ConvParams& p, hipStream_t s);
