// conv_kq.hip — translation unit of the wave-split 3x3 convolution (conv3x3_kq.hpp, tile-config kind 9 of conv_igemm.hip's launcher).
#include <stdlib.h>
#include "conv_shared.hpp"

namespace {
#include "conv3x3_kq.hpp"
#include "conv3x3_kqp.hpp"
typedef void (*KqFn)(KqParams);
struct KqVariant { int KG, NG, NF, NB, NFR, S2; KqFn fn; };
const KqVariant kVariants[] = {
    {4, 1, 4, 1, 13, 0, conv3x3_kq_kernel<4, 1, 4, 1>},
    {2, 2, 4, 1, 13, 0, conv3x3_kq_kernel<2, 2, 4, 1>},
    {4, 1, 2, 1, 13, 0, conv3x3_kq_kernel<4, 1, 2, 1>},
    {2, 2, 2, 1, 13, 0, conv3x3_kq_kernel<2, 2, 2, 1>},
    {2, 2, 2, 2, 13, 0, conv3x3_kq_kernel<2, 2, 2, 2>},
    {1, 4, 2, 2, 13, 0, conv3x3_kq_kernel<1, 4, 2, 2>},
    // stride 2 (the patch of a tile is four times its output: one 64-channel slice of a 13-fragment tile, or two of a 7-fragment tile, fill the LDS)
    {1, 4, 2, 1, 13, 1, conv3x3_kq_kernel<1, 4, 2, 1, 13, true>},
    {2, 2, 4, 1, 7, 1, conv3x3_kq_kernel<2, 2, 4, 1, 7, true>},
    {2, 2, 2, 1, 7, 1, conv3x3_kq_kernel<2, 2, 2, 1, 7, true>},
    {1, 4, 2, 1, 7, 1, conv3x3_kq_kernel<1, 4, 2, 1, 7, true>},
};
}  // namespace

size_t kqp_lds_bytes(int KG) { return (size_t)2 * KG * (KQP_TH + 2) * 16 * 128 + 2048; }
int kqp_launch(int KG, int NG, const ConvParams& p, hipStream_t s) {
    KqpParams q;
    q.x = p.x; q.w = p.w; q.bias = p.bias; q.y = p.y;
    q.N = p.N; q.H = p.H; q.W = p.W; q.Cin = p.Cin; q.Cout = p.Cout; q.K = p.K; q.x_ld = p.x_ld; q.y_ld = p.y_ld; q.act = p.act; q.slope = p.slope; q.ups = p.ups;
    q.tiles_x = (p.W + KQP_TW - 1) / KQP_TW; q.tiles_y = (p.H + KQP_TH - 1) / KQP_TH; q.ntiles = p.N * q.tiles_x * q.tiles_y;
    const int grid = q.ntiles < SM_NUM_CU ? q.ntiles : SM_NUM_CU;
    q.tpb = (q.ntiles + grid - 1) / grid;
    sm_fastdiv_make((unsigned)q.tiles_x, &q.tx_mul, &q.tx_sh); sm_fastdiv_make((unsigned)(q.tiles_x * q.tiles_y), &q.txy_mul, &q.txy_sh);
    const int blocks = (q.ntiles + q.tpb - 1) / q.tpb;
    if (KG == 2 && NG == 2) hipLaunchKernelGGL((conv3x3_kqp_kernel<2, 2>), dim3(blocks), dim3(256), kqp_lds_bytes(2), s, q);
    else if (KG == 1 && NG == 4) hipLaunchKernelGGL((conv3x3_kqp_kernel<1, 4>), dim3(blocks), dim3(256), kqp_lds_bytes(1), s, q);
    else return -1;
    return 0;
}
int kq_variant(int KG, int NG, int NF, int NB, int NFR, int S2) {
    for (int i = 0; i < (int)(sizeof(kVariants) / sizeof(kVariants[0])); i++)
        if (kVariants[i].KG == KG && kVariants[i].NG == NG && kVariants[i].NF == NF && kVariants[i].NB == NB && kVariants[i].NFR == NFR && kVariants[i].S2 == S2) return i;
    return -1;
}
#ifdef SM_KQ_TL
extern "C" int sm_kq_timeline_read(unsigned long long* dst, int blocks) {   // diagnostic build: the stamps of the last launch (blocks x 4 waves x 16)
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_kq_tl), (size_t)blocks * 4 * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
void kq_launch(int variant, const ConvParams& p, unsigned grid, size_t lds, hipStream_t s) {
    KqParams q;
    q.x = p.x; q.w = p.w; q.bias = p.bias; q.y = p.y;
    q.N = p.N; q.H = p.H; q.W = p.W; q.Cin = p.Cin; q.Cout = p.Cout; q.K = p.K; q.x_ld = p.x_ld; q.y_ld = p.y_ld; q.act = p.act; q.slope = p.slope; q.Ho = p.Ho; q.Wo = p.Wo;
    q.nt = p.nt; q.RS = p.kq_RS; q.NI = p.kq_NI; q.slice_bytes = p.kq_slice_bytes; q.TH = p.kq_TH; q.TI = p.kq_TI; q.mt = p.kq_mt; q.bands = p.kq_bands; q.mpx = p.kq_mpx; q.nsb = p.kq_nsb;
    q.nt_mul = p.nt_mul; q.nt_sh = p.nt_sh; q.dhw_mul = p.dhw_mul; q.dhw_sh = p.dhw_sh; q.dw_mul = p.dw_mul; q.dw_sh = p.dw_sh;
    q.h1_mul = p.kq_h1_mul; q.h1_sh = p.kq_h1_sh; q.bands_mul = p.kq_bands_mul; q.bands_sh = p.kq_bands_sh; q.rs_mul = p.kq_rs_mul; q.rs_sh = p.kq_rs_sh;
    // swizzle key: the usual one, except where the table-addressed instantiations (NF <= 2) have a conflict-free alternative — 28-pixel rows in 32-slot patch
    // rows: (2 (cc / 2) + 4 r) & 7 serves every 16-lane group of every (fragment, tap) in 1.08 LDS cycles instead of 2.0 (tools/lds_swizzle_search.py --tw 28 --rs 32 --mfma 16 --pixels 196)
    q.ka = 1; q.kb = (p.kq_RS / 2) & 7;
    if (!kVariants[variant].S2 && kVariants[variant].NF <= 2 && p.kq_RS == 32 && p.W == 28) { q.ka = 2; q.kb = 4; }
    hipLaunchKernelGGL(kVariants[variant].fn, dim3(grid), dim3(256), lds, s, q);
}
