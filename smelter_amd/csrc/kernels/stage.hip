// stage.hip — a RUN of convolution layers at one resolution (the bottleneck blocks of a ResNet stage: 1x1 reduce -> 3x3 -> 1x1 expand + residual, ...) as ONE
// persistent launch: sm_stage_fwd (include/smelter_hip.h).  Replaces, per phase, an MPSCNNConvolutionNode (+ MPSNNAdditionNode + MPSCNNNeuronReLUNode) of the
// reference (/root/reference/Sources/Smelter/Converters.swift:187-338, :430-451, :342-359) — the same arithmetic as sm_conv2d_fwd / sm_conv2d_dual_fwd_cfg.
//
// Why: at batch 64 every late layer of ResNet-50 gives a CU ONE tile, so every launch pays kernel entry + a cold first fill (operands the
// previous launch wrote come from the Infinity Cache at ~13 B/clk/CU: the L2 is invalidated at the kernel boundary) + epilogue + drain, 7.7 us per launch, 48 times.
// Inference has no cross-sample dependency (BN uses stored statistics, Converters.swift:760-773), so:
//   * 256 blocks (one per CU) form 8 TEAMS of 32 — the blocks of one XCD (HW_REG_XCC_ID + a per-XCD ticket; placement is verified at kernel entry, see below);
//   * team t owns images [t N/8, (t+1) N/8) for the WHOLE run: what a phase writes is read by the next phase from the SAME XCD's L2 (~27 B/clk/CU), with plain
//     stores on the producer side and L1-bypassing (sc1) loads on the consumer side — no cache write-back, no invalidate (tools/ubench/xcdphase.hip: 0 stale words of 168 M);
//   * hand-off = one counter per (phase, image): a tile adds 1 to the counters of the images it covers once its stores are drained (s_waitcnt vmcnt(0) in every wave,
//     barrier, one relaxed agent-scope atomic); a consumer tile polls the counters of ITS images only (one lane, sc1 loads, s_sleep, bounded) — no team-wide barrier,
//     images advance independently; weights of the next tile are requested before the poll;
//   * a phase's tile = TP output pixels (whole images or an aligned part of one) x BN couts, K reduction in "units" of 32 channels:
//       kinds R (1x1, many input channels, few couts) and T (3x3): K split over the four waves (each wave owns the whole tile, partial sums meet in LDS);
//       kinds E (1x1, few input channels, many couts; optionally a second source = projection shortcut concatenated along K): couts split over the four waves;
//       the same three with EIGHT waves (forms 6-8: two waves per SIMD, <= 256 registers each, 32 couts per wave): with one wave per SIMD every cycle the wave spends issuing its
//       memory instructions or waiting on a counter is a cycle its matrix unit idles (profiles/r05_stage_ablation.txt part 8); with two, one's issue and waits sit under the
//       other's MFMAs.  The planner takes the 8-wave forms when every phase of the run has one (a launch has one block size);
//     the input patch goes HBM/L2 -> LDS by buffer LDS-DMA (out-of-range lanes = zeros without a memory access: padding, halo, dead pixels), 64-channel slices
//     [slot][128 B] XOR-swizzled by slot/2, NB buffers deep; weights go global -> registers in fragment layout (never through the LDS), two units per request so that
//     both halves of a 128-byte line are asked for together; v_mfma_f32_16x16x32_f16, fp32 accumulators; bias + residual + activation + one rounding in the epilogue.
// Numerics: fp32 accumulation over (channel block, tap, channel) ascending inside a wave's K slice, slices added in ascending order — the same as the wave-split
// per-layer kernel (conv3x3_kq.hpp); a phase launched ALONE through this file is bit-identical to the same phase inside a run (tests/test_gpu_stage.py).
//
// Two modes, decided once per launch by the blocks themselves: every block takes a ticket from its XCD's counter; if all 256 arrive within a short bound and every XCD holds exactly 32, the
// launch is STATIC and XCD-local as above.  Otherwise (blocks of another process on some CUs, a partitioned device, another placement) the tiles of a phase are CLAIMED from one counter by
// whichever blocks are running, and every hand-off uses the architecturally sanctioned agent-scope release / acquire (buffer_wbl2 sc1 / buffer_inv sc1, MI355X_MICROARCH.md §Workgroup
// dispatch) — slower, never wrong, and it needs no co-residency at all (a claimed tile belongs to a running block; a phase is claimed out before the next one opens).  Every spin is bounded;
// sm_stage_status reports the modes seen and whether a hand-off wait ever ran out.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "conv_shared.hpp"

namespace {

typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
constexpr int ST_MAXPH = SM_STAGE_MAX_PHASES, ST_TEAM = 32, ST_NTEAM = 8, ST_GRID = ST_TEAM * ST_NTEAM;
constexpr int ST_CTL_HDR = 48;                  // ctl words: [0..7] tickets per XCD, [8] arrived, [9] done, [10] error bits, [11] modes seen (1 = XCD-local, 2 = claimed + fenced), [12] this launch's mode,
                                                // [16 .. 16 + phases) next unclaimed tile per phase (mode 2), then cnt[phase][cnt_ld]
constexpr int ST_MODE = 12, ST_CLAIM = 16;

struct StPhase {
    const half_t* x; const half_t* x2; const half_t* w; const half_t* wp; const float* bias; const half_t* res; half_t* y;   // wp: the weights in fragment order (sm_stage_pack_weights) or nullptr
    int kind, H, W, HW, K1, K, Cout;            // K1 = channels of x (per tap); K = row length of w = TAPS K1 (+ K2)
    int x_ld, x2_ld, res_ld, y_ld, x2_W, x2_HW, x2_s;
    int act; float slope;
    int TP, TI, MT, NT, nsb1, nsb, RS, NI;      // pixels / whole images per tile; tiles per team; super-blocks from x / in all; patch row slots; DMA instructions per 64-channel slice
    int dep0, dep1, dep2, expect;               // producing phase of x / x2 / res (-1: outside the run); arrivals per image that complete THIS phase's output
    unsigned x_bytes, x2_bytes, res_bytes, y_bytes, wt_bytes;   // wt_bytes: packed weights of ONE cout tile (a wave reads a fourth of it, front to back)
    unsigned hw_mul, hw_sh, w_mul, w_sh, h1_mul, h1_sh, kpr_mul, kpr_sh;
};
struct StArgs { unsigned* ctl; int nphase, N, ipt, spin_limit, arrive_limit, cnt_ld, lds_bytes, touch; StPhase ph[ST_MAXPH]; };   // touch: bytes of a tile's weights a wave asks the L2 for ahead of its ring (0: none)
static_assert(sizeof(StArgs) <= 4096 && ST_CLAIM + ST_MAXPH <= ST_CTL_HDR, "kernel argument block, control words");

#ifndef ST_EPI_DIRECT
#define ST_EPI_DIRECT 1
#endif
// KG x NG = 4 waves: KG = K groups (a wave multiplies channels [kg WCH, (kg + 1) WCH) of every super-block of KG WCH channels), NG = cout groups of NF 16-cout fragments;
// NFR = 16-pixel fragments per tile; TAPS = 1 | 9; SBG = super-blocks per unrolled loop body; NB = patch buffers.
template <int KG_, int NG_, int NF_, int NFR_, int TAPS_, int WCH_, int SBG_, int NB_, int RW_>
struct StKind {
    static constexpr int KG = KG_, NG = NG_, NF = NF_, NFR = NFR_, TAPS = TAPS_, WCH = WCH_, SBG = SBG_, NB = NB_;
    static constexpr int QPW = WCH / 32, UPS = TAPS * QPW, UB = SBG * UPS, NSL = WCH / 64, SBCH = KG * WCH, WN = NF * 16, BN = NG * WN, PD = NB - 1;
    static constexpr int RW = RW_, PA = RW - 2;                 // weight ring (units) and how many units ahead a PAIR of units is requested.  (Requesting unit U + PA fragment by
                                                                // fragment between the MFMAs of unit U instead of as a burst at its top: 136 -> 155 us on the 7x7 stage, profiles/r05_stage_ablation.txt)
    static constexpr int NI_LDS = 157 / (NB_ * KG_ * (WCH_ / 64));              // ... that fit the LDS (NB buffers of KG x WCH / 64 slices of NI KB)
    static constexpr int NI_MAX = TAPS == 1 ? (NFR * 2 < NI_LDS ? NFR * 2 : NI_LDS) : (NFR <= 7 ? 18 : 37);  // DMA instructions per slice the kind can take
    static constexpr bool FIXS = TAPS == 1;                     // 1x1 forms: a slice is NI_MAX KB whatever the tile holds — slice and channel half of a fragment read are an immediate offset and a
                                                                // choice between two address registers, not arithmetic per fragment and unit
    static constexpr int NIW_MAX = (NSL * NI_MAX + NG - 1) / NG;               // ... per wave and super-block
    static constexpr int DPU = NB > 1 ? (NIW_MAX + UPS - 1) / UPS : 0, V = NF + DPU;   // DMA slots per unit; VMEM operations per unit
    static constexpr bool DUAL = TAPS == 1 && KG == 1, SHARED = NG > 1;
    static constexpr int TPMAX = NFR * 16;
    static constexpr bool XCH = ST_EPI_DIRECT && KG_ == 1;   // weight rows assigned so that an accumulator lane holds eight consecutive couts: the tile leaves from the registers
    static constexpr int NW = KG * NG;                          // waves of the block: 4 (one per SIMD, up to 512 registers each) or 8 (two per SIMD, 256 each)
    static_assert((NW == 4 || NW == 8) && UB % RW == 0 && UB % 2 == 0 && QPW % 2 == 0 && (NB > 1 || DPU == 0), "shape of the loop body");
};
#ifndef ST_RW1
#define ST_RW1 4          // weight ring of the 1x1 forms, in units (a pair is requested ST_RW1 - 2 units ahead); 8 needs loop bodies of 8 units
#endif
#define ST_SBG1 (ST_RW1 / 4)
#define ST_SBG_R7 (ST_RW1 / 2)
typedef StKind<4, 1, 4, 7, 1, 64, ST_SBG_R7, 3, ST_RW1> StR7;     // 0: 1x1, K % 512 == 0, tile <= 112 px x 64 couts      (7x7: 2048 -> 512)
typedef StKind<4, 1, 4, 7, 9, 64, 1, 2, 6> StT7;     // 1: 3x3, K1 % 256 == 0, images of <= 56 px, 2+ per tile x 64 couts (7x7 512 -> 512)
typedef StKind<1, 4, 4, 7, 1, 128, ST_SBG1, 3, ST_RW1> StE7;    // 2: 1x1 (+ second source), K % 128 == 0, <= 112 px x 256 couts (7x7: 512 -> 2048)
typedef StKind<2, 2, 4, 7, 1, 128, ST_SBG1, 3, ST_RW1> StR14;   // 3: 1x1, K % 256 == 0, <= 112 px x 128 couts          (14x14: 1024 -> 256)
typedef StKind<1, 4, 4, 13, 1, 128, 1, 3, 4> StE14;  // 4: 1x1 (+ second source), <= 208 px x 256 couts       (14x14: 256 -> 1024)
typedef StKind<4, 1, 4, 13, 9, 64, 1, 1, 6> StT14;   // 5: 3x3, one image of <= 208 px x 64 couts, patch single-buffered (14x14 256 -> 256)
// EIGHT-wave forms (two waves per SIMD, <= 256 registers each: one wave's memory issue and counter waits sit under its SIMD partner's MFMAs; a run is all-4 or all-8)
typedef StKind<1, 8, 2, 7, 1, 128, 1, 3, 4> StE8;    // 6: form 2's tile (<= 112 px x 256 couts), couts split over eight waves of 32 (every wave reads every pixel fragment: twice form 2's LDS reads, no exchange of partial sums)
typedef StKind<4, 2, 2, 7, 1, 64, 2, 3, 4> StR8;     // 7: form 0's tile (<= 104 px x 64 couts): K split 4 ways, two waves of 32 couts per K group
typedef StKind<4, 2, 2, 7, 9, 64, 1, 2, 6> StT8;     // 8: form 1's tile (3x3, images of <= 56 px, 2+ per tile x 64 couts), K split 4 x 2 waves of 32 couts
constexpr int ST_NKINDS = 9;
// Epilogue of the cout-split forms: 1 (default) = straight from the accumulator layout (16 x 4 lanes = 16 pixels x 64 bytes per instruction: half lines); 0 = through the LDS like the
// K-split forms (a pixel's 64 couts leave as one 128-byte piece per 8 lanes: whole lines for stores and residual loads).  Measured (profiles/r05_stage_ablation.txt part 5): the LDS round
// trip costs more than the whole lines save — 14 x 14 expand epilogue 8.2 -> 11.1 us, 7 x 7 3.6 -> 4.6 us, the 7 x 7 run 135 -> 143 us.
// Plain ReLU in the epilogues as round + packed max instead of compare / multiply / select per value (the 7 x 7 run 125.2 -> 124.4 us; the same values).
#ifndef ST_RELU_MAX
#define ST_RELU_MAX 1
#endif
// Weight touch (st_touch): 1 = each wave asks the L2 for its fourth of the tile's weight slice at tile start, in front of the wait for the producers; 0 = off.
#ifndef ST_TOUCH
#define ST_TOUCH 0
#endif
#ifndef ST_TOUCH_BYTES
#define ST_TOUCH_BYTES (160 * 1024)     // per wave and tile (a 7 x 7 3x3 tile's fourth is 147 KB)
#endif

#ifdef SM_STAGE_TL   // diagnostic build (tools/stage_tl.py): per block and tile, wall clock (100 MHz, the same on every XCD) at 8 points
__device__ unsigned long long g_st_tl[ST_GRID * 64 * 8];
__device__ unsigned g_st_tl_n[ST_GRID];
#define ST_TL(i) do { if (threadIdx.x == 0 && tl_slot < 64) g_st_tl[(blockIdx.x * 64 + tl_slot) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ST_TL(i)
#endif

#ifndef ST_DMA_AUX
#define ST_DMA_AUX " sc1"
#endif
__device__ __forceinline__ void st_bdma16(unsigned voff, st_u32x4 rsrc, unsigned lds_dst, unsigned soff) {   // 16 bytes per lane, buffer -> LDS, L1 bypassed (sc1): the producer may be another CU of this XCD
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen" ST_DMA_AUX " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
// One dword per 128-byte line of [base, base + bytes): 64 lines = 8 KB per instruction, into an LDS word nobody reads.  Nothing waits for the data: the point is that the XCD's L2 asks the
// Infinity Cache / HBM for the lines NOW, so that the weight ring's requests (a few KB in flight per wave) find them at L2 latency instead of serialising on the far latency.
__device__ __forceinline__ void st_touch(const void* base, unsigned bytes, unsigned sink_lds, unsigned lane128) {
    const uintptr_t b = (uintptr_t)base;
    const st_u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu)), (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)sink_lds);
    for (unsigned o = 0; o < bytes; o += 8192u) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane128), "s"(rs), "s"(dst), "s"(o) : "memory");
    }
}
__device__ __forceinline__ void st_bdma4_dead(st_u32x4 rsrc, unsigned lds_dst) {   // a DMA slot that can never be a real one (the counted waits see every slot): 4 bytes per lane, out of range, into the sink
    unsigned keep; const unsigned voff = 0x80000000u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
template <int OFF> __device__ __forceinline__ void st_gload16(half8_t& d, unsigned voff, const half_t* sbase) { asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory"); }
template <int OFF> __device__ __forceinline__ void st_dsread16(half8_t& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ void st_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned st_ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int N> __device__ __forceinline__ void st_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory"); }

// One tile of one phase: output pixels pix0 .. pix0 + PTe - 1 (global pixel index n HW + y W + x), couts of cout tile `ntile`.
template <class KD, class Sync>
__device__ __forceinline__ void st_tile(const StPhase& p, const int pix0, const int PTe, const int ntile, const unsigned lds0, unsigned char* const smem, const unsigned smem_lds, Sync&& sync, const int tl_slot) {
    constexpr int KG = KD::KG, NG = KD::NG, NF = KD::NF, NFR = KD::NFR, TAPS = KD::TAPS, WCH = KD::WCH, SBG = KD::SBG, NB = KD::NB;
    constexpr int QPW = KD::QPW, UPS = KD::UPS, UB = KD::UB, NSL = KD::NSL, SBCH = KD::SBCH, WN = KD::WN, BN = KD::BN, PD = KD::PD, RW = KD::RW, PA = KD::PA, DPU = KD::DPU, V = KD::V;
    constexpr bool DUAL = KD::DUAL, SHARED = KD::SHARED, T3 = TAPS == 9;
    constexpr int NTAB = DPU * UPS;                                    // DMA slots of one super-block (per wave)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4, c8 = lane & 7, l8 = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = wave / NG, ng = wave % NG;
    const int H = p.H, W = p.W, HW = p.HW, RS = p.RS, NI = p.NI, K1 = p.K1, K = p.K, Cout = p.Cout, nsb = p.nsb, nsb1 = p.nsb1;
    const unsigned SLB = KD::FIXS ? (unsigned)KD::NI_MAX * 1024u : (unsigned)NI * 1024u, BUFB = (unsigned)(KG * NSL) * SLB, sink = lds0 + (unsigned)NB * BUFB;
    const unsigned myslices = lds0 + (unsigned)(kg * NSL) * SLB;       // this K group's slices inside buffer 0
    const int co0 = ntile * BN + ng * WN;
    const int n0 = (int)sm_fastdiv((unsigned)pix0, p.hw_mul, p.hw_sh);
    // 3x3: a tile is whole images, or (TP < HW) a BAND of whole rows of one image — its patch is then rows y0 - 1 .. y0 + TP / W of that image
    const bool band = T3 && p.TP < HW;
    const int imgs = !T3 ? 1 : band ? 1 : (int)sm_fastdiv((unsigned)PTe, p.hw_mul, p.hw_sh);
    const int t3_y0 = band ? (int)sm_fastdiv((unsigned)(pix0 - n0 * HW), p.w_mul, p.w_sh) - 1 : -1;   // image row of patch row 0

    // ---- weight fragments: lane (l15, g) of fragment j holds one cout row, channels 8 g .. 8 g + 7 of the unit's 32.  KG == 1 (register epilogue): row l15 of fragment j is
    // cout (j/2) 32 + (l15/4) 8 + (j%2) 4 + l15%4, so that an accumulator lane holds eight CONSECUTIVE couts across fragments 2 jp, 2 jp + 1 (one 16-byte store) ----
    unsigned boff[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
        int co = KD::XCH ? co0 + (j >> 1) * 32 + (l15 >> 2) * 8 + (j & 1) * 4 + (l15 & 3) : co0 + j * 16 + l15;
        if (co >= Cout) co = Cout - 1;
        boff[j] = (unsigned)(co * K + g * 8) * 2u;
    }
    half8_t wb[RW][NF];
    // Packed weights (sm_stage_pack_weights): the fragments of this wave, unit after unit, are ONE linear stream — [cout tile][ng][kg][super-block][unit][fragment][lane] x 16 bytes:
    // a request is 1 KB contiguous (8 full lines) instead of 16 row pieces of 64 bytes, which cost a request each (measured: 24-30 GB/s per CU against 55 for the patch stream)
    const bool packed = p.wp != nullptr;
    const half_t* const wstream = p.wp + ((size_t)((ntile * NG + ng) * KG + kg) * (size_t)(nsb * UPS)) * (size_t)(NF * 512);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_b = [&](auto slotc, int sb, int ul) {                   // weight fragments of unit `ul` of super-block `sb` -> ring slot
        constexpr int SLOT = decltype(slotc)::value;
#ifndef ST_ABL_NOW
        if (packed) {
            const half_t* ub = wstream + (size_t)(sb * UPS + ul) * (size_t)(NF * 512);
            sm_static_for<0, NF>([&](auto jc) { constexpr int J = decltype(jc)::value; st_gload16<J * 1024>(wb[SLOT][J], lane16, ub); });
            return;
        }
#endif
        int k1o = K1; asm volatile("" : "+s"(k1o));                  // (opaque: the nine tap offsets are NOT loop invariants worth 18 scalar registers)
        const half_t* ub = p.w + ((ul / QPW) * k1o + sb * SBCH + kg * WCH + (ul % QPW) * 32);
#ifdef ST_ABL_NOW
        for (int j = 0; j < NF; j++) { half8_t& d = wb[SLOT][j]; const unsigned bo = 0x80000000u; const st_u32x4 rz = {0u, 0u, 0u, 0x00020000u}; asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(d) : "v"(bo), "s"(rz) : "memory"); }
#else
#pragma unroll
        for (int j = 0; j < NF; j++) { half8_t& d = wb[SLOT][j]; const unsigned bo = boff[j]; asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(bo), "s"(ub) : "memory"); }
#endif
    };
    // requests of the first PA units: nothing here depends on another tile
    sm_static_for<0, PA>([&](auto uc) { constexpr int U = decltype(uc)::value; load_b(uc, U / UPS, U % UPS); });
    sync();                                                           // the producers of this tile's inputs have published; the block is through with the previous tile's LDS

    // ---- patch geometry ----
    const unsigned ld2 = (unsigned)p.x_ld * 2u;
    const st_u32x4 rs_x = {(unsigned)(uintptr_t)p.x, (unsigned)((uintptr_t)p.x >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    st_u32x4 rs_x2 = rs_x;
    if constexpr (DUAL) { if (p.x2) rs_x2 = st_u32x4{(unsigned)(uintptr_t)p.x2, (unsigned)((uintptr_t)p.x2 >> 32) & 0xffffu, p.x2_bytes, 0x00020000u}; }
    // byte offset (channel base excluded) of this lane's 16 bytes of DMA instruction i of a slice: slots 8 i .. 8 i + 7, lane = (slot 8 i + l8, chunk c8); dead: beyond the buffer.
    // 1x1, first source: slot = pixel, the offset is (pix0 + 8 i) ld2 + [l8 ld2 + swizzled chunk], and the swizzle key (slot / 2) & 7 = (l8 / 2) | 4 (i & 1): two lane constants, no
    // table (slots beyond the tile copy whatever follows it — their pixels are never stored; beyond the tensor the descriptor returns zeros)
    const unsigned lo0 = (unsigned)l8 * ld2 + ((((unsigned)c8) ^ (unsigned)(l8 >> 1)) << 4), lo1 = (unsigned)l8 * ld2 + ((((unsigned)c8) ^ (unsigned)((l8 >> 1) | 4)) << 4);
    auto dma_off = [&](int i) -> unsigned { return (unsigned)(pix0 + 8 * i) * ld2 + ((i & 1) ? lo1 : lo0); };   // (1x1, first source)
    // 3x3: the instructions of a slice in order — patch row r = 0 .. (row 0 = halo above the first image; images H + 1 rows apart: the row between two images is zero), 8-slot group k
    // of the row.  Which image row a patch row is, whether it exists, its byte offset: wave-uniform running counters (no division); per lane only the column test and the add.
    // Swizzle key of slot r RS + 8 k + l8 = ((r RS / 2 + 4 k) & 4) | l8 / 2 (RS % 8 == 0): the two lane constants of the 1x1 case.
    int t3_k = 0, t3_y = t3_y0, t3_im = 0, t3_r = 0;
    auto t3_next = [&]() -> unsigned {
        const bool rok = t3_y >= 0 && t3_y < H && t3_im < imgs;
        const int cc0 = 8 * t3_k;
        const unsigned rowb = (unsigned)(((n0 + t3_im) * H + t3_y) * W - 1 + cc0) * ld2;
        const bool okc = (unsigned)(cc0 + l8 - 1) < (unsigned)W;
        const unsigned v = rowb + ((((t3_r * (RS >> 1) + 4 * t3_k) & 4) != 0) ? lo1 : lo0);
        if (++t3_k == (RS >> 3)) { t3_k = 0; t3_r++; if (++t3_y == H + 1) { t3_y = 0; t3_im++; } }
        return (rok && okc) ? v : 0x80000000u;
    };
    auto t3_skip = [&]() { if (++t3_k == (RS >> 3)) { t3_k = 0; t3_r++; if (++t3_y == H + 1) { t3_y = 0; t3_im++; } } };   // (another wave's instruction: the counters only)
    auto dma_off2 = [&](int i) -> unsigned {                          // second source: output pixel (n, oy, ox) reads pixel (n, oy s, ox s) of x2
        const int s = 8 * i + l8;
        const unsigned chunk = (((unsigned)c8 ^ (unsigned)((s >> 1) & 7)) << 4);
        const unsigned P = (unsigned)(pix0 + (s < PTe ? s : 0));
        const unsigned n = sm_fastdiv(P, p.hw_mul, p.hw_sh), r = P - n * (unsigned)HW, oy = sm_fastdiv(r, p.w_mul, p.w_sh), ox = r - oy * (unsigned)W;
        return s < PTe ? (n * (unsigned)p.x2_HW + (oy * (unsigned)p.x2_W + ox) * (unsigned)p.x2_s) * ((unsigned)p.x2_ld * 2u) + chunk : 0x80000000u;
    };
    // DMA slot m of a super-block (this wave): instruction idx = ng + m NG of the K group's NSL NI instructions.  Tables (one register per slot): 3x3 patches and second sources
    constexpr int NTAB1 = T3 ? NTAB : 0, NTAB2 = DUAL ? NTAB : 0;
    unsigned dmo[NTAB1 > 0 ? NTAB1 : 1], dmo2[NTAB2 > 0 ? NTAB2 : 1];
    const int nidx = NSL * NI;
    auto issue_slot = [&](auto mc, int sbn, unsigned bufo) {           // one DMA slot: real if the super-block and the instruction exist, else zeros into the sink (the counted waits see every slot)
        constexpr int M = decltype(mc)::value;
        if constexpr (M * NG >= NSL * KD::NI_MAX) { st_bdma4_dead(rs_x, (unsigned)__builtin_amdgcn_readfirstlane((int)sink)); return; }   // (slots per unit are a whole number: the last ones of a super-block never carry an instruction)
        const int idx = ng + M * NG;
#ifdef ST_ABL_NODMA
        const bool real = false;
#else
        const bool real = sbn < nsb && idx < nidx;
#endif
        const int sl = idx >= NI ? 1 : 0;
        const bool second = DUAL && sbn >= nsb1;
        const unsigned chb = (unsigned)(((second ? sbn - nsb1 : sbn) * SBCH + kg * WCH + sl * 64) * 2);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(real ? myslices + bufo + (unsigned)idx * 1024u + (sl ? SLB - (unsigned)NI * 1024u : 0u) : sink));
        if constexpr (DUAL) {
            if (second) { st_bdma16(real ? dmo2[M] : 0x80000000u, rs_x2, dst, chb); return; }
        }
        unsigned vo;
        if constexpr (T3) vo = dmo[M]; else vo = dma_off(idx - sl * NI);
        st_bdma16(real ? vo : 0x80000000u, rs_x, dst, chb);
    };

    // ---- prologue: the patches of the first PD super-blocks ----
    if constexpr (NB > 1) {
        sm_static_for<0, NTAB>([&](auto mc) {
            constexpr int M = decltype(mc)::value;
            const int idx = ng + M * NG, sl = idx >= NI ? 1 : 0, i = idx - sl * NI;
            if constexpr (T3) {                                        // the running counters walk the slice's instructions in order: with NG waves per slice every NG-th is this wave's
                static_assert(!T3 || NSL == 1, "3x3 forms: one 64-channel slice per K group");
                unsigned v = 0x80000000u;
                if constexpr (M * NG < NSL * KD::NI_MAX) {
#pragma unroll
                    for (int w = 0; w < NG; w++) { if (w == ng) v = t3_next(); else t3_skip(); }
                }
                dmo[M] = idx < nidx ? v : 0x80000000u;
            }
            if constexpr (DUAL) { dmo2[M] = 0x80000000u; if (idx < nidx && nsb1 < nsb) dmo2[M] = dma_off2(i); }
            issue_slot(mc, 0, 0u);
        });
        sm_static_for<1, PD>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            sm_static_for<0, NTAB>([&](auto mc) { issue_slot(mc, S, (unsigned)S * BUFB); });
        });
    }
    // (while they are in flight) pixel fragments: slice-relative LDS byte offset of this lane's slot — 1x1: swizzled, channel half 0; 3x3: slot of tap (0, 0), before the swizzle
    constexpr bool FIXS = KD::FIXS;
    unsigned pf[T3 ? NFR : 1];
    constexpr bool AD1 = FIXS && KD::NW == 4;                          // (8-wave forms: 256 registers — the second address is an XOR at the read)
    unsigned ad[NFR], ad1[AD1 ? NFR : 1];                             // 1x1: this lane's slot in slice 0 of the current buffer, channel half 0 / half 1 (the swizzle puts the half into bit 6: an XOR, not an offset)
    floatx4 acc[NFR][NF];
#pragma unroll
    for (int f = 0; f < NFR; f++) {
        const int t = f * 16 + l15, tc = t < PTe ? t : 0;
        if constexpr (T3) {
            const int im = (int)sm_fastdiv((unsigned)tc, p.hw_mul, p.hw_sh), rr = tc - im * HW, yy = (int)sm_fastdiv((unsigned)rr, p.w_mul, p.w_sh), xx = rr - yy * W;
            pf[f] = (unsigned)(((im * (H + 1) + yy) * RS + xx) << 7);
        } else { ad[f] = myslices + ((unsigned)tc << 7) + ((((unsigned)g) ^ (unsigned)((tc >> 1) & 7)) << 4); if constexpr (AD1) ad1[f] = ad[f] ^ 64u; }
#pragma unroll
        for (int j = 0; j < NF; j++) acc[f][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned G0 = (unsigned)g << 4;
    half8_t af[NFR];                                                   // ONE buffer: fragment F of the next unit is requested into slot F right behind the last MFMA that reads it (an MFMA takes its operands at issue)
    unsigned bufo_at = 0u;                                             // (1x1) the buffer ad / ad1 point into
    // LDS address of fragment F under unit `ul` (tap ul / QPW, 32-channel piece ul % QPW) of the buffer at `bufo`
    auto frag_addr = [&](int F, int ul, unsigned bufo) -> unsigned {
        const int tap = ul / QPW, q = ul % QPW;
        const unsigned base = myslices + bufo + (unsigned)(q >> 1) * SLB;
        if constexpr (T3) {
            unsigned to = (unsigned)(((tap / 3) * RS + tap % 3) << 7); asm volatile("" : "+s"(to));
            const unsigned rel = pf[F] + to;
            return ((rel ^ G0 ^ ((rel >> 4) & 0x70u)) ^ (unsigned)((q & 1) << 6)) + base;
        } else return 0u;
    };
    // (1x1) fragment F of unit `ul` of the current buffer: half = ul % 2 picks the register, slice = ul / 2 is the immediate offset
    auto read_frag = [&](auto fc, auto ulc) {
        constexpr int F = decltype(fc)::value, Q = decltype(ulc)::value % QPW;
        static_assert((Q >> 1) * KD::NI_MAX * 1024 < 65536, "ds_read offset field");
        if constexpr ((Q & 1) && AD1) st_dsread16<(Q >> 1) * KD::NI_MAX * 1024>(af[F], ad1[F]);
        else if constexpr (Q & 1) st_dsread16<(Q >> 1) * KD::NI_MAX * 1024>(af[F], ad[F] ^ 64u);
        else st_dsread16<(Q >> 1) * KD::NI_MAX * 1024>(af[F], ad[F]);
    };

    // ---- main loop.  VMEM operations per unit, in program order: [even units: 2 NF weight loads of the pair PA units ahead] then DPU patch slots of super-block sb + PD ----
    int bcur = 0;                                                      // buffer of the current super-block
#pragma unroll 1
    for (int sb0 = 0; sb0 < nsb; sb0 += SBG) {
        sm_static_for<0, UB>([&](auto uc) {
            constexpr int U = decltype(uc)::value, UL = U % UPS, SBL = U / UPS;
            const int sb = sb0 + SBL;
            const unsigned bufo = (unsigned)bcur * BUFB;
            if constexpr (UL == 0) {                                   // ---- super-block boundary: its patch has landed, everybody has left the buffer the next patch goes to ----
                if constexpr (NB == 1) {
                    if (sb > 0) { st_lds_barrier(); t3_k = 0; t3_y = t3_y0; t3_im = 0; t3_r = 0; }
                    if constexpr (T3 && NG > 1) { for (int w = 0; w < ng; w++) t3_skip(); }   // (the slice's instructions in order; every NG-th is this wave's)
#pragma unroll 1
                    for (int idx = ng; idx < nidx; idx += NG) {
                        const int sl = idx >= NI ? 1 : 0;
                        unsigned vo;
                        if constexpr (T3) { vo = t3_next(); if constexpr (NG > 1) { for (int w = 1; w < NG; w++) t3_skip(); } } else vo = dma_off(idx - sl * NI);
                        st_bdma16(vo, rs_x, (unsigned)__builtin_amdgcn_readfirstlane((int)(myslices + (unsigned)idx * 1024u)), (unsigned)((sb * SBCH + kg * WCH + sl * 64) * 2));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (SHARED) __builtin_amdgcn_s_barrier();
                } else {
                    // younger than the last slot of this patch: the slots of the PD - 1 later prologue patches (first super-blocks) or everything issued during the PD - 1 super-blocks since
                    if (sb0 == 0 && SBL == 0) st_wait_vm<(PD - 1) * NTAB>();
                    else st_wait_vm<(PD - 1) * UPS * V + (UPS / 2) * 0>();
                    if (SHARED) __builtin_amdgcn_s_barrier();
                }
                if (sb0 == 0 && SBL == 0) ST_TL(2);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FIXS) {
                    const unsigned dlt = bufo - bufo_at; bufo_at = bufo;
                    sm_static_for<0, NFR>([&](auto fc) { constexpr int F = decltype(fc)::value; ad[F] += dlt; if constexpr (AD1) ad1[F] += dlt; read_frag(fc, std::integral_constant<int, 0>()); });
                } else {
#pragma unroll
                    for (int f = 0; f < NFR; f++) { const unsigned a0 = frag_addr(f, 0, bufo); if constexpr (!(T3 && KD::NW == 8)) ad[f] = a0; st_dsread16<0>(af[f], a0); }
                }
            }
            if constexpr (U % 2 == 0) {                                // weights of the pair PA units ahead (beyond the last unit: a harmless reload keeps the counted waits exact)
                constexpr int U2 = U + PA;
                int sb2 = sb0 + (U2 % UB) / UPS + (U2 >= UB ? SBG : 0);
                if (sb2 >= nsb) sb2 = sb;
                load_b(std::integral_constant<int, U2 % RW>(), sb2, U2 % UPS);
                load_b(std::integral_constant<int, (U2 + 1) % RW>(), sb2, (U2 + 1) % UPS);
                st_wait_vm<NF + PA * DPU + PA * NF>();                 // unit U's fragments: younger = its partner's NF, the slots of units U - PA .. U - 1, the pairs requested since
            } else st_wait_vm<(PA + 1) * DPU + PA * NF>();
            __builtin_amdgcn_sched_barrier(0);
            sm_static_for<0, NFR>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                constexpr bool LASTU = UL == UPS - 1;                  // the last unit of a super-block fetches nothing
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LASTU ? NFR - 1 - F : NFR - 1) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NF; j++) {
#ifdef ST_ABL_NOMFMA
                    asm volatile("" ::"v"(wb[U % RW][j]), "v"(af[F]));
#else
                    acc[F][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[U % RW][j], af[F], acc[F][j], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!LASTU) {
                        constexpr int VL = UL + 1;
                        if constexpr (!FIXS && !(T3 && KD::NW == 8)) {
                            if (j == 0) {
                                if constexpr ((VL % QPW) % 2 == 1) ad[F] ^= 64u;       // same tap, same slice, other channel half
                                else ad[F] = frag_addr(F, VL, bufo);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if (j == NF - 1) {
                            if constexpr (FIXS) read_frag(fc, std::integral_constant<int, VL>());
                            else if constexpr (T3 && KD::NW == 8) st_dsread16<0>(af[F], frag_addr(F, VL, bufo));   // (256 registers: the address is made where it is used, not kept per fragment)
                            else { half8_t& d = af[F]; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(ad[F])); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if constexpr (DPU > 0) {
                    sm_static_for<0, DPU>([&](auto dc) {
                        constexpr int D = decltype(dc)::value;
                        if constexpr (F == ((2 * D + 1) * NFR) / (2 * DPU)) {
                            constexpr int M = UL * DPU + D;
                            int bn = bcur + PD; if (bn >= NB) bn -= NB;
                            issue_slot(std::integral_constant<int, M>(), sb + PD, (unsigned)bn * BUFB);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                }
            });
            if constexpr (UL == UPS - 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); bcur = bcur + 1 == NB ? 0 : bcur + 1; }
        });
    }

    // ---- epilogue ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ST_TL(3);
    // (ring slots requested beyond the last unit are never multiplied: keep them allocated up to here, behind the wait — to the register allocator an asm output is
    // written when the statement executes, and a dead destination would be handed out while its load is in flight)
#pragma unroll
    for (int q = 0; q < RW; q++)
#pragma unroll
        for (int j = 0; j < NF; j++) asm volatile("" ::"v"(wb[q][j]));
    const float sl = p.act == SM_ACT_NONE ? 1.f : p.slope;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.y), 0, (int)(p.res ? p.res_bytes : 0u), 0x00020000);
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    const bool relu0 = p.act != SM_ACT_NONE && sl == 0.f; (void)relu0;
    auto finish = [&](floatx4 a0, floatx4 a1, half8_t rv) -> u32x4s {   // 8 consecutive couts of one pixel: + residual, activation, one rounding
        half8_t ov;
#pragma unroll
        for (int e = 0; e < 4; e++) { a0[e] += (float)rv[e]; a1[e] += (float)rv[4 + e]; }
#if ST_RELU_MAX
        if (relu0) {                                                  // plain ReLU: round, then four packed maxima (rounding is monotone: the same values as max-then-round)
#pragma unroll
            for (int e = 0; e < 4; e++) { ov[e] = (half_t)a0[e]; ov[4 + e] = (half_t)a1[e]; }
            ov = __builtin_elementwise_max(ov, half8_t{0, 0, 0, 0, 0, 0, 0, 0});
            return __builtin_bit_cast(u32x4s, ov);
        }
#endif
#pragma unroll
        for (int e = 0; e < 4; e++) { ov[e] = (half_t)(a0[e] < 0.f ? a0[e] * sl : a0[e]); ov[4 + e] = (half_t)(a1[e] < 0.f ? a1[e] * sl : a1[e]); }
        return __builtin_bit_cast(u32x4s, ov);
    };
    const unsigned ybase = (unsigned)pix0 * (unsigned)p.y_ld * 2u, rbase = (unsigned)pix0 * (unsigned)p.res_ld * 2u;
    // registers -> memory, fragments [F0, F1): lane (l15, g) of fragments 2 jp, 2 jp + 1 holds couts co0 + 32 jp + 8 g .. + 7 of pixel 16 F + l15.  Two steps, so that a form can do
    // something else while the residual rows are on their way: every residual load of the range in flight (ONE memory latency per tile, not one per fragment: the weight ring and
    // the pixel fragments are dead by now, the registers are there), then sums + stores
    constexpr int RVF = KD::XCH ? NFR : 1;
    half8_t rv[RVF][NF / 2];
    floatx4 bq[NF];
    auto direct_load = [&](auto f0c, auto f1c) {
        constexpr int F0 = decltype(f0c)::value, F1 = decltype(f1c)::value;
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int co = co0 + (j >> 1) * 32 + g * 8 + (j & 1) * 4;
            bq[j] = (p.bias && co + 3 < Cout) ? *(const floatx4*)(p.bias + co) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int f = 0; f < F1 - F0; f++)
#pragma unroll
            for (int jp = 0; jp < NF / 2; jp++) {
                const int t = (F0 + f) * 16 + l15, co = co0 + jp * 32 + g * 8;
                rv[f][jp] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (p.res) rv[f][jp] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)((t < PTe && co < Cout) ? rbase + (unsigned)t * (unsigned)p.res_ld * 2u + (unsigned)co * 2u : 0x80000000u), 0, 16));
            }
    };
    auto direct_store = [&](auto f0c, auto f1c) {
        constexpr int F0 = decltype(f0c)::value, F1 = decltype(f1c)::value;
        sm_static_for<F0, F1>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            const int t = F * 16 + l15;
#pragma unroll
            for (int jp = 0; jp < NF / 2; jp++) {
                const int co = co0 + jp * 32 + g * 8;
                const bool ok = t < PTe && co < Cout;
                const u32x4s ov = finish(acc[F][2 * jp] + bq[2 * jp], acc[F][2 * jp + 1] + bq[2 * jp + 1], rv[F - F0][jp]);
                __builtin_amdgcn_raw_buffer_store_b128(ov, rs_y, (int)(ok ? ybase + (unsigned)t * (unsigned)p.y_ld * 2u + (unsigned)co * 2u : 0x80000000u), 0, 0);
            }
        });
    };
    constexpr auto c0 = std::integral_constant<int, 0>(); constexpr auto cn = std::integral_constant<int, NFR>();
    if constexpr (KD::XCH) {
        direct_load(c0, cn);
        direct_store(c0, cn);
    } else {
        // the KG partial sums meet in LDS (fp32 rows [pixel][WN + 4] per wave, in rounds of <= RFR fragments), are added in the order kg = 0 .. KG - 1, then as above
        constexpr int RFRW = (140 * 1024) / (KD::NW * 16 * (WN + 4) * 4) < 7 ? (140 * 1024) / (KD::NW * 16 * (WN + 4) * 4) : 7, RFR = NFR > RFRW ? RFRW : NFR, ROUNDS = (NFR + RFR - 1) / RFR, RLD = WN + 4, RROWS = RFR * 16, OCT = WN / 8, PPS = 64 / OCT;
        st_lds_barrier();                                             // everybody has left the patch: its LDS becomes the exchange area
        float* const red = (float*)(smem + (lds0 - smem_lds));
        const int o = lane % OCT, co = co0 + o * 8;
        floatx4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && co + 7 < Cout) { bq0 = *(const floatx4*)(p.bias + co); bq1 = *(const floatx4*)(p.bias + co + 4); }
        sm_static_for<0, ROUNDS>([&](auto rc) {
            constexpr int RND = decltype(rc)::value, F0 = RND * RFR, NFRR = (NFR - F0) < RFR ? (NFR - F0) : RFR;
#pragma unroll
            for (int fi = 0; fi < NFRR; fi++)
#pragma unroll
                for (int j = 0; j < NF; j++) *(floatx4*)&red[(wave * RROWS + fi * 16 + l15) * RLD + j * 16 + g * 4] = acc[F0 + fi][j];
            st_lds_barrier();
            constexpr int ITS = (NFRR * 16 + KG * PPS - 1) / (KG * PPS);
            half8_t rv[ITS];                                          // every residual piece of the round in flight before the first sum (the partial sums are on their way through the LDS meanwhile)
#pragma unroll
            for (int it = 0; it < ITS; it++) {
                const int pp = kg * PPS + lane / OCT + it * KG * PPS, t = F0 * 16 + pp;
                const bool ok = pp < NFRR * 16 && t < PTe && co < Cout;
                rv[it] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (p.res) rv[it] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(ok ? rbase + (unsigned)t * (unsigned)p.res_ld * 2u + (unsigned)co * 2u : 0x80000000u), 0, 16));
            }
#pragma unroll
            for (int it = 0; it < ITS; it++) {
                const int pp = kg * PPS + lane / OCT + it * KG * PPS, ppc = pp < NFRR * 16 ? pp : 0, t = F0 * 16 + pp;
                const bool ok = pp < NFRR * 16 && t < PTe && co < Cout;
                floatx4 a0 = *(const floatx4*)&red[((0 * NG + ng) * RROWS + ppc) * RLD + o * 8], a1 = *(const floatx4*)&red[((0 * NG + ng) * RROWS + ppc) * RLD + o * 8 + 4];
#pragma unroll
                for (int k = 1; k < KG; k++) {
                    a0 += *(const floatx4*)&red[((k * NG + ng) * RROWS + ppc) * RLD + o * 8];
                    a1 += *(const floatx4*)&red[((k * NG + ng) * RROWS + ppc) * RLD + o * 8 + 4];
                }
                const u32x4s ov = finish(a0 + bq0, a1 + bq1, rv[it]);
                __builtin_amdgcn_raw_buffer_store_b128(ov, rs_y, (int)(ok ? ybase + (unsigned)t * (unsigned)p.y_ld * 2u + (unsigned)co * 2u : 0x80000000u), 0, 0);
            }
            if (RND + 1 < ROUNDS) st_lds_barrier();
        });
    }
}

// MASK: the tile forms this instantiation holds (bit k = form k); NW: waves of a block (4 or 8: a launch has ONE block size).  One kernel with all six 4-wave forms pays for the union of their registers (the 208-accumulator forms push the
// others' lane constants to scratch memory); a run of forms 0-2 or 3-5 only gets a kernel of its own.
template <int MASK, int NW = 4>
__global__ __launch_bounds__(NW * 64) void stage_kernel(const StArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
    __shared__ int s_info[4];
    __shared__ unsigned s_sink[64];
    const int tid = threadIdx.x;
    unsigned* const ctl = a.ctl;
    unsigned* const cnt = ctl + ST_CTL_HDR;
    if (tid == 0) {
        // Mode of this launch, decided ONCE, by the first block to arrive (a single writer: 256 compare-and-swaps on one word cost a 10 us start ramp): 1 = every one of the 256 blocks arrived within the bound and every XCD
        // holds 32 of them -> static tile assignment per XCD team, XCD-local hand-off; 2 = anything else (blocks of another process on some CUs, a partitioned device, a placement
        // that is not 32 per XCD) -> tiles are CLAIMED from a counter per phase by whichever blocks are running, every hand-off through agent-scope release / acquire.  Mode 2 needs no
        // co-residency at all: a claimed tile belongs to a running block, every phase is claimed out before the next one opens, so the earliest unfinished tile always progresses; a block
        // that becomes resident late adopts the mode it finds and claims what is left (nothing: it leaves).
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u;
        const unsigned slot = __hip_atomic_fetch_add(&ctl[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned gi = __hip_atomic_fetch_add(&ctl[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned mode = 0u;
        if (gi == 0u) {                                               // the FIRST block to arrive decides (it is running: nobody waits on a block that is not), everybody else reads the word
            int spins = 0;
            while (st_ld_sc1(&ctl[8]) < (unsigned)ST_GRID && spins < a.arrive_limit) { __builtin_amdgcn_s_sleep(1); spins++; }
            bool local = st_ld_sc1(&ctl[8]) >= (unsigned)ST_GRID;
            for (int i = 0; i < ST_NTEAM; i++) local = local && st_ld_sc1(&ctl[i]) == (unsigned)ST_TEAM;
            mode = local ? 1u : 2u;
            if (st_ld_sc1(&ctl[ST_MODE]) == 2u) mode = 2u;              // (test hook: the host preset the word — the claimed mode is forced)
            __hip_atomic_store(&ctl[ST_MODE], mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while ((mode = st_ld_sc1(&ctl[ST_MODE])) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 8 * a.arrive_limit + 1000) { __hip_atomic_fetch_or(&ctl[10], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mode = 2u; break; }   // (cannot happen while block 0 runs; flagged, not hung)
            }
        }
        s_info[0] = (int)xcc; s_info[1] = (int)slot; s_info[2] = (int)mode;
        __hip_atomic_fetch_or(&ctl[11], mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool local = s_info[2] == 1;
    const int member = s_info[1];
    const unsigned smem_lds = ws_lds_addr(st_smem), lds0 = (smem_lds + 1023u) & ~1023u;
    int tl_slot = 0;                                                  // (tiles this block has run: index of the diagnostic build's stamps)
#pragma unroll 1
    for (int ph = 0; ph < a.nphase; ph++) {
        const StPhase& p = a.ph[ph];
        const int ntiles = p.MT * p.NT, HW = p.HW;
        // mode 1: tiles member, member + 32, ... of MY team; mode 2: the next unclaimed (team, tile) of the phase
        int jj = local ? member : 0;
#pragma unroll 1
        for (;;) {
            int team, j;
            if (local) { team = s_info[0]; j = jj; jj += ST_TEAM; if (j >= ntiles) break; }
            else {
                __syncthreads();
                if (tid == 0) s_info[3] = (int)__hip_atomic_fetch_add(&ctl[ST_CLAIM + ph], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const int c = s_info[3];
                if (c >= ntiles * ST_NTEAM) break;
                team = c / ntiles; j = c - team * ntiles;
            }
            const int i0 = team * a.ipt, i1 = min(a.N, i0 + a.ipt);

            const int mt = (int)((unsigned)j / (unsigned)p.NT), nt = j - mt * p.NT;
            const int pix0 = i0 * HW + mt * p.TP, PTe = min(p.TP, i1 * HW - pix0);
            if (PTe <= 0) continue;
            const int ia = (int)sm_fastdiv((unsigned)pix0, p.hw_mul, p.hw_sh), ib = (int)sm_fastdiv((unsigned)(pix0 + PTe - 1), p.hw_mul, p.hw_sh);
            ST_TL(0);
#if ST_TOUCH
            if (p.wp && a.touch) {                                    // this wave's fourth of the tile's weights (ST_TOUCH 2: of the NEXT phase's tile with the same index, for whoever runs it)
                const int php = ST_TOUCH == 2 ? ph + 1 : ph;
                if (php < a.nphase && a.ph[php].wp) {
                    const StPhase& pt = a.ph[php];
                    const int ntp = ST_TOUCH == 2 ? j % pt.NT : nt;
                    const unsigned q4 = pt.wt_bytes >> 2;
                    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                    st_touch((const char*)pt.wp + (size_t)ntp * pt.wt_bytes + (size_t)wv * q4, min(q4, (unsigned)a.touch), ws_lds_addr(s_sink), (unsigned)(tid & 63) * 128u);
                }
            }
#endif
            // ---- wait for the tiles that produce this tile's inputs (wave 0, lane 0), called by the tile body behind its first weight requests ----
            auto sync = [&]() {
                if (tid == 0) {
                    const int deps[3] = {p.dep0, p.dep1, p.dep2};
                    bool waited = false;
                    for (int d = 0; d < 3; d++) {
                        if (deps[d] < 0) continue;
                        const unsigned want = (unsigned)a.ph[deps[d]].expect;
                        for (int im = ia; im <= ib; im++) {
                            int spins = 0;
                            while (st_ld_sc1(&cnt[deps[d] * a.cnt_ld + im]) < want) {
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > a.spin_limit) { __hip_atomic_fetch_or(&ctl[10], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            }
                            waited = true;
                        }
                    }
                    if (waited && !local) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                ST_TL(1);
            };
#define ST_CASE(k) if constexpr ((MASK >> k) & 1)
            switch (p.kind) {
            case 0: ST_CASE(0) st_tile<StR7>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 1: ST_CASE(1) st_tile<StT7>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 2: ST_CASE(2) st_tile<StE7>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 3: ST_CASE(3) st_tile<StR14>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 4: ST_CASE(4) st_tile<StE14>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 5: ST_CASE(5) st_tile<StT14>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 6: ST_CASE(6) st_tile<StE8>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            case 7: ST_CASE(7) st_tile<StR8>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            default: ST_CASE(8) st_tile<StT8>(p, pix0, PTe, nt, lds0, st_smem, smem_lds, sync, tl_slot); break;
            }
            ST_TL(4);
            // ---- publish: every wave's stores are drained, then one arrival per covered image ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (!local) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                for (int im = ia; im <= ib; im++) __hip_atomic_fetch_add(&cnt[ph * a.cnt_ld + im], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ST_TL(5);
#ifdef SM_STAGE_TL
            if (tid == 0 && tl_slot < 64) { g_st_tl[(blockIdx.x * 64 + tl_slot) * 8 + 7] = ((unsigned long long)ph << 32) | (unsigned)j | ((unsigned long long)team << 48); }
#endif
            tl_slot++;
        }
    }
#ifdef SM_STAGE_TL
    if (tid == 0) g_st_tl_n[blockIdx.x] = (unsigned)tl_slot;
#endif
    // ---- exit: the last block out resets the control words for the next launch ----
    __syncthreads();
    if (tid == 0) {
        const unsigned d = __hip_atomic_fetch_add(&ctl[9], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_info[3] = d == (unsigned)ST_GRID - 1u;
    }
    __syncthreads();
    if (s_info[3]) {
        for (int i = tid; i < a.nphase * a.cnt_ld; i += NW * 64) __hip_atomic_store(&cnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 10 || tid == ST_MODE || (tid >= ST_CLAIM && tid < ST_CLAIM + ST_MAXPH)) __hip_atomic_store(&ctl[tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (error bits [10] and the modes-seen word [11] stay for sm_stage_status)
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct KindInfo { int KG, NG, NF, NFR, TAPS, WCH, SBG, NB, NI_MAX, NW; bool XCH; };
template <class KD> constexpr KindInfo kind_info() { return {KD::KG, KD::NG, KD::NF, KD::NFR, KD::TAPS, KD::WCH, KD::SBG, KD::NB, KD::NI_MAX, KD::NW, KD::XCH}; }
const KindInfo kKinds[ST_NKINDS] = {kind_info<StR7>(), kind_info<StT7>(), kind_info<StE7>(), kind_info<StR14>(), kind_info<StE14>(), kind_info<StT14>(), kind_info<StE8>(), kind_info<StR8>(), kind_info<StT8>()};

bool overlap(const void* a, size_t an, const void* b, size_t bn) { return a && b && (const char*)a < (const char*)b + bn && (const char*)b < (const char*)a + an; }

// largest tile (pixels) <= cap that is a whole number of images, or an aligned part of one image
int tile_pixels(int HW, int cap) {
    if (HW <= cap) return cap / HW * HW;
    for (int d = cap; d >= 16; d--) if (HW % d == 0) return d;
    return 0;
}

// the tile form of one phase among the forms with `nw` waves (ph.form > 0: that form or nothing), and its tile's pixels; -1: none takes it
int pick_kind(const sm_stage_phase& ph, int ipt, int nw, int* tp_out) {
    const sm_conv_desc& d = ph.d;
    const bool dual = ph.x2 != nullptr, t3 = d.kH == 3;
    const int K2 = dual ? ph.K2 : 0, K1 = d.Cin - K2, HW = d.H * d.W;
    if (HW <= 0 || K1 <= 0 || d.Cout <= 0) return -1;
    const int team_px = ipt * HW;
    int best = -1; double best_cost = 0;
    for (int k = 0; k < ST_NKINDS; k++) {
        const KindInfo& ki = kKinds[k];
        if ((ki.TAPS == 9) != t3 || ki.NW != nw || (ph.form > 0 && ph.form != k + 1)) continue;
        const int sbch = ki.KG * ki.WCH;
        if (K1 % sbch || (dual && (K2 % sbch || !(ki.TAPS == 1 && ki.KG == 1))) || ((K1 + K2) / sbch) % ki.SBG) continue;
        int tp, ni;
        if (t3) {
            const int rs = (d.W + 1 + 7) / 8 * 8, extra = rs == d.W + 1 ? 1 : 0;
            if (HW > ki.NFR * 16) {                                   // a band of TR whole rows of one image (TR | H): patch = TR + 2 rows
                int tr = ki.NFR * 16 / d.W;
                while (tr > 0 && (d.H % tr || (tr + 2) * (rs / 8) + extra > ki.NI_MAX)) tr--;
                if (tr < 1) continue;
                tp = tr * d.W; ni = (tr + 2) * (rs / 8) + extra;
            } else {
                int ti = ki.NFR * 16 / HW;                           // whole images per tile: as many as the fragments and the patch instructions allow
                while (ti > 1 && (ti * (d.H + 1) + 1) * (rs / 8) + extra > ki.NI_MAX) ti--;
                tp = ti * HW; ni = (ti * (d.H + 1) + 1) * (rs / 8) + extra;
            }
        } else { tp = tile_pixels(HW, std::min(ki.NFR * 16, ki.NI_MAX * 8)); if (!tp) continue; ni = (tp + 7) / 8; }
        if (ni > ki.NI_MAX) continue;
        const size_t need = (size_t)ki.NB * ki.KG * (ki.WCH / 64) * (t3 ? ni : ki.NI_MAX) * 1024 + 1024 + 1024;
        const size_t red = (ki.KG > 1 || !ST_EPI_DIRECT) ? (size_t)ki.NW * std::min(ki.NFR, std::min(7, (140 * 1024) / (ki.NW * 16 * (ki.NF * 16 + 4) * 4))) * 16 * (ki.NF * 16 + 4) * 4 + 1024 : 0;
        if (std::max(need, red) > 160 * 1024 - 320) continue;
        const int mt = sm_ceil_div(team_px, tp), nt = sm_ceil_div(d.Cout, ki.NG * ki.NF * 16);
        const double cost = (double)sm_ceil_div(mt * nt, ST_TEAM) * ((double)tp + ki.NG * ki.NF * 16) * (K1 + K2) * 2 * (ki.KG > 1 ? 1.05 : 1.0);
        if (best < 0 || cost < best_cost) { best = k; best_cost = cost; *tp_out = tp; }
    }
    return best;
}

// fills `a` from the caller's phase list; returns SM_OK or an error code with the reason in smelter_last_error_string
int stage_plan(const sm_stage_phase* ph, int n, StArgs& a, bool quiet) {
#define ST_REQ(cond, ...) do { if (!(cond)) { if (!quiet) sm_set_error(__VA_ARGS__); return SM_ERR_INVALID_ARGUMENT; } } while (0)
    ST_REQ(ph && n >= 1 && n <= ST_MAXPH, "sm_stage: 1 .. %d phases, got %d", ST_MAXPH, n);
    memset(&a, 0, sizeof a);
    const int N = ph[0].d.N;
    ST_REQ(N >= 1 && N <= 4096, "sm_stage: batch %d", N);
    a.nphase = n; a.N = N; a.ipt = sm_ceil_div(N, ST_NTEAM); a.spin_limit = 1000000; a.arrive_limit = 800;   /* polls of 64 clk: ~20 us for the last block of the previous launch to make room before the launch gives up on the static mode */ a.touch = ST_TOUCH_BYTES; a.cnt_ld = (N + 15) / 16 * 16;
    size_t lds = 0;
    // block size of the launch: the 8-wave forms (two waves per SIMD) when EVERY phase has one whose tile is no smaller than its 4-wave form's — same tile, the waves' stalls overlapped:
    // the 7 x 7 stage; at 14 x 14 the 8-wave forms hold half an image where forms 4 / 5 hold a whole one, and two tiles per CU pay the per-tile chain twice (365 against 352 us).
    // A phase that names its form (form > 0) decides; form < 0 on any phase keeps the 4-wave forms
    int nw_run = 8;
    for (int i = 0; i < n; i++) {
        int tp8 = 0, tp4 = 0;
        if (ph[i].form > 0 && ph[i].form <= ST_NKINDS) { nw_run = kKinds[ph[i].form - 1].NW; break; }
        const int k8 = ph[i].form < 0 ? -1 : pick_kind(ph[i], a.ipt, 8, &tp8), k4 = pick_kind(ph[i], a.ipt, 4, &tp4);
        if (k8 < 0 || (k4 >= 0 && (long)tp8 * kKinds[k8].NG * kKinds[k8].NF < (long)tp4 * kKinds[k4].NG * kKinds[k4].NF)) nw_run = 4;
    }
    struct Range { const void* p; size_t n; };
    std::vector<Range> ys(n);
    for (int i = 0; i < n; i++) {
        const sm_conv_desc& d = ph[i].d; StPhase& q = a.ph[i];
        const bool dual = ph[i].x2 != nullptr;
        const int K2 = dual ? ph[i].K2 : 0, K1 = d.Cin - K2;
        ST_REQ(d.N == N && d.groups == 1 && d.dH == 1 && d.dW == 1 && d.sH == 1 && d.sW == 1 && d.Ho == d.H && d.Wo == d.W && !d.x_ups && !d.y_sH && !d.y_sW && !d.y_H,
               "sm_stage: phase %d is not a stride-1, same-size, dense convolution", i);
        const bool t3 = d.kH == 3 && d.kW == 3 && d.pt == 1 && d.pl == 1 && d.pb == 1 && d.pr == 1, t1 = d.kH == 1 && d.kW == 1 && d.pt == 0 && d.pl == 0 && d.pb == 0 && d.pr == 0;
        ST_REQ(t1 || t3, "sm_stage: phase %d: 1x1 / no padding or 3x3 / padding 1 only", i);
        ST_REQ(d.act >= SM_ACT_NONE && d.act <= SM_ACT_PRELU, "sm_stage: phase %d: activation %d (none, ReLU, leaky / PReLU only)", i, d.act);
        ST_REQ(!(t3 && dual) && K1 > 0, "sm_stage: phase %d: a second source needs a 1x1 phase", i);
        const int HW = d.H * d.W, x_ld = d.x_ld ? d.x_ld : K1, y_ld = d.y_ld ? d.y_ld : d.Cout, res_ld = d.res_ld ? d.res_ld : d.Cout, x2_ld = dual ? (ph[i].x2_ld ? ph[i].x2_ld : K2) : 0;
        ST_REQ((d.Cout & 7) == 0 && (x_ld & 7) == 0 && (y_ld & 7) == 0 && (res_ld & 7) == 0 && (x2_ld & 7) == 0 && !((uintptr_t)ph[i].x & 15) && !((uintptr_t)ph[i].y & 15) && !((uintptr_t)ph[i].w_ohwi & 15) &&
               !((uintptr_t)ph[i].w_packed & 15) && !((uintptr_t)ph[i].bias & 15) && !((uintptr_t)ph[i].residual & 15) && !((uintptr_t)ph[i].x2 & 15) && ph[i].x && ph[i].y && ph[i].w_ohwi, "sm_stage: phase %d: 16-byte aligned operands, channel counts and strides % 8", i);
        const long xb = (long)N * HW * x_ld * 2, yb = (long)N * HW * y_ld * 2, rb = ph[i].residual ? (long)N * HW * res_ld * 2 : 0, wb = (long)d.Cout * d.kH * d.kW * d.Cin * 2;
        const long x2b = dual ? (long)N * ph[i].x2_H * ph[i].x2_W * x2_ld * 2 : 0;
        ST_REQ(xb < 0x7fff0000L && yb < 0x7fff0000L && rb < 0x7fff0000L && wb < 0x7fff0000L && x2b < 0x7fff0000L, "sm_stage: phase %d: a tensor of 2 GB or more", i);
        if (dual) ST_REQ(ph[i].x2_stride >= 1 && (long)(d.H - 1) * ph[i].x2_stride < ph[i].x2_H && (long)(d.W - 1) * ph[i].x2_stride < ph[i].x2_W && K2 > 0, "sm_stage: phase %d: second source geometry", i);
        // ---- kind: the cheapest by bytes a CU pulls through its L1 (tiles per CU x (pixels + couts) x K x 2), among the forms of the run's block size ----
        int best_tp = 0;
        const int team_px = a.ipt * HW, best = pick_kind(ph[i], a.ipt, nw_run, &best_tp);
        ST_REQ(best >= 0, "sm_stage: phase %d (%dx%d, %d -> %d channels, %dx%d kernel): no tile form takes it", i, d.H, d.W, d.Cin, d.Cout, d.kH, d.kW);
        const KindInfo& ki = kKinds[best];
        q.x = (const half_t*)ph[i].x; q.x2 = (const half_t*)ph[i].x2; q.w = (const half_t*)ph[i].w_ohwi; q.wp = (const half_t*)ph[i].w_packed; q.bias = ph[i].bias; q.res = (const half_t*)ph[i].residual; q.y = (half_t*)ph[i].y;
        q.kind = best; q.H = d.H; q.W = d.W; q.HW = HW; q.K1 = K1; q.K = d.kH * d.kW * K1 + K2; q.Cout = d.Cout;
        q.x_ld = x_ld; q.x2_ld = x2_ld; q.res_ld = res_ld; q.y_ld = y_ld; q.x2_W = dual ? ph[i].x2_W : 0; q.x2_HW = dual ? ph[i].x2_H * ph[i].x2_W : 0; q.x2_s = dual ? ph[i].x2_stride : 1;
        q.act = d.act; q.slope = d.act == SM_ACT_RELU ? 0.f : d.slope;
        q.TP = best_tp; q.TI = t3 ? best_tp / HW : 0; q.MT = sm_ceil_div(team_px, best_tp); q.NT = sm_ceil_div(d.Cout, ki.NG * ki.NF * 16);
        const int sbch = ki.KG * ki.WCH;
        q.nsb1 = K1 / sbch; q.nsb = (K1 + K2) / sbch;
        q.RS = t3 ? (d.W + 1 + 7) / 8 * 8 : 8;
        q.NI = !t3 ? (best_tp + 7) / 8 : ((best_tp < HW ? best_tp / d.W + 2 : q.TI * (d.H + 1) + 1) * (q.RS / 8) + (q.RS == d.W + 1 ? 1 : 0));   // (3x3: patch rows x 8-slot groups; a band of rows has one halo row above and one below)
        q.x_bytes = (unsigned)xb; q.x2_bytes = (unsigned)x2b; q.res_bytes = (unsigned)rb; q.y_bytes = (unsigned)yb; q.wt_bytes = (unsigned)(ki.NG * ki.NF * 16 * q.K * 2);
        sm_fastdiv_make((unsigned)HW, &q.hw_mul, &q.hw_sh); sm_fastdiv_make((unsigned)d.W, &q.w_mul, &q.w_sh); sm_fastdiv_make((unsigned)(d.H + 1), &q.h1_mul, &q.h1_sh);
        sm_fastdiv_make((unsigned)(q.RS / 8), &q.kpr_mul, &q.kpr_sh);
        // arrivals per image: every cout tile of every pixel tile that overlaps the image
        q.expect = q.NT * (HW > best_tp ? HW / best_tp : 1);
        lds = std::max(lds, std::max((size_t)ki.NB * ki.KG * (ki.WCH / 64) * (t3 ? q.NI : ki.NI_MAX) * 1024 + 1024, (ki.KG > 1 || !ST_EPI_DIRECT) ? (size_t)ki.NW * std::min(ki.NFR, std::min(7, (140 * 1024) / (ki.NW * 16 * (ki.NF * 16 + 4) * 4))) * 16 * (ki.NF * 16 + 4) * 4 : (size_t)0) + 1024);
        ys[i] = {ph[i].y, (size_t)yb};
        // ---- producers: the LATEST earlier phase whose output holds the operand ----
        q.dep0 = q.dep1 = q.dep2 = -1;
        for (int e = 0; e < i; e++) {
            if (overlap(ph[i].x, (size_t)xb, ys[e].p, ys[e].n)) q.dep0 = e;
            if (dual && overlap(ph[i].x2, (size_t)x2b, ys[e].p, ys[e].n)) q.dep1 = e;
            if (ph[i].residual && overlap(ph[i].residual, (size_t)rb, ys[e].p, ys[e].n)) q.dep2 = e;
        }
        // a producer hands over whole images: its image grid must be this phase's (same N is given; a second source at another resolution is still per image)
    }
    // ---- no phase may overwrite what another phase of the run reads or writes (images advance independently: there is no run-wide order between phases) ----
    for (int i = 0; i < n; i++)
        for (int e = 0; e < n; e++) {
            if (e == i) {
                const sm_conv_desc& d = ph[i].d; const int HW = d.H * d.W; const bool dual = ph[i].x2 != nullptr; const int K2 = dual ? ph[i].K2 : 0, K1 = d.Cin - K2;
                ST_REQ(!overlap(ys[i].p, ys[i].n, ph[i].x, (size_t)N * HW * (d.x_ld ? d.x_ld : K1) * 2) && !overlap(ys[i].p, ys[i].n, ph[i].residual, ph[i].residual ? (size_t)N * HW * (d.res_ld ? d.res_ld : d.Cout) * 2 : 0) &&
                       !overlap(ys[i].p, ys[i].n, ph[i].x2, dual ? (size_t)N * ph[i].x2_H * ph[i].x2_W * (ph[i].x2_ld ? ph[i].x2_ld : K2) * 2 : 0), "sm_stage: phase %d works in place", i);
                continue;
            }
            const sm_conv_desc& d = ph[e].d;
            const int HW = d.H * d.W;
            const bool dual = ph[e].x2 != nullptr; const int K2 = dual ? ph[e].K2 : 0, K1 = d.Cin - K2;
            const size_t xb = (size_t)N * HW * (d.x_ld ? d.x_ld : K1) * 2, rb = ph[e].residual ? (size_t)N * HW * (d.res_ld ? d.res_ld : d.Cout) * 2 : 0;
            const size_t x2b = dual ? (size_t)N * ph[e].x2_H * ph[e].x2_W * (ph[e].x2_ld ? ph[e].x2_ld : K2) * 2 : 0;
            const bool reads = overlap(ys[i].p, ys[i].n, ph[e].x, xb) || overlap(ys[i].p, ys[i].n, ph[e].x2, x2b) || overlap(ys[i].p, ys[i].n, ph[e].residual, rb);
            ST_REQ(!(reads && e < i), "sm_stage: phase %d overwrites an input of phase %d", i, e);
            ST_REQ(!(reads && e > i && a.ph[e].dep0 != i && a.ph[e].dep1 != i && a.ph[e].dep2 != i), "sm_stage: phase %d's output is overwritten before phase %d reads it", i, e);
            ST_REQ(!overlap(ys[i].p, ys[i].n, ys[e].p, ys[e].n), "sm_stage: phases %d and %d write overlapping tensors", i, e);
        }
    for (int i = 1; i < n; i++) ST_REQ(kKinds[a.ph[i].kind].NW == kKinds[a.ph[0].kind].NW, "sm_stage: phase %d's tile form has %d waves, phase 0's %d (one block size per launch)", i, kKinds[a.ph[i].kind].NW, kKinds[a.ph[0].kind].NW);
    a.lds_bytes = (int)lds;
    ST_REQ(lds <= 160 * 1024 - 320, "sm_stage: %zu bytes of LDS", lds);   // (+ 272 bytes of static LDS)
    return SM_OK;
#undef ST_REQ
}

}  // namespace

// 1 if the current device runs a launch in its fast mode when nothing else occupies it: a gfx950 / gfx942 part in its full configuration — 256 compute units in 8 XCDs, so that block b
// lands on XCD b % 8 and the 32 blocks of a team share that XCD's L2 (the hand-off of mode 1 relies on exactly that; ADVICE r5: a partitioned or harvested device has the unit
// count of a smaller multiple and another block -> XCD map).  The executor keeps its layers apart otherwise.
extern "C" int sm_stage_device_fast(void) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const bool arch = !strncmp(prop.gcnArchName, "gfx950", 6) || !strncmp(prop.gcnArchName, "gfx942", 6);
    return arch && prop.multiProcessorCount == SM_NUM_XCD * 32 && prop.multiProcessorCount >= ST_GRID ? 1 : 0;
}
extern "C" int sm_stage_supported(const sm_stage_phase* ph, int32_t n) { StArgs a; return stage_plan(ph, n, a, true) == SM_OK ? 1 : 0; }
extern "C" size_t sm_stage_ctl_bytes(const sm_stage_phase* ph, int32_t n) {
    if (!ph || n < 1) return 0;
    return (size_t)(ST_CTL_HDR + (size_t)ST_MAXPH * ((ph[0].d.N + 15) / 16 * 16)) * 4;
}
extern "C" int sm_stage_phase_kind(const sm_stage_phase* ph, int32_t n, int32_t i) { StArgs a; if (stage_plan(ph, n, a, true) != SM_OK || i < 0 || i >= n) return -1; return a.ph[i].kind; }
// phase i's weights in the fragment order its tile form reads: [cout tile][ng][kg][super-block][unit = (tap, 32-channel piece)][fragment j][lane (l15, g)][8 halfs],
// element e of lane (l15, g) = w[cout(tile, ng, j, l15)][tap][channel sb SBCH + kg WCH + piece 32 + 8 g + e]; couts beyond Cout are zero rows.  Host memory on both sides.
extern "C" size_t sm_stage_packed_weight_bytes(const sm_stage_phase* ph, int32_t n, int32_t i) {
    StArgs a;
    if (stage_plan(ph, n, a, true) != SM_OK || i < 0 || i >= n) return 0;
    const KindInfo& ki = kKinds[a.ph[i].kind];
    return (size_t)a.ph[i].NT * ki.NG * ki.NF * 16 * (size_t)a.ph[i].K * 2;
}
extern "C" int sm_stage_pack_weights(const sm_stage_phase* ph, int32_t n, int32_t i, const void* w_ohwi_host, void* dst_host) {
    StArgs a;
    const int rc = stage_plan(ph, n, a, false);
    if (rc) return rc;
    SM_REQUIRE(i >= 0 && i < n && w_ohwi_host && dst_host, "sm_stage_pack_weights: phase %d of %d", i, n);
    const StPhase& q = a.ph[i]; const KindInfo& ki = kKinds[q.kind];
    const int KG = ki.KG, NG = ki.NG, NF = ki.NF, WCH = ki.WCH, QPW = WCH / 32, UPS = ki.TAPS * QPW, SBCH = KG * WCH, BN = NG * NF * 16;
    const uint16_t* w = (const uint16_t*)w_ohwi_host; uint16_t* dst = (uint16_t*)dst_host;
    size_t o = 0;
    for (int nt = 0; nt < q.NT; nt++)
        for (int ng = 0; ng < NG; ng++)
            for (int kg = 0; kg < KG; kg++)
                for (int sb = 0; sb < q.nsb; sb++)
                    for (int ul = 0; ul < UPS; ul++)
                        for (int j = 0; j < NF; j++)
                            for (int lane = 0; lane < 64; lane++) {
                                const int l15 = lane & 15, g = lane >> 4, co0 = nt * BN + ng * NF * 16;
                                const int co = ki.XCH ? co0 + (j >> 1) * 32 + (l15 >> 2) * 8 + (j & 1) * 4 + (l15 & 3) : co0 + j * 16 + l15;
                                const size_t k = (size_t)(ul / QPW) * q.K1 + (size_t)sb * SBCH + kg * WCH + (ul % QPW) * 32 + g * 8;
                                if (co < q.Cout) memcpy(dst + o, w + (size_t)co * q.K + k, 16); else memset(dst + o, 0, 16);
                                o += 8;
                            }
    return SM_OK;
}
extern "C" int sm_stage_fwd(const sm_stage_phase* ph, int32_t n, void* ctl, sm_stream_t stream) {
    StArgs a;
    const int rc = stage_plan(ph, n, a, false);
    if (rc) return rc;
    SM_REQUIRE(ctl && !((uintptr_t)ctl & 15), "sm_stage_fwd: ctl = %p (sm_stage_ctl_bytes of zeroed device memory, 16-byte aligned)", ctl);
    a.ctl = (unsigned*)ctl;
    // (no co-residency requirement: a device that cannot hold the 256 blocks together — fewer CUs, a compute partition, other work — completes the launch in the claimed mode)
    int mask = 0;
    for (int i = 0; i < n; i++) mask |= 1 << a.ph[i].kind;
    if (kKinds[a.ph[0].kind].NW == 8) {
        hipLaunchKernelGGL((stage_kernel<0x1c0, 8>), dim3(ST_GRID), dim3(512), (size_t)a.lds_bytes, sm_s(stream), a);   // forms 6, 7, 8
    }
    else if ((mask & ~0x7) == 0) hipLaunchKernelGGL(stage_kernel<0x7>, dim3(ST_GRID), dim3(256), (size_t)a.lds_bytes, sm_s(stream), a);
    else if ((mask & ~0x38) == 0) hipLaunchKernelGGL(stage_kernel<0x38>, dim3(ST_GRID), dim3(256), (size_t)a.lds_bytes, sm_s(stream), a);
    else hipLaunchKernelGGL(stage_kernel<0x3f>, dim3(ST_GRID), dim3(256), (size_t)a.lds_bytes, sm_s(stream), a);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
// error bit 1: a hand-off wait ran out.  mode: modes seen since the last call (1 = static + XCD-local, 2 = claimed + fenced, 3 = both).  Synchronises the stream.  Clears the words.
extern "C" int sm_stage_status(void* ctl, sm_stream_t stream, int32_t* error_bits, int32_t* mode) {
    unsigned h[2] = {0, 0};
    SM_HIP_CHECK(hipMemcpyAsync(h, (char*)ctl + 10 * 4, 8, hipMemcpyDeviceToHost, sm_s(stream)));
    SM_HIP_CHECK(hipStreamSynchronize(sm_s(stream)));
    SM_HIP_CHECK(hipMemsetAsync((char*)ctl + 10 * 4, 0, 8, sm_s(stream)));
    if (error_bits) *error_bits = (int32_t)h[0];
    if (mode) *mode = (int32_t)h[1];
    return SM_OK;
}
#ifdef SM_STAGE_TL
extern "C" int sm_stage_timeline_read(unsigned long long* dst, unsigned* counts) {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_st_tl), sizeof(unsigned long long) * ST_GRID * 64 * 8) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_st_tl_n), sizeof(unsigned) * ST_GRID) == hipSuccess ? 0 : -1;
}
#endif
