#!/bin/bash
# where a launch's batch-independent time goes, per kernel.  For the ten most frequent (layer shape, tile configuration) pairs of the
# ResNet-50 bs=64 step (32 of its 48 launches), one launch of the SM_TIMELINE build records per-block stamps (inputs and weights evicted by a 768 MB memset first:
# SM_TIME_COLD, i.e. HBM-cold — harsher than the graph, where the input sits in the Infinity Cache); sm_conv2d_time prints, averaged over blocks:
#   the clocks between consecutive stamps (the eight labels are conv_igemm64's phases; the ps / rwl kernels stamp their own points in the same slots:
#   ps: entry | ring prologue | tiles 0..3 | drain | end;  rwl: entry | weight slice in registers | first stages ... | end),
#   "block wall avg" (entry -> exit of a block), "first start -> last end" (the grid: start skew + tail imbalance on top of a block), and "event" (HIP events
#   around the launch: + launch boundary).  The p9 / strip kernels carry no stamps: event time only.
#   make -j6 OBJDIR=build/obj_tl LIB=build/libsmelter_hip_tl.so FLAGSTAMP=build/.flags_tl EXTRA=-DSM_TIMELINE build/libsmelter_hip_tl.so   (no GPU needed)
#   bash tools/kq_tl.sh                                                                                                                       (no GPU needed)
#   bash tools/fixed_cost_breakdown.sh > profiles/r04_fixed_cost_breakdown.txt                                                                (on the GPU box)
cd "$(dirname "$0")/.."
run() {   # shape "label"
  echo "== $1   $2   x$3 launches per step"
  SM_TIME_COLD=1 python tools/conv_bench.py --lib build/libsmelter_hip_tl.so --shape "$1" --cfg "$2" --iters 20 $4 2>&1 | grep -v amdgpu.ids | sed 's/ | /\n      /g'
}
run 64,14,14,256,256,3,1,1   "p9:128x128(32x64) s3"        5
run 64,14,14,1024,256,1,1,0  "ps:128x128(64x32) s4"        5
run 64,14,14,256,1024,1,1,0  "rwl:k256 n256 s32x4 +res"    5 --res
run 64,28,28,128,512,1,1,0   "rwl:k128 n256 s32x4 +res"    3 --res
run 64,28,28,512,128,1,1,0   "rwl:k512 n128 s32x3"         3
run 64,7,7,512,512,3,1,1     "strip w4:128x32"             2
run 64,7,7,2048,512,1,1,0    "ps:128x64 s4"                2
run 64,7,7,512,2048,1,1,0    "rwl:k512 n128 s32x3 +res"    2 --res
run 64,56,56,256,64,1,1,0    "rwl:k256 n64 s64x4"          2
echo "== wave-split 3x3 (conv3x3_kq.hpp; its own stamps: tools/kq_tl.py)   x3 launches per step"
python tools/kq_tl.py --shape 64,28,28,128,128 --cfg "kq:k1n4 196x128 b2" --flush 2>&1 | grep -v amdgpu.ids
