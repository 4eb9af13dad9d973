#!/bin/bash
# End-of-round evidence, one GPU call: per-layer tables, fixed-cost fit, same-box A/B (needs build/r03tree), rocprofv3 kernel stats + PMC passes + bench line per model.
TAG=${1:-r03}; HEAD=${2:-unrecorded}
OUT=gpurun_out
for m in r50:resnet50 mbv2:mobilenetv2 unet:unet; do
  python tools/gpu_probe.py --model ${m%%:*} --ops --check > $OUT/${TAG}_layers_${m##*:}.txt 2>&1
done
python tools/fixed_cost_fit.py --out $OUT/${TAG}_fixed_cost_fit.json --head $HEAD > $OUT/${TAG}_fit.log 2>&1
[ -d build/r03tree ] && bash tools/ab_round.sh > $OUT/${TAG}_ab_same_box.txt 2>&1
for m in r50 mbv2 unet; do bash tools/profile.sh $TAG $HEAD $m > $OUT/${TAG}_profile_$m.log 2>&1; done
SMELTER_ROCTX=1 python tools/gpu_probe.py --model r50 2>&1 | grep "bs=64" > $OUT/${TAG}_roctx_smoke.txt
grep "bs=" $OUT/${TAG}_layers_*.txt; tail -12 $OUT/${TAG}_fixed_cost_fit.json; cat $OUT/${TAG}_ab_same_box.txt | cut -c1-70; cat $OUT/${TAG}_roctx_smoke.txt
