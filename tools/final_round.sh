#!/bin/bash
# End-of-round evidence, one GPU call: per-layer tables, fixed-cost fit, same-box A/B against the previous round (needs build/r04tree) and of the persistent runs, the runs' per-phase
# timeline (needs tools/stage_tl.sh's diagnostic library), rocprofv3 kernel stats + PMC passes + bench line per model.  Afterwards: copy gpurun_out/<tag>_* into profiles/ and run tools/doc_numbers.py.
TAG=${1:-r03}; HEAD=${2:-unrecorded}
OUT=gpurun_out
for m in r50:resnet50 mbv2:mobilenetv2 unet:unet; do
  python tools/gpu_probe.py --model ${m%%:*} --ops --check > $OUT/${TAG}_layers_${m##*:}.txt 2>&1
done
python tools/fixed_cost_fit.py --out $OUT/${TAG}_fixed_cost_fit.json --head $HEAD > $OUT/${TAG}_fit.log 2>&1
[ -d build/${PREV:-r04tree} ] && bash tools/ab_round.sh > $OUT/${TAG}_ab_same_box.txt 2>&1
bash tools/ab_stage.sh > $OUT/${TAG}_ab_stage_same_box.txt 2>&1
[ -f build/libsmelter_hip_stagetl.so ] && for s in stage4 stage3; do SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so python tools/stage_tl.py $s 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_stage_timeline.txt
for m in r50 mbv2 unet; do bash tools/profile.sh $TAG $HEAD $m > $OUT/${TAG}_profile_$m.log 2>&1; done
SMELTER_ROCTX=1 python tools/gpu_probe.py --model r50 2>&1 | grep "bs=64" > $OUT/${TAG}_roctx_smoke.txt
grep "bs=" $OUT/${TAG}_layers_*.txt; tail -12 $OUT/${TAG}_fixed_cost_fit.json; cat $OUT/${TAG}_ab_same_box.txt | cut -c1-70; cat $OUT/${TAG}_roctx_smoke.txt
