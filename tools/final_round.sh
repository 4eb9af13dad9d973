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
[ -f build/libsmelter_hip_stagetl.so ] && for s in "stage4" "stage4 4w" "stage3"; do SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so python tools/stage_tl.py $s 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_stage_timeline.txt
for m in r50 mbv2 unet; do bash tools/profile.sh $TAG $HEAD $m > $OUT/${TAG}_profile_$m.log 2>&1; done
SMELTER_ROCTX=1 python tools/gpu_probe.py --model r50 2>&1 | grep "bs=64" > $OUT/${TAG}_roctx_smoke.txt
# where a launch's time goes: instruction mix of every launch of the ResNet-50 step, SQ / TCP / TCC counters of the persistent runs alone, what the memory system delivers to 256 streaming CUs
bash tools/pmc_mix.sh r50 > $OUT/pmc_mix.log 2>&1; { echo "# tools/pmc_mix.sh r50 (rocprofv3 --pmc, two counters per pass, --kernel-trace only), binary $HEAD: instruction mix per launch of the last encode() step of ResNet-50 bs 64."; echo "# x/m = wave-instructions of that class per MFMA instruction; wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue% = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES)"; cat $OUT/pmc_mix_r50.txt; } > $OUT/${TAG}_pmc_mix.txt
for s in stage4 stage3; do bash tools/pmc_stage.sh $s > $OUT/pmc_$s.log 2>&1; done
{ echo "# tools/pmc_stage.sh (rocprofv3 --pmc, two counters per pass), binary $HEAD: the persistent run alone, ResNet-50 bs 64 stage shapes, per launch.  SQ_* wave counters are in units of 4 cycles (SQ_WAVE_CYCLES x 4 = 1024 waves x the launch's cycles)."; cat $OUT/pmc_stage_stage4.txt $OUT/pmc_stage_stage3.txt; } > $OUT/${TAG}_pmc_stage.txt
[ -x build/fabric_bw ] && { echo "# tools/micro/fabric_bw.hip on one MI355X: bytes per second that reach the CUs when EVERY CU streams at once, by the level that holds the data (see the file header)"; timeout 300 build/fabric_bw; } > $OUT/${TAG}_fabric_bw.txt 2>&1
rm -rf $OUT/prof_kt $OUT/pmc        # (raw traces: tens of MB; the summaries above are what travels back)
grep "bs=" $OUT/${TAG}_layers_*.txt; tail -12 $OUT/${TAG}_fixed_cost_fit.json; cat $OUT/${TAG}_ab_same_box.txt | cut -c1-70; cat $OUT/${TAG}_roctx_smoke.txt
