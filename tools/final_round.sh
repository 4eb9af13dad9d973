#!/bin/bash
# End-of-round evidence, one GPU call: per-layer tables, fixed-cost fit, same-box A/B against the previous round (needs a built git worktree of its head: build/r05tree, PREV=...),
# of the persistent runs and of the whole-block launches (pass 3l), full-size parity of the final binary, the admission of the runs over five compiles, the per-phase timelines
# (need tools/stage_tl.sh's / tools/bneck_tl.sh's diagnostic libraries), rocprofv3 kernel stats + PMC passes + bench line per model.
# Afterwards: copy gpurun_out/<tag>_* into profiles/, run tools/doc_numbers.py, delete the diagnostic libraries and the worktree.
TAG=${1:-r03}; HEAD=${2:-unrecorded}
OUT=gpurun_out
for m in r50:resnet50 mbv2:mobilenetv2 unet:unet; do
  python tools/gpu_probe.py --model ${m%%:*} --ops --check > $OUT/${TAG}_layers_${m##*:}.txt 2>&1
done
python tools/fixed_cost_fit.py --out $OUT/${TAG}_fixed_cost_fit.json --head $HEAD > $OUT/${TAG}_fit.log 2>&1
export PREV=${PREV:-r05tree}
[ -d build/$PREV ] && bash tools/ab_round.sh > $OUT/${TAG}_ab_same_box.txt 2>&1
bash tools/ab_stage.sh > $OUT/${TAG}_ab_stage_same_box.txt 2>&1
# pass 3l (whole bottleneck blocks as one launch) against the same binary with SMELTER_BNECK=0: three alternating rounds
for i in 1 2 3; do
  SMELTER_BNECK=0 python tools/gpu_probe.py --model r50 2>&1 | grep " bs=" | sed 's/^/blocks as reduce + (3x3 >> expand)   /'
  python tools/gpu_probe.py --model r50 2>&1 | grep " bs=" | sed 's/^/whole blocks, one launch (default)   /'
done > $OUT/${TAG}_ab_bneck_same_box.txt 2>&1
# the admission of the persistent runs, five compiles in one process
python tools/run_admission.py --compiles 5 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_run_admission.txt
# full-size parity of THIS binary (tests/test_gpu_full.py writes gpurun_out/parity_full_size.json)
rm -f $OUT/parity_full_size.json
python -m pytest tests/test_gpu_full.py -q -k "tight_parity_at_full_size" > $OUT/${TAG}_parity_full_size.log 2>&1; cp $OUT/parity_full_size.json $OUT/${TAG}_parity_full_size.json
# the whole-block kernels alone: microbenchmark (one launch against reduce + (3x3 >> expand)), timelines, L2 / fabric / SQ counters
{ for g in "--hw 56 --c0 256 --cm 64" "--hw 28 --c0 512 --cm 128"; do echo "== tools/bneck_bench.py $g"; python tools/bneck_bench.py $g 2>&1 | grep -v amdgpu.ids; done; } > $OUT/${TAG}_bneck_bench.txt
[ -f build/libsmelter_hip_bntl.so ] && { SMELTER_LIB_PATH=build/libsmelter_hip_bntl.so python tools/bneck_tl.py; SMELTER_LIB_PATH=build/libsmelter_hip_bntl.so python tools/bneck128_tl.py; } 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_bneck_timeline.txt
bash tools/pmc_bneck.sh one > $OUT/${TAG}_pmc_bneck.txt 2>&1
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
