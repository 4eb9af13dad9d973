#!/bin/bash
# instruction mix of every launch of one model's step (rocprofv3 --pmc passes, two counters each, --kernel-trace only): bash tools/pmc_mix.sh [r50|mbv2|unet]
MODEL=${1:-r50}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_mix_$MODEL; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=$'\n'; SETS=("SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"); unset IFS
i=0
for ctr in "${SETS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p -- python $ROOT/bench.py --model $MODEL --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/p$i.log 2>&1
done
cd $ROOT
python tools/pmc_mix.py $(find $OUT -name "p_counter_collection.csv" | sort) | tee gpurun_out/pmc_mix_$MODEL.txt
rm -rf $OUT        # (the raw passes are tens of MB: only the summary travels back)
