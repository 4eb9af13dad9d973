#!/bin/bash
# SQ / LDS counters of ONE conv shape + tile config (rocprofv3 --pmc passes with --kernel-trace only): bash tools/pmc_conv.sh <shape> <cfg>
SHAPE=${1:-64,14,14,1024,256,1,1,0}; CFG=${2:-32}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_conv; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS=$'\n'; SETS=(
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
"SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
"SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
"TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
"TCC_HIT_sum TCC_MISS_sum"
"TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum"
"TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
"TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
"TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum"
"TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
"TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"
"TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
); unset IFS
if [ -n "$PMC_ONLY_LAST" ]; then SETS=("${SETS[@]: -$PMC_ONLY_LAST}"); fi
for ctr in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p -- python $ROOT/tools/conv_bench.py --shape $SHAPE --cfg $CFG --iters 3 > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_conv/p*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"] or "conv3x3" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print("%-28s per launch %.4g  (launches %d)" % (k, sum(v) / len(v), len(v)))
PY
