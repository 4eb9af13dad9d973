#!/bin/bash
# SQ / TA / TCP / TCC counters of the persistent stage kernel alone (rocprofv3 --pmc passes with --kernel-trace only, two counters per pass): bash tools/pmc_stage.sh [stage4|stage3]
# Per launch of stage_kernel (256 blocks of 4 waves, one wave per SIMD): where the wave cycles go.  Output: gpurun_out/pmc_stage_<which>.txt
WHICH=${1:-stage4}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_stage_$WHICH; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=$'\n'; SETS=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
"SQ_WAIT_ANY SQ_WAIT_INST_ANY"
"SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
"SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM"
"SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
"SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
"SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU"
"SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
"SQ_INSTS_VALU SQ_INSTS_MFMA"
"SQ_INSTS_SALU SQ_INSTS_SMEM"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
"SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
"TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
"TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
"TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
"TCC_HIT_sum TCC_MISS_sum"
"TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
"TCC_TAG_STALL_sum TCC_REQ_sum"
"TD_TD_BUSY_sum TD_TC_STALL_sum"
"GRBM_GUI_ACTIVE SQ_CYCLES"
); unset IFS
i=0
for ctr in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p -- python $ROOT/tools/stage_run.py $WHICH 5 > $OUT/p$i.log 2>&1
done
cd $ROOT
python - $WHICH <<'PY' | tee gpurun_out/pmc_stage_$WHICH.txt
import csv, glob, collections, sys
which = sys.argv[1]
tot = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_stage_%s/p*/**/p_counter_collection.csv" % which, recursive=True)):
    for r in csv.DictReader(open(f)):
        if "stage_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# %s: per launch of stage_kernel (average over the launches of the pass)" % which)
for k, v in tot.items():
    print("%-36s %.6g  (launches %d)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $OUT        # (only the summary travels back)
