#!/bin/bash
# SQ counters of the fused inverted-residual kernel (rocprofv3 --pmc passes with --kernel-trace only): bash tools/pmc_mb.sh <shape index> <tag>
IDX=${1:-0}; TAG=${2:-v2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_mb_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS=$'\n'; SETS=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
"SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
"SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"
"SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
"SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH"
"SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
"SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
); unset IFS
for ctr in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p -- python $ROOT/tools/mb_one.py $IDX > $OUT/p$i.log 2>&1
done
cd $ROOT
python - $TAG <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_mb_%s/p*/**/p_counter_collection.csv" % sys.argv[1], recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mbconv" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print("%-28s per launch %.4g  (launches %d)" % (k, sum(v) / len(v), len(v)))
PY
