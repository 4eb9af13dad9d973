#!/bin/bash
# L2 / fabric / SQ counters of the fused bottleneck launch (rocprofv3 --pmc passes with --kernel-trace only): bash tools/pmc_bneck.sh [one|two]
FORM=${1:-one}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_bneck_$FORM; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS=$'\n'; SETS=(
"TCC_HIT_sum TCC_MISS_sum"
"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
"TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
"SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
"SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
"SQ_INSTS_VALU SQ_INSTS_MFMA"
"SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
"SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
); unset IFS
for ctr in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p -- python $ROOT/tools/bneck_bench.py --only $FORM --iters 6 > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_bneck_$FORM/p*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bneck" in k or "conv" in k:
            tot[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in tot.items():
    print(kn)
    for k, v in d.items():
        print("   %-32s per launch %.5g  (launches %d)" % (k, sum(v) / len(v), len(v)))
PY
