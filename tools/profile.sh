#!/bin/bash
# Reproduces the profiles committed under profiles/ (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats over `bench.py --steps 20`           -> gpurun_out/prof_kt/
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC hit+miss / MFMA busy, each in its OWN pass with --kernel-trace only
#      (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE do not fit one pass)          -> gpurun_out/pmc/pass*/
# Summaries: tools/rocprof_summary.py, tools/pmc_summary.py (run here or in the build container).
set -u
TAG=${1:-r03}
HEAD=${2:-unrecorded}
MODEL=${3:-r50}                      # r50 | mbv2 | unet (bench.py --model); file names get a _<model> suffix for the other two
SUF=""; [ "$MODEL" != "r50" ] && SUF="_$MODEL"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --model $MODEL --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
rm -rf $OUT/prof_kt $OUT/pmc
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o r -- $BENCH > $OUT/prof_kt.log 2>&1
i=0
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" SQ_VALU_MFMA_BUSY_CYCLES; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc/pass$i -o p -- python $ROOT/bench.py --model $MODEL --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/pmc_pass$i.log 2>&1
done
cd $ROOT
python tools/rocprof_summary.py $(ls $OUT/prof_kt/*/r_results.db $OUT/prof_kt/r_results.db 2>/dev/null | head -1) --steps 20 --model $MODEL > $OUT/${TAG}_kernel_stats${SUF}.txt 2>&1
python tools/pmc_summary.py $(find $OUT/pmc -name "p_counter_collection.csv" | sort) --head $HEAD --out $OUT/${TAG}_pmc_traffic${SUF}.json > $OUT/${TAG}_pmc_per_op${SUF}.txt 2>&1
cp $OUT/${TAG}_pmc_traffic${SUF}.json $ROOT/profiles/ 2>/dev/null   # bench.py quotes the newest profiles/r*_pmc_traffic<suffix>.json as roofline.traffic
python bench.py --model $MODEL --steps 50 --warmup 10 > $OUT/${TAG}_bench${SUF}.json 2> $OUT/${TAG}_bench${SUF}.err
tail -3 $OUT/${TAG}_kernel_stats${SUF}.txt; tail -2 $OUT/${TAG}_pmc_per_op${SUF}.txt; tail -1 $OUT/${TAG}_bench${SUF}.json | cut -c1-300
