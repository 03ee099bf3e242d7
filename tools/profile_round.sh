#!/bin/bash
# One-shot profile of the default benchmark for profiles/: kernel-trace statistics, then (separate runs, --kernel-trace only
# besides the counter, as the pool requires) the FETCH_SIZE and WRITE_SIZE PMC passes.
#   usage (on a GPU box, from the repo root):  bash tools/profile_round.sh r02_a            -> gpurun_out/r02_a_*
set -e -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT/prof_$TAG
cd /tmp && export TMPDIR=/tmp
REPO=$OLDPWD
STEPS=${STEPS:-10}; WARM=${WARM:-3}; BATCH=${BATCH:-96}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -o stats -- python3 $REPO/bench.py --lean --batch $BATCH --steps $STEPS --warmup $WARM > $OUT/${TAG}_bench_under_profiler.json 2> $OUT/${TAG}_stats.err
python3 $REPO/tools/prof_summary.py $(find $OUT/prof_$TAG/stats -name '*kernel_stats.csv' | head -1) $((STEPS + WARM + 2)) $OUT/${TAG}_default_bench_b${BATCH}_kernel_stats.csv > /dev/null
# the framework's own small kernels by call site (previous / next kernel), per step
python3 $REPO/tools/trace_glue.py $(find $OUT/prof_$TAG/stats -name '*kernel_trace.csv' | head -1) $((STEPS + WARM + 2)) > $OUT/${TAG}_glue.txt 2>&1 || true
if [ "${PMC:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_$TAG/fetch -o fetch -- python3 $REPO/bench.py --lean --batch $BATCH --steps 3 --warmup 2 > /dev/null 2> $OUT/${TAG}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_$TAG/write -o write -- python3 $REPO/bench.py --lean --batch $BATCH --steps 3 --warmup 2 > /dev/null 2> $OUT/${TAG}_write.err
  python3 $REPO/tools/hbm_traffic.py $(find $OUT/prof_$TAG/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/prof_$TAG/write -name '*counter_collection.csv' | head -1) \
      $OUT/${TAG}_hbm_traffic_b${BATCH}.csv "python3 bench.py --lean --batch $BATCH --steps 3 --warmup 2" > /dev/null
fi
rm -rf $OUT/prof_$TAG
head -30 $OUT/${TAG}_default_bench_b${BATCH}_kernel_stats.csv
