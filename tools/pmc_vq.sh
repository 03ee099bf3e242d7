#!/bin/bash
# HBM-side traffic of the VQ sweep per code-group count: FETCH_SIZE and WRITE_SIZE passes of tools/bench_vq.py (separate runs,
# --kernel-trace only besides the counter).   usage (GPU box, repo root): B=64 bash tools/pmc_vq.sh  -> gpurun_out/vq_traffic_g*.csv
set -e -o pipefail
OUT=$PWD/gpurun_out
REPO=$PWD
export B=${B:-64}
cd /tmp && export TMPDIR=/tmp
for G in ${VQ_GROUPS_LIST:-1 2 4 8}; do
  export VQ_GROUPS=$G
  rm -rf $OUT/prof_vq_$G
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_vq_$G/fetch -o fetch -- python3 $REPO/tools/bench_vq.py > /dev/null 2> $OUT/vq_fetch_$G.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_vq_$G/write -o write -- python3 $REPO/tools/bench_vq.py > /dev/null 2> $OUT/vq_write_$G.err
  python3 $REPO/tools/hbm_traffic.py $(find $OUT/prof_vq_$G/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/prof_vq_$G/write -name '*counter_collection.csv' | head -1) \
      $OUT/vq_traffic_b${B}_g$G.csv "B=$B VQ_GROUPS=$G python3 tools/bench_vq.py" > /dev/null
  rm -rf $OUT/prof_vq_$G
  grep -i "vq_" $OUT/vq_traffic_b${B}_g$G.csv
done
