#!/bin/bash
# Where do the tubelet kernels' memory-side read requests come from?  L2 line misses against fabric read requests (all / 32-byte) of
# the unfused gather + LayerNorm kernel (patch_ln_fwd_fast) and of the fused embedding (patch_gemm_fwd_kernel), tools/bench_patch_fused.py
# at B pairs: separate rocprofv3 passes (--kernel-trace only besides the counters).
#   usage (GPU box, repo root): B=32 bash tools/pmc_patch_fetch.sh   -> gpurun_out/patch_fetch_b<B>.txt
set -e -o pipefail
OUT=$PWD/gpurun_out
REPO=$PWD
export B=${B:-32}
cd /tmp && export TMPDIR=/tmp
: > $OUT/patch_fetch_b$B.txt
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf $OUT/prof_pf
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_pf -o c -- python3 $REPO/tools/bench_patch_fused.py > /dev/null 2> $OUT/patch_fetch.err || { echo "# counters '$C' not collected" >> $OUT/patch_fetch_b$B.txt; continue; }
  python3 - "$(find $OUT/prof_pf -name '*counter_collection.csv' | head -1)" >> $OUT/patch_fetch_b$B.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    for k in ("patch_ln_fwd_fast", "patch_gemm_fwd_kernel", "patch_wgrad_kernel"):
        if k in name:
            a = agg[(k, r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:24s} {c:32s} launches {n:3d}  per launch {v / n:.5g}")
PY
done
rm -rf $OUT/prof_pf
cat $OUT/patch_fetch_b$B.txt
