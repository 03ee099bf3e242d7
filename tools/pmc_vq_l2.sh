#!/bin/bash
# L2 hit / miss counts of the VQ nearest-code sweep (one code group: the product form), summed over the XCDs and per channel group:
# separate rocprofv3 passes (--kernel-trace only besides the counters), tools/bench_vq.py at B pairs.
#   usage (GPU box, repo root): B=64 bash tools/pmc_vq_l2.sh   -> gpurun_out/vq_l2_b<B>.txt
set -e -o pipefail
OUT=$PWD/gpurun_out
REPO=$PWD
export B=${B:-64} VQ_GROUPS=1
cd /tmp && export TMPDIR=/tmp
: > $OUT/vq_l2_b$B.txt
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf $OUT/prof_vql2
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_vql2 -o c -- python3 $REPO/tools/bench_vq.py > /dev/null 2> $OUT/vq_l2.err || { echo "# counters '$C' not collected" >> $OUT/vq_l2_b$B.txt; continue; }
  python3 - "$(find $OUT/prof_vql2 -name '*counter_collection.csv' | head -1)" >> $OUT/vq_l2_b$B.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "vq_topk3" not in r["Kernel_Name"]:
        continue
    a = agg[r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"vq_topk3_kernel  {k:32s} launches {n:3d}  per launch {v / n:.4g}")
PY
done
rm -rf $OUT/prof_vql2
cat $OUT/vq_l2_b$B.txt
