#!/bin/bash
# FETCH_SIZE per launch of single GEMM shapes of the step (tools/bench_gemm.py, ONLY=<shape>), against the operand bytes: how
# often is the row operand A re-read when N spans several column tiles?   usage (GPU box, repo root): B=64 bash tools/pmc_gemm_shapes.sh
set -e -o pipefail
OUT=$PWD/gpurun_out
REPO=$PWD
export B=${B:-64}
cd /tmp && export TMPDIR=/tmp
IFS=';'
for S in ${SHAPES:-kv fwd;q fwd;ff1 fwd;ff2 fwd f32;ff1 dgrad f32;ff2 dgrad;kv wgrad}; do
  export ONLY="$S"
  D=$OUT/prof_gs
  rm -rf $D
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -o f -- python3 $REPO/tools/bench_gemm.py > $OUT/gs_bench.txt 2> $OUT/gs.err
  python3 - "$S" $(find $D -name '*counter_collection.csv' | head -1) $OUT/gs_bench.txt <<'PY'
import csv, collections, sys
tot, n = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[2])):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
line = [l for l in open(sys.argv[3]) if "TFLOP" in l]
print(sys.argv[1], "|", line[0].strip() if line else "")
for k in sorted(tot, key=lambda k: -tot[k])[:3]:
    print(f"    {k:60s} launches {len(n[k]):3d}  fetch {2.0 * tot[k] / len(n[k]) * 1024 / 1e6:9.1f} MB per launch (x2-corrected)")
PY
  rm -rf $D
done
