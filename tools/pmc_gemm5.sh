#!/bin/bash
# SQ counters of one GEMM shape through gemm5.hip and gemm3.hip (diag build, CTCLIP_GEMM5_MINK): LDS bank conflicts, issue stalls,
# matrix-pipe busy cycles.   usage (GPU box, repo root): SHAPE="sq4096" bash tools/pmc_gemm5.sh
set -e -o pipefail
OUT=$PWD/gpurun_out; REPO=$PWD
export B=${B:-32} ONLY="${SHAPE:-sq4096}"
export CTCLIP_HIP_LIB=$REPO/ct-clip-ut_amd/ctclip_hip/libctclip_hip_diag.so
cd /tmp && export TMPDIR=/tmp
for MINK in 128 1000000; do
  export CTCLIP_GEMM5_MINK=$MINK
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    D=$OUT/prof_g5; rm -rf $D
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $D -o f -- python3 $REPO/tools/bench_gemm.py > $OUT/g5_pmc_bench.txt 2> $OUT/g5_pmc.err
    python3 - $(find $D -name '*counter_collection.csv' | head -1) <<'PY'
import csv, collections, sys
tot, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:50]
    if "gemm" not in k: continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in tot:
    print(k, "launches", len(n[k]))
    for c, v in sorted(tot[k].items()):
        print(f"    {c:28s} {v / len(n[k]):16.0f} per launch")
PY
    rm -rf $D
  done
done
