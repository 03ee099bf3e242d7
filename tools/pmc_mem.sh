#!/bin/bash
# Memory-side counters of the GEMM kernels (this repo's gemm3 / gemm5 and the vendor library's kernel) on one shape: L1 -> L2 read
# requests and their mean latency, L1 stalls, L2 hit rate, fabric reads.   usage (GPU box, repo root): SHAPE="sq4096" bash tools/pmc_mem.sh
set -e -o pipefail
OUT=$PWD/gpurun_out; REPO=$PWD
export B=${B:-32} R=2 ONLY="${SHAPE:-sq4096}"
export CTCLIP_HIP_LIB=$REPO/ct-clip-ut_amd/ctclip_hip/libctclip_hip_diag.so
cd /tmp && export TMPDIR=/tmp
for MINK in 128 1000000; do
  export CTCLIP_GEMM5_MINK=$MINK
  for SET in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
             "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
             "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
    D=$OUT/prof_pm; rm -rf $D
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $D -o f -- python3 $REPO/tools/gemm_vs_vendor.py > $OUT/pm_bench.txt 2> $OUT/pm.err || { tail -5 $OUT/pm.err; continue; }
    python3 - $(find $D -name '*counter_collection.csv' | head -1) $MINK <<'PY'
import csv, collections, sys
tot, n, dur = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set), collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    if not ("gemm" in k or "Cijk" in k): continue
    if int(sys.argv[2]) > 1000 and "Cijk" in k: continue          # the vendor rows once
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in tot:
    print(k, "| launches", len(n[k]), "|", "  ".join(f"{c} {v / len(n[k]):.4g}" for c, v in sorted(tot[k].items())))
PY
    rm -rf $D
  done
done
