#!/bin/bash
# A/B of gemm5.hip (one wave per SIMD) against gemm3.hip on the k-major shapes of the step, one process per arm, interleaved:
# the diag build (CTCLIP_EXTRA_HIPCC_FLAGS=-DCTCLIP_TUNING_KNOBS) honours CTCLIP_GEMM5_MINK.   bash tools/ab_gemm5.sh [ONLY list]
export ONLY=${1:-"sq4096,sq8192,ff1 fwd,ff2 fwd,q fwd,kv fwd,out fwd,patch fwd,ff2 dgrad,ff1 dgrad,kv dgrad"}; export B=${BATCH:-32}
export CTCLIP_HIP_LIB=$PWD/ct-clip-ut_amd/ctclip_hip/libctclip_hip_diag.so
for i in 1 2; do
  echo "== gemm3"; CTCLIP_GEMM5_MINK=1000000 python3 tools/bench_gemm.py 2>/dev/null
  echo "== gemm5"; CTCLIP_GEMM5_MINK=128 python3 tools/bench_gemm.py 2>/dev/null
done
