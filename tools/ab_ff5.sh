#!/bin/bash
# fused feed-forward epilogues: gemm3 vs gemm5 (diag build), interleaved
export B=${BATCH:-32}; export CTCLIP_HIP_LIB=$PWD/ct-clip-ut_amd/ctclip_hip/libctclip_hip_diag.so
for i in 1 2 3; do
  echo "== gemm3"; CTCLIP_GEMM5_MINK=1000000 python3 tools/bench_ff.py 2>/dev/null
  echo "== gemm5"; CTCLIP_GEMM5_MINK=128 python3 tools/bench_ff.py 2>/dev/null
done
