#!/bin/bash
# interleaved A/B of two builds on bench_gemm shapes with gemm5 forced on:  bash tools/ab_lib2.sh <a.so> <b.so> [ONLY]
A=$1; B_=$2; export ONLY=${3:-"sq4096,ff2 fwd,ff1 dgrad,kv fwd,ff1 fwd"}; export B=${BATCH:-32}; export CTCLIP_GEMM5_MINK=128
for i in 1 2; do
  echo "== $A"; CTCLIP_HIP_LIB=$A python3 tools/bench_gemm.py 2>/dev/null
  echo "== $B_"; CTCLIP_HIP_LIB=$B_ python3 tools/bench_gemm.py 2>/dev/null
done
