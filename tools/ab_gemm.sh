#!/bin/bash
# A/B of two builds on the GEMM shapes of the step:  bash tools/ab_gemm.sh <base.so> <new.so> [ONLY list]
LIB_A=$1; LIB_B=$2; export ONLY=${3:-"ff1 fwd,ff2 fwd,q fwd,kv fwd,out fwd,ff2 dgrad,ff1 dgrad,sq4096"}; export B=${BATCH:-32}
for i in 1 2; do
  echo "== base"; CTCLIP_HIP_LIB=$LIB_A python3 tools/bench_gemm.py 2>/dev/null
  echo "== new";  CTCLIP_HIP_LIB=$LIB_B python3 tools/bench_gemm.py 2>/dev/null
done
