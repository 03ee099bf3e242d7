#!/bin/bash
# A/B of two builds of the library on one box, interleaved:  bash tools/ab_lib.sh <base.so> <new.so> [WHAT]
A=$1; B=$2; export WHAT=${3:-fwd,bwd,bwdt}
for i in 1 2; do
  CTCLIP_HIP_LIB=$A TAG="base" python3 tools/bench_attn_hm.py 2>/dev/null
  CTCLIP_HIP_LIB=$B TAG="new " python3 tools/bench_attn_hm.py 2>/dev/null
done
