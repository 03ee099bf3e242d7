#!/bin/bash
# forward ablations: which part of a tile's work the time follows
D=$PWD/ct-clip-ut_amd/ctclip_hip
for v in "$@"; do
  CTCLIP_HIP_LIB=$D/libctclip_hip_$v.so TAG="$v" WHAT=${WHAT:-fwd} python3 tools/bench_attn_hm.py 2>/dev/null
done
