#!/bin/bash
# timing-only ablations of gemm5's K-step (-DG5_ABL=<mask> builds)
export ONLY=${1:-"sq4096,ff1 dgrad"}; export B=32; export CTCLIP_GEMM5_MINK=128
for A in 0 1 2 3 7 15 0; do
  echo "== ABL $A"; CTCLIP_HIP_LIB=$PWD/ct-clip-ut_amd/ctclip_hip/libctclip_hip_abl$A.so python3 tools/bench_gemm.py 2>/dev/null
done
