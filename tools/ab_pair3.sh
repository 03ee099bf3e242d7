#!/bin/bash
# gemm3 with / without paired LDS-DMA issue (gemm5 off), interleaved; then the k-major GEMM tests on the paired build
export B=${BATCH:-32}; export CTCLIP_GEMM5_MINK=1000000
export ONLY=${1:-"sq4096,ff1 fwd,ff2 fwd,q fwd,kv fwd,out fwd,patch fwd,ff2 dgrad,ff1 dgrad,kv dgrad"}
for i in 1 2; do
  for v in base pair3; do echo "== $v"; CTCLIP_HIP_LIB=$PWD/ct-clip-ut_amd/ctclip_hip/libctclip_hip_$v.so python3 tools/bench_gemm.py 2>/dev/null; done
done
for v in base pair3; do echo "== $v (fused)"; CTCLIP_HIP_LIB=$PWD/ct-clip-ut_amd/ctclip_hip/libctclip_hip_$v.so python3 tools/bench_ff.py 2>/dev/null; done
