#!/bin/bash
# several builds on the same shapes, interleaved:  ONLY="..." bash tools/ab_many.sh name1 name2 ...   (libctclip_hip_<name>.so; "prod" = the product build)
D=$PWD/ct-clip-ut_amd/ctclip_hip; export B=${BATCH:-32}
for v in "$@"; do
  L=$D/libctclip_hip_$v.so; [ "$v" = prod ] && L=$D/libctclip_hip.so
  echo "== $v"; CTCLIP_HIP_LIB=$L python3 tools/bench_gemm.py 2>/dev/null
done
