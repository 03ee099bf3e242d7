#!/bin/bash
# Interleaved A/B of library variants (tools/build_variant.sh names; "base" = the product library) on tools/bench_gemm.py shapes,
# medians over REPS repetitions.   usage (GPU box): LIBS="base g4s1" ONLY=wgrad REPS=4 B=32 bash tools/ab_libs.sh out.txt
OUT=${1:-gpurun_out/ab_libs.txt}; : > $OUT
L=$PWD/ct-clip-ut_amd/ctclip_hip
for i in $(seq ${REPS:-4}); do for V in $LIBS; do
  echo "lib=$V" >> $OUT
  if [ $V = base ]; then env -u CTCLIP_HIP_LIB python tools/bench_gemm.py >> $OUT 2>&1; else CTCLIP_HIP_LIB=$L/libctclip_hip_$V.so python tools/bench_gemm.py >> $OUT 2>&1; fi
done; done
python - $OUT <<'PY'
import collections, statistics, sys
d=collections.defaultdict(list); lib=None
for l in open(sys.argv[1]):
    if l.startswith('lib='): lib=l.strip()[4:]
    elif 'TFLOP/s' in l:
        d[(l[:20].strip(),lib)].append(float(l.split()[-2]))
for k in sorted(d): print("%-18s %-8s median %7.1f  %s" % (k[0], k[1], statistics.median(d[k]), d[k]))
PY
