#!/bin/bash
# Diagnosis of an intermittent NaN loss of `bench.py --gpus 2 --small --batch 2 --local-negatives` with two gloo ranks on ONE device:
# N runs per configuration, final losses listed.   usage (GPU box): N=4 bash tools/diag_two_rank_nan.sh > gpurun_out/diag_nan.txt
N=${N:-4}
run() {   # label, extra env assignments...
  local label=$1; shift
  local out=""
  for i in $(seq $N); do
    v=$(env "$@" CTCLIP_GEMM_V2_ALL=1 CTCLIP_DIST_BACKEND=gloo timeout -k 10 300 python bench.py $ARGS --small --steps 2 --warmup 1 --batch 2 --lean 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        print(json.loads(l)['config'].get('final_loss'))
")
    out="$out $v"
  done
  echo "$label:$out"
}
ARGS="--gpus 2 --local-negatives"
ARGS="--gpus 2 --local-negatives"; run "two ranks, default" X=1
ARGS=""; run "one rank, default" X=1
ARGS="--gpus 2 --local-negatives"
run "two ranks, text stream off" CTCLIP_TEXT_STREAM=0
run "two ranks, patch unfused" CTCLIP_PATCH_FUSED=0
run "two ranks, PEG unfused" CTCLIP_PEG_FUSED=0
run "two ranks, head-norm separate" CTCLIP_HEADNORM_IN_GEMM=0
