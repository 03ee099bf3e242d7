# SQ counter passes over the spatial-attention micro-benchmark + per-kernel durations; summary of the ws_* kernels
# (the TCP_* counters abort rocprofv3 on this image: not collected)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out
: > $OUT/r2_attn_bwd_pmc.txt
pass() {
  rm -rf $OUT/pmc_attn; mkdir -p $OUT/pmc_attn
  B=64 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_attn -o a -- python3 $REPO/tools/bench_attn.py > $OUT/pmc_attn.log 2>&1
  f=$(find $OUT/pmc_attn -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 $REPO/tools/pmc_summary.py $f ${FILTER:-ws_} >> $OUT/r2_attn_bwd_pmc.txt; else echo "pass $* failed" >> $OUT/r2_attn_bwd_pmc.txt; tail -3 $OUT/pmc_attn.log >> $OUT/r2_attn_bwd_pmc.txt; fi
  rm -rf $OUT/pmc_attn
}
rm -rf $OUT/kt_attn; mkdir -p $OUT/kt_attn
B=64 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_attn -o a -- python3 $REPO/tools/bench_attn.py > $OUT/kt_attn.log 2>&1
f=$(find $OUT/kt_attn -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/r2_attn_kernel_stats.csv
rm -rf $OUT/kt_attn
pass SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA
cut -d, -f1-4,6-8 $OUT/r2_attn_kernel_stats.csv | head -12
cat $OUT/r2_attn_bwd_pmc.txt
