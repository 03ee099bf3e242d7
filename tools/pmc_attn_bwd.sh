# several --pmc passes over the spatial-attention micro-benchmark; summary of the ws_bwd kernels per pass
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out
: > $OUT/r2_attn_bwd_pmc.txt
pass() {
  rm -rf $OUT/pmc_attn; mkdir -p $OUT/pmc_attn
  B=64 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_attn -o a -- python3 $REPO/tools/bench_attn.py > $OUT/pmc_attn.log 2>&1
  f=$(find $OUT/pmc_attn -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 $REPO/tools/pmc_summary.py $f ${FILTER:-ws_bwd} >> $OUT/r2_attn_bwd_pmc.txt; else echo "pass $* failed" >> $OUT/r2_attn_bwd_pmc.txt; tail -3 $OUT/pmc_attn.log >> $OUT/r2_attn_bwd_pmc.txt; fi
  rm -rf $OUT/pmc_attn
}
pass SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM
pass TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
pass TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass GRBM_GUI_ACTIVE FETCH_SIZE
cat $OUT/r2_attn_bwd_pmc.txt
