cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out
rm -rf $OUT/pmc_peg; mkdir -p $OUT/pmc_peg
B=64 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_peg -o a -- python3 $REPO/tools/bench_peg.py > $OUT/pmc_peg.log 2>&1
python3 $REPO/tools/pmc_summary.py $(find $OUT/pmc_peg -name '*counter_collection.csv' | head -1) peg_ > $OUT/r2_peg_pmc.txt
rm -rf $OUT/pmc_peg
cat $OUT/r2_peg_pmc.txt; grep peg $OUT/pmc_peg.log
