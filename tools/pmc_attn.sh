set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out
rm -rf $OUT/pmc_attn; mkdir -p $OUT/pmc_attn
B=64 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_attn -o a -- python3 $REPO/tools/bench_attn.py > $OUT/pmc_attn.log 2>&1
TAG=${1:-r3}
python3 $REPO/tools/pmc_summary.py $(find $OUT/pmc_attn -name '*counter_collection.csv' | head -1) hm_fwd > $OUT/${TAG}_attn_pmc.txt
python3 $REPO/tools/pmc_summary.py $(find $OUT/pmc_attn -name '*counter_collection.csv' | head -1) hm_bwd >> $OUT/${TAG}_attn_pmc.txt
python3 $REPO/tools/pmc_summary.py $(find $OUT/pmc_attn -name '*counter_collection.csv' | head -1) ws_ >> $OUT/${TAG}_attn_pmc.txt
rm -rf $OUT/pmc_attn
cat $OUT/${TAG}_attn_pmc.txt
