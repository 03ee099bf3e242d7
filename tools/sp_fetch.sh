#!/bin/bash
# FETCH_SIZE of the sequence-persistent attention kernels at several sequences-per-workgroup settings
cd /tmp && export TMPDIR=/tmp
for c in 8 32 96 220; do
  export CTCLIP_ATTN_SP_CHUNK=$c
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/spf
  B=64 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/spf -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py > $GRAFT_REPO_ROOT/gpurun_out/spf_$c.txt 2>&1
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/spf/*/*counter_collection.csv | head -1)
  echo "chunk $c"; grep -E "spatial +(fwd|bwd \(\+table)" $GRAFT_REPO_ROOT/gpurun_out/spf_$c.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f sp_ | grep -E "^sp_|FETCH"
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/spf
