#!/bin/bash
# round 5, job 12: where a round's 146 us go -- the rounds kernel without its loads / without its arithmetic (OSQ_MSE_DBG build)
mkdir -p gpurun_out/r05/job12
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
  rm -rf /tmp/p$d
  MSE_DBG=$d OSQ_BENCH_SHORT=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p$d -o t -- python $GRAFT_REPO_ROOT/tools/mse_dbg_probe.py > /tmp/p$d.log 2>&1
  f=$(find /tmp/p$d -name "*kernel_stats.csv" | head -1)
  echo "== MSE_DBG=$d" >> $GRAFT_REPO_ROOT/gpurun_out/r05/job12/summary.txt
  grep -i "ordered_multi\|Name" $f | head -3 >> $GRAFT_REPO_ROOT/gpurun_out/r05/job12/summary.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r05/job12/summary.txt
