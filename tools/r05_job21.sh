#!/bin/bash
mkdir -p gpurun_out/r05
S=$SECONDS
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/job21_gputests.txt 2>&1
echo "gpu tests: rc $? in $((SECONDS - S)) s"; tail -2 gpurun_out/r05/job21_gputests.txt
timeout 200 python tools/mse_dbg_probe.py 2>&1 | grep "round_groups  8"
for i in 1 2; do OSQ_MSE_STREAMS=2 timeout 200 python tools/mse_rounds_sweep.py 8 2>/dev/null | tail -1; done
