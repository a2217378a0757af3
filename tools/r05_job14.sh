#!/bin/bash
mkdir -p gpurun_out/r05
for st in 2 1; do for g in 12 16 24 32 48; do
  OSQ_MSE_STREAMS=$st timeout 200 python tools/mse_rounds_sweep.py $g 2>/dev/null | tail -1 | tee -a gpurun_out/r05/job14_sweep.txt
done; done
