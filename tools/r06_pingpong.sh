#!/bin/bash
# ping-pong round direction of the MSEFast rounds: same-box A/B on configs[3] (tunable build)
export OSQ_HIP_LIBRARY=$PWD/outlier_suppression_amd/libosq_hip_dbg.so
for r in 1 2 3; do
  for pp in 0 1; do python tools/mse_rounds_sweep.py mse_pingpong=$pp 2>&1 | grep "^streams"; done
done
OSQ_MSE_STREAMS=1 python tools/mse_rounds_sweep.py mse_pingpong=0 2>&1 | grep "^streams"
OSQ_MSE_STREAMS=1 python tools/mse_rounds_sweep.py mse_pingpong=1 2>&1 | grep "^streams"
