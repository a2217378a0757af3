#!/bin/bash
# A/B in one job: cascade_rotate as explicit v_mov (rotate) vs plain assignment (plain), configs[3] and the rounds probe, alternating
for rep in 1 2; do for v in rotate plain; do
  echo "== $v ($rep)"
  OSQ_HIP_LIBRARY=$PWD/outlier_suppression_amd/libosq_hip_$v.so OSQ_MSE_STREAMS=2 timeout 200 python tools/mse_rounds_sweep.py 8 2>/dev/null | tail -1
  OSQ_HIP_LIBRARY=$PWD/outlier_suppression_amd/libosq_hip_$v.so timeout 200 python tools/mse_dbg_probe.py 2>&1 | grep "round_groups  8"
done; done
