#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python tools/mse_dbg_probe.py 2>&1 | grep "round_groups"
for g in 8 12; do OSQ_MSE_STREAMS=2 timeout 200 python tools/mse_rounds_sweep.py $g 2>/dev/null | tail -1; done
