#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05; mkdir -p $O
OSQ_BENCH_SHORT=1 OSQ_MSE_STREAMS=1 bash tools/pmc_kernel.sh mse msefast_tensor_ordered_multi python bench.py --steps 5 --warmup 2 --settle 0 --preroll 0.05 --no-cpu-baseline --no-kernel-table --calib-configs 3 > $O/pmc_mse_rounds.txt 2>&1
cat $O/pmc_mse_rounds.txt | tail -30
