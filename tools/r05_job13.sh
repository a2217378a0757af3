#!/bin/bash
# round 5, job 13: rounds kernel with branch-free prefetch, uniform lean loop, immediate offsets, global loads: parity + configs[3]
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r05/job13_tests.txt 2>&1
tail -3 gpurun_out/r05/job13_tests.txt
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --calib-configs 3 > gpurun_out/r05/job13_bench_$i.json 2> gpurun_out/r05/job13_bench_$i.err
python - <<PY
import json
j = json.loads(open('gpurun_out/r05/job13_bench_$i.json').read().strip().splitlines()[-1])
v = j['calibration_config3']; print(v['wall_s'], v['phases_s'])
PY
done
timeout 200 python tools/mse_dbg_probe.py 2>&1 | grep "persistent    0"
