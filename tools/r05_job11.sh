#!/bin/bash
# round 5, job 11: the rounds kernel with GLOBAL loads of x (was flat_load_dword: lgkmcnt-coupled to the LDS waits)
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_parity.py -m gpu -x -q -k "msefast or ordered or strict or lean" > gpurun_out/r05/job11_tests.txt 2>&1
tail -3 gpurun_out/r05/job11_tests.txt
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --calib-configs 3 > gpurun_out/r05/job11_bench_$i.json 2> gpurun_out/r05/job11_bench_$i.err
python - <<PY
import json
j = json.loads(open('gpurun_out/r05/job11_bench_$i.json').read().strip().splitlines()[-1])
for k, v in j['config'].items():
    if k.startswith('calibration'):
        print(k, json.dumps(v)[:600])
PY
done
