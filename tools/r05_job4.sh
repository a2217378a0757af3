#!/bin/bash
# round 5, job 4: MSEFast rounds of several groups on concurrent streams -- correctness subset, then configs[3] per stream count
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_deferred.py tests/test_gpu_site_size.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "mse or MSE or deferred or ordered or rounds or site_size" > $O/gpu_tests4.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests4.log
tail -4 $O/gpu_tests4.log
for n in 1 2 3 4 6; do
  OSQ_MSE_STREAMS=$n OSQ_BENCH_NO_STRICT=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --calib-configs 3 --steps 20 --warmup 5 > $O/mse_streams_$n.json 2> $O/mse_streams_$n.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mse_streams_$n.json").read().strip().splitlines()[-1])
    print("OSQ_MSE_STREAMS=$n:", json.dumps(d["calibration_config3"])[:300])
except Exception as e:
    print("OSQ_MSE_STREAMS=$n: failed", e); import subprocess; print(open("$O/mse_streams_$n.err").read()[-1500:])
PY
done
