#!/bin/bash
# round 5, job 6: the lean float64 term of the MSEFast rounds: correctness (reference-generated fixtures, lean == full chain), then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_deferred.py tests/test_gpu_site_size.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider -k "mse or MSE or deferred or ordered or rounds or site_size or lean" > $O/gpu_tests6.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests6.log
tail -6 $O/gpu_tests6.log
for lean in 1 0 1 0; do
  OSQ_BENCH_TUNING="mse_lean=$lean" OSQ_BENCH_NO_STRICT=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --calib-configs 3 --steps 20 --warmup 5 > $O/mse_lean_$lean.json 2> $O/mse_lean_$lean.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mse_lean_$lean.json").read().strip().splitlines()[-1])
    print("mse_lean=$lean:", json.dumps(d["calibration_config3"])[:300])
except Exception as e:
    print("mse_lean=$lean: failed", e); print(open("$O/mse_lean_$lean.err").read()[-1500:])
PY
done
