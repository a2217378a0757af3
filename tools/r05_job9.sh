#!/bin/bash
# round 5, job 9: reference-order MSEFast searches RESIDENT on the chip: correctness, then configs[3] per variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
for ki in 8 12; do
OSQ_BENCH_TUNING_TEST="mse_resident_items=$ki" timeout 600 python - <<PY > $O/gpu_tests9a_$ki.log 2>&1
import os, sys, subprocess
from outlier_suppression_amd import ops
ops.set_tuning("mse_resident_items", $ki)
import pytest
sys.exit(pytest.main(["tests/test_gpu_strict_order.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", "resident_ordered or rounds_equal"]))
PY
echo "KI=$ki tests rc=$?"; tail -3 $O/gpu_tests9a_$ki.log | cut -c1-300
done
for v in "1 8" "1 10" "1 12" "0 8"; do
  set -- $v
  OSQ_MSE_RESIDENT=$1 OSQ_BENCH_TUNING="mse_resident_items=$2" OSQ_BENCH_NO_STRICT=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --calib-configs 3 --steps 20 --warmup 5 > $O/mse_res_$1_$2.json 2> $O/mse_res_$1_$2.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mse_res_$1_$2.json").read().strip().splitlines()[-1])
    print("OSQ_MSE_RESIDENT=$1 items=$2:", json.dumps(d["calibration_config3"])[:300])
except Exception as e:
    print("OSQ_MSE_RESIDENT=$1 items=$2: failed", e); print(open("$O/mse_res_$1_$2.err").read()[-2500:])
PY
done
