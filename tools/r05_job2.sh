#!/bin/bash
# round 5, job 2: GPU tests (q/k/v one launch, LN site default, bench refuse test), q/k/v A/B, MSEFast round-geometry sweep,
# the driver's bench command, profile collection
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_tests2.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests2.log
tail -5 $O/gpu_tests2.log
timeout 300 python tools/qkv_ab.py > $O/qkv_ab.txt 2>&1; tail -12 $O/qkv_ab.txt
for g in 2 4 8 16 32; do
  OSQ_BENCH_TUNING="mse_round_groups=$g" OSQ_BENCH_NO_STRICT=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --calib-configs 3 --steps 20 --warmup 5 > $O/mse_rg_$g.json 2> $O/mse_rg_$g.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mse_rg_$g.json").read().strip().splitlines()[-1])
    print("mse_round_groups $g:", json.dumps(d["calibration_config3"])[:300])
except Exception as e:
    print("mse_round_groups $g: failed", e)
PY
done
/usr/bin/time -v python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; grep -E "Elapsed|Maximum resident" $O/bench_steps20.err
python - <<PY
import json
d = json.loads(open("$O/bench_steps20.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["roofline"]["frac"], d["config"].get("graph_us_per_step"), d["config"].get("eager_us_per_step"))
print(json.dumps(d["calibration_summary"])[:1500])
PY
