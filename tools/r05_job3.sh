#!/bin/bash
# round 5, job 3: the judged profile artefacts (kernel trace, PMC passes, default bench line), the driver's bench command
# with its wall time, kernel traces of the calibration flows of configs 1 and 3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; tail -3 $O/collect.log
ls $O
SECONDS=0
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
echo "driver command wall: ${SECONDS}s" | tee $O/bench_steps20.wall
python - <<PY
import json
d = json.loads(open("$O/bench_steps20.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
      d["config"].get("graph_us_per_step"), d["config"].get("eager_us_per_step"), d["config"].get("launch_picked"))
print(json.dumps(d["calibration_summary"])[:2000])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
PMC=0 bash tools/collect_calibration_profiles.sh r05 1 3 > $O/collect_cal.log 2>&1; tail -3 $O/collect_cal.log
