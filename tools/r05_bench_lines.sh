#!/bin/bash
# round 5, bench lines: the two bench lines of record at the round's last kernels (default flags; the driver's flags)
mkdir -p gpurun_out/r05
S=$SECONDS
timeout 600 python bench.py > gpurun_out/r05/final_bench.json 2> gpurun_out/r05/final_bench.err
echo "default bench: rc $? in $((SECONDS - S)) s"
S=$SECONDS
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/final_bench_steps20.json 2> gpurun_out/r05/final_bench_steps20.err
echo "driver-flag bench: rc $? in $((SECONDS - S)) s"
python - <<'PY'
import json
for f in ("final_bench", "final_bench_steps20"):
    j = json.loads(open(f"gpurun_out/r05/{f}.json").read().strip().splitlines()[-1])
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], {k: v.get("wall_s") for k, v in j.items() if k.startswith("calibration_config")})
PY
