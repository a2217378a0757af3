cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_steps20.json 2> gpurun_out/r04/bench_steps20.err
python - <<PY
import json
for f in ("bench.json", "bench_steps20.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/r04/" + f) if l.startswith("{")][0])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["launch"][:60])
        print({k: v["wall_s"] for k, v in d.items() if k.startswith("calibration") and isinstance(v, dict) and "wall_s" in v})
    except Exception as e:
        print(f, "unreadable:", e)
PY
bash tools/collect_calibration_profiles.sh r04 1 2 3 4 2>&1 | tail -5
ls gpurun_out/r04_cal
