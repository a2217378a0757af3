# Round-end run on ONE box: GPU tests, the bench profiles (kernel trace + PMC passes + the default bench line), the driver's flags.
# usage: bash tools/final_run.sh <tag>     (raw databases stay in /tmp on the box; summaries go to gpurun_out/<tag>/)
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -x -q > gpurun_out/$tag/gpu_tests.log 2>&1; tail -2 gpurun_out/$tag/gpu_tests.log
bash tools/collect_profiles.sh $tag > gpurun_out/$tag/collect.log 2>&1; tail -3 gpurun_out/$tag/collect.log
python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_steps20.json 2> gpurun_out/$tag/bench_steps20.err
rm -rf gpurun_out/$tag/trace gpurun_out/$tag/pmc_fetch gpurun_out/$tag/pmc_write
ls gpurun_out/$tag
python - <<PY
import json
for f in ("bench.json", "bench_steps20.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/$tag/" + f) if l.startswith("{")][0])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["launch"][:90])
    except Exception as e:
        print(f, "unreadable:", e)
PY
