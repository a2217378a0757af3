#!/bin/bash
# round 5, job 10: the whole GPU suite + the default bench line after the resident-search kernel was removed again
mkdir -p gpurun_out/r05
S=$SECONDS
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/job10_gputests.txt 2>&1
echo "gpu tests: rc $? in $((SECONDS - S)) s" >> gpurun_out/r05/job10_gputests.txt
tail -3 gpurun_out/r05/job10_gputests.txt
S=$SECONDS
timeout 600 python bench.py > gpurun_out/r05/job10_bench.json 2> gpurun_out/r05/job10_bench.err
echo "bench: rc $? in $((SECONDS - S)) s"
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r05/job10_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['unit'], j['ms_per_step'], j['roofline'])
print({k: (v.get('wall_s') if isinstance(v, dict) else v) for k, v in j['config'].items() if k.startswith('calibration')})
PY
