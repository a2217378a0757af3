# scratch command file for gpurun calls (rewritten per call)
timeout 900 python -m pytest tests/test_gpu_onelaunch.py -x -q 2>&1 | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-calib --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        for k,v in d['kernels'].items():
            if 'token' in k or 'observ' in k: print(k, v.get('avg_us'), v.get('frac_of_8TBps'))
"
