for i in 1 2 3 4; do python bench.py --steps 20 --warmup 5 --no-calib --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['config']['launch'][-70:], d['config']['eager_ms_per_step'], d['roofline']['avg_launch_us'])
"; done
