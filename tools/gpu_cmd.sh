python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_site_size.py -q --durations=12 2>&1 | tail -45
for m in "" "--eager" "" "--eager"; do python bench.py --steps 20 --warmup 5 --no-calib --no-kernel-table --no-cpu-baseline $m 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('MODE [$m]', d['ms_per_step'], 'eager', d['config']['eager_ms_per_step'], 'host', d['config']['host_enqueue_ms_per_step'], 'kernel', d['roofline']['avg_launch_us'])
"; done
