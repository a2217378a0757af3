timeout 600 python -m pytest tests/test_gpu_strict_order.py tests/test_gpu_parity.py -m gpu -x -q -k "order or lsq or backward or rounds" 2>&1 | tail -3
timeout 300 python tools/bwd_order_ab.py 2>&1 | tail -4
timeout 300 python tools/strict_one_shape.py 32,128,768 | tail -1
timeout 300 python tools/strict_one_shape.py 32,128,3072 full | tail -1
for g in 4 8 16; do
OSQ_BENCH_TUNING=mse_round_groups=$g timeout 800 python bench.py --steps 5 --warmup 2 --settle 0 --preroll 0.05 --no-cpu-baseline --no-kernel-table --calib-configs 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['calibration_config3']
print('groups $g: default(strict)', c['wall_s'], 'order-free', c['order_free']['wall_s'], c['order_free']['activation_scale_rel_diff_vs_default']['max'])
"
done
