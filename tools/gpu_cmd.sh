python -m pytest tests/test_gpu_strict_order.py -q -x 2>&1 | tail -3
python tools/strict_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_strict_bench.txt
