python tools/mse_grid_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_mse_grid_ab.txt
python -m pytest tests -q -m gpu -x -k "mse_grid or other_observers or reciprocal_division or golden or fuzz" 2>&1 | tail -3
