set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
bash tools/collect_profiles.sh r02 > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_steps20.json 2> gpurun_out/r02/bench_steps20.err
bash tools/prof.sh calib1 python bench.py --steps 20 --warmup 5 --settle 0 --no-cpu-baseline --no-kernel-table --calib-configs 1 > gpurun_out/calib1_table.txt 2>&1
bash tools/prof.sh calib3 python bench.py --steps 20 --warmup 5 --settle 0 --no-cpu-baseline --no-kernel-table --calib-configs 3 > gpurun_out/calib3_table.txt 2>&1
python tools/fused_timing.py > gpurun_out/fused_timing_r02_final.log 2>&1 || true
ls gpurun_out/r02
