#!/bin/bash
# round 6, first GPU call: the bench line (driver flags, default flags), the same command under rocprofv3 --kernel-trace
# (do the in-run dispatch events agree with the trace?), then the GPU suite with the bit-strict comparison helpers
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
S=$SECONDS
timeout 900 python bench.py --steps 20 --warmup 5 --detail-file $out/bench_steps20_detail.json > $out/bench_steps20.stdout 2> $out/bench_steps20.err
echo "driver-flag bench: rc $? in $((SECONDS - S)) s"; tail -c 2500 $out/bench_steps20.stdout; echo
S=$SECONDS
rm -rf /tmp/osq_trace
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/osq_trace -o r -- python bench.py --steps 50 --warmup 10 --settle 0 --no-cpu-baseline --no-calib --no-kernel-table --detail-file $out/bench_traced_detail.json > $out/bench_traced.stdout 2> $out/bench_traced.err
echo "traced bench: rc $? in $((SECONDS - S)) s"; tail -1 $out/bench_traced.stdout
python tools/summarize_rocprof.py $(find /tmp/osq_trace -name "*.db" | head -1) 2>&1 | head -12 > $out/bench_traced_kernel_stats.txt; cat $out/bench_traced_kernel_stats.txt
S=$SECONDS
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
echo "pytest: rc $? in $((SECONDS - S)) s"; tail -15 $out/pytest_gpu.log
