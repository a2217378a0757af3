#!/bin/bash
# round 6, second GPU call: untraced vs traced bench at the same flags (does the profiler change the step time?), the
# trace split into phases, then the GPU suite
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
flags="--steps 50 --warmup 10 --settle 0 --no-cpu-baseline --no-calib --no-kernel-table"
python bench.py $flags --detail-file $out/bench_untraced_detail.json > $out/bench_untraced.stdout 2> $out/bench_untraced.err
echo "untraced:"; tail -1 $out/bench_untraced.stdout | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['ms_per_step_regions'], d['roofline']['avg_launch_us'])"
rm -rf /tmp/osq_trace
rocprofv3 --kernel-trace --stats -d /tmp/osq_trace -o r -- python bench.py $flags --detail-file $out/bench_traced_detail.json > $out/bench_traced.stdout 2> $out/bench_traced.err
echo "traced:"; tail -1 $out/bench_traced.stdout | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['ms_per_step_regions'], d['roofline']['avg_launch_us'])"
db=$(find /tmp/osq_trace -name "*.db" | head -1)
python tools/trace_phases.py $db observe_fq_fused_kernel --min-run 40 > $out/bench_traced_phases.txt 2>&1; cat $out/bench_traced_phases.txt
S=$SECONDS
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1
echo "pytest: rc $? in $((SECONDS - S)) s"; tail -40 $out/pytest_gpu.log
