#!/bin/bash
# Collect the judged profile artefacts on the GPU box (run from the repo root):
#   tools/collect_profiles.sh <round-tag>            e.g. r01
# 1. rocprofv3 --kernel-trace --stats on the graded bench command  -> per-kernel duration table
# 2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, each in its OWN pass (no trace flags mixed in;
#    MI355X_MICROARCH.md "rocprofv3 PMC slots": the two counters do not fit one pass)
# Raw databases stay under gpurun_out/ (scratch); the summaries are written to gpurun_out/<tag>/ and
# copied into profiles/ by hand (tracked).
set -e
tag=${1:-r02}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"; rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write"
export TMPDIR=/tmp
cmd="python bench.py --steps 50 --warmup 10 --settle 0 --no-cpu-baseline --no-calib --no-kernel-table"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o r -- $cmd > "$out/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o r -- $cmd > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o r -- $cmd > "$out/pmc_write.log" 2>&1
# what the trace says launch by launch: back-to-back runs (graph replays, eager loops) against launches that carry dispatch events
python tools/trace_phases.py "$(find "$out/trace" -name '*.db' | head -1)" observe_fq_fused_kernel --min-run 40 > "$out/${tag}_bench_trace_phases.md" 2>&1
python tools/summarize_profiles.py "$out" "$tag"
# the counters of THESE kernel sources first (bench.py takes `roofline.traffic` from the table only while its source hash matches)
cp "$out/roofline_traffic.json" profiles/roofline_traffic.json
# the bench lines of record: default flags, and the driver's flags; last stdout line = the compact line, the rest in *_detail.json
python bench.py --detail-file "$out/${tag}_bench_detail.json" > "$out/bench.stdout" 2> "$out/bench.err"; tail -1 "$out/bench.stdout" > "$out/${tag}_bench.json"
python bench.py --steps 20 --warmup 5 --detail-file "$out/${tag}_bench_steps20_detail.json" > "$out/bench_steps20.stdout" 2> "$out/bench_steps20.err"; tail -1 "$out/bench_steps20.stdout" > "$out/${tag}_bench_steps20.json"
# keep the scratch directory small (gpurun copies back at most 64 MiB)
find "$out" -name '*.db' -delete
