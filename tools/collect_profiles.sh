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
python bench.py > "$out/bench.json" 2> "$out/bench.err"
python tools/summarize_profiles.py "$out" "$tag"
# keep the scratch directory small (gpurun copies back at most 64 MiB)
find "$out" -name '*.db' -delete
