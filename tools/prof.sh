#!/bin/bash
# Kernel durations from rocprofv3 (HIP-event timing from Python is launch-bound below ~10 us).
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> <command...>
# writes gpurun_out/<tag>/ and prints the per-kernel table (grouped by grid with BYGRID=1)
set -e
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o r -- "$@" > "$out/run.log" 2>&1 || { tail -20 "$out/run.log"; exit 1; }
db=$(find "$out/trace" -name '*.db' | head -1)
python tools/summarize_rocprof.py "$db" "$out/kernels.md" ${BYGRID:+--by-grid} | grep -v "at::native\|rocclr\|^$" | head -40; rm -rf "$out/trace"
