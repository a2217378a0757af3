#!/bin/bash
# Profiles of the 256-sample calibration flows (BASELINE configs 1-4) on the GPU box, from the repo root:
#   tools/collect_calibration_profiles.sh <round-tag> [configs...]      e.g. r04 1 2 3 4
# per config: rocprofv3 --kernel-trace --stats of the FULL flow (bench.py's calibration section), then --pmc FETCH_SIZE and
# --pmc WRITE_SIZE -- each in its OWN pass, never mixed with trace domains -- of the SHORT flow (OSQ_BENCH_SHORT=1: the same
# kernels in the same states on 2 batches / 3 candidates / 1 epoch; a PMC pass costs milliseconds per dispatch, the full
# flows have 10^5-10^6 of them).  Summaries: gpurun_out/<tag>_cal/<tag>_calibration_config<N>_kernel_stats.md.
tag=${1:-r04}; shift
cfgs=${@:-1 2 3 4}
export TMPDIR=/tmp
mkdir -p "$PWD/gpurun_out/${tag}_cal"
for c in $cfgs; do
  out=/tmp/osq_prof_${tag}_c$c          # raw databases never enter gpurun_out/ (64 MiB limit on what travels back)
  rm -rf "$out"; mkdir -p "$out"
  cmd="python bench.py --steps 5 --warmup 2 --settle 0 --preroll 0.05 --no-cpu-baseline --no-kernel-table --calib-configs $c"
  OSQ_BENCH_NO_STRICT=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/trace" -o r -- $cmd > "$out/trace.log" 2>&1 || tail -5 "$out/trace.log"
  if [ "${PMC:-1}" = "1" ]; then
    OSQ_BENCH_SHORT=1 timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o r -- $cmd > "$out/pmc_fetch.log" 2>&1 || tail -5 "$out/pmc_fetch.log"
    OSQ_BENCH_SHORT=1 timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o r -- $cmd > "$out/pmc_write.log" 2>&1 || tail -5 "$out/pmc_write.log"
  fi
  python tools/summarize_calibration_profile.py "$out" "$tag" "$c" "$cmd"
  cp "$out/${tag}_calibration_config${c}_kernel_stats.md" "$PWD/gpurun_out/${tag}_cal/"
  rm -rf "$out"
done
