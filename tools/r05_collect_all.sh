#!/bin/bash
# round 5, closing job: after the rounds-kernel work (branch-free prefetch, uniform lean loop, immediate offsets, 8 groups per workgroup):
# whole GPU suite, the judged profile set (trace + PMC passes + default bench line), configs[3]'s kernel trace
mkdir -p gpurun_out/r05
S=$SECONDS
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/collect_gputests.txt 2>&1
echo "gpu tests: rc $? in $((SECONDS - S)) s" >> gpurun_out/r05/collect_gputests.txt
tail -3 gpurun_out/r05/collect_gputests.txt
S=$SECONDS
timeout 900 bash tools/collect_profiles.sh r05 > gpurun_out/r05/collect_collect.log 2>&1
echo "collect_profiles: rc $? in $((SECONDS - S)) s"
S=$SECONDS
PMC=0 timeout 700 bash tools/collect_calibration_profiles.sh r05 3 > gpurun_out/r05/collect_cal.log 2>&1
echo "calibration profile: rc $? in $((SECONDS - S)) s"
ls gpurun_out/r05 gpurun_out/r05_cal 2>/dev/null | head -40
