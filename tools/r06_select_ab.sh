#!/bin/bash
# selection changes: parity (fused step, fuzz, parity files), then A/B against the previous commit's library on one box
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
S=$SECONDS
timeout 1200 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_deferred.py -m gpu -q > $out/pytest_prelist.log 2>&1
echo "pytest: rc $? in $((SECONDS - S)) s"; tail -8 $out/pytest_prelist.log
{
echo "# bench lengths"; python tools/ab_step.py libosq_hip_head.so libosq_hip.so
echo "# all tokens valid"; python tools/ab_step.py libosq_hip_head.so libosq_hip.so full
} > $out/prelist_ab.txt 2>&1
for L in bench full; do python tools/select_loop.py --n 200 --lengths $L --events; python tools/select_loop.py --n 200 --lengths $L --events --state; done >> $out/prelist_ab.txt 2>&1
python tools/final_timing.py >> $out/prelist_ab.txt 2>&1; python tools/fused_timing.py 2>&1 | sed -n 2,9p >> $out/prelist_ab.txt; cat $out/prelist_ab.txt | grep -v amdgpu.ids
