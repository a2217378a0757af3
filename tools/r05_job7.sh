#!/bin/bash
# round 5, job 7: the whole GPU suite at the round's defaults + a long seeded random walk against the oracle
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_tests7.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests7.log
tail -4 $O/gpu_tests7.log
OSQ_FUZZ_CASES=6000 OSQ_FUZZ_SEED=505 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider > $O/fuzz_walk.log 2>&1; echo "fuzz rc=$?" >> $O/fuzz_walk.log
tail -4 $O/fuzz_walk.log
