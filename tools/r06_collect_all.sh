#!/bin/bash
# round 6, the judged profile set at the round's kernels: bench trace + PMC passes + bench lines (tools/collect_profiles.sh), the
# calibration flows of configs[1..4] with their PMC passes (tools/collect_calibration_profiles.sh), BERT-base integer tensors
export TMPDIR=/tmp
S=$SECONDS
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1; echo "collect_profiles: rc $? in $((SECONDS - S)) s"; tail -3 gpurun_out/r06_collect.log
cat gpurun_out/r06/r06_bench_steps20.json; echo
S=$SECONDS
PMC=1 bash tools/collect_calibration_profiles.sh r06 1 2 3 4 > gpurun_out/r06_cal_collect.log 2>&1; echo "calibration profiles: rc $? in $((SECONDS - S)) s"
ls gpurun_out/r06_cal/
S=$SECONDS
mkdir -p gpurun_out/r06_parity
OSQ_PARITY_REPORT_DIR=$PWD/gpurun_out/r06_parity timeout 900 python -m pytest tests/test_gpu_model_base.py -m gpu -q > gpurun_out/r06_parity/pytest.log 2>&1; echo "model base: rc $? in $((SECONDS - S)) s"; tail -3 gpurun_out/r06_parity/pytest.log
