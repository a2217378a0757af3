# scratch command file for gpurun calls (rewritten per call)
mkdir -p gpurun_out/r04_parity
OSQ_REPORT_BASE=1 OSQ_PARITY_REPORT_DIR=gpurun_out/r04_parity python -m pytest tests/test_gpu_model_base.py -q -x -s 2>&1 | grep -E "bert-base|passed|failed|Error|assert" | head -40
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_model_base.py 2>&1 | tail -5
