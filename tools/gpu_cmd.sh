# scratch command file for `gpurun -- 'bash tools/gpu_cmd.sh'` calls (rewritten per call during development)
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
