#!/bin/bash
# the GPU suite (release library; the tunable build in its subprocess)
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
S=$SECONDS
timeout 1700 python -m pytest tests -m gpu -q ${PYTEST_ARGS} > $out/pytest_gpu.log 2>&1
echo "pytest: rc $? in $((SECONDS - S)) s"; tail -${TAIL:-25} $out/pytest_gpu.log
