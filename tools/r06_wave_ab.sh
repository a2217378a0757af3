#!/bin/bash
# the two forms of a round's first stage on the same box: a row per thread, two chunks per group of eight waves (cascade_chunks_pipelined,
# round 5) against a wave per chunk with a ring of 16-byte loads (cascade_chunks_wave), the latter budgeted for 4 or 5 waves per SIMD
L=$PWD/outlier_suppression_amd
echo "== pipelined (mse_wave 0)"; OSQ_HIP_LIBRARY=$L/libosq_hip_dbg.so MSE_WAVE=0 python tools/mse_round_probe.py 4 8 16 2>&1 | grep round_groups
echo "== wave, 4 waves per SIMD"; OSQ_HIP_LIBRARY=$L/libosq_hip_dbg.so MSE_WAVE=1 python tools/mse_round_probe.py 2 4 8 2>&1 | grep round_groups
echo "== wave, 5 waves per SIMD"; OSQ_HIP_LIBRARY=$L/libosq_hip_dbg5.so MSE_WAVE=1 python tools/mse_round_probe.py 2 4 8 2>&1 | grep round_groups
