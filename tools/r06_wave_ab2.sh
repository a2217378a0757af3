L=$PWD/outlier_suppression_amd
echo "== pipelined, four-wide lean term (mse_wave 0)"; OSQ_HIP_LIBRARY=$L/libosq_hip_dbg.so MSE_WAVE=0 python tools/mse_round_probe.py 4 8 12 16 2>&1 | grep round_groups
echo "== wave, 4 waves per SIMD (reference for the box)"; OSQ_HIP_LIBRARY=$L/libosq_hip_dbg.so MSE_WAVE=1 MSE_PROBE_CASES="0 1" python tools/mse_round_probe.py 4 2>&1 | grep round_groups
