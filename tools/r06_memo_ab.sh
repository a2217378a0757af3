#!/bin/bash
# the loss memo of the MSEFast searches: same-box A/B on configs[3] (release library: "mse_memo" is a shipped switch), then the
# number of rounds between two looks at the all-done flag, then streams
for r in 1 2; do
  for m in 0 1; do python tools/mse_rounds_sweep.py mse_memo=$m 2>&1 | grep "^streams"; done
done
for c in 8 16 32 64; do echo "chunk $c"; OSQ_MSE_CHUNK=$c python tools/mse_rounds_sweep.py mse_memo=1 2>&1 | grep "^streams"; done
for s in 1 3 4; do OSQ_MSE_STREAMS=$s python tools/mse_rounds_sweep.py mse_memo=1 2>&1 | grep "^streams"; done
