#!/usr/bin/env python3
"""configs[3] (RoBERTa-base, MSEFast per-tensor activations) wall-clock against the knobs of the strict rounds:
OSQ_MSE_STREAMS (concurrent groups of nested searches; environment, read at import) and "mse_round_groups" (chunk groups per
workgroup).  One process per row: python tools/mse_rounds_sweep.py <round_groups>; prints the second (steady) run."""
import io
import json
import os
import runpy
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OSQ_BENCH_NO_STRICT", "1")
from outlier_suppression_amd import ops  # noqa: E402

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ops.set_tuning("mse_round_groups", groups)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--settle", "0", "--preroll", "0.05", "--no-cpu-baseline", "--no-kernel-table",
            "--calib-configs", "3"]
buf = io.StringIO()
with redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
j = json.loads(buf.getvalue().strip().splitlines()[-1])
v = j["calibration_config3"]
print(f"streams {os.environ.get('OSQ_MSE_STREAMS', 'default')}  round_groups {groups}:  wall {v['wall_s']:.3f} s  activation phase "
      f"{v['phases_s']['activation_calibration_msefast_per_tensor']:.3f} s")
