#!/usr/bin/env python3
"""configs[3] (RoBERTa-base, MSEFast per-tensor activations) wall-clock against the A/B knobs of the strict rounds:
OSQ_MSE_STREAMS (concurrent groups of nested searches; environment, read at import) and osq_set_tuning keys given as
key=value arguments ("mse_round_groups=8", "mse_pingpong=0": these exist only in the -DOSQ_TUNABLE build -> run with
OSQ_HIP_LIBRARY=.../libosq_hip_dbg.so).  One process per row; prints the second (steady) run."""
import io
import os
import runpy
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OSQ_BENCH_NO_STRICT", "1")
from outlier_suppression_amd import ops  # noqa: E402
from benchlib import line as BL  # noqa: E402

knobs = [a for a in sys.argv[1:] if "=" in a]
for kv in knobs:
    k, v = kv.split("=")
    ops.set_tuning(k, int(v))
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--settle", "0", "--preroll", "0.05", "--regions", "1", "--kernel-launches", "10",
            "--no-cpu-baseline", "--no-kernel-table", "--calib-configs", "3", "--detail-file", "/tmp/mse_sweep_detail.json"]
buf = io.StringIO()
with redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
_, detail = BL.parse_stdout(buf.getvalue())
v = detail["calibration_config3"]
print(f"streams {os.environ.get('OSQ_MSE_STREAMS', 'default')}  {' '.join(knobs) or '(defaults)'}:  wall {v['wall_s']:.3f} s  activation phase "
      f"{v['phases_s']['activation_calibration_msefast_per_tensor']:.3f} s", flush=True)
