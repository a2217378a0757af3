"""The tests that need the performance A/B knobs as VARIABLES, against the -DOSQ_TUNABLE development build.

The release library holds those knobs as compile-time constants (include/osq_hip.h, osq_set_tuning): it refuses their keys,
and the tests that flip them skip themselves there.  `make dbg` (run by __graft_entry__.build()) builds libosq_hip_dbg.so
from the same sources with the knobs as variables; a library is loaded once per process, so those tests run HERE in a
subprocess with OSQ_HIP_LIBRARY pointing at it."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "outlier_suppression_amd", "libosq_hip_dbg.so")


def test_knob_dependent_tests_pass_on_the_tunable_build():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(DBG):
        pytest.skip("libosq_hip_dbg.so not built (make -C outlier_suppression_amd/csrc dbg)")
    env = dict(os.environ, OSQ_HIP_LIBRARY=DBG)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_strict_order.py"), "-k",
                        "test_rounds_do_not_depend_on_the_workgroups_share_of_chunk_groups or test_lean_float64_term_equals_the_full_chain"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert "5 passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_tunable_build_accepts_what_the_release_library_refuses():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(DBG):
        pytest.skip("libosq_hip_dbg.so not built")
    code = ("from outlier_suppression_amd import ops, _hip; _hip.load(); assert ops.tunable_build(); "
            "[ops.set_tuning(k, v) for k, v in (('fq_unroll', 4), ('mse_round_groups', 3), ('obs_blocks', 512), ('select_hint', 0))]; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OSQ_HIP_LIBRARY=DBG), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
