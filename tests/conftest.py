import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]
    return load


def same_f32(a, b):
    """Bit-for-bit equality of two fp32 arrays, treating NaN == NaN and -0.0 == +0.0."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(both_nan | (a == b)))


@pytest.fixture(scope="session")
def eq32():
    return same_f32


@pytest.fixture()
def order_free():
    """The ORDER-FREE tier for one test (outlier_suppression_amd.set_strict(False)): exact / float64 sums, the resident
    per-tensor MSEFast searches; the package's default tier is restored afterwards."""
    import outlier_suppression_amd as osq
    osq.set_strict(False)
    yield
    osq.reset_tier()


def aten_order_mean(sq):
    """torch's CPU mean on a one-thread host with 8 fp32 SIMD lanes (float64: 4) -- the order the package's default sums follow."""
    from oracle.aten_sum import aten_mean_flat
    sq = np.asarray(sq)
    dt = np.float64 if sq.dtype == np.float64 else np.float32
    return aten_mean_flat(sq.reshape(-1), 4 if dt is np.float64 else 8, dt)


@pytest.fixture(params=["reference-order", "order-free"])
def sum_tier(request):
    """Both tiers of the two whole-tensor sums for one test: the package default (the reference's one-thread order;
    the oracle's MSE loss is then summed by oracle/aten_sum.py in that order) and set_strict(False) (exact sums on the
    device, the oracle's plain float64 mean)."""
    import outlier_suppression_amd as osq
    from oracle import observer_oracle as OB
    old = OB.MEAN_LIKE_TORCH
    if request.param == "order-free":
        osq.set_strict(False)
    else:
        osq.set_strict(True)                      # both sums in the reference's order (the backward's is opt-in since round 5)
        OB.MEAN_LIKE_TORCH = aten_order_mean
    yield request.param
    OB.MEAN_LIKE_TORCH = old
    osq.reset_tier()
