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


def f32_bits(a):
    """uint32 view of an fp32 array with every NaN mapped to one canonical pattern (NaN payloads / signs are not part of the
    contract: torch, NumPy and the device produce different quiet NaNs for inf - inf)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    bits = a.view(np.uint32).copy()
    bits[np.isnan(a)] = np.uint32(0x7FC00000)
    return bits


def same_f32(a, b):
    """BIT-for-bit equality of two fp32 arrays: compared as uint32 words, so -0.0 and +0.0 DIFFER (the
    reference dequantizes -0.0 to +0.0 -- a sign-of-zero slip must show); NaN equals NaN whatever its payload.  Until round 5
    this was `a == b`, which let the sign of a zero through."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    return bool(np.array_equal(f32_bits(a), f32_bits(b)))


def bits_equal(a, b, equal_nan=True):
    """np.array_equal made a BIT comparison for floating-point arrays (words compared, -0.0 != +0.0, NaN == NaN); integer,
    bool and mixed arrays go to np.array_equal.  The `-m gpu` tests compare the device's results with the oracle through this."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float32 and b.dtype == np.float32:
        return same_f32(a, b)
    if a.dtype == np.float64 and b.dtype == np.float64:
        ab, bb = np.ascontiguousarray(a).view(np.uint64).copy(), np.ascontiguousarray(b).view(np.uint64).copy()
        ab[np.isnan(a)] = np.uint64(0x7FF8000000000000)
        bb[np.isnan(b)] = np.uint64(0x7FF8000000000000)
        return bool(np.array_equal(ab, bb))
    return bool(np.array_equal(a, b, equal_nan=equal_nan and a.dtype.kind == "f" and b.dtype.kind == "f"))


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
