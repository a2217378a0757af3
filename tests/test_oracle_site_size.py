"""The oracle against the REFERENCE's own one-thread run at the site sizes of BASELINE configs[3]
(tests/golden/site_size.npz, made by tests/golden/make_golden_site_size.py; the inputs are re-drawn from the same seeded
generators): per-tensor MSEFast searches on [32,128,768] masked hidden states (nested 2-D, float64 from the second call
on), [32,12,128,128] attention probabilities (1-D, fp32 for ever) and -- OSQ_SLOW_TESTS=1 only: 1.5 minutes of NumPy here, green at the round's last commit --
[32,128,3072] GELU outputs; LSQ+ gradients on [32,128,768] and [32,128,3072].  With the sums in ATen's one-thread order
(oracle/aten_sum.py::aten_sum_flat) every statistic after every call, every evaluation count and both gradients are
EQUAL.  The GPU counterpart (the kernels' strict switch) is tests/test_gpu_strict_order.py."""
import os

import numpy as np
import pytest
import torch

from _site_size import BWD_CASES, MSE_CASES, bwd_case, checksum, site_input, site_lengths
from oracle import fake_quant_oracle as FQ, observer_oracle as OB
from test_oracle_golden import aten_order_mean

SLOW = os.environ.get("OSQ_SLOW_TESTS", "0") not in ("", "0")


@pytest.mark.parametrize("case", [c[0] for c in MSE_CASES])
def test_msefast_site_size_equals_reference(golden, case):
    if case == "gelu3072" and not SLOW:
        pytest.skip("12.6 M elements x 1200 evaluations in NumPy: OSQ_SLOW_TESTS=1 (the GPU test runs it always)")
    g = golden("site_size")
    name, cls, shape, seq_pos, kind, bit, sym, batches, seed = next(c for c in MSE_CASES if c[0] == case)
    gen = torch.Generator().manual_seed(seed)
    st = OB.ObserverState(bit=bit, symmetric=sym, ch_axis=-1)
    counter = [0]
    OB.MEAN_LIKE_TORCH = aten_order_mean
    try:
        for r in range(batches):
            x = site_input(gen, shape, kind, r)
            L = site_lengths(gen, shape, seq_pos)
            if checksum(x) != int(g[f"{name}_xsum"][r]):      # another torch build / CPU draws other tensors: the fixture does not apply
                pytest.skip("the seeded input differs from the fixture's (torch's CPU generator on this host)")
            OB.observe_msefast(st, x.numpy(), L.numpy(), seq_pos, average=cls.startswith("Avg"), counter=counter)
            assert float(st.min_val) == float(g[f"{name}_min"][r]) and float(st.max_val) == float(g[f"{name}_max"][r]), \
                (name, r, st.min_val, g[f"{name}_min"][r], st.max_val, g[f"{name}_max"][r])
            assert counter[0] == int(g[f"{name}_nfev"][r]), (name, r, counter[0], int(g[f"{name}_nfev"][r]))
    finally:
        OB.MEAN_LIKE_TORCH = None
    assert st.one_side_dist == str(g[f"{name}_side"])


@pytest.mark.parametrize("case", [c[0] for c in BWD_CASES])
def test_lsqplus_site_size_gradients_equal_reference(golden, case):
    g = golden("site_size")
    name, shape, kind, seed = next(c for c in BWD_CASES if c[0] == case)
    x, gy, scale, zp, gf = bwd_case(shape, kind, seed)
    if [checksum(x), checksum(gy)] != [int(v) for v in g[f"{name}_xsum"]]:
        pytest.skip("the seeded input differs from the fixture's (torch's CPU generator on this host)")
    assert np.array_equal(scale.numpy(), g[f"{name}_scale"]) and np.array_equal(zp.numpy(), g[f"{name}_zp"])
    dx, ds, dz = FQ.lsqplus_backward_per_tensor_reference_order(x.numpy(), gy.numpy(), scale.numpy(), zp.numpy(), 0, 63, gf)
    assert checksum(torch.from_numpy(dx)) == int(g[f"{name}_dxsum"][0])
    assert np.float32(ds) == g[f"{name}_dscale"][0] and np.float32(dz) == g[f"{name}_dzp"][0], (ds, g[f"{name}_dscale"], dz, g[f"{name}_dzp"])
