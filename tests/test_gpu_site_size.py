"""BASELINE configs[3]'s per-tensor AvgMSEFast searches at their REAL site shapes, DEFAULT configuration, against the oracle.

The kernels real flows launch -- `msefast_resident_kernel<K>` (one persistent launch per search, the tensor in registers)
and `msefast_resident_multi_kernel` (the searches of one forward in one launch, quantization/deferred.py) -- on
[32,128,768] masked hidden states, [32,12,128,128] attention probabilities (axis 2 = tokens) and [32,128,3072] GELU-like
outputs, the seeded tensors of tests/_site_size.py:

  * every fp32 call -- the first call of every observer, and EVERY call on non-negative data, which the reference
    searches in fp32 for ever (observer.py:491) -- is bit-equal to `oracle.observer_oracle.observe_msefast`: the loss is the
    float64 sum of fp32 squares, its mean rounded to fp32 once, on both sides;
  * float64 calls (two-sided data from the second call on, observer.py:524,549) are bit-equal once both sides add
    order-independently (`osq_set_tuning("mse_sum_order", 64)` <-> `exact_mean`), and within the order noise of a plain
    float64 sum otherwise (tests/test_gpu_parity.py::test_msefast_equals_oracle states that bar).

Bit-equality with the REFERENCE's own run at these shapes is the strict switch's: tests/test_gpu_strict_order.py.
"""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from _site_size import MSE_CASES, site_input, site_lengths

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()
    return torch.device("cuda:0")


def N(t):
    return t.detach().cpu().numpy()


def _case(name):
    return next(c for c in MSE_CASES if c[0] == name)


def _stats(ob):
    return float(N(ob.min_val).reshape(-1)[0]), float(N(ob.max_val).reshape(-1)[0])


@pytest.mark.parametrize("case,calls", [("hidden768", 1), ("probs128", 3), ("gelu3072", 1)])
def test_resident_search_at_site_shape_equals_oracle(dev, case, calls, order_free):
    """One resident launch per search (default) against the oracle on the fp32 calls of the site: ranges and evaluation
    counts equal.  hidden768: 4 float4 slots per lane; probs128: 8 (head-split view, tokens on axis 2); gelu3072: 16."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import observer as OBS
    from oracle import observer_oracle as OB
    name, cls, shape, seq_pos, kind, bit, sym, batches, seed = _case(case)
    gen = torch.Generator().manual_seed(seed)
    ob = getattr(OBS, cls)(bit=bit, symmetric=sym, ch_axis=-1).to(dev)
    st = OB.ObserverState(bit=bit, symmetric=sym, ch_axis=-1)
    counter = [0]
    evals = 0
    for r in range(calls):
        x = site_input(gen, shape, kind, r)
        L = site_lengths(gen, shape, seq_pos)
        valid = int(L.sum()) * (x.numel() // (shape[0] * shape[seq_pos]))
        assert ops.msefast_resident_slots(valid) > 0, "the site must take the resident path"
        ob(x.to(dev), L.to(dev), seq_pos)
        evals += int(ob.last_nfev.sum().item())
        OB.observe_msefast(st, x.numpy(), L.numpy(), seq_pos, average=True, counter=counter)
        assert _stats(ob) == (float(st.min_val), float(st.max_val)), (case, r, _stats(ob), float(st.min_val), float(st.max_val))
        assert evals == counter[0], (case, r, evals, counter[0])


def test_float64_call_at_site_shape_equals_oracle_with_exact_sums(dev):
    """[32,128,768] masked, second call = float64 arithmetic on 1.7 M elements: with order-independent sums on both sides the
    running mean after the second batch and the evaluation count are EQUAL; with the default (plain float64 sums, resident
    launch) the range stays within the order noise stated in test_msefast_equals_oracle."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import observer as OBS
    from oracle import observer_oracle as OB
    name, cls, shape, seq_pos, kind, bit, sym, batches, seed = _case("hidden768")
    gen = torch.Generator().manual_seed(seed)
    xs = [(site_input(gen, shape, kind, r), site_lengths(gen, shape, seq_pos)) for r in range(2)]
    st = OB.ObserverState(bit=bit, symmetric=sym, ch_axis=-1)
    counter = [0]
    old = OB.MEAN_LIKE_TORCH
    OB.MEAN_LIKE_TORCH = OB.exact_mean
    try:
        for x, L in xs:
            OB.observe_msefast(st, x.numpy(), L.numpy(), seq_pos, average=True, counter=counter)
    finally:
        OB.MEAN_LIKE_TORCH = old
    ops.set_tuning("mse_sum_order", 64)
    try:
        ob = OBS.AvgMSEFastObserver(bit=bit, symmetric=sym, ch_axis=-1).to(dev)
        evals = 0
        for x, L in xs:
            ob(x.to(dev), L.to(dev), seq_pos)
            evals += int(ob.last_nfev.sum().item())
    finally:
        ops.set_tuning("mse_sum_order", 8)      # the default
    assert _stats(ob) == (float(st.min_val), float(st.max_val)) and evals == counter[0], (_stats(ob), st.min_val, st.max_val, evals, counter[0])
    ob2 = OBS.AvgMSEFastObserver(bit=bit, symmetric=sym, ch_axis=-1).to(dev)          # default: resident, plain float64 sums
    for x, L in xs:
        ob2(x.to(dev), L.to(dev), seq_pos)
    np.testing.assert_allclose(_stats(ob2), (float(st.min_val), float(st.max_val)), rtol=3e-2)


def test_multi_site_launch_at_site_shapes_equals_oracle(dev, order_free):
    """The searches of one forward in ONE launch (msefast_resident_multi_kernel through deferred_observation): a hidden-state
    site, a probabilities site and a second hidden-state site at their BERT-base shapes -- first call of each, fp32 --
    against the oracle, site by site."""
    from outlier_suppression_amd.quantization import Quantizer
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    from oracle import observer_oracle as OB
    cfg = NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1)
    sites = []
    for case, r in (("hidden768", 0), ("probs128", 0), ("hidden768", 1)):
        name, cls, shape, seq_pos, kind, bit, sym, batches, seed = _case(case)
        gen = torch.Generator().manual_seed(seed + 50 * r)
        sites.append((case, site_input(gen, shape, kind, r), site_lengths(gen, shape, seq_pos), seq_pos))
    qs = [Quantizer(None, cfg).to(dev) for _ in sites]
    for q in qs:
        q.enable_observer()
        q.disable_fake_quant()
    with deferred_observation() as pending:
        for q, (case, x, L, seq_pos) in zip(qs, sites):
            q(x.to(dev), L.to(dev), seq_pos)
        assert len(pending.mse) == len(sites)
        pending.flush()
        assert pending.launches == 1, "the three searches must share one resident launch"
    for q, (case, x, L, seq_pos) in zip(qs, sites):
        st = OB.ObserverState(bit=6, symmetric=False, ch_axis=-1)
        counter = [0]
        OB.observe_msefast(st, x.numpy(), L.numpy(), seq_pos, average=True, counter=counter)
        assert _stats(q.observer) == (float(st.min_val), float(st.max_val)), (case, _stats(q.observer), st.min_val, st.max_val)
        assert int(q.observer.last_nfev.sum().item()) == counter[0], (case, int(q.observer.last_nfev.sum().item()), counter[0])
        scale, zp = st.qparams()
        assert np.float32(q.scale.item()) == np.float32(scale) and float(q.zero_point.item()) == float(zp), case
