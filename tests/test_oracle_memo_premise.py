"""The premise of the loss memo of the per-tensor MSEFast searches (csrc/msefast.hip, tensor_search_advance), on the CPU with the oracle.

`loss_fx` (quantization/observer.py:423-432) sees a candidate only through the pair it hands to the fake-quant -- `scale.item()` and
`int(zero_point.item())` -- so two candidates with the same pair have the same loss, bit for bit.  The test runs the oracle's nested
search (observer.py:434-481 restated) with a hook on `calculate_qparams`, and checks (a) that equal pairs did give equal losses, every
time, and (b) how often scipy's bounded search repeats a pair: most evaluations on two-sided data (the inner search moves the shift
of a fixed range: the zero point changes once per quantisation step), few when the data's minimum clips the range (GELU-like outputs:
the scale then moves with every shift).  The kernels answer the repeats from a table (tests/test_gpu_mse_memo.py holds results with
and without it against each other on the GPU); these numbers are why that is worth doing and where it is not."""
import numpy as np
import pytest

from oracle import observer_oracle as O


class _State:
    quant_min, quant_max, symmetric, one_side_dist, ch_axis = 0, 63, False, "no", -1


def _run(x, monkeypatch):
    pairs, losses = [], []
    real_qparams, real_loss = O.calculate_qparams, O.mse_loss

    def qparams(mn, mx, qmin, qmax, sym):
        s, z = real_qparams(mn, mx, qmin, qmax, sym)
        word = np.float64(float(s)).tobytes() if x.dtype == np.float64 else np.float32(float(s)).tobytes()
        pairs.append((word, int(z)))
        return s, z

    def loss(*a, **k):
        v = real_loss(*a, **k)
        losses.append(np.asarray(v).tobytes())
        return v

    monkeypatch.setattr(O, "calculate_qparams", qparams)
    monkeypatch.setattr(O, "mse_loss", loss)
    O.msefast_search_2d(x, x.min(), x.max(), _State())
    assert len(pairs) == len(losses)
    return pairs, losses


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_equal_pairs_have_equal_losses_and_most_pairs_repeat_on_two_sided_data(monkeypatch, dtype):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(20000).astype(np.float32)
    x[::997] *= 11
    pairs, losses = _run(x.astype(dtype), monkeypatch)
    seen = {}
    for p, l in zip(pairs, losses):
        assert seen.setdefault(p, l) == l, "the same (scale, zero point) pair gave two different losses"
    assert len(seen) * 2 < len(pairs), (len(seen), len(pairs))          # measured: 80-110 distinct pairs of 400-700 evaluations


def test_data_clipped_at_its_minimum_repeats_fewer_pairs(monkeypatch):
    """GELU-like data in the float64 arithmetic of a site's second call on: where the range's lower end sits at the data's minimum the
    scale moves with every shift, and a larger share of the evaluations is distinct than on two-sided data."""
    rng = np.random.default_rng(4)
    two_sided = rng.standard_normal(20000)
    clipped = np.maximum(rng.standard_normal(20000), -0.17) * 3
    share = []
    for x in (two_sided, clipped):
        pairs, losses = _run(x.astype(np.float32).astype(np.float64), monkeypatch)
        seen = {}
        for p, l in zip(pairs, losses):
            assert seen.setdefault(p, l) == l
        share.append(len(seen) / len(pairs))
    assert share[0] < 0.5 and share[1] > share[0], share
