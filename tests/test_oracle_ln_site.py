"""The oracle's LayerNorm-site chain (oracle/gamma_oracle.py + observer / fake-quant oracles) against what the
REFERENCE's own wrappers returned (model/util_layernorm.py:6-52): tests/golden/gamma.npz (`ln_full`, `ln_split`, H = 48)
and tests/golden/ln_site.npz (BERT-base width, one [32,128,768] calibration batch; made by
tests/golden/make_golden_ln_site.py).  CPU only.

Bars: the normalisation is a float computation whose last bit depends on how the row moments are accumulated, so the
un-quantised output is held to BASELINE.json's 1e-5 (relative to the output's magnitude); everything downstream of it is
integer work -- with the reference's scale and zero point the oracle's integer tensor may differ from the reference's
only where the normalised value sits on a rounding boundary (a whole step, a vanishing fraction of the entries)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ln_site import CASES, FLOAT_SAMPLES, checksum, ln_site_inputs  # noqa: E402

from oracle import fake_quant_oracle as FQ, gamma_oracle as GM, observer_oracle as OB  # noqa: E402

F32 = np.float32


def test_layernorm_wrappers_tiny(golden):
    g = golden("gamma")
    full = GM.affine_layernorm(g["x"], g["gamma"], g["beta"], 1e-12)
    split = GM.non_scaling_layernorm(g["x"], GM.split_bias(g["beta"], g["gamma"]), 1e-5)
    for ours, ref in ((full, g["ln_full"]), (split, g["ln_split"])):
        assert np.abs(ours - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def oracle_site(name, cls, eps, with_gamma, x, hidden, gamma, beta):
    r = x if with_gamma is None else GM.gamma_residual(x, hidden, gamma if with_gamma else None)
    if cls == "QuantizedSplitLayerNorm":
        return GM.non_scaling_layernorm(r, GM.split_bias(beta, gamma), 1e-5)        # quirk: the split wrapper's eps is torch's default
    return GM.affine_layernorm(r, gamma, beta, eps)


def test_ln_site_chain_matches_reference(golden):
    g = golden("ln_site")
    for name, cls, eps, with_gamma, quantizer, observer, pct, seed in CASES:
        x, hidden, gamma, beta, L = (t.numpy() for t in ln_site_inputs(seed))
        assert [checksum(t) for t in ln_site_inputs(seed)[:4]] == list(g[name + "_sums"][:4]), "the seeded inputs drifted"
        y = oracle_site(name, cls, eps, with_gamma, x, hidden, gamma, beta)
        ref_ln = g[name + "_ln"]
        bar = 1e-5 * max(1.0, float(np.abs(ref_ln).max()))
        assert np.abs(y[:FLOAT_SAMPLES] - ref_ln).max() <= bar, name
        if cls == "QuantizedSplitLayerNorm":
            assert np.array_equal(GM.split_bias(beta, gamma), g[name + "_split_bias"])
        # observer on the oracle's own LayerNorm output: the statistic is an extremum / order statistic of float values
        st = OB.ObserverState(bit=6, symmetric=False, name="encoder.layer.0.output.LayerNorm")
        if pct is not None:
            st.percentile = pct
            OB.observe_avg_prune_minmax(st, y, L, 1)
        else:
            OB.observe_avg_minmax(st, y, L, 1)
        scale, zp = st.qparams()
        np.testing.assert_allclose(F32(scale), g[name + "_scale"][0], rtol=1e-5)
        assert F32(zp) == g[name + "_zp"][0]
        # the integer tensor, with the REFERENCE's parameters (so that only the normalisation is being compared)
        rs, rz = g[name + "_scale"][0], g[name + "_zp"][0]
        if quantizer == "LSQPlusFakeQuantize":
            xq, _ = FQ.fake_quantize_learnableplus_per_tensor(y, rs, rz, 0, 63, FQ.lsqplus_grad_factor(y.size, 63))
        else:
            xq, _ = FQ.fake_quantize_per_tensor_affine(y, rs, rz, 0, 63)
        d = xq.astype(np.int32) - g[name + "_xq"].astype(np.int32)
        assert np.abs(d).max() <= 1, name
        assert (d != 0).mean() <= 2e-5, (name, float((d != 0).mean()))
