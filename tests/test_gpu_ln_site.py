"""The LayerNorm site of a quantized block against the REFERENCE's own wrappers
(/root/reference/quant_transformer/model/util_layernorm.py:6-52, run by tests/golden/make_golden_ln_site.py and
make_golden.py::gen_gamma): QuantizedLayerNorm / QuantizedSplitLayerNorm / GammaResidual of this package, in BOTH forms --
the eager sequence (torch-ROCm's LayerNorm followed by the HIP quantizer) and the one-launch site
(csrc/layernorm.hip) -- on gamma.npz (H = 48) and ln_site.npz (BERT-base width, [32,128,768]).

Bars: un-quantised LayerNorm output within 1e-5 (relative to the output's magnitude) of the reference CPU run; scale to
1e-5 relative, zero point equal; with the reference's scale / zero point the integer tensor differs from the reference's
only by whole steps on a vanishing fraction of entries (values that sit on a rounding boundary)."""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
from conftest import bits_equal
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ln_site import CASES, FLOAT_SAMPLES, checksum, ln_site_inputs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()
    return torch.device("cuda:0")


@pytest.fixture(params=["eager", "one-launch"])
def form(request):
    from outlier_suppression_amd import util_layernorm as UL
    old = UL.FUSE_LAYERNORM
    UL.FUSE_LAYERNORM = request.param == "one-launch"
    yield request.param
    UL.FUSE_LAYERNORM = old


def build_site(cls, eps, with_gamma, quantizer, observer, pct, gamma, beta, dev):
    from outlier_suppression_amd import util_layernorm as UL
    ln = torch.nn.LayerNorm(gamma.numel(), eps=eps)
    with torch.no_grad():
        ln.weight.copy_(gamma)
        ln.bias.copy_(beta)
    ln = ln.to(dev)
    cfg = NS(quantizer=quantizer, observer=observer, bit=6, symmetric=False, ch_axis=-1)
    mod = getattr(UL, cls)(ln, cfg, cfg, qoutput=True).to(dev).eval()
    q = mod.layernorm_post_act_fake_quantize
    q.observer.set_name("encoder.layer.0.output.LayerNorm.layernorm_post_act_fake_quantize.observer")
    if pct is not None:
        q.observer.set_percentile(pct)
    res = UL.GammaResidual()
    if with_gamma:
        res.set_gamma(ln.weight.data)
    return mod, res.to(dev), q


def run_site(mod, res, with_gamma, x, hidden, L):
    from outlier_suppression_amd import util_layernorm as UL
    if with_gamma is None:
        return mod(x.clone(), L)
    return UL.residual_layernorm(res, mod, x, hidden, L)       # what the model files call (quant_bert.py:211-216)


def test_layernorm_wrappers_tiny(golden, form, dev):
    """gamma.npz: the reference's wrappers with their quantizers in the default (all off) state."""
    g = golden("gamma")
    x = torch.from_numpy(g["x"]).to(dev)
    gamma, beta = torch.from_numpy(g["gamma"]), torch.from_numpy(g["beta"])
    for cls, key in (("QuantizedLayerNorm", "ln_full"), ("QuantizedSplitLayerNorm", "ln_split")):
        mod, _, _ = build_site(cls, 1e-12, None, "FixedFakeQuantize", "AvgMinMaxObserver", None, gamma, beta, dev)
        with torch.no_grad():
            y = mod(x.clone()).cpu().numpy()
        ref = g[key]
        assert np.abs(y - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (cls, form)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_ln_site_matches_reference(golden, form, case, dev):
    name, cls, eps, with_gamma, quantizer, observer, pct, seed = case
    g = golden("ln_site")
    x, hidden, gamma, beta, L = ln_site_inputs(seed)
    assert [checksum(t) for t in (x, hidden, gamma, beta)] == list(g[name + "_sums"][:4]), "the seeded inputs drifted"
    mod, res, q = build_site(cls, eps, with_gamma, quantizer, observer, pct, gamma, beta, dev)
    x, hidden, L = x.to(dev), hidden.to(dev), L.to(dev)
    with torch.no_grad():
        q.enable_observer(); q.disable_fake_quant()
        y_obs = run_site(mod, res, with_gamma, x, hidden, L)
        ref_ln = g[name + "_ln"]
        bar = 1e-5 * max(1.0, float(np.abs(ref_ln).max()))
        assert np.abs(y_obs[:FLOAT_SAMPLES].cpu().numpy() - ref_ln).max() <= bar, (name, form)
        if cls == "QuantizedSplitLayerNorm":
            assert bits_equal(mod.bias.data.cpu().numpy(), g[name + "_split_bias"])
        np.testing.assert_allclose(q.scale.detach().cpu().numpy().reshape(-1), g[name + "_scale"], rtol=1e-5)
        assert bits_equal(q.zero_point.detach().cpu().numpy().reshape(-1).astype(np.float32), g[name + "_zp"])
        # quantized pass with the REFERENCE's parameters: only the normalisation is being compared
        q.disable_observer(); q.enable_fake_quant()
        rs, rz = float(g[name + "_scale"][0]), float(g[name + "_zp"][0])
        q.scale.data.fill_(rs)
        q.zero_point.data.fill_(int(rz) if q.zero_point.dtype == torch.int32 else rz)
        y_q = run_site(mod, res, with_gamma, x, hidden, L).cpu().numpy()
    ref_xq = g[name + "_xq"].astype(np.float32)
    ref_y = (ref_xq - np.float32(rz)) * np.float32(rs)                       # util_quant.py:14, exact reconstruction
    steps = np.rint((y_q - ref_y) / np.float32(rs))
    assert np.abs(y_q - ref_y - steps * np.float32(rs)).max() <= 1e-5 * max(1.0, float(np.abs(ref_y).max()))
    assert np.abs(steps).max() <= 1, (name, form)
    assert (steps != 0).mean() <= 2e-5, (name, form, float((steps != 0).mean()))
