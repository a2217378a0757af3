"""Seeded inputs of the LayerNorm-site fixture (tests/golden/ln_site.npz): shared by the generator that runs the
reference (tests/golden/make_golden_ln_site.py) and by the tests that re-draw the same tensors.  The inputs are not
stored (12.6 MB each at [32,128,768]); torch's CPU generator is deterministic for a given build and the fixture keeps
each tensor's bit-pattern checksum (tests/_site_size.py::checksum)."""
import torch

from _site_size import checksum, site_lengths  # noqa: F401  (re-exported)

H = 768
SHAPE = (32, 128, H)          # BERT-base hidden states of one calibration batch (sizes table)
FLOAT_SAMPLES = 2             # samples whose un-quantised LayerNorm output the fixture stores in full

# name, wrapper class, LayerNorm eps, GammaResidual carries gamma, quantizer, observer, percentile, seed
CASES = (
    # before Gamma Migration: affine LayerNorm (BERT eps), plain residual, the MinMax flow's quantizer pair
    ("full_fixed", "QuantizedLayerNorm", 1e-12, False, "FixedFakeQuantize", "AvgMinMaxObserver", None, 5101),
    # after Gamma Migration: non-scaling LayerNorm + beta/gamma, shortcut * gamma, the twc_fine_gamma quantizer pair
    ("split_lsqplus", "QuantizedSplitLayerNorm", 1e-12, True, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", 0.95, 5102),
    # the embedding LayerNorm keeps its affine pair but sits behind a migrated shortcut in no model; without a residual
    ("full_lsqplus_nores", "QuantizedLayerNorm", 1e-12, None, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", 0.9, 5103),
)


def ln_site_inputs(seed, shape=SHAPE):
    """(x, hidden, gamma, beta, lengths): the shortcut (hidden states with six outlier channels), the sub-layer output
    added to it, a LayerNorm weight with outlier entries (what Gamma Migration exists for), its bias, valid lengths."""
    gen = torch.Generator().manual_seed(seed)
    h = shape[-1]
    x = torch.randn(*shape, generator=gen)
    idx = torch.randperm(h, generator=gen)[:6]
    x[..., idx] *= 20.0
    hidden = torch.randn(*shape, generator=gen)
    gamma = torch.rand(h, generator=gen) * 1.5 + 0.2
    gamma[idx[:3]] = torch.tensor([6.0, 4.5, 0.05])
    beta = torch.randn(h, generator=gen) * 0.3
    lengths = site_lengths(gen, shape, 1)
    return x, hidden, gamma, beta, lengths
