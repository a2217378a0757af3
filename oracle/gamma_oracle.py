"""Oracle (test infrastructure): Gamma-Migration arithmetic in NumPy fp32.

Follows ``quant_transformer/solver/gamma_migration.py:46-76`` and
``quant_transformer/model/util_layernorm.py:21-52``.
"""
import numpy as np

from .fake_quant_oracle import F32


def fold_gamma_into_weight(weight, gamma):
    """gamma_migration.py:70-71: ``W *= gamma`` with gamma broadcast over the in-features (columns)."""
    w = np.asarray(weight, dtype=F32)
    g = np.asarray(gamma, dtype=F32)
    assert w.shape[-1] == g.shape[0]
    return w * g[None, :]


def split_bias(beta, gamma):
    """util_layernorm.py:27: bias of the non-scaling LayerNorm, ``beta / gamma``."""
    return np.asarray(beta, dtype=F32) / np.asarray(gamma, dtype=F32)


def non_scaling_layernorm(x, bias, eps=1e-5):
    """util_layernorm.py:26,32-34: ``layer_norm(x, no affine, eps=1e-5) + beta/gamma``.

    Normalisation statistics are accumulated in float64 and rounded to fp32
    (torch's CPU layer_norm accumulates in a wider type as well); comparisons use
    a 1e-5 tolerance, not bit equality.
    """
    x64 = np.asarray(x, dtype=np.float64)
    mu = x64.mean(axis=-1, keepdims=True)
    var = x64.var(axis=-1, keepdims=True)
    y = ((x64 - mu) / np.sqrt(var + eps)).astype(F32)
    return y + np.asarray(bias, dtype=F32)


def affine_layernorm(x, weight, bias, eps):
    """util_layernorm.py:15 (QuantizedLayerNorm keeps the model's own ``nn.LayerNorm``): ``layer_norm(x) * weight + bias``
    with the moments in float64, the normalised value rounded to fp32 once, then the affine pair in fp32 -- the order
    of torch's CPU kernel up to its last-bit rounding; compared at 1e-5, not bit for bit."""
    x64 = np.asarray(x, dtype=np.float64)
    mu = x64.mean(axis=-1, keepdims=True)
    var = x64.var(axis=-1, keepdims=True)
    y = ((x64 - mu) / np.sqrt(var + eps)).astype(F32)
    return y * np.asarray(weight, dtype=F32) + np.asarray(bias, dtype=F32)


def gamma_residual(shortcut, hidden, gamma=None):
    """util_layernorm.py:49-52: ``input * gamma + hidden`` (gamma absent before migration)."""
    s = np.asarray(shortcut, dtype=F32)
    if gamma is not None:
        s = s * np.asarray(gamma, dtype=F32)
    return s + np.asarray(hidden, dtype=F32)
