"""Functional fake-quant API with the reference's names (quant_transformer/quantization/util_quant.py).

Each function is ONE fused HIP launch (forward) and, under autograd, one launch for
the backward.  ``scale`` / ``zero_point`` may be Python numbers or device tensors;
numbers are uploaded (the reference passes ``.item()`` values at fake_quant.py:124).
"""
import torch

from .. import ops
from ..ops import PARAM_FIXED, PARAM_LSQ, PARAM_LSQPLUS


def _as_scale(v, like):
    if torch.is_tensor(v):
        return v if v.dim() else v.reshape(1)
    return torch.tensor([float(v)], dtype=torch.float32, device=like.device)


def _as_zero_point(v, like, learnable=False):
    if torch.is_tensor(v):
        v = v if v.dim() else v.reshape(1)
        if v.dtype in (torch.int32, torch.float32):
            return v
        return v.to(torch.float32 if v.is_floating_point() else torch.int32)
    if learnable or isinstance(v, float) and not float(v).is_integer():
        return torch.tensor([float(v)], dtype=torch.float32, device=like.device)
    return torch.tensor([int(v)], dtype=torch.int32, device=like.device)


def round_ste(x):
    """util_quant.py:4-8.  Kept for API completeness: value round-half-even(x), gradient 1."""
    return (x.round() - x).detach() + x


def grad_scale(t, scale):
    """util_quant.py:70-71."""
    return (t - (t * scale)).detach() + (t * scale)


def fake_quantize_per_tensor_affine(x, scale, zero_point, quant_min, quant_max):
    """util_quant.py:11-15."""
    return ops.fake_quant(x, _as_scale(scale, x), _as_zero_point(zero_point, x), -1, quant_min, quant_max, PARAM_FIXED)


def fake_quantize_per_channel_affine(x, scale, zero_point, ch_axis, quant_min, quant_max):
    """util_quant.py:18-26."""
    return ops.fake_quant(x, _as_scale(scale, x), _as_zero_point(zero_point, x), ch_axis, quant_min, quant_max, PARAM_FIXED)


def fake_quantize_learnable_per_tensor_affine_training(x, scale, zero_point, quant_min, quant_max, grad_factor):
    """util_quant.py:29-34 (LSQ)."""
    return ops.fake_quant(x, _as_scale(scale, x), _as_zero_point(zero_point, x), -1, quant_min, quant_max,
                          PARAM_LSQ, grad_factor)


def fake_quantize_learnable_per_channel_affine_training(x, scale, zero_point, ch_axis, quant_min, quant_max, grad_factor):
    """util_quant.py:37-45 (LSQ)."""
    return ops.fake_quant(x, _as_scale(scale, x), _as_zero_point(zero_point, x), ch_axis, quant_min, quant_max,
                          PARAM_LSQ, grad_factor)


def fake_quantize_learnableplus_per_tensor_affine_training(x, scale, zero_point, quant_min, quant_max, grad_factor):
    """util_quant.py:48-55 (LSQ+)."""
    return ops.fake_quant(x, _as_scale(scale, x), _as_zero_point(zero_point, x, True), -1, quant_min, quant_max,
                          PARAM_LSQPLUS, grad_factor)


def fake_quantize_learnableplus_per_channel_affine_training(x, scale, zero_point, ch_axis, quant_min, quant_max,
                                                            grad_factor):
    """util_quant.py:58-67 (LSQ+)."""
    return ops.fake_quant(x, _as_scale(scale, x), _as_zero_point(zero_point, x, True), ch_axis, quant_min, quant_max,
                          PARAM_LSQPLUS, grad_factor)
