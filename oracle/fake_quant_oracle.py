"""Oracle (test infrastructure): fake-quant forward / LSQ+ backward in NumPy fp32.

Follows ``quant_transformer/quantization/util_quant.py`` of the reference.  Every
operation is a separately rounded fp32 operation, in the reference's order: true
division, round-half-even, add zero-point, clamp, subtract zero-point, multiply.
"""
import numpy as np

F32 = np.float32


def _f32(a):
    return np.asarray(a, dtype=F32)


def round_ste_value(u):
    """Forward value of ``round_ste`` (util_quant.py:4-8): ``(u.round() - u) + u``.

    For finite ``u`` this equals ``rint(u)`` exactly; for +-inf it is NaN
    (inf - inf), which the reference inherits and so do we.
    """
    u = _f32(u)
    with np.errstate(invalid="ignore"):
        return (np.round(u) - u) + u


def grad_scale_value(t, g):
    """Forward value of ``grad_scale`` (util_quant.py:70-71): ``(t - t*g) + t*g``."""
    t = _f32(t)
    tg = t * F32(g)
    return (t - tg) + tg


def quantize_affine(x, scale, zero_point, quant_min, quant_max):
    """Integer-valued tensor ``x_quant`` of util_quant.py:12-13 (fp32 storage)."""
    x = _f32(x)
    scale = _f32(scale)
    zero_point = _f32(zero_point)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        x_int = round_ste_value(x / scale) + zero_point
        # np.clip propagates NaN exactly as torch.clamp does
        return np.clip(x_int, F32(quant_min), F32(quant_max))


def dequantize_affine(x_quant, scale, zero_point):
    """util_quant.py:14: ``(x_quant - zero_point) * scale``."""
    with np.errstate(invalid="ignore", over="ignore"):
        return (_f32(x_quant) - _f32(zero_point)) * _f32(scale)


def fake_quantize_per_tensor_affine(x, scale, zero_point, quant_min, quant_max):
    """util_quant.py:11-15.  ``scale``/``zero_point`` are scalars.  Returns (x_quant, x_dequant)."""
    xq = quantize_affine(x, F32(scale), F32(zero_point), quant_min, quant_max)
    return xq, dequantize_affine(xq, F32(scale), F32(zero_point))


def _broadcast_shape(x, ch_axis):
    shape = [1] * x.ndim
    shape[ch_axis] = x.shape[ch_axis]
    return shape


def fake_quantize_per_channel_affine(x, scale, zero_point, ch_axis, quant_min, quant_max):
    """util_quant.py:18-26.  ``scale``/``zero_point`` have one entry per index of ``ch_axis``."""
    x = _f32(x)
    shp = _broadcast_shape(x, ch_axis)
    s = _f32(scale).reshape(shp)
    z = _f32(zero_point).reshape(shp)
    xq = quantize_affine(x, s, z, quant_min, quant_max)
    return xq, dequantize_affine(xq, s, z)


def lsq_effective_scale(scale, grad_factor):
    """util_quant.py:30,40: forward value of ``grad_scale(scale, g)``."""
    return grad_scale_value(scale, grad_factor)


def lsqplus_effective_params(scale, zero_point, grad_factor):
    """util_quant.py:49-51 / 59-63: values of scale and zero-point that reach the quantizer.

    zero_point <- (zp.round() - zp) + zp ; scale <- grad_scale(scale) ; zp <- grad_scale(zp).
    """
    zp = _f32(zero_point)
    zp = (np.round(zp) - zp) + zp
    return grad_scale_value(scale, grad_factor), grad_scale_value(zp, grad_factor)


def fake_quantize_learnableplus_per_tensor(x, scale, zero_point, quant_min, quant_max, grad_factor):
    """util_quant.py:48-55 forward.  Returns (x_quant, x_dequant)."""
    s, z = lsqplus_effective_params(scale, zero_point, grad_factor)
    s = s.reshape(()) if s.size == 1 else s
    z = z.reshape(()) if z.size == 1 else z
    xq = quantize_affine(x, s, z, quant_min, quant_max)
    return xq, dequantize_affine(xq, s, z)


def fake_quantize_learnableplus_per_channel(x, scale, zero_point, ch_axis, quant_min, quant_max, grad_factor):
    """util_quant.py:58-67 forward."""
    x = _f32(x)
    s, z = lsqplus_effective_params(scale, zero_point, grad_factor)
    shp = _broadcast_shape(x, ch_axis)
    s, z = s.reshape(shp), z.reshape(shp)
    xq = quantize_affine(x, s, z, quant_min, quant_max)
    return xq, dequantize_affine(xq, s, z)


def fake_quantize_learnable_per_tensor(x, scale, zero_point, quant_min, quant_max, grad_factor):
    """util_quant.py:29-34 forward (LSQ: integer zero-point, learnable scale)."""
    s = lsq_effective_scale(scale, grad_factor)
    s = s.reshape(()) if s.size == 1 else s
    xq = quantize_affine(x, s, F32(zero_point), quant_min, quant_max)
    return xq, dequantize_affine(xq, s, F32(zero_point))


def lsqplus_backward_per_tensor(x, grad_out, scale, zero_point, quant_min, quant_max, grad_factor):
    """Gradients autograd produces for util_quant.py:48-55 (per-tensor LSQ+).

    Elementwise parts are written with the fp32 operations autograd executes
    (mul backward ``gy*s``, clamp mask, div backward ``g/s`` and
    ``-g*((x/s)/s)``); the two reductions are accumulated in float64 because the
    reference's own fp32 summation order is an implementation detail of torch.
    Returns (dx [fp32], dscale [float64 scalar], dzero_point [float64 scalar]).
    """
    x = _f32(x)
    gy = _f32(grad_out)
    s, z = lsqplus_effective_params(scale, zero_point, grad_factor)
    s = F32(s.reshape(-1)[0])
    z = F32(z.reshape(-1)[0])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        u = x / s
        x_int = round_ste_value(u) + z
        inside = (x_int >= F32(quant_min)) & (x_int <= F32(quant_max))
        xq = np.clip(x_int, F32(quant_min), F32(quant_max))
        g_mul = gy * s                       # d/d(xq - z) of (xq - z) * s
        g_in = np.where(inside, g_mul, F32(0))
        dx = g_in / s                        # div backward wrt numerator
        ds_mul = gy * (xq - z)               # mul backward wrt s
        ds_div = -g_in * ((x / s) / s)       # div backward wrt denominator
        dz_elem = g_in - g_mul               # +1 through x_int, -1 through (xq - z)
    g = float(grad_factor)
    dscale = (ds_mul.astype(np.float64).sum() + ds_div.astype(np.float64).sum()) * g
    dzp = dz_elem.astype(np.float64).sum() * g
    return dx.astype(F32), dscale, dzp


def lsqplus_backward_per_tensor_reference_order(x, grad_out, scale, zero_point, quant_min, quant_max, grad_factor, vec=8):
    """The same gradients with the reductions done the way autograd does them on the reference's CPU (one thread):
    scale.grad and zero_point.grad are each the fp32 sum of TWO ``sum_to_size`` reductions -- mul backward and div
    backward for the scale, add backward and sub backward for the zero point (util_quant.py:48-55) --, every reduction
    torch's fp32 ``sum`` in ATen's order (oracle/aten_sum.py), then grad_scale's factor (util_quant.py:70-71) as an fp32
    multiplication.  Equal to the reference's own run (tests/golden/lsqplus.npz) bit for bit.
    Returns (dx [fp32], dscale [fp32 scalar], dzero_point [fp32 scalar])."""
    from .aten_sum import aten_sum_flat
    x = _f32(x)
    gy = _f32(grad_out)
    s, z = lsqplus_effective_params(scale, zero_point, grad_factor)
    s = F32(s.reshape(-1)[0])
    z = F32(z.reshape(-1)[0])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        u = x / s
        x_int = round_ste_value(u) + z
        inside = (x_int >= F32(quant_min)) & (x_int <= F32(quant_max))
        xq = np.clip(x_int, F32(quant_min), F32(quant_max))
        g_mul = gy * s
        g_in = np.where(inside, g_mul, F32(0))
        dx = g_in / s
        ds_mul = gy * (xq - z)
        ds_div = -g_in * ((x / s) / s)
    total = lambda a: aten_sum_flat(a, vec, np.float32)   # noqa: E731  (any length: the one-thread order)
    g = F32(grad_factor)
    dscale = F32(F32(total(ds_mul) + total(ds_div)) * g)
    dzp = F32(F32(total(g_in) + total(-g_mul)) * g)
    return dx.astype(F32), dscale, dzp


def lsqplus_backward_per_channel_reference_order(x, grad_out, scale, zero_point, quant_min, quant_max, grad_factor, vec=8):
    """Per-channel (ch_axis = 0, x = [channels, inner]) counterpart: sum_to_size reduces every row with torch's fp32 ``sum``
    over the contiguous inner axis (the same cascade per row).  Returns (dx, dscale [channels], dzero_point [channels]),
    equal to tests/golden/lsqplus.npz's ``pc_*`` bit for bit."""
    from .aten_sum import aten_sum
    x = _f32(x)
    gy = _f32(grad_out)
    s, z = lsqplus_effective_params(scale, zero_point, grad_factor)
    s = np.asarray(s, F32).reshape(-1, 1)
    z = np.asarray(z, F32).reshape(-1, 1)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        u = x / s
        x_int = round_ste_value(u) + z
        inside = (x_int >= F32(quant_min)) & (x_int <= F32(quant_max))
        xq = np.clip(x_int, F32(quant_min), F32(quant_max))
        g_mul = gy * s
        g_in = np.where(inside, g_mul, F32(0))
        dx = g_in / s
        ds_mul = gy * (xq - z)
        ds_div = -g_in * ((x / s) / s)
    rows = lambda a: aten_sum(np.ascontiguousarray(a, dtype=F32), vec, np.float32)   # noqa: E731
    g = F32(grad_factor)
    dscale = ((rows(ds_mul) + rows(ds_div)).astype(F32) * g).astype(F32)
    dzp = ((rows(g_in) + rows(-g_mul)).astype(F32) * g).astype(F32)
    return dx.astype(F32), dscale, dzp


def lsqplus_grad_factor(numel, quant_max, channels=None):
    """fake_quant.py:195-204: ``1/sqrt(numel*qmax)`` or ``1/sqrt(numel/C*qmax)`` (Python float)."""
    if channels is None:
        return 1.0 / (numel * quant_max) ** 0.5
    return 1.0 / (numel / channels * quant_max) ** 0.5
