"""Tensor-level entry points: torch tensors in, C-ABI kernel launches out.

Everything here runs on the current HIP stream and never synchronises with the
host.  No arithmetic on tensor data happens in Python; torch only allocates.
"""
import ctypes
import os

import torch

from . import _hip
from ._hip import (PARAM_FIXED, PARAM_LSQ, PARAM_LSQPLUS, PARAM_MODE_MASK, PARAM_SANITIZE, PARAM_NO_PERSISTENT, UPDATE_AVERAGE,  # noqa: F401
                   UPDATE_NONE, UPDATE_RUNNING, ZP_FLOAT32, ZP_INT32)


def _zp_type(zero_point):
    if zero_point.dtype == torch.int32:
        return ZP_INT32
    if zero_point.dtype == torch.float32:
        return ZP_FLOAT32
    raise TypeError(f"zero_point must be int32 or float32, got {zero_point.dtype}")


def _check_f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"outlier_suppression_amd kernels compute in float32, got {t.dtype}")


def is_dense(x):
    """True if x's elements occupy one gap-free block of memory (any dim order)."""
    if x.is_contiguous():
        return True
    expected = 1
    for size, stride in sorted(zip(x.shape, x.stride()), key=lambda p: (p[1], p[0])):
        if size == 1:
            continue
        if stride != expected:
            return False
        expected *= size
    return True


def _i64x4(vals):
    vals = list(vals)
    vals = [1] * (4 - len(vals)) + vals if len(vals) <= 4 else None
    if vals is None:
        raise NotImplementedError("strided fake-quant supports at most 4 dims")
    return (ctypes.c_int64 * 4)(*vals)


def _pad4_strides(x):
    st = list(x.stride())
    return (ctypes.c_int64 * 4)(*([0] * (4 - len(st)) + st))


# ---------------------------------------------------------------------------------------
# fake-quant forward
# ---------------------------------------------------------------------------------------

def fake_quant_per_tensor(x, scale, zero_point, quant_min, quant_max, mode=PARAM_FIXED, grad_factor=1.0,
                          return_quantized=False):
    """(x_dequant[, x_quant]) of util_quant.py:11-15 / 29-34 / 48-55 (forward).  scale/zero_point: 1-element device tensors."""
    lib = _hip.load()
    _hip.require_device(x, scale, zero_point)
    _check_f32(x, scale)
    st = _hip.stream_ptr(x.device)
    if (x.dim() == 4 and not return_quantized and not x.is_contiguous() and x.numel()
            and (x.stride(-1) == 1 or x.stride(-2) == 1)):
        # Head-split views of attention ([B,h,T,d] / [B,h,d,T] seen through [B,T,h,d] memory): quantise into the
        # layout the batched matmul that follows wants -- contiguous, or the transpose of a contiguous tensor for the
        # key -- instead of x's own strides, which torch.matmul would first copy out (reshape of a permuted view).
        xt = x if x.stride(-1) == 1 else x.transpose(-1, -2)
        if xt.shape[-1] % 4 == 0 and all(sv % 4 == 0 for sv in xt.stride()[:3]) and xt.data_ptr() % 16 == 0:
            yt = torch.empty(xt.shape, dtype=x.dtype, device=x.device)
            _hip.check(lib.osq_fake_quant_per_tensor_strided(
                xt.data_ptr(), yt.data_ptr(), None, _i64x4(xt.shape), _pad4_strides(xt), _pad4_strides(yt),
                _hip.ptr(scale), _hip.ptr(zero_point), _zp_type(zero_point), mode, float(grad_factor),
                int(quant_min), int(quant_max), st), "fake_quant_per_tensor_strided")
            return yt if xt is x else yt.transpose(-1, -2)
    if is_dense(x):
        y = torch.empty_like(x)
        xq = torch.empty_like(x) if return_quantized else None
        _hip.check(lib.osq_fake_quant_per_tensor(_hip.ptr(x), _hip.ptr(y), _hip.ptr(xq), x.numel(), _hip.ptr(scale),
                                                 _hip.ptr(zero_point), _zp_type(zero_point), mode, float(grad_factor),
                                                 int(quant_min), int(quant_max), st), "fake_quant_per_tensor")
    else:
        if x.dim() > 4:
            x = x.contiguous()
            return fake_quant_per_tensor(x, scale, zero_point, quant_min, quant_max, mode, grad_factor, return_quantized)
        y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        xq = torch.empty_like(y) if return_quantized else None
        _hip.check(lib.osq_fake_quant_per_tensor_strided(
            _hip.ptr(x), _hip.ptr(y), _hip.ptr(xq), _i64x4(x.shape), _pad4_strides(x), _pad4_strides(y),
            _hip.ptr(scale), _hip.ptr(zero_point), _zp_type(zero_point), mode, float(grad_factor),
            int(quant_min), int(quant_max), st), "fake_quant_per_tensor_strided")
    return (y, xq) if return_quantized else y


def _channel_split(x, ch_axis):
    ch_axis = ch_axis % x.dim()
    outer = 1
    for s in x.shape[:ch_axis]:
        outer *= s
    inner = 1
    for s in x.shape[ch_axis + 1:]:
        inner *= s
    return outer, x.shape[ch_axis], inner


def fake_quant_per_channel(x, scale, zero_point, ch_axis, quant_min, quant_max, mode=PARAM_FIXED, grad_factor=1.0,
                           return_quantized=False):
    """util_quant.py:18-26 / 37-45 / 58-67 (forward)."""
    lib = _hip.load()
    _hip.require_device(x, scale, zero_point)
    _check_f32(x, scale)
    x = x.contiguous()
    outer, channels, inner = _channel_split(x, ch_axis)
    if scale.numel() != channels or zero_point.numel() != channels:
        raise ValueError(f"per-channel fake-quant: {channels} channels but scale/zero_point have "
                         f"{scale.numel()}/{zero_point.numel()} entries")
    y = torch.empty_like(x)
    xq = torch.empty_like(x) if return_quantized else None
    _hip.check(lib.osq_fake_quant_per_channel(_hip.ptr(x), _hip.ptr(y), _hip.ptr(xq), outer, channels, inner,
                                              _hip.ptr(scale), _hip.ptr(zero_point), _zp_type(zero_point), mode,
                                              float(grad_factor), int(quant_min), int(quant_max),
                                              _hip.stream_ptr(x.device)), "fake_quant_per_channel")
    return (y, xq) if return_quantized else y


# ---------------------------------------------------------------------------------------
# backward (STE for x; LSQ/LSQ+ for scale / zero_point)
# ---------------------------------------------------------------------------------------

def _like_layout(g, x):
    """grad_out arranged in memory exactly like x (x is dense)."""
    if g.stride() == x.stride() and g.is_cuda:
        return g
    out = torch.empty_like(x)
    out.copy_(g)
    return out


def lsq_backward_per_tensor(x, grad_out, scale, zero_point, quant_min, quant_max, mode, grad_factor,
                            need_scale=True, need_zp=True):
    lib = _hip.load()
    _hip.require_device(x, grad_out, scale, zero_point)
    _check_f32(x, grad_out, scale)
    if not is_dense(x):
        x = x.contiguous()
    elif x.data_ptr() % 16:
        x = x.clone()              # the kernel reads 16 bytes per lane: a misaligned saved input (a slice of a larger buffer) is copied once
    g = _like_layout(grad_out, x)
    dx = torch.empty_like(x)
    ds = torch.empty(1, dtype=torch.float32, device=x.device) if need_scale else None
    dz = torch.empty(1, dtype=torch.float32, device=x.device) if need_zp else None
    ws = _hip.workspace(x.device)
    order = reference_sum_order("bwd")
    if order and x.numel() > 0 and ordered_sum_fits(x.numel(), order):     # set_strict(backward=True): autograd's four fp32 reductions in torch's one-thread order
        scratch, nbytes = _ordered_scratch(x.device, x.numel(), 4)
        _hip.check(lib.osq_lsq_backward_per_tensor_ordered(_hip.ptr(x), _hip.ptr(g), _hip.ptr(dx), x.numel(), _hip.ptr(scale),
                                                           _hip.ptr(zero_point), _zp_type(zero_point), mode, float(grad_factor),
                                                           int(quant_min), int(quant_max), _hip.ptr(ds), _hip.ptr(dz), order,
                                                           _hip.ptr(scratch), nbytes, _hip.ptr(ws), _hip.stream_ptr(x.device)),
                   "lsq_backward_per_tensor_ordered")
        return dx, ds, dz
    # order-free sums (the default): float64 accumulation rounded once; also what a tensor beyond the ordered kernels' capacity takes
    _hip.check(lib.osq_lsq_backward_per_tensor(_hip.ptr(x), _hip.ptr(g), _hip.ptr(dx), x.numel(), _hip.ptr(scale),
                                               _hip.ptr(zero_point), _zp_type(zero_point), mode, float(grad_factor),
                                               int(quant_min), int(quant_max), _hip.ptr(ds), _hip.ptr(dz), _hip.ptr(ws),
                                               _hip.stream_ptr(x.device)), "lsq_backward_per_tensor")
    return dx, ds, dz


def lsq_backward_per_channel(x, grad_out, scale, zero_point, ch_axis, quant_min, quant_max, mode, grad_factor,
                             need_scale=True, need_zp=True):
    lib = _hip.load()
    _hip.require_device(x, grad_out, scale, zero_point)
    _check_f32(x, grad_out, scale)
    x = x.contiguous()
    g = grad_out.contiguous()
    outer, channels, inner = _channel_split(x, ch_axis)
    dx = torch.empty_like(x)
    ds = torch.empty(channels, dtype=torch.float32, device=x.device) if need_scale else None
    dz = torch.empty(channels, dtype=torch.float32, device=x.device) if need_zp else None
    _hip.check(lib.osq_lsq_backward_per_channel(_hip.ptr(x), _hip.ptr(g), _hip.ptr(dx), outer, channels, inner,
                                                _hip.ptr(scale), _hip.ptr(zero_point), _zp_type(zero_point), mode,
                                                float(grad_factor), int(quant_min), int(quant_max), _hip.ptr(ds),
                                                _hip.ptr(dz), reference_sum_order("bwd"), _hip.stream_ptr(x.device)),
               "lsq_backward_per_channel")
    return dx, ds, dz


def lsq_sanitize_(scale, zero_point, eps, quant_min, quant_max):
    """In place: scale <- max(|scale|, eps); fp32 zero_point <- clamp(zero_point, qmin, qmax).  One launch."""
    lib = _hip.load()
    _hip.require_device(scale, zero_point)
    _check_f32(scale, zero_point)
    _hip.check(lib.osq_lsq_sanitize(_hip.ptr(scale), _hip.ptr(zero_point), scale.numel(), float(eps), int(quant_min),
                                    int(quant_max), _hip.stream_ptr(scale.device)), "lsq_sanitize")


class _FakeQuantFn(torch.autograd.Function):
    """Differentiable fake-quant: forward = one HIP launch, backward = one HIP launch."""

    @staticmethod
    def forward(ctx, x, scale, zero_point, ch_axis, quant_min, quant_max, mode, grad_factor):
        ctx.cfg = (ch_axis, quant_min, quant_max, mode & PARAM_MODE_MASK, grad_factor)    # PARAM_SANITIZE is forward-only
        ctx.save_for_backward(x, scale, zero_point)
        if ch_axis == -1:
            return fake_quant_per_tensor(x, scale, zero_point, quant_min, quant_max, mode, grad_factor)
        return fake_quant_per_channel(x, scale, zero_point, ch_axis, quant_min, quant_max, mode, grad_factor)

    @staticmethod
    def backward(ctx, grad_out):
        x, scale, zero_point = ctx.saved_tensors
        ch_axis, quant_min, quant_max, mode, grad_factor = ctx.cfg
        need_s = ctx.needs_input_grad[1]
        need_z = ctx.needs_input_grad[2] and zero_point.dtype == torch.float32
        if ch_axis == -1:
            dx, ds, dz = lsq_backward_per_tensor(x, grad_out, scale, zero_point, quant_min, quant_max, mode, grad_factor,
                                                 need_s, need_z)
        else:
            dx, ds, dz = lsq_backward_per_channel(x, grad_out, scale, zero_point, ch_axis, quant_min, quant_max, mode,
                                                  grad_factor, need_s, need_z)
        if ds is not None:
            ds = ds.reshape(scale.shape)
        if dz is not None:
            dz = dz.reshape(zero_point.shape)
        return (dx if ctx.needs_input_grad[0] else None), ds, dz, None, None, None, None, None


def fake_quant(x, scale, zero_point, ch_axis, quant_min, quant_max, mode=PARAM_FIXED, grad_factor=1.0):
    """Fake-quant with autograd when any input needs a gradient, a bare launch otherwise."""
    needs = torch.is_grad_enabled() and (x.requires_grad or scale.requires_grad or
                                         (zero_point.is_floating_point() and zero_point.requires_grad))
    if needs:
        return _FakeQuantFn.apply(x, scale, zero_point, ch_axis, quant_min, quant_max, mode, grad_factor)
    sd, zd = scale.detach(), zero_point.detach()
    if ch_axis == -1:
        return fake_quant_per_tensor(x, sd, zd, quant_min, quant_max, mode, grad_factor)
    return fake_quant_per_channel(x, sd, zd, ch_axis, quant_min, quant_max, mode, grad_factor)


# ---------------------------------------------------------------------------------------
# observers
# ---------------------------------------------------------------------------------------

def calculate_qparams(min_val, max_val, quant_min, quant_max, symmetric, scale_out=None, zero_point_out=None,
                      zp_dtype=None):
    """observer.py:101-119 on device.  Returns (scale fp32, zero_point) with zero_point int32 when
    symmetric and fp32 otherwise unless ``zp_dtype`` / ``zero_point_out`` says differently."""
    lib = _hip.load()
    _hip.require_device(min_val, max_val)
    f64 = torch.float64 in (min_val.dtype, max_val.dtype)          # torch's promotion: float64 as soon as either statistic is
    mn = min_val.detach().to(torch.float64 if f64 else torch.float32).contiguous()
    mx = max_val.detach().to(torch.float64 if f64 else torch.float32).contiguous()
    if scale_out is None:
        scale_out = torch.empty(mn.shape, dtype=torch.float32, device=mn.device)
    if zero_point_out is None:
        dt = zp_dtype if zp_dtype is not None else (torch.int32 if symmetric else torch.float32)
        zero_point_out = torch.empty(mn.shape, dtype=dt, device=mn.device)
    fn = lib.osq_calculate_qparams_f64 if f64 else lib.osq_calculate_qparams
    _hip.check(fn(_hip.ptr(mn), _hip.ptr(mx), mn.numel(), int(quant_min), int(quant_max),
                                         int(bool(symmetric)), _hip.ptr(scale_out), _hip.ptr(zero_point_out),
                                         _zp_type(zero_point_out), _hip.stream_ptr(mn.device)), "calculate_qparams")
    return scale_out, zero_point_out


class QParamSink:
    """Where a fused observer launch should write scale / zero_point (None = do not compute)."""
    __slots__ = ("scale", "zero_point")

    def __init__(self, scale=None, zero_point=None):
        self.scale, self.zero_point = scale, zero_point

    def args(self):
        if self.scale is None:
            return None, None, ZP_INT32
        return _hip.ptr(self.scale), _hip.ptr(self.zero_point), _zp_type(self.zero_point)


_NO_SINK = QParamSink()


def observe_flat(x, rule, cnt, min_val, max_val, quant_min, quant_max, symmetric, sink=None, cur=None):
    """Global min/max of a dense tensor + running statistic (+ qparams): ONE launch."""
    lib = _hip.load()
    _hip.require_device(x, min_val, max_val)
    _check_f32(x, min_val, max_val)
    if not is_dense(x):
        x = x.contiguous()
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    ws = _hip.workspace(x.device)
    _hip.check(lib.osq_observe_flat(_hip.ptr(x), x.numel(), rule, int(cnt), _hip.ptr(min_val), _hip.ptr(max_val),
                                    _hip.ptr(cur), int(quant_min), int(quant_max), int(bool(symmetric)), s_ptr, z_ptr,
                                    z_type, _hip.ptr(ws), _hip.stream_ptr(x.device)), "observe_flat")


def observe_channels(x, ch_axis, rule, cnt, min_val, max_val, quant_min, quant_max, symmetric, sink=None):
    """Per-channel min/max + running statistic (+ qparams): ONE launch."""
    lib = _hip.load()
    _hip.require_device(x, min_val, max_val)
    _check_f32(x, min_val, max_val)
    x = x.contiguous()
    outer, channels, inner = _channel_split(x, ch_axis)
    if min_val.numel() != channels or max_val.numel() != channels:
        raise ValueError("observe_channels: statistic buffers must have one entry per channel")
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    _hip.check(lib.osq_observe_channels(_hip.ptr(x), outer, channels, inner, rule, int(cnt), _hip.ptr(min_val),
                                        _hip.ptr(max_val), int(quant_min), int(quant_max), int(bool(symmetric)), s_ptr,
                                        z_ptr, z_type, _hip.stream_ptr(x.device)), "observe_channels")


_view_cache = {}


def token_view(x, seq_pos, n_lengths=None):
    """Describe x as [batch, tokens, feat_outer, feat_inner] the way observer.py:72-80 permutes it.

    With a length-B mask on a tensor whose dim 0 is larger (BART's [B*h, T, S] attention
    probabilities) ``zip`` in observer.py:82 only visits the first B rows: ``n_lengths`` trims batch.
    """
    key = (x.shape, x.stride(), seq_pos, n_lengths)
    view = _view_cache.get(key)          # a model calls each site with the same geometry over and over
    if view is not None:
        return view
    if x.dim() not in (3, 4):
        raise NotImplementedError("masked observers support 3-D and 4-D activations (observer.py:76-79)")
    seq_pos = seq_pos % x.dim()
    if seq_pos == 0:
        raise ValueError("seq_pos must not be the batch axis")
    others = [d for d in range(x.dim()) if d != seq_pos]
    sz, st = x.shape, x.stride()
    batch = sz[0] if n_lengths is None else min(sz[0], n_lengths)
    if len(others) == 3:
        fo, fi = others[1], others[2]
        outer, inner, s_outer, s_inner = sz[fo], sz[fi], st[fo], st[fi]
    else:
        fi = others[1]
        outer, inner, s_outer, s_inner = 1, sz[fi], 0, st[fi]
    view = _hip.TokenView(batch, sz[seq_pos], outer, inner, st[0], st[seq_pos], s_outer, s_inner)
    if len(_view_cache) < 4096:
        _view_cache[key] = view
    return view


_token_scratch = {}
_wide_min_slots = 32769


_tuning = {}          # what set_tuning has been given (the summation-order switches decide which entry point a call takes)


def tunable_build():
    """True when the loaded library is the -DOSQ_TUNABLE development build (`make dbg`), whose osq_set_tuning also accepts
    the performance A/B knobs; the release library holds their measured winners as compile-time constants."""
    return bool((_hip._lib or _hip.load()).osq_build_flags() & 1)


def set_tuning(key, value, lib=None):
    """Performance / path-selection knobs of the library (osq_set_tuning).  Results never change -- except for the two
    summation-order switches ("mse_sum_order", "bwd_sum_order": 8 / 16 = the reference's one-thread CPU order, see
    outlier_suppression_amd.set_strict), which pick WHICH of two roundings of the same sum is returned.
    "bwd_sum_order" is host state only: the backward's entry points take the order as an argument per call."""
    if key == "bwd_sum_order":
        if int(value) not in (0, 8, 16):
            raise ValueError("bwd_sum_order must be 0, 8 or 16")
    else:
        _hip.check((lib or _hip.load()).osq_set_tuning(key.encode(), int(value)), f"set_tuning({key})")
    _tuning[key] = int(value)


def reference_sum_order(kind):
    """SIMD width (8 / 16) of the reference host whose summation order the "mse" / "bwd" sums follow, or 0."""
    v = _tuning.get("mse_sum_order" if kind == "mse" else "bwd_sum_order", 0)
    return v if v in (8, 16) else 0


def ordered_sum_fits(n, lanes):
    """Whether the reference-order kernels take a vector of n elements summed on `lanes` SIMD lanes (csrc/aten_order.h:
    cascade step 2^P with P <= 5, i.e. up to 2^23 rows of 4 * lanes columns: 268 M fp32 / 134 M float64 elements at 8 lanes).
    Larger tensors keep the order-free sums."""
    size = (int(n) // lanes) // 4
    p = max(4, ((size - 1).bit_length() if size > 1 else 0) // 4)
    return p <= 5


def _ordered_scratch(device, n, n_sums):
    nbytes = int(_hip.load().osq_ordered_sum_scratch_bytes(int(n), int(n_sums)))
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def set_wide_min_slots(slots):
    """Token-slot count from which the finaliser uses its three multi-workgroup launches (default 32769:
    above the register capacity of the two-workgroup kernel)."""
    global _wide_min_slots
    _hip.check(_hip.load().osq_set_wide_min_slots(int(slots)), "set_wide_min_slots")
    _wide_min_slots = int(slots)



# ---------------------------------------------------------------------------------------
# persistent launches: time-outs must not stay silent
# ---------------------------------------------------------------------------------------
# The one-launch observe + fake-quant step and the resident MSEFast searches are grids whose workgroups wait for each
# other; when they are not resident together (another process on the GPU, another stream's long kernel on the CUs) a
# bounded wait expires, the launch writes NaN and raises a sticky flag in its workspace.  Every such launch marks its
# (device, stream) workspace here; check_persistent() -- called at the host's synchronisation points: the state
# togglers, state_dict(), the end of calibrate / find_ratio / learn_scale / calibrate_sharded / ptq.run, the flush of a
# deferred observer pass that ran searches -- reads the flags of the marked workspaces (one stream synchronisation
# each, nothing when nothing persistent ran), resets the state blocks and raises.
_persistent_dirty = set()


class PersistentLaunchTimeout(RuntimeError):
    pass


def _mark_persistent(device):
    _persistent_dirty.add((_hip._device_index(device), _hip.raw_stream(device)))


def check_persistent(where=""):
    """Raise PersistentLaunchTimeout if a persistent launch since the last check timed out.  Synchronises the streams
    that ran persistent launches since then; free when there were none."""
    if not _persistent_dirty:
        return
    lib = _hip.load()
    failed = []
    for key in sorted(_persistent_dirty):
        ws = _hip._workspaces.get(key)
        if ws is None:
            continue
        f, r = ctypes.c_int(0), ctypes.c_int(0)
        with torch.cuda.device(key[0]):
            _hip.check(lib.osq_persistent_status(ws.data_ptr(), ctypes.byref(f), ctypes.byref(r), 1, key[1]), "persistent_status")
        if f.value or r.value:
            failed.append((key[0], f.value, r.value))
    _persistent_dirty.clear()
    if failed:
        what = "; ".join(f"device {d}: fused observe+fake-quant step status {f}, resident MSEFast search status {r}" for d, f, r in failed)
        raise PersistentLaunchTimeout(
            f"outlier_suppression_amd{' (' + where + ')' if where else ''}: a persistent launch timed out waiting for its own "
            f"workgroups ({what}).  Its outputs and the statistics / scale / zero_point it wrote are NaN-poisoned: everything "
            "computed since the previous check is invalid.  Cause: the grid was not resident together -- another process is "
            "using this GPU, or another stream of this process kept CUs busy.  Set OSQ_FUSED_STEP=0 (launch-per-stage "
            "kernels, no cross-workgroup waits) when the GPU is shared.  The launch state has been reset.")


def _scratch(device, n):
    """(token_min, token_max, list_scratch): per (device, stream) scratch, grown on demand.  The three
    views are cached per slot count: building tensor views costs microseconds on a path that runs per site."""
    key = (device.index, _hip.raw_stream(device))
    entry = _token_scratch.get(key)
    if entry is not None:
        views = entry[1].get(n)
        if views is not None:
            return views
    if entry is None or entry[0].numel() < 4 * n:
        entry = (torch.empty(4 * max(n, 4096), dtype=torch.float32, device=device), {})
        _token_scratch[key] = entry
    buf = entry[0]
    views = (buf[:n], buf[n:2 * n], buf[2 * n:4 * n])
    if len(entry[1]) < 64:
        entry[1][n] = views
    return views


def token_minmax(x, seq_pos, lengths=None, out=None):
    """Per-token (min, max) over features for tokens t < lengths[b]; padded slots untouched."""
    lib = _hip.load()
    _hip.require_device(x, lengths)
    _check_f32(x)
    if lengths is not None and lengths.dtype != torch.int64:
        lengths = lengths.to(torch.int64)
    view = token_view(x, seq_pos, None if lengths is None else lengths.numel())
    n = view.batch * view.tokens
    tmin, tmax = out if out is not None else _scratch(x.device, n)[:2]
    if tmin.numel() < n or tmax.numel() < n:
        raise ValueError(f"token_minmax: the output rows hold {tmin.numel()} slots, this tensor has {n} (batch x tokens)")
    _hip.check(lib.osq_token_minmax(_hip.ptr(x), ctypes.byref(view), _hip.ptr(lengths), _hip.ptr(tmin), _hip.ptr(tmax),
                                    _hip.stream_ptr(x.device)), "token_minmax")
    return tmin, tmax, view.batch, view.tokens, lengths


def token_range_finalize(tmin, tmax, batch, tokens, lengths, prune, percentile, rule, cnt, min_val, max_val,
                         quant_min, quant_max, symmetric, sink=None, cur=None):
    """Percentile pruning / plain extrema over valid tokens + running statistic (+ qparams): ONE launch."""
    lib = _hip.load()
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    dev = tmin.device
    lst = _scratch(dev, batch * tokens)[2] if batch * tokens >= _wide_min_slots else None
    _hip.check(lib.osq_token_range_finalize(_hip.ptr(tmin), _hip.ptr(tmax), batch, tokens, _hip.ptr(lengths),
                                            int(bool(prune)), float(percentile if prune else 1.0), rule, int(cnt),
                                            _hip.ptr(min_val), _hip.ptr(max_val), _hip.ptr(cur), int(quant_min),
                                            int(quant_max), int(bool(symmetric)), s_ptr, z_ptr, z_type,
                                            _hip.ptr(_hip.workspace(dev)), _hip.ptr(lst), _hip.stream_ptr(dev)),
               "token_range_finalize")


def observe_tokens(x, seq_pos, lengths, prune, percentile, rule, cnt, min_val, max_val, quant_min, quant_max,
                   symmetric, sink=None, cur=None):
    """token_minmax + token_range_finalize behind ONE call of the binding (the per-site hot path of a
    calibration forward: host time per quantizer call is of the order of the kernels' own run time).
    Returns (batch, tokens, lengths_int64)."""
    lib = _hip.load()
    _hip.require_device(x, lengths)
    _check_f32(x)
    if lengths is not None and lengths.dtype != torch.int64:
        lengths = lengths.to(torch.int64)
    view = token_view(x, seq_pos, None if lengths is None else lengths.numel())
    dev = x.device
    n = view.batch * view.tokens
    tmin, tmax, lst = _scratch(dev, n)
    s_ptr, z_ptr, z_type = (sink or _NO_SINK).args()
    rc = lib.osq_observe_tokens(x.data_ptr(), ctypes.byref(view), _hip.ptr(lengths), tmin.data_ptr(), tmax.data_ptr(),
                                1 if prune else 0, float(percentile) if prune else 1.0, rule, int(cnt),
                                _hip.ptr(min_val), _hip.ptr(max_val), _hip.ptr(cur), int(quant_min), int(quant_max),
                                1 if symmetric else 0, s_ptr, z_ptr, z_type, _hip.workspace(dev).data_ptr(),
                                lst.data_ptr() if n >= _wide_min_slots else None, _hip.raw_stream(dev))
    if rc != 0:
        _hip.check(rc, "observe_tokens")
    return view.batch, view.tokens, lengths


def observe_tokens_fake_quant(x, seq_pos, lengths, prune, percentile, rule, cnt, min_val, max_val, quant_min, quant_max,
                              symmetric, scale, zero_point, mode, grad_factor, cur=None, persistent=True):
    """A whole quantizer call (observe the masked activation, refresh scale / zero_point, fake-quantise) behind ONE
    call of the binding.  x: dense fp32 on the device.  cur: 2-float device slot that also receives this batch's own
    (min, max) while the running statistic moves as usual.  persistent=False: THIS call runs as three ordinary launches
    instead of the one-launch persistent form (a grid that needs every CU of the device for itself -- a caller that shares the
    GPU across streams or tenants opts out per call; same results).  Returns (y, batch, tokens, lengths_int64)."""
    if not persistent:
        mode = mode | PARAM_NO_PERSISTENT
    lib = _hip._lib or _hip.load()
    if not (lengths.is_cuda and scale.is_cuda and zero_point.is_cuda and min_val.is_cuda):
        _hip.require_device(x, lengths, scale, zero_point, min_val, max_val)
    if lengths.dtype != torch.int64:
        if lengths.is_floating_point():
            raise TypeError("observation_mask must hold integer lengths")
        lengths = lengths.to(torch.int64)
    view = token_view(x, seq_pos, lengths.numel())
    dev = x.device
    n = view.batch * view.tokens
    tmin, tmax, lst = _scratch(dev, n)
    y = torch.empty_like(x)
    rc = lib.osq_observe_tokens_fake_quant(x.data_ptr(), ctypes.byref(view), lengths.data_ptr(), tmin.data_ptr(), tmax.data_ptr(),
                                           1 if prune else 0, float(percentile) if prune else 1.0, rule, cnt,
                                           min_val.data_ptr(), max_val.data_ptr(), None if cur is None else cur.data_ptr(),
                                           quant_min, quant_max, 1 if symmetric else 0,
                                           scale.data_ptr(), zero_point.data_ptr(), _zp_type(zero_point), y.data_ptr(), x.numel(),
                                           mode, grad_factor, _hip.workspace(dev).data_ptr(),
                                           lst.data_ptr() if n >= _wide_min_slots else None, _hip.raw_stream(dev))
    if rc != 0:
        _hip.check(rc, "observe_tokens_fake_quant")
    _persistent_dirty.add((dev.index, _hip.raw_stream(dev)))
    return y, view.batch, view.tokens, lengths


def token_range_finalize_batched(token_min, token_max, n_quantizers, n_batches, batch, tokens, lengths, prune_flags,
                                 percentile, cur_table):
    """Re-threshold the cached per-token extrema of every (quantizer, batch) pair in ONE launch.
    token_min/token_max: [n_quantizers, n_batches, batch*tokens] fp32; lengths: [n_batches, batch] int64 (one mask for
    every quantizer), [n_quantizers, n_batches, batch] (every quantizer its own) or None; prune_flags: [n_quantizers]
    int32; cur_table: [n_batches, n_quantizers, 2] fp32 (written)."""
    lib = _hip.load()
    _hip.require_device(token_min, token_max, lengths, prune_flags, cur_table)
    per_q = 0
    if lengths is not None:
        if lengths.dtype != torch.int64 or not lengths.is_contiguous():
            raise ValueError("token_range_finalize_batched: lengths must be a contiguous int64 tensor")
        if tuple(lengths.shape) == (n_quantizers, n_batches, batch):
            per_q = 1
        elif tuple(lengths.shape) != (n_batches, batch):
            raise ValueError(f"token_range_finalize_batched: lengths has shape {tuple(lengths.shape)}, expected "
                             f"({n_batches}, {batch}) or ({n_quantizers}, {n_batches}, {batch})")
    if token_min.shape[-1] < batch * tokens or tuple(token_min.shape[:2]) != (n_quantizers, n_batches):
        raise ValueError("token_range_finalize_batched: token arrays do not cover [n_quantizers, n_batches, batch*tokens]")
    _hip.check(lib.osq_token_range_finalize_batched(_hip.ptr(token_min), _hip.ptr(token_max), token_min.stride(1),
                                                    int(n_quantizers), int(n_batches), int(batch), int(tokens),
                                                    _hip.ptr(lengths), per_q, _hip.ptr(prune_flags), float(percentile),
                                                    _hip.ptr(cur_table), _hip.ptr(_hip.workspace(token_min.device)),
                                                    _hip.stream_ptr(token_min.device)),
               "token_range_finalize_batched")


def observer_update(cur_min, cur_max, rule, cnt, min_val, max_val):
    lib = _hip.load()
    _hip.require_device(cur_min, cur_max, min_val, max_val)
    _check_f32(cur_min, cur_max, min_val, max_val)
    _hip.check(lib.osq_observer_update(_hip.ptr(cur_min), _hip.ptr(cur_max), cur_min.numel(), rule, int(cnt),
                                       _hip.ptr(min_val), _hip.ptr(max_val), _hip.stream_ptr(min_val.device)),
               "observer_update")


# ---------------------------------------------------------------------------------------
# MSEFast
# ---------------------------------------------------------------------------------------

SIDE = {"no": 0, "pos": 1, "neg": 2}


def batch_minmax(x, observation_mask=None, seq_pos=-1):
    """(min, max) of this tensor alone (padding excluded) as a 2-float device tensor; no state touched."""
    cur = torch.empty(2, dtype=torch.float32, device=x.device)
    if observation_mask is not None:
        tmin, tmax, batch, tokens, lengths = token_minmax(x, seq_pos, observation_mask)
        token_range_finalize(tmin, tmax, batch, tokens, lengths, False, 1.0, UPDATE_NONE, 0, None, None, 0, 1, False,
                             None, cur)
    else:
        observe_flat(x, UPDATE_NONE, 0, None, None, 0, 1, False, None, cur)
    return cur


def msefast_rows(w, ch_axis, quant_min, quant_max, symmetric, one_side, two_d):
    """One bounded-Brent search per channel (observer.py:496-517), all rows in one launch."""
    lib = _hip.load()
    _hip.require_device(w)
    _check_f32(w)
    ch_axis = ch_axis % w.dim()
    if ch_axis != 0:                                  # observer.py:11-21: channel axis first, rest flattened
        order = list(range(w.dim()))
        order[ch_axis], order[0] = 0, ch_axis
        w = w.permute(order)
    rows2d = w.reshape(w.shape[0], -1).contiguous()
    rows, cols = rows2d.shape
    bmin = torch.empty(rows, dtype=torch.float32, device=w.device)
    bmax = torch.empty(rows, dtype=torch.float32, device=w.device)
    nfev = torch.empty(rows, dtype=torch.int32, device=w.device)
    _hip.check(lib.osq_msefast_rows(_hip.ptr(rows2d), rows, cols, int(quant_min), int(quant_max), int(bool(symmetric)),
                                    SIDE[one_side], int(bool(two_d)), _hip.ptr(bmin), _hip.ptr(bmax), _hip.ptr(nfev),
                                    _hip.stream_ptr(w.device)), "msefast_rows")
    return bmin, bmax, nfev


class MseSearch:
    """One per-tensor MSEFast search between its begin and its commit (state on the device, what the launches need)."""
    __slots__ = ("state", "x", "view", "lengths", "args", "elems")


def msefast_tensor_begin(x, cur, observation_mask, seq_pos, quant_min, quant_max, symmetric, one_side, two_d,
                         float64_input=False):
    lib = _hip.load()
    dev = x.device
    r = MseSearch()
    r.state = torch.zeros(int(lib.osq_msefast_state_bytes()), dtype=torch.uint8, device=dev)
    _hip.check(lib.osq_msefast_tensor_begin(_hip.ptr(r.state), _hip.ptr(cur), int(quant_min), int(quant_max),
                                            int(bool(symmetric)), SIDE[one_side], int(bool(two_d)), int(bool(float64_input)),
                                            _hip.stream_ptr(dev)), "msefast_begin")
    if observation_mask is not None or not is_dense(x):
        lengths = observation_mask
        if lengths is not None and lengths.dtype != torch.int64:
            lengths = lengths.to(torch.int64)
        if observation_mask is None:
            x = x.contiguous()
            view = None
        else:
            view = token_view(x, seq_pos, lengths.numel())
    else:
        view, lengths = None, None
    if view is None and x.data_ptr() % 16:
        x = x.clone()          # the flat kernels read 16 bytes per lane: a misaligned buffer (a slice of a larger one) is copied once
    r.x, r.view, r.lengths, r.elems = x, view, lengths, x.numel()
    r.args = (int(quant_min), int(quant_max), int(bool(symmetric)))
    return r


def msefast_ordered_fits(r):
    """The float64 form of a search's sum is the finer one (half the lanes): it decides."""
    order = reference_sum_order("mse")
    return bool(order) and ordered_sum_fits(r.elems, order // 2)


def msefast_tensor_run(r, chunk=None, two_d=True):
    """The loss evaluations of one search: one persistent launch when the tensor fits the grid's registers, otherwise one
    launch per evaluation, enqueued in chunks (the converged flag is read back once per chunk; the reference syncs on
    every evaluation)."""
    lib = _hip.load()
    dev = r.x.device
    st = _hip.stream_ptr(dev)
    ws = _hip.workspace(dev)
    order = reference_sum_order("mse")
    if order and msefast_ordered_fits(r):
        return _msefast_tensor_run_ordered(r, chunk, two_d)
    # order-free sums.  (With the reference order set and a tensor beyond the ordered kernels' capacity the resident form
    # answers OSQ_ERR_UNSUPPORTED and the streaming evaluations below run: no library state is toggled for one call.)
    rc = lib.osq_msefast_tensor_search(_hip.ptr(r.state), _hip.ptr(r.x), r.x.numel(), None if r.view is None else ctypes.byref(r.view),
                                       _hip.ptr(r.lengths), _hip.ptr(ws), st)
    if rc not in (0, _hip.ERR_UNSUPPORTED):
        _hip.check(rc, "msefast_tensor_search")
    if rc == 0:
        _mark_persistent(dev)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    chunk = chunk or (64 if two_d else 32)
    launched = 0
    while rc != 0:
        if r.view is None:
            _hip.check(lib.osq_msefast_tensor_evals_flat(_hip.ptr(r.state), _hip.ptr(r.x), r.x.numel(), chunk, _hip.ptr(ws), st),
                       "msefast_evals_flat")
        else:
            _hip.check(lib.osq_msefast_tensor_evals_tokens(_hip.ptr(r.state), _hip.ptr(r.x), ctypes.byref(r.view),
                                                           _hip.ptr(r.lengths), chunk, _hip.ptr(ws), st), "msefast_evals_tokens")
        launched += chunk
        _hip.check(lib.osq_msefast_tensor_done(_hip.ptr(r.state), _hip.ptr(done), st), "msefast_done")
        if int(done.item()) or launched > 500 * 500:
            break


def _msefast_tensor_run_ordered(r, chunk, two_d):
    """The strict form of a per-tensor search ("mse_sum_order" 8 / 16): the site is laid out once the way the reference's
    remove_padding / flatten does (observer.py:72-84) and every loss evaluation is one launch that adds the squared errors
    in the order of torch.sum on a one-thread host (csrc/aten_order.h)."""
    lib = _hip.load()
    dev = r.x.device
    st = _hip.stream_ptr(dev)
    ws = _hip.workspace(dev)
    if r.view is None:
        flat, n, n_dev = r.x, r.x.numel(), None
    else:
        n = r.view.batch * r.view.tokens * r.view.feat_outer * r.view.feat_inner
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        n_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        _hip.check(lib.osq_gather_valid_tokens(_hip.ptr(r.x), ctypes.byref(r.view), _hip.ptr(r.lengths), _hip.ptr(flat),
                                               _hip.ptr(n_dev), st), "gather_valid_tokens")
    scratch, nbytes = _ordered_scratch(dev, n, 1)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    chunk = chunk or (64 if two_d else 32)
    launched = 0
    while True:
        _hip.check(lib.osq_msefast_tensor_evals_ordered(_hip.ptr(r.state), _hip.ptr(flat), n, _hip.ptr(n_dev), chunk,
                                                        _hip.ptr(scratch), nbytes, _hip.ptr(ws), st), "msefast_evals_ordered")
        launched += chunk
        _hip.check(lib.osq_msefast_tensor_done(_hip.ptr(r.state), _hip.ptr(done), st), "msefast_done")
        if int(done.item()) or launched > 500 * 500:
            break


ORDERED_GROUP_SITES = 128     # searches one table of osq_msefast_ordered_multi_* holds
# Bytes of site data (and of gathered copies of masked sites, alive together) one group of rounds may hold: a MEMORY bound,
# not a speed knob -- a group runs its own ~600 rounds, so splitting a forward's searches multiplies the launches, and a
# round's cost is dominated by its ~14 000 workgroups' start-up, not by where the bytes come from (measured: groups that fit
# the 256 MiB Infinity Cache are 1.4-2.6x SLOWER than one group, profiles/r05_mse_group_ab.txt).  Masked sites count with all
# their slots (the valid share is only known on the device).  BERT-base [32,128]: 0.57 GB, one group.  0 = no bound.
ORDERED_GROUP_BYTES = int(os.environ.get("OSQ_MSE_GROUP_MIB", "2048")) << 20


ORDERED_CHUNK = max(1, int(os.environ.get("OSQ_MSE_CHUNK", "64")))          # rounds enqueued between two looks at a group's all-done flag
ORDERED_STREAMS = max(1, int(os.environ.get("OSQ_MSE_STREAMS", "2")))     # concurrent groups of nested searches (see msefast_ordered_groups)


def msefast_ordered_groups(searches, two_d):
    """Partition the searches of a flush into the groups whose rounds run together.

    A round is one launch = one loss evaluation of every unfinished search of its group.  Rounds of DIFFERENT groups are
    independent, so the groups run on separate streams (msefast_tensor_run_ordered_groups): the short 1-D searches (~20
    rounds: attention probabilities, one-sided) no longer sit in the tables of the nested ones (hundreds of rounds), and
    one group's ramp and tail are filled by the other's workgroups.  Measured on BASELINE configs[3]: 1.06 s with one
    table per forward -> 0.98-1.03 s with the 1-D searches on their own stream and the nested ones dealt into 1-6 groups
    (the number of groups does not matter beyond that: a round is bound by its float64 arithmetic and its loads, not by
    launch overhead -- profiles/r05_mse_streams_ab.txt).  Hence: the nested (2-D) searches are dealt into ORDERED_STREAMS
    groups of about equal bytes (largest first onto the lightest group), the 1-D searches form one more group; every group
    is bounded by ORDERED_GROUP_SITES and ORDERED_GROUP_BYTES.  Grouping never changes a result: every search's
    evaluations are its own (tests/test_gpu_strict_order.py)."""
    groups = []
    nested = [r for r, nd in zip(searches, two_d) if nd]
    flat = [r for r, nd in zip(searches, two_d) if not nd]
    if nested:
        k = min(ORDERED_STREAMS, len(nested))
        lanes, load = [[] for _ in range(k)], [0] * k
        for r in sorted(nested, key=lambda r: -int(r.elems)):            # stable: equal sizes keep their forward order
            i = load.index(min(load))
            lanes[i].append(r)
            load[i] += int(r.elems)
        groups += [g for g in lanes if g]
    if flat:
        groups.append(flat)
    out = []
    for g in groups:                                                      # the table's and the memory bound's limits
        cur, used = [], 0
        for r in g:
            nbytes = 4 * int(r.elems)
            if cur and (len(cur) == ORDERED_GROUP_SITES or (ORDERED_GROUP_BYTES and used + nbytes > ORDERED_GROUP_BYTES)):
                out.append(cur)
                cur, used = [], 0
            cur.append(r)
            used += nbytes
        if cur:
            out.append(cur)
    return out


def _ordered_group_prepare(group, launch_stream=None):
    """Lay out the masked sites of a group (remove_padding order) and build the group's table.  Every ALLOCATION (gathered
    copies, scratch, table) and its zero fill happens on the CURRENT stream -- the caller's: the blocks come from, and go back
    to, the pool the rest of the calibration allocates from, not a side stream's private pool --; the gathers and the table's
    preparation are launched on `launch_stream` (default: the current one), which first waits for those fills."""
    lib = _hip.load()
    n_sites = len(group)
    assert 0 < n_sites <= ORDERED_GROUP_SITES
    dev = group[0].x.device
    flats, ns, n_devs = [], [], []
    for r in group:
        if r.view is None:
            flats.append(r.x)
            ns.append(r.x.numel())
            n_devs.append(None)
        else:
            n = r.view.batch * r.view.tokens * r.view.feat_outer * r.view.feat_inner
            flats.append(torch.empty(n, dtype=torch.float32, device=dev))
            ns.append(n)
            n_devs.append(torch.zeros(1, dtype=torch.int64, device=dev))
    sizes = [int(lib.osq_ordered_sum_scratch_bytes(int(n), 1)) for n in ns]
    offs, total = [], 0
    for b in sizes:
        offs.append(total)
        total += (b + 255) // 256 * 256
    scratch = torch.empty(total, dtype=torch.uint8, device=dev)
    table_bytes = int(lib.osq_msefast_ordered_multi_bytes(n_sites))
    table = torch.zeros(table_bytes, dtype=torch.uint8, device=dev)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = {"table": table, "n_sites": n_sites, "blocks": 0, "done": done, "keep": (flats, n_devs, scratch, group), "launched": 0}

    def launches():
        st = _hip.stream_ptr(dev)
        for r, flat, n_dev in zip(group, flats, n_devs):
            if r.view is not None:
                _hip.check(lib.osq_gather_valid_tokens(_hip.ptr(r.x), ctypes.byref(r.view), _hip.ptr(r.lengths), _hip.ptr(flat),
                                                       _hip.ptr(n_dev), st), "gather_valid_tokens")
        vp = ctypes.c_void_p
        states = (vp * n_sites)(*[_hip.ptr(r.state) for r in group])
        xs = (vp * n_sites)(*[_hip.ptr(f) for f in flats])
        n_arr = (ctypes.c_int64 * n_sites)(*ns)
        nd_arr = (vp * n_sites)(*[_hip.ptr(t) for t in n_devs])
        sc_arr = (vp * n_sites)(*[scratch.data_ptr() + o for o in offs])
        sb_arr = (ctypes.c_size_t * n_sites)(*sizes)
        blocks = ctypes.c_int(0)
        _hip.check(lib.osq_msefast_ordered_multi_prepare(_hip.ptr(table), table_bytes, states, xs, n_arr, nd_arr, sc_arr, sb_arr, n_sites,
                                                         ctypes.byref(blocks), st), "msefast_ordered_multi_prepare")
        ctx["blocks"] = blocks.value

    if launch_stream is None:
        launches()
    else:
        filled = torch.cuda.Event()
        filled.record(torch.cuda.current_stream(dev))
        launch_stream.wait_event(filled)
        with torch.cuda.stream(launch_stream):
            launches()
    return ctx


def _ordered_group_rounds(ctx, chunk):
    """Enqueue `chunk` rounds of a prepared group and its all-done check on the CURRENT stream (nothing waits)."""
    lib = _hip.load()
    dev = ctx["table"].device
    _hip.check(lib.osq_msefast_ordered_multi_evals(_hip.ptr(ctx["table"]), ctx["n_sites"], ctx["blocks"], chunk, _hip.ptr(ctx["done"]),
                                                   _hip.stream_ptr(dev)), "msefast_ordered_multi_evals")
    ctx["launched"] += chunk


def msefast_tensor_run_ordered_group(group, chunk=None):
    """The strict form of several searches (MseSearch records of one device, e.g. the MSEFast observers of one forward):
    rounds of ONE launch = one loss evaluation of every unfinished search, each sum in the order of torch.sum on a
    one-thread host (csrc/aten_order.h, osq_msefast_ordered_multi_*).  Same numbers as _msefast_tensor_run_ordered
    search by search (tests/test_gpu_strict_order.py)."""
    chunk = chunk or ORDERED_CHUNK
    ctx = _ordered_group_prepare(group)
    while True:
        _ordered_group_rounds(ctx, chunk)
        if int(ctx["done"].item()) or ctx["launched"] > 500 * 500:
            break
    return ctx["launched"]


_side_streams = {}


def msefast_tensor_run_ordered_groups(groups, chunk=None):
    """Several groups of strict searches CONCURRENTLY (see msefast_ordered_groups for why), at most ORDERED_STREAMS + 1 at a
    time: a group is prepared -- its gathered copies of masked sites, scratch and table allocated -- only when a stream is
    free, and dropped as soon as its searches have converged, so ORDERED_GROUP_BYTES bounds the memory of
    ORDERED_STREAMS + 1 groups, not of a whole flush (a flush that the byte cap splits into many groups used to hold all of
    them at once).  The side streams start behind everything the caller's stream has been given; the caller's stream
    continues behind all of them -- ALSO when a launch, a prepare or a read-back raises: the records' tensors must not be
    freed under rounds that are still queued.  The records of `groups` must stay referenced by the caller until then (they
    are: the flush commits them afterwards)."""
    chunk = chunk or ORDERED_CHUNK
    groups = [g for g in groups if g]
    if not groups:
        return 0
    if len(groups) == 1:
        return msefast_tensor_run_ordered_group(groups[0], chunk)
    dev = groups[0][0].x.device
    main = torch.cuda.current_stream(dev)
    start = torch.cuda.Event()
    start.record(main)
    lanes = min(len(groups), ORDERED_STREAMS + 1)
    streams = []
    for i in range(lanes):
        key = (_hip._device_index(dev), i)
        s = _side_streams.get(key)
        if s is None:
            s = _side_streams[key] = torch.cuda.Stream(device=dev)
        s.wait_event(start)
        streams.append(s)
    queue = list(groups)
    free = list(streams)
    pending, launched = [], 0
    try:
        while queue or pending:
            while queue and free:                    # a free stream takes the next group: allocations on the caller's stream
                s = free.pop(0)
                pending.append((s, _ordered_group_prepare(queue.pop(0), launch_stream=s)))
            for s, ctx in pending:                   # every unfinished group gets its next rounds before anybody waits
                with torch.cuda.stream(s):
                    _ordered_group_rounds(ctx, chunk)
            still = []
            for s, ctx in pending:
                with torch.cuda.stream(s):
                    finished = int(ctx["done"].item())      # synchronises with s: the group's rounds so far are complete
                if not finished and ctx["launched"] <= 500 * 500:
                    still.append((s, ctx))
                else:
                    launched += ctx["launched"]
                    ctx.clear()                      # the group's copies, scratch and table go back to the pool now
                    free.append(s)
            pending = still
    finally:
        for s in streams:                            # normal end or exception: nothing the caller frees is still in use
            main.wait_stream(s)
    return launched


def msefast_tensor_run_group(group):
    """Several searches (MseSearch records of one device) in ONE persistent launch; False = the group does not fit (nothing
    launched).  The caller sizes groups with msefast_resident_slots."""
    lib = _hip.load()
    n = len(group)
    dev = group[0].x.device
    states = (ctypes.c_void_p * n)(*[_hip.ptr(r.state) for r in group])
    xs = (ctypes.c_void_p * n)(*[_hip.ptr(r.x) for r in group])
    ns = (ctypes.c_int64 * n)(*[r.x.numel() for r in group])
    views = (_hip.TokenView * n)()
    for i, r in enumerate(group):
        if r.view is not None:
            views[i] = r.view
    lens = (ctypes.c_void_p * n)(*[_hip.ptr(r.lengths) for r in group])
    rc = lib.osq_msefast_tensor_search_multi(states, xs, ns, views, lens, n, _hip.ptr(_hip.workspace(dev)), _hip.stream_ptr(dev))
    if rc == _hip.ERR_UNSUPPORTED:
        return False
    _hip.check(rc, "msefast_tensor_search_multi")
    _mark_persistent(dev)
    return True


def msefast_resident_slots(elems):
    return int(_hip.load().osq_msefast_resident_slots(int(elems)))


_resident_limits = None


def msefast_resident_limits():
    """(float4 slots per lane, searches) one osq_msefast_tensor_search_multi launch can hold."""
    global _resident_limits
    if _resident_limits is None:
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _hip.check(_hip.load().osq_msefast_resident_limits(ctypes.byref(a), ctypes.byref(b)), "msefast_resident_limits")
        _resident_limits = (a.value, b.value)
    return _resident_limits


def msefast_tensor_commit(r, rule, cnt, min_val, max_val, sink=None, ref_float64=None):
    """ref_float64: int32[2] on the device (the reference's dtype of min_val / max_val so far, see osq_hip.h) or None."""
    lib = _hip.load()
    dev = r.x.device
    quant_min, quant_max, symmetric = r.args
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    nfev = torch.empty(1, dtype=torch.int32, device=dev)
    _hip.check(lib.osq_msefast_tensor_commit(_hip.ptr(r.state), rule, int(cnt), _hip.ptr(min_val), _hip.ptr(max_val),
                                             quant_min, quant_max, symmetric, s_ptr, z_ptr, z_type,
                                             _hip.ptr(nfev), _hip.ptr(ref_float64), _hip.stream_ptr(dev)), "msefast_commit")
    return nfev


def msefast_tensor_stats(r):
    """int32[4] on the device: evaluations so far (nfev), pairs the search's loss memo holds, evaluations it answered, converged
    flag (osq_msefast_tensor_stats; the memo: include/osq_hip.h)."""
    out = torch.empty(4, dtype=torch.int32, device=r.x.device)
    _hip.check(_hip.load().osq_msefast_tensor_stats(_hip.ptr(r.state), _hip.ptr(out), _hip.stream_ptr(r.x.device)), "msefast_tensor_stats")
    return out


def msefast_tensor(x, cur, observation_mask, seq_pos, quant_min, quant_max, symmetric, one_side, two_d,
                   rule, cnt, min_val, max_val, sink=None, chunk=None, float64_input=False):
    """Per-tensor search (observer.py:497-499), begin -> loss evaluations -> commit.  min_val/max_val: float64."""
    r = msefast_tensor_begin(x, cur, observation_mask, seq_pos, quant_min, quant_max, symmetric, one_side, two_d, float64_input)
    msefast_tensor_run(r, chunk, two_d)
    return msefast_tensor_commit(r, rule, cnt, min_val, max_val, sink)


# ---------------------------------------------------------------------------------------
# remaining observers: LSQPlus (moments), AvgQuantile (histogram), MSE (grid)
# ---------------------------------------------------------------------------------------

def _source(x, observation_mask, seq_pos):
    """(x, n, view_or_None, lengths) for kernels that stream either a flat tensor or valid tokens."""
    if observation_mask is None:
        if not is_dense(x):
            x = x.contiguous()
        return x, x.numel(), None, None
    lengths = observation_mask if observation_mask.dtype == torch.int64 else observation_mask.to(torch.int64)
    return x, x.numel(), token_view(x, seq_pos, lengths.numel()), lengths


def observe_moments(x, ch_axis, min_val, max_val, quant_min, quant_max, symmetric, sink=None):
    lib = _hip.load()
    _hip.require_device(x, min_val, max_val)
    _check_f32(x, min_val, max_val)
    x = x.contiguous()
    outer, channels, inner = (1, 1, x.numel()) if ch_axis == -1 else _channel_split(x, ch_axis)
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    _hip.check(lib.osq_observe_moments(_hip.ptr(x), outer, channels, inner, _hip.ptr(min_val), _hip.ptr(max_val),
                                       int(quant_min), int(quant_max), int(bool(symmetric)), s_ptr, z_ptr, z_type,
                                       _hip.ptr(_hip.workspace(x.device)), _hip.stream_ptr(x.device)), "observe_moments")


def observe_quantile(x, observation_mask, seq_pos, cur, threshold, hist_scratch, rule, cnt, min_val, max_val,
                     quant_min, quant_max, symmetric, sink=None):
    lib = _hip.load()
    _hip.require_device(x, cur, hist_scratch, min_val, max_val)
    _check_f32(x, min_val, max_val)
    x, n, view, lengths = _source(x, observation_mask, seq_pos)
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    _hip.check(lib.osq_observe_quantile(_hip.ptr(x), n, ctypes.byref(view) if view is not None else None, _hip.ptr(lengths),
                                        _hip.ptr(cur), float(threshold), _hip.ptr(hist_scratch), rule, int(cnt),
                                        _hip.ptr(min_val), _hip.ptr(max_val), int(quant_min), int(quant_max),
                                        int(bool(symmetric)), s_ptr, z_ptr, z_type, _hip.stream_ptr(x.device)),
               "observe_quantile")


def mse_grid_tensor(x, observation_mask, seq_pos, cur, quant_min, quant_max, symmetric, one_side, two_d, rule, cnt,
                    min_val, max_val, sink=None):
    lib = _hip.load()
    _hip.require_device(x, cur, min_val, max_val)
    _check_f32(x, min_val, max_val)
    x, n, view, lengths = _source(x, observation_mask, seq_pos)
    n_cand = int(lib.osq_mse_grid_candidates(int(quant_min), int(quant_max), int(bool(two_d))))
    nbytes = int(lib.osq_mse_grid_scratch_bytes(int(quant_min), int(quant_max), int(bool(two_d))))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)      # losses + every workgroup's partial sums: the grid is one launch
    losses = scratch[:4 * n_cand].view(torch.float32)
    s_ptr, z_ptr, z_type = (sink or QParamSink()).args()
    _hip.check(lib.osq_mse_grid_tensor(_hip.ptr(x), n, ctypes.byref(view) if view is not None else None, _hip.ptr(lengths),
                                       _hip.ptr(cur), int(quant_min), int(quant_max), int(bool(symmetric)), SIDE[one_side],
                                       int(bool(two_d)), _hip.ptr(scratch), nbytes, rule, int(cnt), _hip.ptr(min_val),
                                       _hip.ptr(max_val), s_ptr, z_ptr, z_type, _hip.ptr(_hip.workspace(x.device)),
                                       _hip.stream_ptr(x.device)), "mse_grid_tensor")
    return losses


def mse_grid_rows(w, ch_axis, quant_min, quant_max, symmetric, one_side, two_d):
    lib = _hip.load()
    _hip.require_device(w)
    _check_f32(w)
    ch_axis = ch_axis % w.dim()
    if ch_axis != 0:
        order = list(range(w.dim()))
        order[ch_axis], order[0] = 0, ch_axis
        w = w.permute(order)
    rows2d = w.reshape(w.shape[0], -1).contiguous()
    rows, cols = rows2d.shape
    bmin = torch.empty(rows, dtype=torch.float32, device=w.device)
    bmax = torch.empty(rows, dtype=torch.float32, device=w.device)
    _hip.check(lib.osq_mse_grid_rows(_hip.ptr(rows2d), rows, cols, int(quant_min), int(quant_max), int(bool(symmetric)),
                                     SIDE[one_side], int(bool(two_d)), _hip.ptr(bmin), _hip.ptr(bmax),
                                     _hip.stream_ptr(w.device)), "mse_grid_rows")
    return bmin, bmax


# ---------------------------------------------------------------------------------------
# gamma migration
# ---------------------------------------------------------------------------------------

# moved by every kernel of this package that writes a WEIGHT through its raw pointer (torch's ``_version`` does not see
# those writes): cached fake-quantised weights (quantization/weight_cache.py) are keyed on it
weight_epoch = 0


def gamma_fold_(weight, gamma):
    """In place W[:, j] *= gamma[j] (gamma_migration.py:70-71)."""
    global weight_epoch
    weight_epoch += 1
    lib = _hip.load()
    _hip.require_device(weight, gamma)
    _check_f32(weight, gamma)
    if not weight.is_contiguous():
        raise ValueError("gamma_fold_: weight must be contiguous")
    cols = weight.shape[-1]
    if gamma.numel() != cols:
        raise ValueError("gamma_fold_: gamma must have one entry per input feature")
    _hip.check(lib.osq_gamma_fold(_hip.ptr(weight), _hip.ptr(gamma.contiguous()), weight.numel() // cols, cols,
                                  _hip.stream_ptr(weight.device)), "gamma_fold")
    return weight


def gamma_split_bias(beta, gamma):
    lib = _hip.load()
    _hip.require_device(beta, gamma)
    _check_f32(beta, gamma)
    out = torch.empty_like(beta, memory_format=torch.contiguous_format)
    _hip.check(lib.osq_gamma_split_bias(_hip.ptr(beta.contiguous()), _hip.ptr(gamma.contiguous()), _hip.ptr(out),
                                        beta.numel(), _hip.stream_ptr(beta.device)), "gamma_split_bias")
    return out


def gamma_residual(inp, hidden, gamma=None):
    """input * gamma + hidden (util_layernorm.py:49-52), inference path (no autograd)."""
    lib = _hip.load()
    _hip.require_device(inp, hidden, gamma)
    _check_f32(inp, hidden, gamma)
    a, h = inp.contiguous(), hidden.contiguous()
    cols = a.shape[-1]
    out = torch.empty_like(a)
    _hip.check(lib.osq_gamma_residual(_hip.ptr(a), _hip.ptr(h), _hip.ptr(gamma), _hip.ptr(out), a.numel() // cols, cols,
                                      _hip.stream_ptr(a.device)), "gamma_residual")
    return out


# ---------------------------------------------------------------------------------------
# residual + LayerNorm + fake-quant in one pass
# ---------------------------------------------------------------------------------------

def layernorm_fusable(x, *operands):
    """Layout rules of osq_residual_layernorm_fake_quant."""
    cols = x.shape[-1] if x.dim() else 0
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() and cols % 4 == 0 and cols <= 4096
            and x.data_ptr() % 16 == 0):
        return False
    for t in operands:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0):
            return False
    return True


def residual_layernorm_fake_quant(x, hidden, gamma, weight, bias, eps, quant=None):
    """y = fake_quant(layer_norm(x*gamma + hidden) * weight + bias) in ONE launch; every operand but x may be None.
    quant: None (no fake-quant) or (scale, zero_point, quant_min, quant_max, mode, grad_factor)."""
    lib = _hip.load()
    _hip.require_device(x, hidden, gamma, weight, bias)
    cols = x.shape[-1]
    rows = x.numel() // cols
    if hidden is not None and hidden.shape != x.shape:
        raise ValueError("residual_layernorm_fake_quant: hidden must have x's shape")
    for name, t in (("gamma", gamma), ("weight", weight), ("bias", bias)):
        if t is not None and t.numel() != cols:
            raise ValueError(f"residual_layernorm_fake_quant: {name} must have {cols} entries")
    y = torch.empty_like(x)
    if quant is None:
        s_ptr, z_ptr, z_type, mode, gf, qmin, qmax = None, None, ZP_INT32, PARAM_FIXED, 1.0, 0, 1
    else:
        scale, zero_point, qmin, qmax, mode, gf = quant
        _hip.require_device(scale, zero_point)
        s_ptr, z_ptr, z_type = scale.data_ptr(), zero_point.data_ptr(), _zp_type(zero_point)
    _hip.check(lib.osq_residual_layernorm_fake_quant(x.data_ptr(), _hip.ptr(hidden), _hip.ptr(gamma), _hip.ptr(weight),
                                                     _hip.ptr(bias), float(eps), y.data_ptr(), rows, cols, s_ptr, z_ptr,
                                                     z_type, int(mode), float(gf), int(qmin), int(qmax),
                                                     _hip.raw_stream(x.device)), "residual_layernorm_fake_quant")
    return y


# ---------------------------------------------------------------------------------------
# GELU + fake-quant in one pass (the intermediate-activation site)
# ---------------------------------------------------------------------------------------

def is_exact_gelu(fn):
    """True for the callables HF models use for hidden_act='gelu' (erf form)."""
    import torch.nn.functional as F
    if fn is F.gelu:
        return True
    if isinstance(fn, torch.nn.GELU):
        return getattr(fn, "approximate", "none") == "none"
    inner = getattr(fn, "act", None)            # transformers.activations.GELUActivation(use_gelu_python=False)
    return type(fn).__name__ == "GELUActivation" and inner is F.gelu


def fake_quant_headsplit_multi(xs, params, heads):
    """The query / key / value sites of one attention block as ONE launch (osq_fake_quant_headsplit_multi).
    xs: 1..4 contiguous [B, T, heads * d] fp32 tensors of one shape; params[i] = (scale, zero_point, quant_min, quant_max,
    mode, grad_factor) of site i.  Returns the dense [B, heads, T, d] fake-quantised head-split tensors -- the bits of
    fake_quant_per_tensor on each ``x.view(B, T, heads, d).permute(0, 2, 1, 3)`` -- or None when the kernel does not take the
    geometry (nothing was launched)."""
    lib = _hip.load()
    n = len(xs)
    b, t, width = xs[0].shape
    d = width // heads
    table = (_hip.HeadSplitSite * n)()
    ys = []
    for i, (x, (scale, zero_point, quant_min, quant_max, mode, grad_factor)) in enumerate(zip(xs, params)):
        y = torch.empty((b, heads, t, d), dtype=x.dtype, device=x.device)
        ys.append(y)
        e = table[i]
        e.x, e.y, e.scale, e.zero_point = x.data_ptr(), y.data_ptr(), scale.data_ptr(), zero_point.data_ptr()
        e.zp_type, e.mode, e.grad_factor, e.quant_min, e.quant_max = _zp_type(zero_point), int(mode), float(grad_factor), int(quant_min), int(quant_max)
    rc = lib.osq_fake_quant_headsplit_multi(table, n, b, t, heads, d, _hip.stream_ptr(xs[0].device))
    if rc == _hip.ERR_UNSUPPORTED:
        return None
    _hip.check(rc, "fake_quant_headsplit_multi")
    return ys


def gelu_fake_quant_per_tensor(x, scale, zero_point, quant_min, quant_max, mode=PARAM_FIXED, grad_factor=1.0):
    """fake_quant(F.gelu(x)) in ONE launch; x dense fp32 on the device."""
    lib = _hip.load()
    _hip.require_device(x, scale, zero_point)
    _check_f32(x, scale)
    if not x.is_contiguous():
        x = x.contiguous()
    if x.data_ptr() % 16:          # a slice of a larger buffer: the two launches this one replaces (same numbers)
        return fake_quant_per_tensor(torch.nn.functional.gelu(x), scale, zero_point, quant_min, quant_max, mode, grad_factor)
    y = torch.empty_like(x)
    _hip.check(lib.osq_gelu_fake_quant_per_tensor(x.data_ptr(), y.data_ptr(), x.numel(), scale.data_ptr(),
                                                  zero_point.data_ptr(), _zp_type(zero_point), int(mode),
                                                  float(grad_factor), int(quant_min), int(quant_max),
                                                  _hip.raw_stream(x.device)), "gelu_fake_quant_per_tensor")
    return y
