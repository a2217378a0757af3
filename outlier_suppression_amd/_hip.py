"""ctypes binding of ``libosq_hip.so`` (the C ABI declared in ``include/osq_hip.h``).

There is no CPU fallback anywhere in this package: if the library is missing, or a
tensor is not on a HIP device, the call raises.  PyTorch is used only for device
memory, streams and ``torch.distributed``.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSQ_HIP_LIBRARY: another build of the same ABI (A/B runs of a kernel variant, the -DOSQ_MSE_DBG probe build); the ABI version
# check applies to it as to the in-tree library
LIB_PATH = os.environ.get("OSQ_HIP_LIBRARY") or os.path.join(_HERE, "libosq_hip.so")

# zero-point storage / parameter mode / update rule (mirrors include/osq_hip.h)
ZP_INT32, ZP_FLOAT32 = 0, 1
PARAM_FIXED, PARAM_LSQ, PARAM_LSQPLUS = 0, 1, 2
PARAM_MODE_MASK, PARAM_SANITIZE, PARAM_NO_PERSISTENT = 3, 16, 32
TIME_FAKE_QUANT, TIME_LSQ_BACKWARD, TIME_OBSERVE_FLAT, TIME_TOKEN_MINMAX, TIME_TOKEN_SELECT = 1, 2, 3, 4, 5
TIME_LAYERNORM, TIME_FUSED_STEP = 6, 7
TIME_FAKE_QUANT_STRIDED, TIME_FAKE_QUANT_CHANNEL, TIME_OBSERVE_CHANNELS, TIME_TOKEN_MINMAX_MULTI, TIME_MSEFAST_ROWS = 8, 9, 10, 11, 12
TIME_OBSERVE_TOKENS = 13
UPDATE_NONE, UPDATE_RUNNING, UPDATE_AVERAGE = 0, 1, 2
ERR_UNSUPPORTED = -3          # OSQ_ERR_UNSUPPORTED: nothing was launched, the caller takes its other path
ABI_VERSION = 7               # OSQ_ABI_VERSION of include/osq_hip.h this file was written against

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float
_D = ctypes.c_double


class TokenView(ctypes.Structure):
    """``osq_token_view``: logical [batch, tokens, feat_outer, feat_inner] with element strides."""
    _fields_ = [(n, _L) for n in ("batch", "tokens", "feat_outer", "feat_inner",
                                  "stride_batch", "stride_token", "stride_outer", "stride_inner")]


class WeightDesc(ctypes.Structure):
    """``osq_weight_desc``: one entry of the table of osq_fake_quant_weights_multi."""
    _fields_ = [("x", _P), ("y", _P), ("scale", _P), ("zero_point", _P), ("rows", _L), ("channels", _L), ("inner", _L),
                ("zp_type", ctypes.c_int32), ("mode", ctypes.c_int32), ("grad_factor", _F), ("quant_min", ctypes.c_int32),
                ("quant_max", ctypes.c_int32), ("pad", ctypes.c_int32)]


class HeadSplitSite(ctypes.Structure):
    """``osq_headsplit_site``: one entry of the table of osq_fake_quant_headsplit_multi."""
    _fields_ = [("x", _P), ("y", _P), ("scale", _P), ("zero_point", _P), ("zp_type", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("grad_factor", _F), ("quant_min", ctypes.c_int32), ("quant_max", ctypes.c_int32), ("pad", ctypes.c_int32)]


class SiteDesc(ctypes.Structure):
    """``osq_site_desc``: one entry of the table of osq_token_minmax_multi."""
    _fields_ = [("x", _P), ("lengths", _P), ("token_min", _P), ("token_max", _P), ("view", TokenView),
                ("vec", ctypes.c_int32), ("pad", ctypes.c_int32)]


# name -> (restype, argtypes); one entry per symbol declared in include/osq_hip.h
SIGNATURES = {
    "osq_last_error": (ctypes.c_char_p, []),
    "osq_abi_version": (_I, []),
    "osq_build_flags": (_I, []),
    "osq_workspace_bytes": (ctypes.c_size_t, []),
    "osq_set_tuning": (_I, [ctypes.c_char_p, _I]),
    "osq_timing_events_create": (_I, [ctypes.POINTER(_P), ctypes.POINTER(_P)]),
    "osq_timing_events_destroy": (_I, [_P, _P]),
    "osq_time_next_launch": (_I, [_I, _P, _P]),
    "osq_timing_elapsed_us": (_I, [_P, _P, ctypes.POINTER(_F)]),
    "osq_fake_quant_per_tensor": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _F, _I, _I, _P]),
    "osq_gelu_fake_quant_per_tensor": (_I, [_P, _P, _L, _P, _P, _I, _I, _F, _I, _I, _P]),
    "osq_fake_quant_per_tensor_strided": (_I, [_P, _P, _P, ctypes.POINTER(_L), ctypes.POINTER(_L), ctypes.POINTER(_L),
                                               _P, _P, _I, _I, _F, _I, _I, _P]),
    "osq_fake_quant_headsplit_multi": (_I, [ctypes.POINTER(HeadSplitSite), _I, _L, _L, _L, _L, _P]),
    "osq_fake_quant_per_channel": (_I, [_P, _P, _P, _L, _L, _L, _P, _P, _I, _I, _F, _I, _I, _P]),
    "osq_fake_quant_weights_multi": (_I, [_P, _P, _I, _L, _P]),
    "osq_lsq_backward_per_tensor": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _F, _I, _I, _P, _P, _P, _P]),
    "osq_ordered_sum_scratch_bytes": (ctypes.c_size_t, [_L, _I]),
    "osq_lsq_backward_per_tensor_ordered": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _F, _I, _I, _P, _P, _I, _P, ctypes.c_size_t, _P, _P]),
    "osq_lsq_backward_per_channel": (_I, [_P, _P, _P, _L, _L, _L, _P, _P, _I, _I, _F, _I, _I, _P, _P, _I, _P]),
    "osq_lsq_sanitize": (_I, [_P, _P, _L, _F, _I, _I, _P]),
    "osq_calculate_qparams": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _I, _P]),
    "osq_observe_flat": (_I, [_P, _L, _I, _L, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "osq_observe_channels": (_I, [_P, _L, _L, _L, _I, _L, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "osq_token_minmax": (_I, [_P, ctypes.POINTER(TokenView), _P, _P, _P, _P]),
    "osq_token_minmax_multi": (_I, [_P, _P, _I, _L, _P]),
    "osq_token_range_finalize": (_I, [_P, _P, _L, _L, _P, _I, _D, _I, _L, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "osq_observe_tokens": (_I, [_P, ctypes.POINTER(TokenView), _P, _P, _P, _I, _D, _I, _L, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "osq_observe_tokens_fake_quant": (_I, [_P, ctypes.POINTER(TokenView), _P, _P, _P, _I, _D, _I, _L, _P, _P, _P, _I, _I, _I, _P, _P, _I,
                                           _P, _L, _I, _F, _P, _P, _P]),
    "osq_fused_step_status": (_I, [_P, ctypes.POINTER(_I), _P]),
    "osq_persistent_status": (_I, [_P, ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _P]),
    "osq_set_wide_min_slots": (_I, [_L]),
    "osq_token_range_finalize_batched": (_I, [_P, _P, _L, _I, _I, _L, _L, _P, _I, _P, _D, _P, _P, _P]),
    "osq_observer_update": (_I, [_P, _P, _L, _I, _L, _P, _P, _P]),
    "osq_replay_statistics": (_I, [_P, _I, _I, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "osq_msefast_rows": (_I, [_P, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "osq_msefast_state_bytes": (ctypes.c_size_t, []),
    "osq_msefast_tensor_begin": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "osq_msefast_tensor_evals_flat": (_I, [_P, _P, _L, _I, _P, _P]),
    "osq_msefast_tensor_evals_tokens": (_I, [_P, _P, ctypes.POINTER(TokenView), _P, _I, _P, _P]),
    "osq_gather_valid_tokens": (_I, [_P, ctypes.POINTER(TokenView), _P, _P, _P, _P]),
    "osq_msefast_tensor_evals_ordered": (_I, [_P, _P, _L, _P, _I, _P, ctypes.c_size_t, _P, _P]),
    "osq_msefast_ordered_multi_bytes": (ctypes.c_size_t, [_I]),
    "osq_msefast_ordered_multi_prepare": (_I, [_P, ctypes.c_size_t, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_L),
                                               ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_size_t), _I,
                                               ctypes.POINTER(_I), _P]),
    "osq_msefast_ordered_multi_evals": (_I, [_P, _I, _I, _I, _P, _P]),
    "osq_msefast_tensor_search": (_I, [_P, _P, _L, ctypes.POINTER(TokenView), _P, _P, _P]),
    "osq_msefast_resident_slots": (_I, [_L]),
    "osq_msefast_resident_limits": (_I, [ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "osq_msefast_tensor_search_multi": (_I, [ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_L), ctypes.POINTER(TokenView),
                                             ctypes.POINTER(_P), _I, _P, _P]),
    "osq_msefast_tensor_done": (_I, [_P, _P, _P]),
    "osq_msefast_tensor_stats": (_I, [_P, _P, _P]),
    "osq_calculate_qparams_f64": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _I, _P]),
    "osq_msefast_tensor_commit": (_I, [_P, _I, _L, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "osq_observe_moments": (_I, [_P, _L, _L, _L, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "osq_observe_quantile": (_I, [_P, _L, ctypes.POINTER(TokenView), _P, _P, _D, _P, _I, _L, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "osq_mse_grid_candidates": (_I, [_I, _I, _I]),
    "osq_mse_grid_scratch_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "osq_mse_grid_tensor": (_I, [_P, _L, ctypes.POINTER(TokenView), _P, _P, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _I, _L, _P, _P, _P, _P, _I, _P, _P]),
    "osq_selftest_division": (_I, [_P, _P, _L, _P, _P]),
    "osq_mse_grid_rows": (_I, [_P, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P]),
    "osq_gamma_fold": (_I, [_P, _P, _L, _L, _P]),
    "osq_gamma_split_bias": (_I, [_P, _P, _P, _L, _P]),
    "osq_gamma_residual": (_I, [_P, _P, _P, _P, _L, _L, _P]),
    "osq_residual_layernorm_fake_quant": (_I, [_P, _P, _P, _P, _P, _D, _P, _L, _L, _P, _P, _I, _I, _F, _I, _I, _P]),
}

_lib = None
_lock = threading.Lock()


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {os.path.join(_HERE, 'csrc')}`.  outlier_suppression_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)   # torch is already imported: its libamdhip64.so.7 satisfies the dependency
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        built = int(lib.osq_abi_version())
        if built != ABI_VERSION:
            raise HipLibraryMissing(
                f"{LIB_PATH} was built against ABI {built}, this package needs ABI {ABI_VERSION}: rebuild it "
                f"(`make -C {os.path.join(_HERE, 'csrc')}`); signatures or the workspace layout changed.")
        from . import _apply_environment
        _apply_environment(lib)   # OSQ_STRICT / OSQ_FAST -- before the library is published: no thread launches in another tier
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().osq_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"osq_hip {what} failed (status {rc}): {msg}")


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("outlier_suppression_amd: tensors must live on a HIP device "
                               f"(got device={t.device}); there is no CPU path.")


# The three helpers below run on every launch: they return plain ints (ctypes converts an int to a
# void* argument by itself) and use the raw-stream query, which costs a fraction of building a
# torch.cuda.Stream object.  A quantizer call is three launches of 10-35 us: host time per launch matters.
def ptr(t):
    return None if t is None else t.data_ptr()


def _device_index(device):
    idx = None if device is None else device.index
    return torch.cuda.current_device() if idx is None else idx


def raw_stream(device=None):
    """hipStream_t of torch's current stream on ``device`` as an int."""
    return torch._C._cuda_getCurrentRawStream(_device_index(device))


def stream_ptr(device=None):
    return raw_stream(device)


_workspaces = {}


def workspace(device):
    """Zero-initialised scratch for (device, current stream); kernels leave their counters zeroed."""
    key = (_device_index(device), raw_stream(device))
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = int(load().osq_workspace_bytes())
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws
