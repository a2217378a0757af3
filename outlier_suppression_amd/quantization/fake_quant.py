"""FakeQuantize modules with the reference's names, flags, buffers and state-dict keys.

Reference: quant_transformer/quantization/fake_quant.py.  ``forward(X, observation_mask=None,
seq_pos=-1)`` does, like there: [observer enabled] observe X and refresh scale / zero_point;
[fake-quant enabled] return the fake-quantised X, else X itself.  Here the observe step is
one or two HIP launches that also write scale / zero_point (no ``.item()``, no host sync),
and the quantise step is one launch, differentiable through one more.
"""
import torch
import torch.nn as nn

from .. import ops
from ..ops import PARAM_FIXED, PARAM_LSQ, PARAM_LSQPLUS, PARAM_SANITIZE, QParamSink
from .observer import MinMaxObserver


class QuantizeBase(nn.Module):
    """fake_quant.py:15-97."""

    param_mode = PARAM_FIXED
    # False (on the class, a module, or per call: forward(..., persistent=False)): the calibrate-and-quantize call never takes
    # its one-launch persistent form, which needs every CU of the device for itself ("persistent launches") --
    # for callers that run several streams or tenants on one GPU.  Results are the same either way.
    persistent = True

    def __init__(self, observer=MinMaxObserver, bit=8, symmetric=False, ch_axis=-1):
        super().__init__()
        self.observer = observer(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.bit, self.symmetric, self.ch_axis = bit, symmetric, ch_axis
        self.observer_enabled = 0
        self.fake_quant_enabled = 0
        self.quant_min, self.quant_max = self.observer.quant_min, self.observer.quant_max

    def set_name(self, name):
        self.name = name

    @torch.jit.export
    def calculate_qparams(self):
        ops.check_persistent("calculate_qparams")
        return self.observer.calculate_qparams(self.observer.min_val, self.observer.max_val)

    @torch.jit.export
    def enable_observer(self):
        self.observer_enabled = 1

    @torch.jit.export
    def disable_observer(self):
        self.observer_enabled = 0

    @torch.jit.export
    def enable_fake_quant(self):
        self.fake_quant_enabled = 1

    @torch.jit.export
    def disable_fake_quant(self):
        self.fake_quant_enabled = 0

    @torch.jit.export
    def extra_repr(self):
        return (f"fake_quant_enabled={self.fake_quant_enabled}, observer_enabled={self.observer_enabled}, "
                f"symmetric={self.symmetric}, bit={self.bit}, ch_axis={self.ch_axis}, "
                f"quant_min={self.quant_min}, quant_max={self.quant_max}")

    # ---- (scale, zero_point) storage ---------------------------------------------------
    def _qparam_storage(self, device, channels):
        """scale / zero_point tensors on ``device`` with ``channels`` entries (1 for per-tensor),
        resized like fake_quant.py:112-114 / 184-186 when the observer turns out per-channel."""
        s, z = self.scale, self.zero_point
        if s.device == device and z.device == device and s.numel() == channels and z.numel() == channels:
            return s.data, z.data               # the steady state: nothing to move or resize
        for name in ("scale", "zero_point"):
            t = getattr(self, name)
            data = t.data if isinstance(t, nn.Parameter) else t
            if data.device != device or data.numel() != channels:
                fresh = (torch.ones if name == "scale" else torch.zeros)(channels, dtype=data.dtype, device=device)
                if isinstance(t, nn.Parameter):
                    t.data = fresh
                else:
                    setattr(self, name, fresh)
        s, z = self.scale, self.zero_point
        return (s.data if isinstance(s, nn.Parameter) else s), (z.data if isinstance(z, nn.Parameter) else z)

    def _touch_qparams(self):
        """A kernel is about to write scale / zero_point through raw pointers (torch's ``_version`` does not move):
        anything derived from them -- the cached fake-quantised weight -- is stale."""
        object.__setattr__(self, "_qparam_epoch", self.__dict__.get("_qparam_epoch", 0) + 1)

    def _observe(self, X, observation_mask, seq_pos):
        self._touch_qparams()
        channels = 1 if self.ch_axis == -1 else X.shape[self.ch_axis]
        scale, zero_point = self._qparam_storage(X.device, channels)
        obs = self.observer
        if hasattr(obs, "observe_into"):
            if X.numel():
                # nothing reads scale / zero_point before the next launch on this stream unless this very call quantises:
                # only then may the observation be recorded and reduced later (quantization/deferred.py)
                object.__setattr__(obs, "_defer_ok", self.fake_quant_enabled != 1)
                obs.observe_into(X.detach(), observation_mask, seq_pos, QParamSink(scale, zero_point))
        else:   # foreign observer object: the reference's three steps, still on the device
            obs(X.detach(), observation_mask=observation_mask, seq_pos=seq_pos)
            ops.calculate_qparams(obs.min_val, obs.max_val, self.quant_min, self.quant_max, self.symmetric,
                                  scale_out=scale, zero_point_out=zero_point)

    def _observe_and_quantize(self, X, observation_mask, seq_pos, persistent=None):
        """Both flags on, masked per-tensor activation, no gradient wanted: the whole call (per-token extrema, range
        selection, running statistic, qparams, fake-quant) is ONE call of the binding.  Returns None when the
        general two-step path has to run.  persistent (None: the module's `persistent` attribute, default True): False
        keeps this call off the one-launch persistent form (ops.observe_tokens_fake_quant)."""
        obs = self.observer
        if (observation_mask is None or self.ch_axis != -1 or not X.is_cuda or X.dtype != torch.float32
                or not X.is_contiguous() or X.numel() == 0 or X.dim() not in (3, 4)
                or getattr(obs, "_capture", True) is not None or obs._token_cache is not None or obs.ch_axis != -1):
            return None
        if torch.is_grad_enabled() and (X.requires_grad or self.scale.requires_grad):
            return None
        prune = obs.token_path_prune()
        if prune is None:
            return None
        self._touch_qparams()
        scale, zero_point = self._qparam_storage(X.device, 1)
        obs._home(X.device)
        gf = self._grad_factor(X) if self.param_mode != PARAM_FIXED else 1.0
        y, batch, tokens, lengths = ops.observe_tokens_fake_quant(
            X, seq_pos, observation_mask, prune, getattr(obs, "percentile", 1.0), obs.update_rule, obs._counter(),
            obs.min_val, obs.max_val, self.quant_min, self.quant_max, self.symmetric, scale, zero_point, self.param_mode, gf,
            obs.__dict__.get("_record"), persistent=self.persistent if persistent is None else persistent)
        object.__setattr__(obs, "_last_site", ("tokens", batch, tokens, lengths))
        obs._bump()
        return y

    def _grad_factor(self, X):
        """fake_quant.py:157-166 / 195-204."""
        if not getattr(self, "use_grad_scaling", False):
            return 1.0
        # numel_multiplier: ranks of a data-parallel learn-scale step each see 1/W of the tensor
        numel = X.numel() * getattr(self, "numel_multiplier", 1)
        if self.ch_axis != -1:
            return 1.0 / (numel / X.shape[self.ch_axis] * self.quant_max) ** 0.5
        return 1.0 / (numel * self.quant_max) ** 0.5

    def _quantize(self, X, flags=0):
        return ops.fake_quant(X, self.scale, self.zero_point, self.ch_axis, self.quant_min, self.quant_max,
                              self.param_mode | flags, self._grad_factor(X) if self.param_mode != PARAM_FIXED else 1.0)

    # ---- state dict: scale / zero_point change size on the first observation ----------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        ops.check_persistent("state_dict")        # what is about to be saved must not come from a timed-out launch
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + "scale"] = self.scale
        destination[prefix + "zero_point"] = self.zero_point

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        for name in ("scale", "zero_point"):
            key = prefix + name
            if key in state_dict:
                stored = state_dict[key]
                mine = getattr(self, name)
                if isinstance(mine, nn.Parameter):
                    mine.data = torch.ones_like(stored, dtype=mine.dtype, device=mine.device)
                elif mine.shape != stored.shape:
                    mine.resize_(stored.shape)
            elif strict:
                missing_keys.append(key)
        for name in ("min_val", "max_val"):     # per-channel statistics also grow after construction
            key = prefix + "observer." + name
            if key in state_dict and hasattr(self.observer, name):
                buf = getattr(self.observer, name)
                if buf.shape != state_dict[key].shape:
                    setattr(self.observer, name, torch.empty_like(state_dict[key], device=buf.device))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


class FixedFakeQuantize(QuantizeBase):
    """fake_quant.py:100-126: scale fp32 buffer, zero_point int32 buffer, not learnable."""

    param_mode = PARAM_FIXED

    def __init__(self, observer, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(observer, bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.register_buffer("scale", torch.tensor([1.0], dtype=torch.float))
        self.register_buffer("zero_point", torch.tensor([0], dtype=torch.int))

    def forward(self, X, observation_mask=None, seq_pos=-1, persistent=None):
        if self.observer_enabled == 1 and self.fake_quant_enabled == 1:
            y = self._observe_and_quantize(X, observation_mask, seq_pos, persistent)
            if y is not None:
                return y
        if self.observer_enabled == 1:
            self._observe(X, observation_mask, seq_pos)
        if self.fake_quant_enabled == 1:
            X = self._quantize(X)
        return X


class _LearnableFakeQuantize(QuantizeBase):
    def __init__(self, observer, bit, symmetric, ch_axis, use_grad_scaling):
        super().__init__(observer, bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.register_buffer("eps", torch.tensor([torch.finfo(torch.float32).eps]))
        self._eps_value = float(torch.finfo(torch.float32).eps)
        self.use_grad_scaling = use_grad_scaling

    def _sanitize(self):
        """fake_quant.py:152-153 / 188-191, one launch."""
        zp = self.zero_point.data if isinstance(self.zero_point, nn.Parameter) else None
        self._touch_qparams()
        if self.scale.is_cuda:
            ops.lsq_sanitize_(self.scale.data, zp, self._eps_value, self.quant_min, self.quant_max)
        elif self.fake_quant_enabled == 1:
            raise RuntimeError("outlier_suppression_amd: quantizer parameters are not on a HIP device; "
                               "move the model with .cuda() (there is no CPU path)")

    def forward(self, X, observation_mask=None, seq_pos=-1, persistent=None):
        flags = 0
        if self.observer_enabled == 1 and self.fake_quant_enabled == 1:
            y = self._observe_and_quantize(X, observation_mask, seq_pos, persistent)
            if y is not None:
                return y
        if self.observer_enabled == 1:
            self._observe(X, observation_mask, seq_pos)
        elif self.fake_quant_enabled == 1 and self.ch_axis == -1 and self.scale.is_cuda and X.numel() > 0:
            flags = PARAM_SANITIZE        # per-tensor: the repair happens inside the fake-quant launch
        else:
            self._sanitize()
        if self.fake_quant_enabled == 1:
            X = self._quantize(X, flags)
        return X


class LSQFakeQuantize(_LearnableFakeQuantize):
    """fake_quant.py:129-167: learnable scale (Parameter), int32 zero_point buffer."""

    param_mode = PARAM_LSQ

    def __init__(self, observer, bit=8, symmetric=False, ch_axis=-1, use_grad_scaling=True):
        super().__init__(observer, bit, symmetric, ch_axis, use_grad_scaling)
        self.scale = nn.Parameter(torch.tensor([1.0], dtype=torch.float))
        self.register_buffer("zero_point", torch.tensor([0], dtype=torch.int))


class LSQPlusFakeQuantize(_LearnableFakeQuantize):
    """fake_quant.py:170-209: learnable scale and zero_point (fp32 Parameters)."""

    param_mode = PARAM_LSQPLUS

    def __init__(self, observer, bit=8, symmetric=False, ch_axis=-1, use_grad_scaling=True):
        super().__init__(observer, bit, symmetric, ch_axis, use_grad_scaling)
        self.scale = nn.Parameter(torch.tensor([1.0], dtype=torch.float))
        self.zero_point = nn.Parameter(torch.tensor([0.0], dtype=torch.float))
