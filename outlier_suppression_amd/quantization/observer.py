"""Observers with the reference's class names and buffers, computed by gfx950 HIP kernels.

Reference: quant_transformer/quantization/observer.py.  State kept exactly as there:
buffers ``min_val`` / ``max_val`` (fp32, start at +inf / -inf), Python attributes
``cnt``, ``percentile``, ``name``; state-dict keys ``observer.min_val`` /
``observer.max_val``.

Differences in mechanism, not in results:
  * padding is never removed with ``cat`` (observer.py:72-84); kernels skip padded tokens;
  * per-token extrema, the two quantiles, the thresholded extrema, the running average
    and calculate_qparams run on the device without a host sync;
  * a whole observation is 1 launch (flat / per-channel) or 2 launches (masked).
"""
import torch
import torch.nn as nn

from .. import ops
from ..ops import UPDATE_AVERAGE, UPDATE_RUNNING, QParamSink


def _quant_range(bit, symmetric):
    if symmetric:
        return -(1 << (bit - 1)), (1 << (bit - 1)) - 1
    return 0, (1 << bit) - 1


# quantization/deferred.py: while a DeferredSites object is installed here, masked activation observers record their
# site and return; the statistics of the whole forward are then computed in a handful of launches
DEFERRED = None


class ObserverBase(nn.Module):
    """observer.py:24-119."""

    update_rule = UPDATE_RUNNING
    # True for observers whose statistic is a per-batch (min, max) folded in by update_rule -- the ones sharded /
    # cached calibration can record per batch (``_capture``) and replay later (calibration.py)
    supports_capture = False

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__()
        self.bit, self.symmetric, self.ch_axis = bit, symmetric, ch_axis
        self.eps = torch.tensor(1e-8, dtype=torch.float32)
        self.quant_min, self.quant_max = _quant_range(bit, symmetric)
        self.register_buffer("min_val", torch.tensor(float("inf")))
        self.register_buffer("max_val", torch.tensor(float("-inf")))
        self._capture = None   # sharded calibration: 2-float device slot that receives this batch's (min, max)
        self._record = None    # one-call observe + quantize: slot that ALSO receives the batch's (min, max) (the running state still moves)
        self._token_cache = None   # cached grid search: (token_min, token_max) rows that receive the per-token extrema
        self._last_site = None     # ("tokens", batch, tokens, lengths) or ("flat",) of the most recent observation

    # -- attributes the solver code sets (state.py:65-69, token_wise_clipping.py:16-17)
    def set_name(self, name):
        self.name = name

    def set_batch(self, batch):
        self.batch = batch

    def set_percentile(self, percentile):
        self.percentile = percentile

    @torch.jit.export
    def calculate_qparams(self, min_val, max_val):
        """observer.py:101-119 on the device: (scale fp32, zero_point int32 if symmetric else fp32)."""
        return ops.calculate_qparams(min_val, max_val, self.quant_min, self.quant_max, self.symmetric)

    # -- helpers -----------------------------------------------------------------------
    def _home(self, device, channels=None):
        """Statistic buffers on x's device (the reference mixes a CPU 0-dim buffer with CUDA
        values freely); per-channel observers grow from the scalar start value to [C]."""
        if self.min_val.device != device:
            self.min_val = self.min_val.to(device)
            self.max_val = self.max_val.to(device)
        if channels is not None and tuple(self.min_val.shape) != (channels,):     # [C] also for C == 1 (torch.min(0-dim, [1]) is [1])
            self.min_val = self.min_val.reshape(-1)[:1].expand(channels).contiguous()
            self.max_val = self.max_val.reshape(-1)[:1].expand(channels).contiguous()

    def _counter(self):
        return getattr(self, "cnt", 0)

    def _bump(self):
        if hasattr(self, "cnt") and self._capture is None and self._token_cache is None:
            self.cnt += 1

    def _observe_tokens(self, x, lengths, seq_pos, prune, sink):
        if self._token_cache is not None:     # keep the per-token extrema; thresholds are applied later, per candidate
            _, _, batch, tokens, lengths = ops.token_minmax(x, seq_pos, lengths, out=self._token_cache)
            object.__setattr__(self, "_last_site", ("tokens", batch, tokens, lengths))   # nn.Module.__setattr__ costs microseconds
            return
        if (DEFERRED is not None and self._capture is None and self.__dict__.get("_defer_ok", False)
                and DEFERRED.add(self, x, lengths, seq_pos, prune, sink)):
            return
        self._home(x.device)
        rule, cur = self.update_rule, None
        if self._capture is not None:      # record this batch only; calibration.replay() applies the rule later
            rule, cur, sink = ops.UPDATE_NONE, self._capture, None
        batch, tokens, lengths = ops.observe_tokens(x, seq_pos, lengths, prune, getattr(self, "percentile", 1.0),
                                                    rule, self._counter(), self.min_val, self.max_val,
                                                    self.quant_min, self.quant_max, self.symmetric, sink, cur)
        object.__setattr__(self, "_last_site", ("tokens", batch, tokens, lengths))

    def token_path_prune(self):
        """None if this observer's masked-activation path is not the standard per-token one; else whether it prunes."""
        return None

    def _observe_flat(self, x, sink):
        object.__setattr__(self, "_last_site", ("flat",))
        self._home(x.device)
        rule, cur = self.update_rule, None
        if self._capture is not None:
            rule, cur, sink = ops.UPDATE_NONE, self._capture, None
        ops.observe_flat(x, rule, self._counter(), self.min_val, self.max_val,
                         self.quant_min, self.quant_max, self.symmetric, sink, cur)

    def _observe_channels(self, x, sink):
        self._home(x.device, x.shape[self.ch_axis])
        ops.observe_channels(x, self.ch_axis, self.update_rule, self._counter(), self.min_val, self.max_val,
                             self.quant_min, self.quant_max, self.symmetric, sink)

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        """Observe ``x`` and, if ``sink`` is given, also write calculate_qparams(min_val, max_val)
        into it in the same launch (what fake_quant.py:108-116 does in three steps)."""
        raise NotImplementedError

    def forward(self, x_orig, observation_mask=None, seq_pos=-1):
        if x_orig.numel() == 0:
            return x_orig
        self.observe_into(x_orig.detach(), observation_mask, seq_pos, None)
        return x_orig


class MinMaxObserver(ObserverBase):
    """observer.py:122-145: running min / max over the calibration set; per-tensor or per-channel."""

    supports_capture = True

    def token_path_prune(self):
        return False

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        if observation_mask is not None:
            assert self.ch_axis == -1
            self._observe_tokens(x, observation_mask, seq_pos, False, sink)
        elif self.ch_axis == -1:
            self._observe_flat(x, sink)
        else:
            self._observe_channels(x, sink)


class AvgMinMaxObserver(ObserverBase):
    """observer.py:176-203: average of the per-batch min / max."""

    supports_capture = True

    update_rule = UPDATE_AVERAGE

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.cnt = 0

    def token_path_prune(self):
        return False

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        assert self.ch_axis == -1
        if observation_mask is not None:
            self._observe_tokens(x, observation_mask, seq_pos, False, sink)
        else:
            self._observe_flat(x, sink)
        self._bump()


class AvgPruneMinMaxObserver(ObserverBase):
    """observer.py:206-237: token-wise clipping -- per-token extrema, percentile over tokens,
    clip range = extrema of the tokens inside the percentile; averaged over batches."""

    supports_capture = True

    update_rule = UPDATE_AVERAGE

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.cnt = 0

    def _prunes(self):
        # observer.py:62-63: attention probabilities are never pruned; the name is set by
        # state.set_observer_name and must be present, as in the reference
        if "attention_probs" in self.name:
            return False
        if getattr(self, "percentile", None) is None:
            raise AttributeError("AvgPruneMinMaxObserver: call set_percentile() before observing "
                                 "(token_wise_clipping.set_ratio does)")
        return True

    def token_path_prune(self):
        return self._prunes()

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        assert self.ch_axis == -1
        if observation_mask is not None:
            self._observe_tokens(x, observation_mask, seq_pos, self._prunes(), sink)
        elif seq_pos != -1:
            self._observe_tokens(x, None, seq_pos, self._prunes(), sink)
        else:
            self._observe_flat(x, sink)   # pooler / classifier inputs: observer.py:220-226
        self._bump()


class MSEFastObserver(ObserverBase):
    """observer.py:412-536: clipping range that minimises the quantisation MSE, found by bounded
    Brent search (1-D for symmetric or one-sided data, nested 2-D otherwise); running min/max.

    Per-tensor statistics are float64 like the reference's (scipy hands float64 results,
    observer.py:481,494); per-channel ones fp32 (observer.py:504,516).  The data's sidedness is
    decided on the first observed tensor (observer.py:528-529) -- one host read, once.
    """

    update_rule = UPDATE_RUNNING

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.p = 2.0
        self.num = 100
        self.one_side_dist = None
        self.last_nfev = None

    def _decide_side(self, cur):
        if self.one_side_dist is None:
            mn, mx = cur.tolist()
            self.one_side_dist = "pos" if mn >= 0.0 else "neg" if mx <= 0.0 else "no"

    def _reference_min_is_float64(self, two_d):
        if not two_d:
            return self.one_side_dist != "pos"
        known = self.__dict__.get("_min_f64_known", False)
        if not known:
            flags = self.__dict__.get("_ref_f64")
            known = flags is not None and bool(flags[0].item())        # one 4-byte read per call, only until it is set
            object.__setattr__(self, "_min_f64_known", known)
        return known

    def _ref_flags(self, device):
        """int32[2] on the device: the reference's dtype of (min_val, max_val) so far; the commit kernel keeps it."""
        flags = self.__dict__.get("_ref_f64")
        if flags is None or flags.device != device:
            flags = torch.zeros(2, dtype=torch.int32, device=device)
            object.__setattr__(self, "_ref_f64", flags)
        return flags

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        if observation_mask is not None:
            assert self.ch_axis == -1
        cur = ops.batch_minmax(x, observation_mask, seq_pos)
        self._decide_side(cur)
        two_d = not (self.one_side_dist != "no" or self.symmetric)
        if self.ch_axis == -1:
            # observer.py:524 / 549: x is cast to min_val's dtype, and min_val is float64 once a per-tensor search has
            # stored its (float64) result -- from the second call on the reference searches on a float64 copy of x
            # WHEN min_val turns float64 is an accident of the reference's Python arithmetic, and parity follows it
            # (oracle.msefast_search_1d / _2d, pinned against the reference run live):
            #   * 1-D, non-negative data: min_val is the float32 zero of observer.py:491 in every batch -- never;
            #   * 1-D otherwise: -torch.tensor(np.float64) -- from the second call on;
            #   * 2-D: `max(tmp_min - shift, x_min)` (observer.py:479) hands back the float32 extremum when the searched
            #     range reaches beyond it, so min_val stays float32 until one batch's best minimum lies INSIDE the data --
            #     a sticky device flag kept by the commit kernel (osq_msefast_tensor_commit, ref_float64), read back once
            #     per call while it is still unset.  The same flags make the commit average a still-float32 statistic in
            #     fp32 and derive the parameters in fp32 while both are.
            # (This package keeps both statistics in float64 storage on the device; the numbers are the reference's.)
            first_call = self.min_val.dtype != torch.float64
            float64_input = (not first_call) and self._reference_min_is_float64(two_d)
            if self.min_val.dtype != torch.float64 or self.min_val.device != x.device:
                self.min_val = self.min_val.to(device=x.device, dtype=torch.float64)
                self.max_val = self.max_val.to(device=x.device, dtype=torch.float64)
            if DEFERRED is not None and self.__dict__.get("_defer_ok", False) and x.is_cuda:
                # an observer pass (fake-quant off): the search joins the other searches of this forward in one launch
                # (quantization/deferred.py); min_val / max_val / scale move at the flush
                search = ops.msefast_tensor_begin(x, cur, observation_mask, seq_pos, self.quant_min, self.quant_max,
                                                  self.symmetric, self.one_side_dist, two_d, float64_input)
                DEFERRED.add_mse(self, search, two_d, sink)
            else:
                search = ops.msefast_tensor_begin(x, cur, observation_mask, seq_pos, self.quant_min, self.quant_max,
                                                  self.symmetric, self.one_side_dist, two_d, float64_input)
                ops.msefast_tensor_run(search, None, two_d)
                self.last_nfev = ops.msefast_tensor_commit(search, self.update_rule, self._counter(), self.min_val, self.max_val, sink,
                                                           self._ref_flags(x.device))
                ops.check_persistent("MSEFast search")     # one synchronisation behind a search of milliseconds
        else:
            bmin, bmax, self.last_nfev = ops.msefast_rows(x, self.ch_axis, self.quant_min, self.quant_max,
                                                          self.symmetric, self.one_side_dist, two_d)
            self._home(x.device, bmin.numel())
            ops.observer_update(bmin, bmax, self.update_rule, self._counter(), self.min_val, self.max_val)
            if sink is not None and sink.scale is not None:
                ops.calculate_qparams(self.min_val, self.max_val, self.quant_min, self.quant_max, self.symmetric,
                                      scale_out=sink.scale, zero_point_out=sink.zero_point)
        self._bump()

    def forward(self, x_orig, observation_mask=None, seq_pos=-1):
        if x_orig.numel() == 0:
            return x_orig
        self.observe_into(x_orig.detach(), observation_mask, seq_pos, None)
        return None     # the reference's MSEFastObserver.forward returns nothing (observer.py:520-536)


class AvgMSEFastObserver(MSEFastObserver):
    """observer.py:539-567: per-batch MSE-optimal range, averaged over batches; per-tensor only."""

    update_rule = UPDATE_AVERAGE

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.cnt = 0
        assert self.ch_axis == -1


class LSQPlusObserver(ObserverBase):
    """observer.py:148-173: weight range = mean -+ 3 std (LSQ+ initialisation); symmetric only; every
    call overwrites the statistic."""

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        assert self.symmetric is True
        self.mean = None
        self.std = None

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        self._home(x.device, None if self.ch_axis == -1 else x.shape[self.ch_axis])
        ops.observe_moments(x, self.ch_axis, self.min_val, self.max_val, self.quant_min, self.quant_max, self.symmetric,
                            sink)


class AvgQuantileObserver(ObserverBase):
    """observer.py:240-282: clip at the histogram bin where the cumulative count of |x| reaches
    ``threshold`` of the elements; averaged over batches; per-tensor only."""

    update_rule = UPDATE_AVERAGE

    def __init__(self, bit=8, symmetric=False, ch_axis=-1, ema_ratio=0.9, threshold=0.99999, bins=2048):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        assert self.ch_axis == -1, "Quantile observer only support in per-tensor scheme."
        if bins != 2048:
            raise NotImplementedError("the HIP histogram is built for the reference's 2048 bins")
        self.ema_ratio, self.threshold, self.bins = ema_ratio, threshold, bins
        self.cnt = 0
        self._hist = None

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        if observation_mask is not None:
            assert self.ch_axis == -1
        cur = ops.batch_minmax(x, observation_mask, seq_pos)
        self._home(x.device)
        if self._hist is None or self._hist.device != x.device:
            self._hist = torch.zeros(self.bins, dtype=torch.int32, device=x.device)
        ops.observe_quantile(x, observation_mask, seq_pos, cur, self.threshold, self._hist, self.update_rule,
                             self._counter(), self.min_val, self.max_val, self.quant_min, self.quant_max,
                             self.symmetric, sink)
        self._bump()


class MSEObserver(ObserverBase):
    """observer.py:285-378: brute-force search of the clipping range by quantisation MSE (100 ranges; times
    every zero-point when the data is two-sided and the scheme asymmetric); running min/max."""

    update_rule = UPDATE_RUNNING

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.p = 2.0
        self.num = 100
        self.one_side_dist = None

    def observe_into(self, x, observation_mask=None, seq_pos=-1, sink=None):
        if observation_mask is not None:
            assert self.ch_axis == -1
        cur = ops.batch_minmax(x, observation_mask, seq_pos)
        if self.one_side_dist is None:          # observer.py:373-374, decided once (one host read)
            mn, mx = cur.tolist()
            self.one_side_dist = "pos" if mn >= 0.0 else "neg" if mx <= 0.0 else "no"
        two_d = not (self.one_side_dist != "no" or self.symmetric)
        if self.ch_axis == -1:
            self._home(x.device)
            ops.mse_grid_tensor(x, observation_mask, seq_pos, cur, self.quant_min, self.quant_max, self.symmetric,
                                self.one_side_dist, two_d, self.update_rule, self._counter(), self.min_val, self.max_val,
                                sink)
        else:
            bmin, bmax = ops.mse_grid_rows(x, self.ch_axis, self.quant_min, self.quant_max, self.symmetric,
                                           self.one_side_dist, two_d)
            self._home(x.device, bmin.numel())
            ops.observer_update(bmin, bmax, self.update_rule, self._counter(), self.min_val, self.max_val)
            if sink is not None and sink.scale is not None:
                ops.calculate_qparams(self.min_val, self.max_val, self.quant_min, self.quant_max, self.symmetric,
                                      scale_out=sink.scale, zero_point_out=sink.zero_point)
        self._bump()

    def forward(self, x_orig, observation_mask=None, seq_pos=-1):
        if x_orig.numel() == 0:
            return x_orig
        self.observe_into(x_orig.detach(), observation_mask, seq_pos, None)
        return None     # as the reference (observer.py:366-378 returns nothing)


class AvgMSEObserver(MSEObserver):
    """observer.py:381-409."""

    update_rule = UPDATE_AVERAGE

    def __init__(self, bit=8, symmetric=False, ch_axis=-1):
        super().__init__(bit=bit, symmetric=symmetric, ch_axis=ch_axis)
        self.cnt = 0
        assert self.ch_axis == -1
