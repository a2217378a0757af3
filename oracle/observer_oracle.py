"""Oracle (test infrastructure): observers of the reference in NumPy.

Follows ``quant_transformer/quantization/observer.py``.  Statistics are fp32 (the
reference casts the observed tensor to the dtype of its ``min_val`` buffer,
observer.py:134), except inside the MSEFast search where scipy hands float64
candidates to ``calculate_qparams`` (observer.py:423-428).
"""
import numpy as np

from .fake_quant_oracle import F32, fake_quantize_per_tensor_affine

EPS = F32(1e-8)  # observer.py:31


def quant_range(bit, symmetric):
    """observer.py:32-37."""
    if symmetric:
        return -2 ** (bit - 1), 2 ** (bit - 1) - 1
    return 0, 2 ** bit - 1


def calculate_qparams(min_val, max_val, quant_min, quant_max, symmetric):
    """observer.py:101-119.  Works in the promoted dtype of the two statistics, as torch's 0-dim / same-shape type promotion
    does: fp32, or float64 as soon as EITHER is float64 (per-tensor MSEFast: a float32 min_val beside a float64 max_val
    -- checked against the reference run live).

    Returns (scale, zero_point); zero_point is int32 zeros when symmetric
    (observer.py:109) and a float array otherwise (observer.py:117-118).
    """
    min_val = np.asarray(min_val)
    max_val = np.asarray(max_val)
    dt = np.dtype(np.float64) if np.float64 in (min_val.dtype, max_val.dtype) else (min_val.dtype if min_val.dtype.kind == "f" else np.dtype(F32))
    zero = dt.type(0)
    min_neg = np.minimum(min_val.astype(dt), zero)
    max_pos = np.maximum(max_val.astype(dt), zero)
    eps = EPS.astype(dt)
    if symmetric:
        max_pos = np.maximum(-min_neg, max_pos)
        scale = max_pos / dt.type(float(quant_max - quant_min) / 2)
        scale = np.maximum(scale, eps)
        zero_point = np.zeros(min_neg.shape, dtype=np.int32)
    else:
        scale = (max_pos - min_neg) / dt.type(float(quant_max - quant_min))
        scale = np.maximum(scale, eps)
        zero_point = dt.type(quant_min) - np.round(min_neg / scale)
        zero_point = np.clip(zero_point, dt.type(quant_min), dt.type(quant_max))
    return scale, zero_point


# ---------------------------------------------------------------------------
# padding removal (observer.py:72-98)
# ---------------------------------------------------------------------------

def _tokens_first(x, seq_pos):
    """observer.py:74-80 / 88-94: move ``seq_pos`` to axis 1 and flatten the rest -> [B, T, F]."""
    x = np.asarray(x)
    other = [d for d in range(x.ndim) if d != seq_pos]
    if len(other) == 3:
        x = np.transpose(x, (other[0], seq_pos, other[1], other[2]))
        x = x.reshape(x.shape[0], x.shape[1], -1)
    elif len(other) == 2:
        x = np.transpose(x, (other[0], seq_pos, other[1]))
    return x


def remove_padding(x, lengths, seq_pos):
    """observer.py:72-84: concatenate the first ``lengths[b]`` tokens of every sample -> [sum L, F].

    ``zip`` (observer.py:82) stops at the shorter of (lengths, batch rows): a
    ``[B*h, T, S]`` BART tensor with a length-B mask keeps only its first B rows.
    """
    xt = _tokens_first(x, seq_pos)
    rows = [seq[: int(n)] for n, seq in zip(np.asarray(lengths).tolist(), xt)]
    if not rows:
        return np.zeros((0,), dtype=F32)
    return np.concatenate(rows, axis=0).astype(F32)


def reshape_batch_embedding(x, seq_pos):
    """observer.py:86-98: as remove_padding with every token kept."""
    xt = _tokens_first(x, seq_pos)
    return xt.reshape(-1, xt.shape[-1]).astype(F32)


# ---------------------------------------------------------------------------
# torch.quantile (linear interpolation) restated; pinned in tests/test_oracle_pinning.py
# ---------------------------------------------------------------------------

def fma_f32(a, b, c):
    """Correctly rounded fp32 ``a*b + c`` (single rounding), as torch's CPU lerp kernel computes.

    The product of two fp32 values is exact in float64; the float64 sum may
    round once, so the TwoSum error term is used to break an fp32 tie the way
    the infinitely precise value would.
    """
    p = np.float64(F32(a)) * np.float64(F32(b))
    c = np.float64(F32(c))
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    r = F32(s)
    if err != 0.0 and np.isfinite(r):
        rd = np.float64(r)
        if rd != s:
            # s lies strictly between fp32 neighbours; a halfway case needs the sign of err
            other = np.nextafter(r, F32(np.inf) if s > rd else F32(-np.inf))
            mid = (rd + np.float64(other)) * 0.5
            if s == mid:
                toward_other = (err > 0) == (np.float64(other) > rd)
                r = other if toward_other else r
    return F32(r)


def torch_quantile_linear(values, q):
    """``torch.quantile(values, q)`` for a 1-D fp32 tensor and scalar ``q`` (observer.py:51).

    rank = fp32(q) * (n-1) in fp32; below/above = floor/ceil; result =
    lerp(below, above, rank - floor(rank)) with torch's two-branch lerp
    evaluated with one fused multiply-add per branch.
    """
    v = np.sort(np.asarray(values, dtype=F32).reshape(-1))
    n = v.size
    if np.isnan(v).any():
        return F32(np.nan)
    rank = F32(q) * F32(n - 1)
    lo = int(np.floor(rank))
    hi = int(np.ceil(rank))
    w = F32(rank - F32(lo))
    a, b = v[lo], v[hi]
    diff = F32(b - a)
    if w < F32(0.5):
        return fma_f32(w, diff, a)
    return fma_f32(F32(w - F32(1)), diff, b)


# ---- zero extrema --------------------------------------------------------------------------------------------------
# torch's CPU min / max of data that holds BOTH -0.0 and +0.0 returns whichever zero sits in the winning SIMD lane
# (`_mm256_min_ps` keeps its second operand when both are zero; the tree over a row depends on its length and the lane
# the zeros fall in) -- the reference itself has no defined answer, and its hard-coded `.cuda()` path has yet another.
# The restatement therefore fixes the one thing the reference leaves open with the IEEE 754-2019 minimum / maximum rule,
# -0 < +0: a minimum that is zero is -0.0 if any -0.0 took part, a maximum that is zero is +0.0 if any +0.0 took part.
# The device follows the same rule (v_min_f32 / v_max_f32 order the zeros that way), so statistics compare as BITS.
# Everywhere the data holds zeros of one sign only -- every fixture captured from the reference -- nothing changes.

def zmin(a, axis=None):
    a = np.asarray(a)
    r = np.asarray(a.min(axis=axis))
    neg0 = np.asarray(((a == 0) & np.signbit(a)).any(axis=axis))
    out = np.where((r == 0) & neg0, a.dtype.type(-0.0), np.where(r == 0, a.dtype.type(0.0), r)).astype(a.dtype)
    return out if out.ndim else out.reshape(())[()]


def zmax(a, axis=None):
    a = np.asarray(a)
    r = np.asarray(a.max(axis=axis))
    pos0 = np.asarray(((a == 0) & ~np.signbit(a)).any(axis=axis))
    out = np.where((r == 0) & ~pos0, a.dtype.type(-0.0), np.where(r == 0, a.dtype.type(0.0), r)).astype(a.dtype)
    return out if out.ndim else out.reshape(())[()]


def zminimum(a, b):
    a, b = np.asarray(a), np.asarray(b)
    r = np.minimum(a, b)
    both0 = (a == 0) & (b == 0)
    return np.where(both0, np.where(np.signbit(a) | np.signbit(b), r.dtype.type(-0.0), r.dtype.type(0.0)), r).astype(r.dtype)


def zmaximum(a, b):
    a, b = np.asarray(a), np.asarray(b)
    r = np.maximum(a, b)
    both0 = (a == 0) & (b == 0)
    return np.where(both0, np.where(np.signbit(a) & np.signbit(b), r.dtype.type(-0.0), r.dtype.type(0.0)), r).astype(r.dtype)


def token_min_max(value):
    """observer.py:64-65: per-token max and min over the feature axis of ``[N_tok, F]``."""
    value = np.asarray(value, dtype=F32)
    return zmin(value, axis=1), zmax(value, axis=1)


def prune_thresholds(token_min, token_max, percentile):
    """observer.py:50-59,66-67: clipping bounds picked by the token-wise percentile.

    upper = quantile(|token_max|, p); lower = -quantile(|token_min|, p);
    up = max(token_max[token_max <= upper]); lo = min(token_min[token_min >= lower]).
    """
    upper = torch_quantile_linear(np.abs(token_max), percentile)
    lower = -torch_quantile_linear(np.abs(token_min), percentile)
    up = zmax(token_max[token_max <= upper])
    lo = zmin(token_min[token_min >= lower])
    return F32(lo), F32(up)


def prune_token(value, percentile, name=""):
    """observer.py:61-70: returns the clipped ``[N_tok, F]`` tensor (unchanged for attention_probs)."""
    value = np.asarray(value, dtype=F32)
    if "attention_probs" in name:
        return value
    tmin, tmax = token_min_max(value)
    lo, up = prune_thresholds(tmin, tmax, percentile)
    # torch.clip(value, min=lo, max=up) == min(max(value, lo), up)
    return zminimum(zmaximum(value, lo), up)       # zero signs by the rule above (torch.clip leaves them to the lane)


def aminmax(x):
    x = np.asarray(x, dtype=F32)
    return F32(zmin(x)), F32(zmax(x))


def _to_channel_rows(x, ch_axis):
    """observer.py:11-21 ``_transform_to_ch_axis``: swap ch_axis with axis 0 and flatten -> [C, -1]."""
    x = np.asarray(x, dtype=F32)
    order = list(range(x.ndim))
    order[ch_axis], order[0] = 0, ch_axis
    return np.transpose(x, order).reshape(x.shape[ch_axis], -1)


# ---------------------------------------------------------------------------
# observer state machines
# ---------------------------------------------------------------------------

class ObserverState:
    """Plain-data stand-in for ObserverBase's buffers (observer.py:26-39)."""

    def __init__(self, bit=8, symmetric=False, ch_axis=-1, name=""):
        self.bit, self.symmetric, self.ch_axis = bit, symmetric, ch_axis
        self.quant_min, self.quant_max = quant_range(bit, symmetric)
        self.min_val = np.asarray(F32(np.inf))
        self.max_val = np.asarray(F32(-np.inf))
        self.cnt = 0
        self.percentile = None
        self.name = name
        self.one_side_dist = None

    def qparams(self):
        return calculate_qparams(self.min_val, self.max_val, self.quant_min, self.quant_max, self.symmetric)

    # observer.py:194-202 (also 228-236, 559-567)
    def _avg_update(self, cur_min, cur_max):
        # Each statistic keeps ITS OWN dtype (tensor * python int, tensor / python int): one-sided positive data under
        # MSEFast leaves min_val a float32 zero beside a float64 max_val (observer.py:491-492), and observer.py:549 casts
        # the next batch to min_val's dtype -- so such data is searched in fp32 in every batch (checked against the
        # reference run live: min_val stays torch.float32).
        first = self.max_val.size <= 1 and np.isinf(self.max_val).all()

        def one(old, cur):
            if first:
                return np.asarray(cur)
            c = F32(self.cnt) if old.dtype == F32 else np.float64(self.cnt)
            return np.asarray(old * c + cur)
        mn, mx = one(self.min_val, cur_min), one(self.max_val, cur_max)
        self.cnt += 1
        self.min_val = np.asarray(mn / (F32(self.cnt) if mn.dtype == F32 else np.float64(self.cnt)))
        self.max_val = np.asarray(mx / (F32(self.cnt) if mx.dtype == F32 else np.float64(self.cnt)))

    # observer.py:143-144 (also 535-536)
    def _running_update(self, cur_min, cur_max):
        self.min_val = zminimum(self.min_val, cur_min)
        self.max_val = zmaximum(self.max_val, cur_max)


def _prepare(x, lengths, seq_pos):
    """Masked: remove_padding's [valid tokens, features] copy.  No mask: the reference works on ``x_orig.clone()``
    (observer.py:134 and the like), which keeps the strides of a dense permuted view, and torch's CPU reductions add such a
    tensor in MEMORY order -- min / max do not care, the MSE loss of a per-tensor search does in its last bits: hand this
    function the array in the memory order of the tensor the reference would see (C order for a contiguous one)."""
    x = np.asarray(x, dtype=F32)
    if lengths is not None:
        return remove_padding(x, lengths, seq_pos)
    return x


def observe_minmax(st, x, lengths=None, seq_pos=-1):
    """MinMaxObserver.forward, observer.py:130-145."""
    x = np.asarray(x)
    if x.size == 0:
        return
    x = _prepare(x, lengths, seq_pos)
    if st.ch_axis == -1:
        cur_min, cur_max = aminmax(x)
    else:
        rows = _to_channel_rows(x, st.ch_axis)
        cur_min, cur_max = zmin(rows, axis=1), zmax(rows, axis=1)
    st._running_update(cur_min, cur_max)


def observe_avg_minmax(st, x, lengths=None, seq_pos=-1):
    """AvgMinMaxObserver.forward, observer.py:184-203."""
    x = np.asarray(x)
    if x.size == 0:
        return
    x = _prepare(x, lengths, seq_pos)
    assert st.ch_axis == -1
    st._avg_update(*aminmax(x))


def observe_avg_prune_minmax(st, x, lengths=None, seq_pos=-1):
    """AvgPruneMinMaxObserver.forward, observer.py:214-237."""
    x = np.asarray(x)
    if x.size == 0:
        return
    x = np.asarray(x, dtype=F32)
    if lengths is not None:
        x = prune_token(remove_padding(x, lengths, seq_pos), st.percentile, st.name)
    elif seq_pos != -1:
        x = prune_token(reshape_batch_embedding(x, seq_pos), st.percentile, st.name)
    assert st.ch_axis == -1
    st._avg_update(*aminmax(x))


# ---------------------------------------------------------------------------
# MSEFast (observer.py:412-567)
# ---------------------------------------------------------------------------

# How the squared errors are averaged.  The reference's ``.pow(2).mean()`` is torch's sum, whose order depends on the
# build's vector width (ATen cascade_sum) -- not part of the algorithm.  Default FOR PER-TENSOR searches (rows: ROW_SUM_VEC
# below): exact (float64) sum, rounded to the
# tensor's dtype once; the device kernels do the same, so kernel and oracle agree bit for bit.  Tests that compare with
# the reference RUN ON THE SAME MACHINE set ``MEAN_LIKE_TORCH = lambda sq: torch.from_numpy(sq).mean().numpy()`` to show
# that the summation order is the only difference (tests/test_oracle_vs_reference_live.py).
MEAN_LIKE_TORCH = None
# Per-channel searches are the exception: a row is shorter than ATen's 32768-element grain, so the reference adds it serially,
# in an order that depends only on the vector width of torch's sum kernel (8 fp32 lanes on x86, AVX2 and AVX-512 alike:
# oracle/aten_sum.py) -- part of what "the reference CPU path" computes, whatever the host's thread count.  The oracle (and
# the kernels, osq_set_tuning("mse_rows_order", 8)) therefore sum ROWS in that order by default; None = the exact sum.
ROW_SUM_VEC = 8


def exact_sum(a):
    """Correctly rounded float64 sum of non-negative values, independent of any order.  Up to 200 000 elements:
    math.fsum (exact by construction).  Beyond (the per-tensor sites of configs[3]: millions of squared errors per loss
    evaluation, hundreds of evaluations per search): a pairwise tree of error-free two-sums in NumPy -- every addition's
    rounding error is kept in a second word, so the pair (hi, lo) carries the sum to ~2^-100 relative; for non-negative
    terms (no cancellation) hi + lo rounds to the same double as the exact sum unless that sum lies within ~2^-100 of a
    rounding boundary.  Pinned against math.fsum in tests/test_oracle_pinning.py::test_exact_sum."""
    import math
    a = np.asarray(a, dtype=np.float64).ravel()
    if a.size <= 200000:
        return np.float64(math.fsum(a.tolist()))
    hi, lo = a, np.zeros_like(a)
    while hi.size > 1:
        if hi.size & 1:
            hi, lo = np.append(hi, 0.0), np.append(lo, 0.0)
        x, y = hi[0::2], hi[1::2]
        s = x + y
        bb = s - x
        err = (x - (s - bb)) + (y - bb)                 # two-sum: x + y = s + err exactly
        lo = (lo[0::2] + lo[1::2]) + err
        hi = s
    return np.float64(hi[0] + lo[0])


def exact_mean(sq):
    """Mean with an exactly rounded sum: the order-independent mean.  ``MEAN_LIKE_TORCH = exact_mean`` is the
    counterpart of the kernels' double-double test mode (osq_set_tuning("mse_sum_order", 64)): with both, kernel and
    oracle agree bit for bit also in float64 arithmetic (a per-tensor observer's second call on), where a plain sum
    carries its order in its last bits (tests/test_gpu_parity.py::test_msefast_float64_equals_oracle_with_exact_sums)."""
    a = np.asarray(sq, dtype=np.float64).ravel()
    return exact_sum(a) / np.float64(a.size)


def mse_loss(x, new_min, new_max, quant_min, quant_max, symmetric):
    """observer.py:423-432 ``loss_fx`` + ``lp_loss`` (p=2).

    ``new_min/new_max`` arrive as float64 (scipy), so qparams are float64; the scale reaches the fake-quant as a
    Python float and the zero-point is truncated with ``int()``.  x is fp32 -- scale applied in fp32, np.float32
    loss, which scipy's bounded minimiser then subtracts in float32 -- or float64: a per-tensor observer casts x to
    ``min_val``'s dtype (observer.py:524,549), and ``min_val`` is float64 after the first call (observer.py:481,494),
    so from the second call on the reference does all of this in float64.
    """
    x = np.asarray(x)
    scale, zp = calculate_qparams(np.float64(new_min), np.float64(new_max), quant_min, quant_max, symmetric)
    if x.dtype == np.float64:
        s, z = np.float64(float(scale)), np.float64(int(zp))
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            u = x / s
            x_int = ((np.round(u) - u) + u) + z
            y = (np.clip(x_int, np.float64(quant_min), np.float64(quant_max)) - z) * s
            d = np.abs(y - x)
            sq = d * d
        return np.float64(MEAN_LIKE_TORCH(sq)) if MEAN_LIKE_TORCH is not None else np.float64(sq.mean())
    x = x.astype(F32, copy=False)
    _, y = fake_quantize_per_tensor_affine(x, F32(float(scale)), F32(int(zp)), quant_min, quant_max)
    d = np.abs(y - x)
    if MEAN_LIKE_TORCH is not None:
        return F32(MEAN_LIKE_TORCH(d * d))
    return F32((d * d).astype(np.float64).mean())


def one_side_dist(x):
    """observer.py:528-529."""
    x = np.asarray(x)
    return "pos" if x.min() >= 0.0 else "neg" if x.max() <= 0.0 else "no"


def msefast_search_1d(x, x_min, x_max, st, counter=None):
    """observer.py:441-445,483-494: bounded Brent over the clipping range, symmetric or one-sided."""
    from scipy.optimize import minimize_scalar
    xr = float(max(abs(float(x_min)), float(x_max)))

    def loss(r):
        if counter is not None:
            counter[0] += 1
        lo = 0.0 if st.one_side_dist == "pos" else -r
        hi = 0.0 if st.one_side_dist == "neg" else r
        return mse_loss(x, lo, hi, st.quant_min, st.quant_max, st.symmetric)

    res = minimize_scalar(loss, bounds=(min(0.1, 0.01 * xr), xr), method="Bounded")
    r = np.float64(res.x)
    # observer.py:491-492: the one-sided zero is zeros_like(fp32 extremum); the searched side is float64
    best_min = F32(0) if st.one_side_dist == "pos" else -r
    best_max = F32(0) if st.one_side_dist == "neg" else r
    return best_min, best_max


def msefast_search_2d(x, x_min, x_max, st, counter=None):
    """observer.py:434-481: outer Brent over range, inner Brent over shift."""
    from scipy.optimize import minimize_scalar
    x_min, x_max = float(x_min), float(x_max)
    span = float(quant_max_minus_min(st))

    def shift_loss(shift, xr):
        if counter is not None:
            counter[0] += 1
        return mse_loss(x, max(0.0 - shift, x_min), min(xr - shift, x_max),
                        st.quant_min, st.quant_max, st.symmetric)

    def shift_bounds(xr):
        delta = xr / span
        return delta * st.quant_min, delta * st.quant_max

    def range_loss(xr):
        return minimize_scalar(shift_loss, args=(xr,), bounds=shift_bounds(xr), method="Bounded").fun

    # observer.py:459: x_max - x_min in the tensor's dtype (fp32 extrema; float64 from a per-tensor observer's second call on)
    xr0 = float(x_max - x_min) if np.asarray(x).dtype == np.float64 else float(F32(x_max) - F32(x_min))
    res = minimize_scalar(range_loss, bounds=(min(0.1, 0.01 * xr0), xr0), method="Bounded")
    final_range = res.x
    sub = minimize_scalar(shift_loss, args=(final_range,), bounds=shift_bounds(final_range), method="Bounded")
    # observer.py:479-480: Python max/min keep the fp32 extremum's dtype when it wins
    lo, hi = np.float64(0.0 - sub.x), np.float64(final_range - sub.x)
    return (F32(x_min) if x_min > lo else lo), (F32(x_max) if x_max < hi else hi)


def quant_max_minus_min(st):
    return st.quant_max - st.quant_min


def observe_msefast(st, x, lengths=None, seq_pos=-1, average=False, counter=None):
    """MSEFastObserver.forward (observer.py:520-536) / AvgMSEFastObserver.forward (545-567)."""
    x = np.asarray(x)
    if x.size == 0:
        return
    x = _prepare(x, lengths, seq_pos)
    if np.asarray(st.min_val).dtype == np.float64:     # observer.py:524 / 549: x.to(self.min_val.dtype)
        x = x.astype(np.float64)
    if st.one_side_dist is None:
        st.one_side_dist = one_side_dist(x)
    search = msefast_search_1d if (st.one_side_dist != "no" or st.symmetric) else msefast_search_2d
    if st.ch_axis == -1:
        x_min, x_max = aminmax(x)
        best_min, best_max = search(x, x_min, x_max, st, counter)
        # per-tensor results keep the dtype the search produced (float64 from scipy, observer.py:481,494)
        best_min, best_max = np.asarray(best_min), np.asarray(best_max)
    else:
        rows = _to_channel_rows(x, st.ch_axis)
        best_min, best_max = zmin(rows, axis=1), zmax(rows, axis=1)
        global MEAN_LIKE_TORCH
        row_order = MEAN_LIKE_TORCH is None and ROW_SUM_VEC is not None and rows.shape[1] < 32768 and rows.dtype == F32
        if row_order:
            from .aten_sum import aten_mean_f32
            MEAN_LIKE_TORCH = lambda sq: aten_mean_f32(sq, ROW_SUM_VEC)      # noqa: E731
        try:
            for c in range(rows.shape[0]):
                lo, hi = search(rows[c], best_min[c], best_max[c], st, counter)
                best_min[c], best_max[c] = lo, hi  # assignment into fp32 tensors (observer.py:504,516)
        finally:
            if row_order:
                MEAN_LIKE_TORCH = None
    if average:
        st._avg_update(best_min, best_max)
    else:
        st._running_update(best_min, best_max)


# ---------------------------------------------------------------------------
# remaining observers of ObserverDict
# ---------------------------------------------------------------------------

def observe_lsqplus(st, x):
    """LSQPlusObserver.forward, observer.py:159-173: range = mean -+ 3 std (unbiased std), not accumulated.
    Moments are taken in float64 and rounded to fp32 (torch's own accumulation order is not part of
    the reference); compare with a 1e-6 tolerance."""
    x = np.asarray(x, dtype=F32)
    if x.size == 0:
        return
    assert st.symmetric
    if st.ch_axis == -1:
        mean = F32(x.astype(np.float64).mean())
        std = F32(x.astype(np.float64).std(ddof=1))
    else:
        rows = _to_channel_rows(x, st.ch_axis).astype(np.float64)
        mean = rows.mean(axis=1).astype(F32)
        std = rows.std(axis=1, ddof=1).astype(F32)
    st.min_val = np.asarray(mean - F32(3) * std)
    st.max_val = np.asarray(mean + F32(3) * std)


def linspace_edge(i, start, end, steps):
    """torch.linspace element i in fp32 (scalar form): start + step*i below the midpoint, end - step*(steps-1-i) above."""
    step = F32(F32(end) - F32(start)) / F32(steps - 1)
    if i < steps // 2:
        return F32(F32(start) + F32(step * F32(i)))
    return F32(F32(end) - F32(step * F32(steps - i - 1)))


def torch_histc(values, bins, vmin, vmax):
    """torch.histc(values, bins, min, max) on the CPU (observer.py:263): linear bin estimate followed by the
    local search against the linspace edges; the last bin is closed on the right; values outside are dropped."""
    v = np.asarray(values, dtype=F32).reshape(-1)
    lo, hi = F32(vmin), F32(vmax)
    counts = np.zeros(bins, dtype=np.int64)
    if not (hi > lo):
        hi = F32(lo + F32(1)) if hi == lo else hi       # torch widens an empty range to [min, min + 1]... only via min==max==0
    edges = np.array([linspace_edge(i, lo, hi, bins + 1) for i in range(bins + 1)], dtype=F32)
    inside = (v >= lo) & (v <= hi)
    vi = v[inside]
    pos = (F32(vi - lo) * F32(bins) / F32(hi - lo)).astype(np.int64)
    for e, p in zip(vi, pos):
        a, b = max(0, p - 1), min(p + 2, bins + 1)
        j = a + int(np.searchsorted(edges[a:b], e, side="right")) - 1
        if j == bins:
            j -= 1
        counts[j] += 1
    return counts.astype(F32)


def quantile_clip_from_hist(hist, numel, threshold, max_range, bins):
    """observer.py:264-270: first bin whose cumulative count reaches threshold*numel -> bin centre."""
    cur = F32(0)
    target = F32(threshold * numel)            # fp32 tensor compared with a Python float
    clip = F32(max_range)
    for i in range(bins):
        if F32(cur + hist[i]) >= target:
            clip = F32(F32(i + 0.5) * F32(F32(max_range) / F32(bins)))
            break
        cur = F32(cur + hist[i])
    return clip


def observe_avg_quantile(st, x, lengths=None, seq_pos=-1, threshold=0.99999, bins=2048):
    """AvgQuantileObserver.forward, observer.py:253-282."""
    x = np.asarray(x)
    if x.size == 0:
        return
    x = _prepare(x, lengths, seq_pos)
    mn, mx = aminmax(x)
    max_range = F32(max(F32(-mn), mx))
    hist = torch_histc(np.abs(x), bins, 0.0, max_range)
    clip = quantile_clip_from_hist(hist, x.size, threshold, max_range, bins)
    st._avg_update(F32(max(mn, F32(-clip))), F32(min(mx, clip)))


def mse_grid_loss(x, new_min, new_max, quant_min, quant_max, symmetric):
    """MSEObserver.loss_fx + lp_loss (observer.py:292-312), per-tensor: fp32 qparams, int zero-point."""
    scale, zp = calculate_qparams(F32(new_min), F32(new_max), quant_min, quant_max, symmetric)
    _, y = fake_quantize_per_tensor_affine(x, F32(scale), F32(int(zp)), quant_min, quant_max)
    d = np.abs(y - x)
    return F32((d * d).astype(np.float64).mean())


def mse_grid_search(x, st, num=100):
    """perform_1D_search / perform_2D_search (observer.py:314-364) for one tensor (or one channel row)."""
    x = np.asarray(x, dtype=F32)
    x_min, x_max = aminmax(x)
    best_score, best_min, best_max = F32(1e10), x_min, x_max
    if st.one_side_dist != "no" or st.symmetric:
        xr = F32(max(abs(x_min), x_max))
        for i in range(1, num + 1):
            thres = F32(F32(xr / F32(num)) * F32(i))
            lo = F32(0) if st.one_side_dist == "pos" else F32(-thres)
            hi = F32(0) if st.one_side_dist == "neg" else thres
            score = mse_grid_loss(x, lo, hi, st.quant_min, st.quant_max, st.symmetric)
            if score < best_score:
                best_score, best_min, best_max = score, lo, hi
        return best_min, best_max
    xr = F32(x_max - x_min)
    span = F32(float(st.quant_max - st.quant_min))
    for i in range(1, num + 1):
        tmp_max = F32(F32(xr / F32(num)) * F32(i))
        delta = F32(tmp_max / span)
        for zp in range(st.quant_min, st.quant_max + 1):
            lo = F32(max(F32(F32(0) - F32(F32(zp) * delta)), x_min))
            hi = F32(min(F32(tmp_max - F32(F32(zp) * delta)), x_max))
            score = mse_grid_loss(x, lo, hi, st.quant_min, st.quant_max, st.symmetric)
            if score < best_score:
                best_score, best_min, best_max = score, lo, hi
    return best_min, best_max


def observe_mse(st, x, lengths=None, seq_pos=-1, average=False):
    """MSEObserver.forward (observer.py:366-378) / AvgMSEObserver.forward (:387-409)."""
    x = np.asarray(x)
    if x.size == 0:
        return
    x = _prepare(x, lengths, seq_pos)
    if st.one_side_dist is None:
        st.one_side_dist = one_side_dist(x)
    if st.ch_axis == -1:
        best_min, best_max = mse_grid_search(x, st)
    else:
        rows = _to_channel_rows(x, st.ch_axis)
        res = [mse_grid_search(r, st) for r in rows]
        best_min = np.array([r[0] for r in res], dtype=F32)
        best_max = np.array([r[1] for r in res], dtype=F32)
    if average:
        st._avg_update(best_min, best_max)
    else:
        st._running_update(best_min, best_max)
