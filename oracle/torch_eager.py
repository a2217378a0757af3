"""Oracle (test infrastructure): the hot path spelled as stock torch CPU ops, multi-threaded.

This is what the reference *executes* -- eager PyTorch op chains on the host
(util_quant.py:11-15, observer.py:50-84,184-237) -- restated functionally so that
``bench.py`` can time "the reference's CPU path" on the GPU box's host cores
(``cpu_baseline``), where /root/reference itself does not exist.  It is checked against the
same golden vectors as the NumPy oracle (tests/test_oracle_golden.py).  Never imported by the
product.
"""
import torch


def fake_quant_chain(x, scale, zero_point, quant_min, quant_max):
    """util_quant.py:11-15 as eight eager ops (div, round, sub, add, add zp, clamp, sub zp, mul)."""
    u = x / scale
    x_int = ((u.round() - u) + u) + zero_point
    x_q = torch.clamp(x_int, quant_min, quant_max)
    return (x_q - zero_point) * scale


def lsqplus_chain(x, scale, zero_point, quant_min, quant_max, g):
    """util_quant.py:48-55 forward with 1-element tensors for scale / zero_point."""
    zp = (zero_point.round() - zero_point) + zero_point
    s = (scale - scale * g) + scale * g
    zp = (zp - zp * g) + zp * g
    return fake_quant_chain(x, s, zp, quant_min, quant_max)


def valid_tokens(x, lengths, seq_pos):
    """observer.py:72-84: permute the sequence axis to dim 1, flatten features, cat the valid prefixes."""
    dims = [d for d in range(x.dim()) if d != seq_pos]
    if len(dims) == 3:
        x = x.permute(dims[0], seq_pos, dims[1], dims[2]).reshape(x.shape[dims[0]], x.shape[seq_pos], -1)
    elif len(dims) == 2:
        x = x.permute(dims[0], seq_pos, dims[1])
    out = torch.empty(0)
    for n, seq in zip(lengths, x):
        out = torch.cat((out, seq[:n]), 0)
    return out


def pruned_minmax(value, percentile):
    """observer.py:50-70 + :227: token-wise clipping, then aminmax of the clipped tensor."""
    token_max = value.max(1)[0]
    token_min = value.min(1)[0]
    upper = torch.quantile(token_max.abs(), percentile)
    lower = -torch.quantile(token_min.abs(), percentile)
    up = token_max[torch.nonzero(token_max <= upper, as_tuple=True)[0]].max()
    lo = token_min[torch.nonzero(token_min >= lower, as_tuple=True)[0]].min()
    return torch.aminmax(torch.clip(value, min=lo, max=up))


def running_average(state, cur_min, cur_max):
    """observer.py:194-202.  state = [min_val, max_val, cnt]."""
    if torch.isinf(state[1]):
        state[0], state[1] = cur_min, cur_max
    else:
        state[0] = state[0] * state[2] + cur_min
        state[1] = state[1] * state[2] + cur_max
    state[2] += 1
    state[0] = state[0] / state[2]
    state[1] = state[1] / state[2]


def qparams(min_val, max_val, quant_min, quant_max, symmetric):
    """observer.py:101-119."""
    mn = torch.min(min_val, torch.zeros_like(min_val))
    mx = torch.max(max_val, torch.zeros_like(max_val))
    eps = torch.tensor(1e-8)
    if symmetric:
        scale = torch.max(torch.max(-mn, mx) / (float(quant_max - quant_min) / 2), eps)
        return scale, torch.zeros_like(scale, dtype=torch.int)
    scale = torch.max((mx - mn) / float(quant_max - quant_min), eps)
    zp = torch.clamp(quant_min - torch.round(mn / scale), quant_min, quant_max)
    return scale, zp


def observe_prune_then_quantize(x, lengths, percentile, state, quant_min=0, quant_max=63):
    """One hot-path step as the reference runs it with observer and fake-quant both on:
    AvgPruneMinMaxObserver.forward -> calculate_qparams -> LSQ+ fake-quant (fake_quant.py:178-208)."""
    v = valid_tokens(x.clone().detach().to(torch.float32), lengths, 1)
    cur_min, cur_max = pruned_minmax(v, percentile)
    running_average(state, cur_min, cur_max)
    scale, zp = qparams(state[0], state[1], quant_min, quant_max, False)
    g = 1.0 / (x.numel() * quant_max) ** 0.5
    return lsqplus_chain(x, scale.reshape(1), zp.reshape(1).float(), quant_min, quant_max, g), scale, zp
