"""Sharded calibration: batches split over ranks, ONE small exchange of per-batch statistics.

The reference calibrates on one GPU (no collectives anywhere).  In every observer pass the
fake-quantizers are off (state.py:18-19, token_wise_clipping.py:18-19), so batch b's
(min, max) at every quantizer is independent of every other batch: rank r observes batches
b = r, r+W, r+2W, ... in capture mode (statistics recorded per batch, running state untouched),
all ranks all-gather the [batches, quantizers, 2] table once over RCCL, and every rank replays
the reference's sequential running mean ``m <- (m*cnt + cur)/(cnt+1)`` (observer.py:194-202) in
GLOBAL batch order on the device -- bit-identical to the single-GPU result, which a SUM
all-reduce would not be.  Payload: 8 B x quantizers x batches (6 KB for BERT-base) -- latency
bound; xGMI bandwidth is irrelevant.
"""
import torch
import torch.distributed as dist

from . import ops
from .quantization.fake_quant import QuantizeBase


def shard_batches(n_batches, rank, world_size):
    """Global batch indices handled by ``rank`` (round-robin keeps every rank within one batch of the others)."""
    return list(range(rank, n_batches, world_size))


def gather_batch_table(local, n_batches, group=None):
    """All-gather per-batch rows and return them in global batch order.

    ``local``: [ceil(n_batches/W), ...] on every rank, row j = this rank's j-th batch (unused
    trailing rows may hold anything).  Returns [n_batches, ...]; identical on every rank.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local[:n_batches].clone()
    world = dist.get_world_size(group)
    rows = (n_batches + world - 1) // world
    if local.shape[0] != rows:
        raise ValueError(f"gather_batch_table: expected {rows} local rows, got {local.shape[0]}")
    # RCCL ("nccl") gathers device tensors directly over xGMI; a gloo group (CPU tests, or several test
    # ranks sharing one GPU) is served by staging the few KB through the host
    staged = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.detach().cpu().contiguous() if staged else local.contiguous()
    flat = torch.empty((world * rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(flat, src, group=group)     # rank-major concatenation
    if staged:
        flat = flat.to(local.device)
    gathered = flat.view((world, rows) + tuple(local.shape[1:]))
    # gathered[r, j] is batch j*W + r  ->  transpose to [j, r] and flatten = global order
    ordered = gathered.transpose(0, 1).reshape((rows * world,) + tuple(local.shape[1:]))
    return ordered[:n_batches].contiguous()


def act_quantizers(model, select=lambda name: "act" in name):
    """Per-tensor quantizers that calibration drives, in named_modules() order (token_wise_clipping.py:13-15)."""
    return [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and select(n)]


class CaptureTable:
    """Per-batch (min, max) of every selected observer, recorded instead of being averaged."""

    def __init__(self, quantizers, rows, device):
        self.quantizers = quantizers
        self.table = torch.zeros(rows, len(quantizers), 2, dtype=torch.float32, device=device)

    def arm(self, row):
        for i, (_, q) in enumerate(self.quantizers):
            q.observer._capture = self.table[row, i]

    def disarm(self):
        for _, q in self.quantizers:
            q.observer._capture = None


def replay(ordered, quantizers):
    """Apply the gathered per-batch statistics in batch order with each observer's own rule, then
    refresh scale / zero_point.  ``ordered``: [n_batches, Q, 2] on the device."""
    n_batches, n_q = ordered.shape[0], ordered.shape[1]
    if n_q == 0:
        return
    dev = ordered.device
    by_rule = {}
    for i, (_, q) in enumerate(quantizers):
        by_rule.setdefault(q.observer.update_rule, []).append(i)
    for rule, idx in by_rule.items():
        sel = torch.tensor(idx, device=dev)
        obs = [quantizers[i][1].observer for i in idx]
        mn = torch.stack([o.min_val.reshape(()).to(dev) for o in obs]).contiguous()
        mx = torch.stack([o.max_val.reshape(()).to(dev) for o in obs]).contiguous()
        cnts = {getattr(o, "cnt", 0) for o in obs}
        if len(cnts) != 1:
            raise RuntimeError("replay: observers disagree on their batch counter")
        cnt0 = cnts.pop()
        cur = ordered.index_select(1, sel)                    # [n_batches, len(idx), 2]
        for b in range(n_batches):
            ops.observer_update(cur[b, :, 0].contiguous(), cur[b, :, 1].contiguous(), rule, cnt0 + b, mn, mx)
        for k, o in enumerate(obs):
            o.min_val = mn[k].clone()
            o.max_val = mx[k].clone()
            if hasattr(o, "cnt"):
                o.cnt = cnt0 + n_batches
    for _, q in quantizers:
        o = q.observer
        s, z = q._qparam_storage(dev, 1)
        ops.calculate_qparams(o.min_val.reshape(1), o.max_val.reshape(1), q.quant_min, q.quant_max, q.symmetric,
                              scale_out=s, zero_point_out=z)


@torch.no_grad()
def calibrate_sharded(model, batches, forward, n_batches=None, group=None, select=lambda name: "act" in name):
    """Observer pass over ``batches`` (this rank's shard, in order) + exchange + replay.

    ``batches``: the batches of ``shard_batches(n_batches, rank, W)``; ``forward(model, batch)`` runs
    the model.  Observers of the selected quantizers must be enabled and fake-quant disabled
    (token_wise_clipping.set_ratio / enable_calibration_woquantization do that).
    """
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if n_batches is None:
        n_batches = len(batches) * world
    rows = (n_batches + world - 1) // world
    qs = act_quantizers(model, select)
    dev = next(model.parameters()).device
    cap = CaptureTable(qs, rows, dev)
    try:
        for j, batch in enumerate(batches):
            cap.arm(j)
            forward(model, batch)
    finally:
        cap.disarm()
    ordered = gather_batch_table(cap.table, n_batches, group)
    replay(ordered, qs)
    return ordered
