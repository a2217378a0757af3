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


def _require_capture_support(quantizers, what):
    """Sharded / cached calibration records each batch's (min, max) and replays the observers' update rules later.
    Observers that keep their own state (MSEFast / MSE searches with float64 statistics, AvgQuantile, LSQPlusObserver)
    neither fill the capture slot nor fit the fp32 replay: refuse instead of folding zeros into their statistics."""
    for name, q in quantizers:
        obs = q.observer
        if not getattr(obs, "supports_capture", False) or obs.min_val.dtype != torch.float32 or obs.ch_axis != -1:
            raise NotImplementedError(
                f"{what}: {type(obs).__name__} at '{name}' cannot be recorded per batch and replayed "
                "(per-tensor MinMax / AvgMinMax / AvgPruneMinMax observers only); calibrate it with the plain loop")


class CaptureTable:
    """Per-batch (min, max) of every selected observer, recorded instead of being averaged."""

    def __init__(self, quantizers, rows, device):
        _require_capture_support(quantizers, "sharded calibration")
        self.quantizers = quantizers
        self.table = torch.zeros(rows, len(quantizers), 2, dtype=torch.float32, device=device)

    def arm(self, row):
        for i, (_, q) in enumerate(self.quantizers):
            q.observer._capture = self.table[row, i]

    def disarm(self):
        for _, q in self.quantizers:
            q.observer._capture = None


class ReplayPlan:
    """Device-side description of a set of quantizers for ``osq_replay_statistics``: per-quantizer update rule,
    quantisation range and the ADDRESSES of its observer's min_val / max_val buffers and of its scale /
    zero_point storage.  Built once; every replay is then one launch, with no per-quantizer tensor traffic.
    The tensors behind the addresses are kept alive (and never reallocated) by the plan."""

    def __init__(self, quantizers, device):
        _require_capture_support(quantizers, "statistics replay")
        self.quantizers = quantizers
        self.device = device
        obs = [q.observer for _, q in quantizers]
        for o in obs:
            o._home(device)
            if o.min_val.numel() != 1 or o.max_val.numel() != 1:
                raise NotImplementedError("replay: per-tensor observers only")
            if o.min_val.dim():                       # 0-dim buffers as the reference registers them
                o.min_val = o.min_val.reshape(())
                o.max_val = o.max_val.reshape(())
        self._keep = []
        i32 = lambda vals: torch.tensor(list(vals), dtype=torch.int32, device=device)
        u64 = lambda vals: torch.tensor(list(vals), dtype=torch.int64, device=device)
        scales, zps = [], []
        for _, q in quantizers:
            s, z = q._qparam_storage(device, 1)
            scales.append(s)
            zps.append(z)
        self._keep += [o.min_val for o in obs] + [o.max_val for o in obs] + scales + zps
        self.rules = i32(o.update_rule for o in obs)
        self.min_ptrs = u64(o.min_val.data_ptr() for o in obs)
        self.max_ptrs = u64(o.max_val.data_ptr() for o in obs)
        self.qmin = i32(q.quant_min for _, q in quantizers)
        self.qmax = i32(q.quant_max for _, q in quantizers)
        self.sym = i32(int(bool(q.symmetric)) for _, q in quantizers)
        self.scale_ptrs = u64(t.data_ptr() for t in scales)
        self.zp_ptrs = u64(t.data_ptr() for t in zps)
        self.zp_types = i32(ops._zp_type(t) for t in zps)
        self._addresses = [(o.min_val.data_ptr(), o.max_val.data_ptr()) for o in obs]

    def valid(self):
        """False once somebody re-assigned an observer buffer (the plan's addresses would be stale)."""
        return all((o.min_val.data_ptr(), o.max_val.data_ptr()) == adr
                   for (_, q), adr in zip(self.quantizers, self._addresses) for o in (q.observer,))

    def run(self, ordered, fresh=False):
        """Fold ``ordered`` [n_batches, Q, 2] into every observer's running statistic, in batch order, and refresh
        scale / zero_point.  ``fresh``: start from the untouched (+inf, -inf) state (what the reference gets by
        ``cnt = 0`` on an observer whose first-batch test then fires, token_wise_clipping.py:17, observer.py:194)."""
        n_batches, n_q = ordered.shape[0], ordered.shape[1]
        if n_q == 0:
            return
        from . import _hip
        obs = [q.observer for _, q in self.quantizers]
        cnts = {0 if fresh else getattr(o, "cnt", 0) for o in obs}
        if len(cnts) != 1:
            raise RuntimeError("replay: observers disagree on their batch counter")
        cnt0 = cnts.pop()
        for _, q in self.quantizers:          # the launch writes scale / zero_point through raw pointers
            q._touch_qparams()
        table = ordered.contiguous()
        lib = _hip.load()
        _hip.check(lib.osq_replay_statistics(table.data_ptr(), int(n_batches), int(n_q), self.rules.data_ptr(), int(cnt0),
                                             1 if fresh else 0, self.min_ptrs.data_ptr(), self.max_ptrs.data_ptr(),
                                             self.qmin.data_ptr(), self.qmax.data_ptr(), self.sym.data_ptr(),
                                             self.scale_ptrs.data_ptr(), self.zp_ptrs.data_ptr(), self.zp_types.data_ptr(),
                                             _hip.raw_stream(self.device)), "replay_statistics")
        for o in obs:
            if hasattr(o, "cnt"):
                object.__setattr__(o, "cnt", cnt0 + n_batches)


def replay(ordered, quantizers, fresh=False, plan=None):
    """Apply the gathered per-batch statistics in batch order with each observer's own rule, then
    refresh scale / zero_point.  ``ordered``: [n_batches, Q, 2] on the device.  One launch."""
    if ordered.shape[1] == 0:
        return plan
    if plan is None or not plan.valid():
        plan = ReplayPlan(quantizers, ordered.device)
    plan.run(ordered, fresh)
    return plan


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
