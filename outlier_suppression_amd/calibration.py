"""Sharded calibration: batches split over ranks, ONE small exchange of per-batch statistics -- and, for observers whose
batches cannot be separated (MSEFast and friends), SITES split over ranks with one exchange of their final state
(``calibrate_owned_sites``, at the end of this file).

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
    neither fill the capture slot nor fit the fp32 replay: they are sharded by SITE instead (calibrate_owned_sites)."""
    for name, q in quantizers:
        obs = q.observer
        if not getattr(obs, "supports_capture", False) or obs.min_val.dtype != torch.float32 or obs.ch_axis != -1:
            raise NotImplementedError(
                f"{what}: {type(obs).__name__} at '{name}' cannot be recorded per batch and replayed "
                "(per-tensor MinMax / AvgMinMax / AvgPruneMinMax observers only); use calibration.calibrate_owned_sites, "
                "which deals the sites -- not the batches -- over the ranks")


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
    check_persistent_collectively("calibrate_sharded", group, cap.table.device)     # before the gather: no rank enters it alone
    ordered = gather_batch_table(cap.table, n_batches, group)
    replay(ordered, qs)
    return ordered


# ---------------------------------------------------------------------------------------------
# Observers that cannot be recorded per batch: shard the SITES instead of the batches
# ---------------------------------------------------------------------------------------------
#
# MSEFast / AvgMSEFast (observer.py:412-567), MSE / AvgMSE, AvgQuantile and LSQPlusObserver keep state that the next
# batch's arithmetic depends on -- a per-tensor MSEFast observer searches its SECOND batch on a float64 copy of x if and
# only if its first search left a float64 min_val behind (observer.py:481,494,524), which for the nested 2-D search
# depends on where the first batch's optimum fell.  Dealing the batches of ONE such observer to different ranks would
# need that decision before the batch that makes it has been searched.  What is independent is the SITE: with
# fake-quant off (state.py:18-19) no observer reads another one's result.  So every rank runs every forward (7 ms of a
# 270 ms observer pass of RoBERTa-base: the searches are the work), but only the observers it OWNS observe; each owned
# observer sees all batches in order, exactly as in one process -- bit-identical by construction -- and one all-gather
# of the final (min_val, max_val, scale, zero_point, counters, dtype flags) hands every rank every site's result.
# Per-channel weight observers (MSEFast: one bounded-Brent search per row, observer.py:496-517) are sites like any
# other: their [C] statistics travel in the same table.

_SIDE_CODE = {None: -1.0, "no": 0.0, "pos": 1.0, "neg": 2.0}
_SIDE_NAME = {v: k for k, v in _SIDE_CODE.items()}
_META = 8          # per site: cnt, one_side code, ref-float64 flags (2), statistics dtype (0 fp32 / 1 float64), nfev, has-flags, 1 spare


def _site_cost(name, q, numel, channels):
    """Relative cost of observing one site once (only the balance of the deal depends on it, never a result)."""
    kind = type(q.observer).__name__
    if "MSEFast" in kind:
        if channels > 1:
            return numel * 15.0                                        # ~15 loss evaluations per row
        nested = not q.observer.symmetric and "attention_probs" not in name     # softmax outputs are one-sided: 1-D search
        return numel * (350.0 if nested else 20.0)
    if "MSE" in kind:
        return numel * (200.0 if not q.observer.symmetric else 4.0)
    return float(numel)


def probe_sites(model, batch, forward, quantizers):
    """One forward with EVERY quantizer of the model switched off -- observers and fake-quant of the selected sites and of
    all others (a weight observer left enabled would otherwise see batch 0 twice, and the one-process pass, which never
    probes, would no longer be what the ranks reproduce) -- returning (numel, channels) of the tensor each selected
    quantizer is called with (0, 1 for quantizers the forward does not reach).  Weight quantizers are called with their
    operator's weight.  Every flag is restored afterwards."""
    from .quantization.fake_quant import QuantizeBase
    seen = {}
    handles = []
    for i, (_, q) in enumerate(quantizers):
        def hook(mod, args, kwargs, i=i):
            x = args[0] if args else kwargs.get("X")
            if x is not None and i not in seen:
                seen[i] = (x.numel(), 1 if mod.ch_axis == -1 else x.shape[mod.ch_axis])
        handles.append(q.register_forward_pre_hook(hook, with_kwargs=True))
    every = [m for m in model.modules() if isinstance(m, QuantizeBase)]
    every += [q for _, q in quantizers if not any(q is m for m in every)]
    saved = [(m.observer_enabled, m.fake_quant_enabled) for m in every]
    for m in every:
        m.observer_enabled, m.fake_quant_enabled = 0, 0
    try:
        with torch.no_grad():
            forward(model, batch)
    finally:
        for h in handles:
            h.remove()
        for m, (o, f) in zip(every, saved):
            m.observer_enabled, m.fake_quant_enabled = o, f
    return [seen.get(i, (0, 1)) for i in range(len(quantizers))]


def check_persistent_collectively(where, group=None, device=None):
    """ops.check_persistent for code that is about to enter (or has just left) a collective: a time-out of a persistent
    launch on ONE rank must not leave the others waiting in the gather, or carrying on with that rank's NaN-poisoned
    rows.  Every rank catches its own time-out, the group agrees on MAX(failed), and every rank raises."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    failure = None
    try:
        ops.check_persistent(where)
    except ops.PersistentLaunchTimeout as exc:
        if world == 1:
            raise
        failure = exc
    if world == 1:
        return
    on_host = device is None or dist.get_backend(group) == "gloo"      # gloo (CPU tests, ranks sharing a GPU): a host flag
    flag = torch.tensor([1.0 if failure is not None else 0.0], dtype=torch.float32, device="cpu" if on_host else device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if failure is not None:
        raise failure
    if flag.item() != 0.0:
        raise ops.PersistentLaunchTimeout(f"outlier_suppression_amd ({where}): a persistent launch timed out on another rank of "
                                          "the group; this rank's copy of the exchanged statistics is invalid as well.")


def deal_sites(costs, world):
    """Owner rank of every site: longest-processing-time-first onto the least loaded rank (ties: lowest rank, lowest
    index) -- the same deal on every rank."""
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += costs[i]
    return owner


def _pack_site(q, channels, out):
    """Final state of an owned quantizer as float64 numbers (fp32 / int32 / float64 values all convert exactly)."""
    obs = q.observer
    c = channels
    flags = obs.__dict__.get("_ref_f64")
    nfev = getattr(obs, "last_nfev", None)
    meta = torch.zeros(_META, dtype=torch.float64, device=out.device)
    meta[0] = float(getattr(obs, "cnt", 0))
    meta[1] = _SIDE_CODE[getattr(obs, "one_side_dist", None)]
    if flags is not None:
        meta[2:4] = flags.to(torch.float64)
        meta[6] = 1.0
    meta[4] = 1.0 if obs.min_val.dtype == torch.float64 else 0.0
    if nfev is not None:
        meta[5] = nfev.to(torch.float64).sum()
    out[:_META] = meta
    body = out[_META:]
    for k, t in enumerate((obs.min_val, obs.max_val, q.scale.detach(), q.zero_point.detach())):
        flat = t.reshape(-1).to(device=out.device, dtype=torch.float64)
        if flat.numel() == c:
            body[k * c:(k + 1) * c] = flat
        elif flat.numel() == 1:                    # never observed (a site no forward reached): the start values
            body[k * c:(k + 1) * c] = flat
        else:
            raise RuntimeError(f"site state has {flat.numel()} entries, expected {c}")


def _unpack_site(q, channels, row, device):
    """Install an owner's final state in this rank's copy of the quantizer (same dtypes and shapes as the owner's)."""
    obs = q.observer
    c = channels
    meta = row[:_META].cpu()
    body = row[_META:]
    stat_dtype = torch.float64 if meta[4].item() == 1.0 else torch.float32
    shape = (c,) if (obs.ch_axis != -1 or obs.min_val.dim()) else ()
    obs.min_val = body[0:c].to(stat_dtype).reshape(shape).clone()
    obs.max_val = body[c:2 * c].to(stat_dtype).reshape(shape).clone()
    scale, zero_point = q._qparam_storage(device, c)
    q._touch_qparams()
    scale.copy_(body[2 * c:3 * c].to(scale.dtype))
    zero_point.copy_(body[3 * c:4 * c].to(zero_point.dtype))
    if hasattr(obs, "cnt"):
        object.__setattr__(obs, "cnt", int(meta[0].item()))
    if hasattr(obs, "one_side_dist"):
        obs.one_side_dist = _SIDE_NAME[meta[1].item()]
    if hasattr(obs, "_ref_flags") and meta[6].item() == 1.0:
        obs._ref_flags(device).copy_(meta[2:4].to(torch.int32))
        object.__setattr__(obs, "_min_f64_known", bool(meta[2].item()))


@torch.no_grad()
def exchange_site_states(quantizers, channels, owner, rank, world, device, group=None):
    """The one exchange of a site-sharded pass: every quantizer's final state travels from its owner to everybody
    (one all_gather_into_tensor of a float64 table: 8 meta numbers + 4 x channels values per site; RCCL gathers the
    device tensor, a gloo group is served through the host).  Returns the seconds spent."""
    import time
    sizes = [_META + 4 * ch for ch in channels]
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + sz)
    table = torch.zeros(offs[-1], dtype=torch.float64, device=device)
    for i, ((_, q), r) in enumerate(zip(quantizers, owner)):
        if r == rank:
            _pack_site(q, channels[i], table[offs[i]:offs[i + 1]])
    if table.is_cuda:
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    staged = table.is_cuda and dist.get_backend(group) == "gloo"
    src = table.cpu() if staged else table
    gathered = torch.empty(world * offs[-1], dtype=torch.float64, device=src.device)
    dist.all_gather_into_tensor(gathered, src, group=group)
    gathered = gathered.to(device).view(world, offs[-1])
    if table.is_cuda:
        torch.cuda.synchronize(device)
    seconds = time.perf_counter() - t0
    for i, ((_, q), r) in enumerate(zip(quantizers, owner)):
        if r != rank:
            _unpack_site(q, channels[i], gathered[r, offs[i]:offs[i + 1]], device)
    return seconds


@torch.no_grad()
def calibrate_owned_sites(model, batches, forward, group=None, select=lambda name: "act" in name, defer=True):
    """Observer pass with the SITES dealt over the ranks of ``group`` (see the block comment above): for observers whose
    batches cannot be separated -- MSEFast / AvgMSEFast / MSE / AvgMSE / AvgQuantile / LSQPlusObserver, per-tensor or
    per-channel.  ``batches``: ALL calibration batches, the same on every rank, in order; ``forward(model, batch)`` runs
    the model.  Observers of the selected quantizers are expected on, fake-quant off.  Ends with every rank holding every
    selected quantizer's final statistics, scale and zero_point, bit-identical to a one-process pass.

    Returns {"owner": [...], "collective_s": seconds spent in the exchange}."""
    import contextlib
    import time
    from .quantization.deferred import deferred_observation
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    qs = [(n, q) for n, q in act_quantizers(model, select) if q.observer_enabled == 1]
    if not batches or not qs:
        return {"owner": [], "collective_s": 0.0}
    dev = next(model.parameters()).device
    if world == 1:
        owner = [0] * len(qs)
        geo = None
    else:
        geo = probe_sites(model, batches[0], forward, qs)
        owner = deal_sites([_site_cost(n, q, numel, ch) for (n, q), (numel, ch) in zip(qs, geo)], world)
    for (_, q), r in zip(qs, owner):
        if r != rank:
            q.observer_enabled = 0
    try:
        with (deferred_observation() if defer else contextlib.nullcontext()) as sites:
            for batch in batches:
                forward(model, batch)
                if sites is not None:
                    sites.flush()
    finally:
        for _, q in qs:
            q.observer_enabled = 1
    check_persistent_collectively("calibrate_owned_sites", group, dev)
    if world == 1:
        return {"owner": owner, "collective_s": 0.0}
    collective_s = exchange_site_states(qs, [ch for _, ch in geo], owner, rank, world, dev, group)
    return {"owner": owner, "collective_s": collective_s}
