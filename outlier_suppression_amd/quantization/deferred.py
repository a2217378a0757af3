"""Observer passes with the sites of a forward reduced together.

In an observer pass of the calibration (token_wise_clipping.py:12-19, 29-47: observers on, fake-quant off) a BERT-base
forward calls ~95 masked activation observers, each a few microseconds of kernel behind 20 us of host work and two
launches.  Nothing downstream depends on their results (fake-quant is off: the quantizer returns X itself), so inside

    with deferred_observation() as sites:
        for batch in fp_input:
            model(**batch)
            sites.flush()

the observers only RECORD their site (tensor, mask, sequence axis, prune flag, percentile, batch counter) and ``flush``
does the whole forward's statistics in a handful of launches: ONE ``osq_token_minmax_multi`` over the pointer table of all
recorded sites, then per geometry group ONE ``osq_token_range_finalize_batched`` and ONE ``osq_replay_statistics``
(update rule + calculate_qparams through per-site pointers).  Same kernels' arithmetic as the per-site path: every
min_val / max_val / scale / zero_point is bit-identical (tests/test_gpu_deferred.py).  The recorded tensors are kept alive
until the flush.  Sites the table cannot hold -- flat (unmasked, no sequence axis) sites, per-channel observers,
observers in capture / cache mode -- run immediately as before.
"""
import contextlib
import ctypes

import torch

from .. import _hip, ops
from . import observer as _observer


# tools / bench: a list here receives, per flushed forward, [(elements, nested search?, int32[4] device tensor of
# ops.msefast_tensor_stats: nfev, pairs kept, evaluations the loss memo answered, converged)] of its per-tensor searches
MSE_STATS_SINK = None


class DeferredSites:
    def __init__(self):
        self.sites = []
        self.mse = []
        self.launches = 0
        self.flushed_sites = 0

    def add_mse(self, obs, search, two_d, sink):
        """A per-tensor MSEFast search that has begun (ops.msefast_tensor_begin): its loss evaluations run at the flush,
        together with the other searches of the forward (up to 16 per persistent launch, 32 float4 slots per lane); its batch counter is the one
        of the call."""
        self.mse.append((obs, search, two_d, sink, obs._counter(), obs.update_rule))

    def _flush_mse(self):
        pending, self.mse = self.mse, []
        if not pending:
            return 0
        if ops.reference_sum_order("mse"):
            # strict sums: one launch per ROUND of loss evaluations of all the forward's searches (128 per table)
            fit = [it for it in pending if ops.msefast_ordered_fits(it[1])]
            groups = ops.msefast_ordered_groups([it[1] for it in fit], [it[2] for it in fit])
            ops.msefast_tensor_run_ordered_groups(groups)          # concurrent: one stream per group
            self.launches += len(groups)
            for it in pending:
                if not ops.msefast_ordered_fits(it[1]):              # beyond the ordered kernels' capacity (> 134 M elements)
                    ops.msefast_tensor_run(it[1], None, it[2])
                    self.launches += 1
            return self._commit_mse(pending)
        # greedy groups: at most max_sites searches and max_slots float4 slots per lane in a launch (16 and 32); what cannot
        # be resident runs alone
        max_slots, max_sites = ops.msefast_resident_limits()
        groups, cur, used = [], [], 0
        for item in pending:
            k = ops.msefast_resident_slots(item[1].elems)
            if k == 0:
                groups.append([item])
                continue
            if cur and (used + k > max_slots or len(cur) == max_sites):
                groups.append(cur)
                cur, used = [], 0
            cur.append(item)
            used += k
        if cur:
            groups.append(cur)
        for g in groups:
            if len(g) == 1 or not ops.msefast_tensor_run_group([it[1] for it in g]):
                for it in g:
                    ops.msefast_tensor_run(it[1], None, it[2])
            self.launches += 1
        return self._commit_mse(pending)

    def _commit_mse(self, pending):
        if MSE_STATS_SINK is not None:
            MSE_STATS_SINK.append([(int(search.elems), bool(two_d), ops.msefast_tensor_stats(search)) for _, search, two_d, _, _, _ in pending])
        for obs, search, two_d, sink, cnt, rule in pending:
            obs.last_nfev = ops.msefast_tensor_commit(search, rule, cnt, obs.min_val, obs.max_val, sink,
                                                      obs._ref_flags(obs.min_val.device))
        self.flushed_sites += len(pending)
        ops.check_persistent("deferred MSEFast searches")     # one synchronisation per flushed forward: the searches took milliseconds
        return len(pending)

    def add(self, obs, x, lengths, seq_pos, prune, sink):
        if sink is None or sink.scale is None or not x.is_cuda or x.dtype != torch.float32 or x.dim() not in (3, 4):
            return False
        if lengths is not None:
            if not lengths.is_cuda:
                return False
            if lengths.dtype != torch.int64:
                lengths = lengths.to(torch.int64)
        view = ops.token_view(x, seq_pos, None if lengths is None else lengths.numel())
        obs._home(x.device)
        self.sites.append((obs, x, lengths, view, bool(prune), float(getattr(obs, "percentile", 1.0)) if prune else 1.0,
                           sink, obs._counter(), obs.update_rule))
        object.__setattr__(obs, "_last_site", ("tokens", view.batch, view.tokens, lengths))
        return True

    def flush(self):
        n_mse = self._flush_mse()
        sites, self.sites = self.sites, []
        if not sites:
            return n_mse
        lib = _hip.load()
        dev = sites[0][1].device
        st = _hip.raw_stream(dev)
        # ---- groups that can share the re-threshold launch and the replay launch
        # (geometry, masked, percentile of the pruning sites, batch counter); sites that do not prune ignore the
        # percentile and join whichever group of their geometry comes first
        groups, loose = {}, []
        for i, s in enumerate(sites):
            obs, x, lengths, view, prune, pct, sink, cnt, rule = s
            if prune:
                groups.setdefault((view.batch, view.tokens, lengths is not None, pct, cnt), []).append(i)
            else:
                loose.append(i)
        for i in loose:
            obs, x, lengths, view, prune, pct, sink, cnt, rule = sites[i]
            home = next((k for k in groups if k[:3] == (view.batch, view.tokens, lengths is not None) and k[4] == cnt), None)
            groups.setdefault(home or (view.batch, view.tokens, lengths is not None, 1.0, cnt), []).append(i)
        descs = (_hip.SiteDesc * len(sites))()
        tok_end, total = [0] * len(sites), 0
        plan = []
        order = []
        for key, idx in groups.items():
            batch, tokens = key[0], key[1]
            slots = batch * tokens
            n = len(idx)
            tmin = torch.empty(n, 1, slots, dtype=torch.float32, device=dev)
            tmax = torch.empty(n, 1, slots, dtype=torch.float32, device=dev)
            plan.append((key, idx, tmin, tmax))
            for k, i in enumerate(idx):
                order.append((i, tmin[k, 0], tmax[k, 0]))
        for pos, (i, tmn, tmx) in enumerate(order):
            obs, x, lengths, view, prune, pct, sink, cnt, rule = sites[i]
            d = descs[pos]
            d.x, d.lengths, d.token_min, d.token_max = x.data_ptr(), _hip.ptr(lengths), tmn.data_ptr(), tmx.data_ptr()
            d.view = view
            d.vec = int(view.stride_inner == 1 and view.feat_inner % 4 == 0 and x.data_ptr() % 16 == 0 and view.stride_batch % 4 == 0
                        and view.stride_token % 4 == 0 and (view.feat_outer == 1 or view.stride_outer % 4 == 0))
            total += view.batch * view.tokens
            tok_end[pos] = total
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev, non_blocking=True)
        ends = torch.tensor(tok_end, dtype=torch.int64).to(dev, non_blocking=True)
        _hip.check(lib.osq_token_minmax_multi(table.data_ptr(), ends.data_ptr(), len(sites), total, st), "token_minmax_multi")
        self.launches += 1
        ws = _hip.workspace(dev)
        keep = [table, ends]
        for key, idx, tmin, tmax in plan:
            batch, tokens, masked, pct, cnt = key
            n = len(idx)
            group = [sites[i] for i in idx]
            if masked:
                first = group[0][2]
                if all(s[2] is first or (s[2].data_ptr() == first.data_ptr() and s[2].numel() == first.numel()) for s in group):
                    lengths = first.reshape(1, batch)                      # one mask for the whole forward: the usual case
                else:
                    lengths = torch.stack([s[2] for s in group]).reshape(n, 1, batch)
            else:
                lengths = None
            flags = torch.tensor([1 if s[4] else 0 for s in group], dtype=torch.int32).to(dev, non_blocking=True)
            cur = torch.empty(1, n, 2, dtype=torch.float32, device=dev)
            ops.token_range_finalize_batched(tmin, tmax, n, 1, batch, tokens, lengths, flags, pct, cur)
            i32 = lambda vals: torch.tensor(list(vals), dtype=torch.int32).to(dev, non_blocking=True)
            u64 = lambda vals: torch.tensor(list(vals), dtype=torch.int64).to(dev, non_blocking=True)
            rules = i32(s[8] for s in group)
            min_ptrs, max_ptrs = u64(s[0].min_val.data_ptr() for s in group), u64(s[0].max_val.data_ptr() for s in group)
            qmin, qmax = i32(s[0].quant_min for s in group), i32(s[0].quant_max for s in group)
            sym = i32(int(bool(s[0].symmetric)) for s in group)
            s_ptrs, z_ptrs = u64(s[6].scale.data_ptr() for s in group), u64(_hip.ptr(s[6].zero_point) or 0 for s in group)
            z_types = i32(ops._zp_type(s[6].zero_point) if s[6].zero_point is not None else 0 for s in group)
            _hip.check(lib.osq_replay_statistics(cur.data_ptr(), 1, n, rules.data_ptr(), int(cnt), 0, min_ptrs.data_ptr(),
                                                 max_ptrs.data_ptr(), qmin.data_ptr(), qmax.data_ptr(), sym.data_ptr(),
                                                 s_ptrs.data_ptr(), z_ptrs.data_ptr(), z_types.data_ptr(), st), "replay_statistics")
            self.launches += 2
            keep += [tmin, tmax, flags, cur, rules, min_ptrs, max_ptrs, qmin, qmax, sym, s_ptrs, z_ptrs, z_types, lengths]
        self.flushed_sites += len(sites)
        self._keep = (keep, sites)        # until the next flush: the launches above are asynchronous
        return len(sites)


@contextlib.contextmanager
def deferred_observation():
    """Observers reached inside the block record their site instead of launching; ``flush()`` (call it after every
    forward: the running statistics advance once per batch) reduces what was recorded.  Leaving the block flushes."""
    sites = DeferredSites()
    previous = _observer.DEFERRED
    _observer.DEFERRED = sites
    try:
        yield sites
        sites.flush()
    finally:
        _observer.DEFERRED = previous
        sites.sites = []
        sites.mse = []
