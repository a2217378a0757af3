"""Weight fake-quant of a whole model: one launch, and none at all while nothing changed.

The reference fake-quantises every weight inside every forward (``F.linear(input, self.weight_fake_quant(self.weight))``,
quantized_module.py:71-72, 97-100): 77 launches per BERT-base forward, 110 M parameters re-quantised although a frozen
model's weights and (scale, zero_point) never change.  Here a weight-quantized operator keeps its fake-quantised weight
as long as

  * the weight tensor is the same storage at the same version (``_version`` moves on every in-place torch op;
    kernels of this package that write weights through raw pointers -- gamma_fold_ -- bump ``ops.weight_epoch``),
  * scale / zero_point are the same storage at the same version, and the quantizer's ``_qparam_epoch`` (moved by every
    call that lets a kernel write them: observation, parameter repair, statistics replay) is unchanged,
  * observer off, fake-quant on, no gradient wanted;

What the key cannot see: in-place edits through ``.data`` (``w.weight.data *= g`` as in the reference's
gamma_migration.py:71, ``p.data.copy_(...)``) -- torch gives ``.data`` its own version counter, so ``weight._version`` does
not move.  In the reference's flows such edits happen before any quantized forward (FP targets and the migration run with
the quantizers off, nothing is cached yet); code that edits weights this way AFTER a quantized forward calls
``invalidate(model)`` (or sets ``enabled = False``).  This package's own ``gamma_migration`` bumps ``ops.weight_epoch``.

``prepare_weights(model)`` refreshes every stale entry of a model in ONE launch (``osq_fake_quant_weights_multi``)
over a pointer table that is itself kept while the pointers stay the same.  Same arithmetic as the per-module launch:
bit-identical weights, hence bit-identical logits.
"""
import weakref

import torch

from .. import _hip, ops
from .fake_quant import QuantizeBase

stats = {"hits": 0, "module_launches": 0, "multi_launches": 0, "multi_tensors": 0}
enabled = True        # False: every forward fake-quantises every weight in its own launch, as the reference does

# Kept beside the modules, not inside them: nothing here is state of the model (state_dict, deepcopy and pickling see
# the reference's attributes only), and entries die with their module.
_CACHE = weakref.WeakKeyDictionary()     # operator -> (key, fake-quantised weight)
_OWNER = weakref.WeakKeyDictionary()     # operator -> weakref(model it was adopted by)
_PLAN = weakref.WeakKeyDictionary()      # model -> (pointer key, device table, row ends, output views, total rows)


def _key(op, fq):
    w, s, z = op.weight, fq.scale, fq.zero_point
    return (w.data_ptr(), w._version, tuple(w.shape), s.data_ptr(), s._version, z.data_ptr(), z._version,
            fq.__dict__.get("_qparam_epoch", 0), ops.weight_epoch, fq.quant_min, fq.quant_max, fq.ch_axis)


def cacheable(op, fq):
    """Frozen quantizer, inference: the only state in which the result can be kept."""
    if fq.fake_quant_enabled != 1 or fq.observer_enabled == 1 or not op.weight.is_cuda:
        return False
    if torch.is_grad_enabled() and (op.weight.requires_grad or fq.scale.requires_grad or
                                    (fq.zero_point.is_floating_point() and fq.zero_point.requires_grad)):
        return False
    return True


def quantized_weight(op):
    """What ``op.weight_fake_quant(op.weight)`` returns, from the cache when it is still valid."""
    fq = op.weight_fake_quant
    if not enabled or not cacheable(op, fq):
        return fq(op.weight)
    entry = _CACHE.get(op)
    if entry is not None and entry[0] == _key(op, fq):
        stats["hits"] += 1
        return entry[1]
    owner = _OWNER.get(op)
    owner = owner() if owner is not None else None
    if owner is not None and prepare_weights(owner):      # a stale entry: refresh the whole model's in one launch
        entry = _CACHE.get(op)
        if entry is not None and entry[0] == _key(op, fq):
            return entry[1]
    y = fq(op.weight)                 # may repair the learnable parameters in place: take the key afterwards
    stats["module_launches"] += 1
    _CACHE[op] = (_key(op, fq), y)
    return y


def adopt(model):
    """Tell every weight-quantized operator of ``model`` which model it belongs to, so that the first stale entry met in
    a forward refreshes all of them in one launch (without it every operator refreshes its own)."""
    ref = weakref.ref(model)
    for m in model.modules():
        if "weight_fake_quant" in m.__dict__.get("_modules", {}):
            _OWNER[m] = ref
    return model


def invalidate(model):
    for m in model.modules():
        _CACHE.pop(m, None)
    _PLAN.pop(model, None)


def _table_entry(op, fq):
    """Layout rules of fq_weights_multi_kernel: rows of the channel axis 0 (or one scale for all rows), inner % 4 == 0,
    aligned, parameters already repaired (a learnable quantizer's first call goes through its own launch)."""
    w = op.weight
    if not w.is_contiguous() or w.dtype != torch.float32 or w.dim() < 2 or w.data_ptr() % 16:
        return None
    rows = w.shape[0]
    inner = w.numel() // rows
    if inner % 4 or fq.ch_axis not in (0, -1) or fq.param_mode != ops.PARAM_FIXED:
        return None
    channels = rows if fq.ch_axis == 0 else 1
    if fq.scale.numel() != channels or fq.zero_point.numel() != channels:
        return None
    return rows, channels, inner


def prepare_weights(model):
    """Refresh every stale cached weight of ``model`` in one launch.  Returns the number of tensors refreshed.
    Operators the table cannot hold (odd row length, learnable weight quantizers, ...) are left to their own launch."""
    lib = _hip.load()
    todo = []
    for m in model.modules():
        fq = m.__dict__.get("_modules", {}).get("weight_fake_quant")
        if fq is None or not isinstance(fq, QuantizeBase) or not hasattr(m, "weight") or not cacheable(m, fq):
            continue
        entry = _CACHE.get(m)
        if entry is not None and entry[0] == _key(m, fq):
            continue
        geo = _table_entry(m, fq)
        if geo is not None:
            todo.append((m, fq, geo))
    if not todo:
        return 0
    dev = todo[0][0].weight.device
    todo = [t for t in todo if t[0].weight.device == dev]
    plan_key = tuple((m.weight.data_ptr(), fq.scale.data_ptr(), fq.zero_point.data_ptr(), geo) for m, fq, geo in todo)
    plan = _PLAN.get(model)
    if plan is None or plan[0] != plan_key:
        total = sum(m.weight.numel() for m, _, _ in todo)
        out = torch.empty(total, dtype=torch.float32, device=dev)      # one allocation, one slice per weight
        descs = (_hip.WeightDesc * len(todo))()
        row_end, views, off, rows_so_far = [], [], 0, 0
        for i, (m, fq, (rows, channels, inner)) in enumerate(todo):
            y = out[off:off + m.weight.numel()].view(m.weight.shape)
            off += m.weight.numel()
            rows_so_far += rows
            row_end.append(rows_so_far)
            views.append(y)
            d = descs[i]
            d.x, d.y, d.scale, d.zero_point = m.weight.data_ptr(), y.data_ptr(), fq.scale.data_ptr(), fq.zero_point.data_ptr()
            d.rows, d.channels, d.inner = rows, channels, inner
            d.zp_type, d.mode, d.grad_factor = ops._zp_type(fq.zero_point), ops.PARAM_FIXED, 1.0
            d.quant_min, d.quant_max = int(fq.quant_min), int(fq.quant_max)
        raw = bytes(descs)
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        ends = torch.tensor(row_end, dtype=torch.int64, device=dev)
        plan = (plan_key, table, ends, views, rows_so_far)
        _PLAN[model] = plan
    _, table, ends, views, total_rows = plan
    _hip.check(lib.osq_fake_quant_weights_multi(table.data_ptr(), ends.data_ptr(), len(todo), total_rows,
                                                _hip.stream_ptr(dev)), "fake_quant_weights_multi")
    for (m, fq, _), y in zip(todo, views):
        _CACHE[m] = (_key(m, fq), y)
    stats["multi_launches"] += 1
    stats["multi_tensors"] += len(todo)
    return len(todo)
