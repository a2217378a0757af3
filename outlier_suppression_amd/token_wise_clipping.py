"""Token-wise clipping driver: coarse percentile search + fine LSQ+ scale learning.

Reference: quant_transformer/solver/token_wise_clipping.py (same function names and order of
operations).  It is the direct caller of the HIP hot path: every ``calibrate`` pass runs every
activation quantizer once per batch.  ``find_ratio_cached`` is the MI355X-first variant of the
coarse search (same results, see its docstring); ``find_ratio`` is the literal one.
"""
import logging

import torch
from torch.nn import MSELoss

from .quantization.fake_quant import LSQFakeQuantize, LSQPlusFakeQuantize, QuantizeBase
from .quantization.state import disable_all

logger = logging.getLogger("transformer")
task_type = None
model_type = None
loss_fct = MSELoss()

a_bit_iters = {8: 0.1, 6: 0.3, 4: 0.9}


def _act_quantizers(model):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]


def set_ratio(model, ratio):
    """token_wise_clipping.py:12-19."""
    for _, q in _act_quantizers(model):
        q.observer.set_percentile(ratio)
        q.observer.cnt = 0
        q.disable_fake_quant()
        q.enable_observer()


def enable_quantization(model):
    """token_wise_clipping.py:22-26: activations quantized with frozen statistics, weights untouched."""
    for _, q in _act_quantizers(model):
        q.disable_observer()
        q.enable_fake_quant()


def batch_loss(outputs, batch, target, task=None):
    """token_wise_clipping.py:35-45 / 93-104: MSE against the FP model's outputs."""
    task = task or task_type
    if task == "glue":
        return loss_fct(outputs[0], target)
    if task in ("squad", "squad_v2"):
        keep = batch["attention_mask"] == 1
        return loss_fct(outputs[0][keep], target[0]) + loss_fct(outputs[1][keep], target[1])
    if task == "summ":
        return loss_fct(outputs[0][batch["decoder_attention_mask"] == 1, :], target)
    raise NotImplementedError(task)


DEFER_OBSERVATION = True     # observer passes record their sites and reduce them per forward (quantization/deferred.py)


def calibrate(model, fp_input, fp_output=None):
    """token_wise_clipping.py:29-47: forward over the cached calibration batches; optional loss sum.  Observers reached
    with their fake-quantizer off are reduced together after each forward (same statistics, a handful of launches per
    forward instead of two per site)."""
    from .quantization.deferred import deferred_observation
    import contextlib
    loss = 0
    with torch.no_grad(), (deferred_observation() if DEFER_OBSERVATION else contextlib.nullcontext()) as sites:
        for i, batch in enumerate(fp_input):
            outputs = model(**batch)
            if sites is not None:
                sites.flush()
            if fp_output is not None:
                loss += batch_loss(outputs, batch, fp_output[i])
    from . import ops
    ops.check_persistent("calibrate")       # free unless a persistent launch ran in this pass (both flags on, or searches)
    return loss


def find_ratio(trainer, fp_input, fp_output, param):
    """token_wise_clipping.py:50-66: grid over the percentile; first best wins (strict '>')."""
    model = trainer.model
    best, best_loss = 0, 10000000
    for i in range(param["iters"]):
        set_ratio(model, 1.0 - param["step"] * i)
        calibrate(model, fp_input)
        enable_quantization(model)
        cur = calibrate(model, fp_input, fp_output)
        logger.info("the ratio is {}, the loss is {}".format(1.0 - param["step"] * i, cur))
        if best_loss > cur:
            best_loss, best = cur, i
    ratio = 1.0 - param["step"] * best
    logger.info("the best percentile is {}".format(ratio))
    set_ratio(model, ratio)
    calibrate(model, fp_input)
    return ratio


class _only_these_learn:
    """While learning the quantizer parameters, every other parameter of the model stops requiring a gradient.
    The reference leaves them as they are (token_wise_clipping.py:81-86 only chooses what the optimiser sees), so
    its backward pass also computes -- and then never reads -- the weight and bias gradients of every Linear and
    LayerNorm: a third GEMM per layer plus a reduction per bias.  The gradients of scale / zero_point do not depend
    on them; skipping them takes about a third off a learn-scale step."""

    def __init__(self, model, learn):
        keep = {id(p) for p in learn}
        self.frozen = [p for p in model.parameters() if p.requires_grad and id(p) not in keep]

    def __enter__(self):
        for p in self.frozen:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.frozen:
            p.requires_grad_(True)


def learn_scale(trainer, fp_input, fp_output, config_quant_learn):
    """token_wise_clipping.py:72-108: Adam on (scale, zero_point) of every activation quantizer."""
    model = trainer.model
    disable_all(model)
    logger.info("*** begin learn the scale now! ***")
    params = []
    for _, q in _act_quantizers(model):
        q.enable_fake_quant()
        q.disable_observer()
        if isinstance(q, LSQPlusFakeQuantize):
            params += [q.scale, q.zero_point]
        elif isinstance(q, LSQFakeQuantize):
            params.append(q.scale)
    opt = torch.optim.Adam(params, lr=config_quant_learn["lr"])
    steps = config_quant_learn["epoch"] * len(fp_input)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=0.0)
    with _only_these_learn(model, params):
        for _ in range(config_quant_learn["epoch"]):
            for i, batch in enumerate(fp_input):
                opt.zero_grad()
                loss = batch_loss(model(**batch), batch, fp_output[i])
                loss.backward()
                opt.step()
                sched.step()
    from . import ops
    ops.check_persistent("learn_scale")


def learn_scale_sharded(trainer, fp_input, fp_output, config_quant_learn, group=None):
    """``learn_scale`` with every optimisation step split over the ranks of ``group`` (data parallel INSIDE
    each batch).  The reference's loop is sequential Adam, so the batches cannot be dealt out the way the
    observer passes are; instead every rank holds all batches and takes samples [r*B/W, (r+1)*B/W) of each:

      * loss: MSE is a mean, and with equal slices the full-batch loss is the average of the ranks'
        losses, so the full-batch gradient is the average of the local gradients -- ONE all-reduce of
        the flattened (scale, zero_point) gradients per step (2 x quantizers floats: latency-bound);
      * grad_factor = 1/sqrt(numel * quant_max) (fake_quant.py:195-204) must see the FULL tensor's numel:
        ``numel_multiplier`` on the quantizers restores it;
      * every rank then takes the same Adam / cosine step on identical parameters.

    Same mathematics as ``learn_scale``; summation order differs (per-rank partial sums), so the learned
    parameters agree to float rounding (~1e-6 relative), not bit for bit -- the single-GPU run itself is
    only within that of the reference.  The masked tasks (SQuAD's two heads over the attended tokens, summarisation over
    the valid decoder tokens) sum squared errors locally, divide by the full batch's element count and SUM the gradients.
    A batch smaller than the group (BART's batches of 4 on 8 ranks) gives its samples to the first ranks, one each; the
    others exchange zero gradients.  Falls back to the replicated loop for any other uneven split.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    sizes = {next(iter(b.values())).shape[0] for b in fp_input}
    # a batch either divides over the ranks, or it is smaller than the group and divides the group's size: then its B samples
    # go to the first B ranks, one each, and the others take part in the gradient exchange with zeros (BART: batches of 4 on 8 ranks)
    if world == 1 or task_type not in ("glue", "squad", "squad_v2", "summ") or len(sizes) != 1 or \
            any(sz % world and not (sz < world and world % sz == 0) for sz in sizes):
        return learn_scale(trainer, fp_input, fp_output, config_quant_learn)
    bsz = next(iter(sizes))
    per = bsz // world if bsz % world == 0 else 1
    lo, hi = min(rank * per, bsz), min((rank + 1) * per, bsz)
    idle = hi == lo
    summed = task_type != "glue" or bsz % world != 0      # local sums over the full batch's denominator, gradients SUMMED
    masked_task = task_type != "glue"
    # Masked tasks (token_wise_clipping.py:38-43): the targets are rows of the KEPT tokens of the whole batch and the MSE's
    # denominator is their number -- a rank's samples hold a data-dependent share of them.  Per batch, once, outside the
    # optimisation loop: where this rank's rows start, how many there are, and the full-batch denominator; the local loss
    # is then (local sum of squares) / (full-batch element count) and the full-batch gradient is the SUM over the ranks.
    spans = []
    if masked_task:
        key = "decoder_attention_mask" if task_type == "summ" else "attention_mask"
        for batch in fp_input:
            counts = (batch[key] == 1).sum(1).cpu()
            spans.append((int(counts[:lo].sum()), int(counts[lo:hi].sum()), int(counts.sum())))
    model = trainer.model
    disable_all(model)
    logger.info("*** begin learn the scale now! (intra-batch data parallel over %d ranks) ***", world)
    params, quantizers = [], []
    for _, q in _act_quantizers(model):
        q.enable_fake_quant()
        q.disable_observer()
        q.numel_multiplier = bsz // per             # full-batch numel = local numel x (samples of the batch / samples here)
        quantizers.append(q)
        if isinstance(q, LSQPlusFakeQuantize):
            params += [q.scale, q.zero_point]
        elif isinstance(q, LSQFakeQuantize):
            params.append(q.scale)
    opt = torch.optim.Adam(params, lr=config_quant_learn["lr"])
    steps = config_quant_learn["epoch"] * len(fp_input)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=0.0)
    shapes = [p.shape for p in params]
    counts = [p.numel() for p in params]
    staged = dist.get_backend(group) == "gloo"        # CPU tests / ranks sharing one GPU: stage through the host
    freeze = _only_these_learn(model, params)
    freeze.__enter__()
    try:
        for _ in range(config_quant_learn["epoch"]):
            for i, batch in enumerate(fp_input):
                local = {k: v[lo:hi] for k, v in batch.items()}
                opt.zero_grad()
                if idle:
                    loss = None
                    for q in quantizers:              # the repair every forward applies to the parameters (fake_quant.py:188-191)
                        q._sanitize()
                elif not masked_task:
                    if summed:
                        got = model(**local)[0]
                        loss = (got - fp_output[i][lo:hi]).square().sum() / fp_output[i].numel()
                    else:
                        loss = batch_loss(model(**local), local, fp_output[i][lo:hi])
                else:
                    off, n_loc, n_tot = spans[i]
                    out = model(**local)
                    if task_type == "summ":
                        got = out[0][local["decoder_attention_mask"] == 1, :]
                        loss = (got - fp_output[i][off:off + n_loc]).square().sum() / (n_tot * got.shape[-1])
                    else:
                        keep = local["attention_mask"] == 1
                        loss = ((out[0][keep] - fp_output[i][0][off:off + n_loc]).square().sum() +
                                (out[1][keep] - fp_output[i][1][off:off + n_loc]).square().sum()) / n_tot
                if loss is not None:
                    loss.backward()
                flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
                if staged:
                    host = flat.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                    flat = host.to(flat.device)
                else:
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
                if not summed:
                    flat /= world
                for p, g, shp in zip(params, torch.split(flat, counts), shapes):
                    p.grad = g.reshape(shp)
                opt.step()
                sched.step()
    finally:
        freeze.__exit__()
        for q in quantizers:
            q.numel_multiplier = 1


def cac_step_iters(a_bit, bs, config_data):
    """token_wise_clipping.py:118-129."""
    seq = config_data.max_seq_length if hasattr(config_data, "max_seq_length") else config_data.max_source_length
    step = min(float(format(128 * 32 * 0.01 / bs / seq, ".2g")), 0.01)
    iters = int(a_bit_iters[a_bit] / step)
    logger.info("the step is {}, the iters is {}".format(step, iters))
    return step, iters


def token_wise_clipping(trainer, fp_input, fp_output, config):
    """token_wise_clipping.py:132-146."""
    global model_type, task_type
    model_type, task_type = config.model.model_type, config.model.task_type
    logger.info("*** Evaluate Token Percentile ***")
    step, iters = cac_step_iters(config.quant.a_qconfig.bit, trainer.args.per_device_eval_batch_size, config.data)
    return find_ratio(trainer, fp_input, fp_output, {"iters": getattr(config.quant, "iters", iters),
                                                     "step": getattr(config.quant, "step", step)})


# ---------------------------------------------------------------------------------------------
# MI355X-first coarse search: per-token extrema cached once, one re-threshold launch per candidate,
# batches sharded over ranks with one small all-gather per candidate.
# ---------------------------------------------------------------------------------------------

def find_ratio_cached(trainer, fp_input, fp_output, param, n_batches=None, group=None):
    """Same result as ``find_ratio`` with half the model forwards.

    In the observer pass of every candidate the fake-quantizers are off (token_wise_clipping.py:12-19),
    so the activations -- and therefore each token's (min, max) at every quantizer -- do not depend on
    the candidate percentile.  Phase A runs the FP model ONCE over the calibration batches and keeps
    the per-token extrema of every (quantizer, batch) pair on the device (BERT-base, 8x32x128:
    98 x 8 x 2 x 4096 floats = 26 MB).  Per candidate: ONE launch re-thresholds all pairs
    (``osq_token_range_finalize_batched``), the [batches, quantizers, 2] table is replayed in batch
    order (running mean, observer.py:194-202) and the quantized forward pass measures the loss.

    Multi-GPU: ``fp_input`` / ``fp_output`` are THIS rank's batches (``calibration.shard_batches``
    order); the statistics table and the per-batch losses are all-gathered (RCCL) and consumed in
    global batch order on every rank, so every rank picks the same percentile and ends with
    bit-identical scales to the single-GPU run.
    """
    import numpy as np
    import torch.distributed as dist
    from . import calibration, ops

    model = trainer.model
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    n_local = len(fp_input)
    n_batches = n_batches if n_batches is not None else n_local * world
    rows = (n_batches + world - 1) // world
    qs = _act_quantizers(model)
    n_q = len(qs)
    dev = next(model.parameters()).device

    # ---- discover each site's kind with one captured forward (also yields batch 0's statistics)
    set_ratio(model, 1.0)
    flat_table = torch.zeros(rows, n_q, 2, dtype=torch.float32, device=dev)
    if n_local == 0:
        raise ValueError("find_ratio_cached: this rank has no calibration batch")
    for i, (_, q) in enumerate(qs):
        q.observer._capture = flat_table[0, i]
    with torch.no_grad():
        model(**fp_input[0])
    sites = [q.observer._last_site for _, q in qs]
    for _, q in qs:
        q.observer._capture = None
    tok_cols = [i for i, s in enumerate(sites) if s is not None and s[0] == "tokens"]
    # Token sites are grouped by what the re-threshold launch must share: (batch, tokens) and whether they are masked.
    # Every site keeps its OWN lengths (BART's cross-attention keys are masked with the decoder's lengths although they
    # have the encoder's geometry, quant_bart.py:167,172,472).
    groups = {}
    for i in tok_cols:
        groups.setdefault((sites[i][1], sites[i][2], sites[i][3] is not None), []).append(i)

    class _Group:
        pass
    tok_groups = []
    for (batch, tokens, masked), cols in groups.items():
        gq = _Group()
        gq.cols, gq.batch, gq.tokens, gq.masked, gq.n = cols, batch, tokens, masked, len(cols)
        slots = batch * tokens
        gq.tok_min = torch.empty(gq.n, rows, slots, dtype=torch.float32, device=dev)
        gq.tok_max = torch.empty(gq.n, rows, slots, dtype=torch.float32, device=dev)
        gq.lengths = torch.zeros(gq.n, rows, batch, dtype=torch.int64, device=dev) if masked else None
        gq.prune_flags = torch.tensor([0 if "attention_probs" in qs[i][1].observer.name else 1 for i in cols],
                                      dtype=torch.int32, device=dev)
        gq.col_index = torch.tensor(cols, device=dev)
        gq.cur = torch.zeros(rows, gq.n, 2, dtype=torch.float32, device=dev)
        tok_groups.append(gq)
    tok_set = set(tok_cols)

    def _release():
        for _, q in qs:
            q.observer._token_cache = None
            q.observer._capture = None

    # ---- phase A: one FP pass, per-token extrema of every (quantizer, batch) kept on the device.  Every batch must
    # show every site with the geometry of batch 0 (fixed-length padding); a batch that does not (dynamic padding, a
    # short last batch) sends the search down the literal path instead of reading rows it did not fill.
    same = True
    with torch.no_grad():
        for j, batch_in in enumerate(fp_input):
            for gq in tok_groups:
                for k, i in enumerate(gq.cols):
                    qs[i][1].observer._token_cache = (gq.tok_min[k, j], gq.tok_max[k, j])
                    object.__setattr__(qs[i][1].observer, "_last_site", None)
            for i, (_, q) in enumerate(qs):
                if i not in tok_set:
                    q.observer._capture = flat_table[j, i]
                    object.__setattr__(q.observer, "_last_site", None)
            try:
                model(**batch_in)
            except ValueError:                      # a site outgrew its cache rows (ops.token_minmax checks the capacity)
                same = False
                break
            for i, (_, q) in enumerate(qs):
                site = q.observer._last_site
                kind = None if site is None else site[0]
                if (i in tok_set) != (kind == "tokens") or (sites[i] is None) != (site is None):
                    same = False
            for gq in tok_groups:
                for k, i in enumerate(gq.cols):
                    site = qs[i][1].observer._last_site
                    if site is None or site[0] != "tokens" or (site[1], site[2], site[3] is not None) != (gq.batch, gq.tokens, gq.masked):
                        same = False
                    elif gq.masked:
                        gq.lengths[k, j].copy_(site[3])
            if not same:
                break
    _release()
    if not same:
        if world > 1:
            raise NotImplementedError("find_ratio_cached: the calibration batches do not share one geometry per site "
                                      "(dynamic padding?); pad to a fixed length or run find_ratio on one rank")
        logger.info("find_ratio_cached: site geometry changes between batches, using the literal search")
        return find_ratio(trainer, fp_input, fp_output, param)

    plan = calibration.ReplayPlan(qs, dev)

    def apply_ratio(ratio):
        """Statistics of every activation quantizer for this candidate (what calibrate() leaves behind)."""
        for _, q in qs:
            object.__setattr__(q.observer, "percentile", ratio)
        table = flat_table.clone()
        for gq in tok_groups:               # one launch per geometry group (one group for BERT / RoBERTa, three for BART)
            ops.token_range_finalize_batched(gq.tok_min, gq.tok_max, gq.n, rows, gq.batch, gq.tokens, gq.lengths, gq.prune_flags,
                                             ratio, gq.cur)
            table.index_copy_(1, gq.col_index, gq.cur)
        ordered = calibration.gather_batch_table(table, n_batches, group)
        plan.run(ordered, fresh=True)       # one launch: running means from scratch + qparams of all quantizers

    # ---- per candidate: re-threshold (1 launch) + replay (1 launch) + quantized forward for the loss.
    # Nothing in the loop reads a result back: the per-batch losses of every candidate stay on the device and
    # are fetched ONCE after the last candidate, so the host keeps enqueueing while the GPU works (a .cpu() per
    # candidate drained the pipeline 30 times).  The reference's log lines and its fp32 batch-order sum are
    # produced afterwards from the same numbers.
    loss_rows = torch.zeros(rows, 1, dtype=torch.float32, device=dev)
    loss_table = torch.zeros(param["iters"], n_batches, dtype=torch.float32, device=dev)
    for it in range(param["iters"]):
        ratio = 1.0 - param["step"] * it
        apply_ratio(ratio)
        for _, q in qs:                 # enable_quantization(model) without walking the module tree again
            q.disable_observer()
            q.enable_fake_quant()
        with torch.no_grad():
            for j, batch_in in enumerate(fp_input):
                loss_rows[j, 0] = batch_loss(model(**batch_in), batch_in, fp_output[j])
        loss_table[it] = calibration.gather_batch_table(loss_rows, n_batches, group).reshape(-1)
    best, best_loss = 0, 10000000
    for it, per_batch in enumerate(loss_table.cpu().numpy().astype(np.float32)):
        cur = np.float32(0)
        for v in per_batch:              # the reference adds fp32 losses in batch order
            cur = np.float32(cur + v)
        cur = float(cur)
        logger.info("the ratio is {}, the loss is {}".format(1.0 - param["step"] * it, cur))
        if best_loss > cur:
            best_loss, best = cur, it
    ratio = 1.0 - param["step"] * best
    logger.info("the best percentile is {}".format(ratio))
    apply_ratio(ratio)
    for _, q in qs:                 # state find_ratio leaves behind: observers on, fake-quant off
        q.disable_fake_quant()
        q.enable_observer()
    ops.check_persistent("find_ratio_cached")
    return ratio
