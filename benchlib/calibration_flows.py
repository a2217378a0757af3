"""256-sample calibration wall-clock of the five BASELINE configs (Metric 2) and the frozen-model forward."""
import os
import time

import torch

from .common import SHORT, _ops_order


class CollectiveClock:
    """Seconds a calibration flow spends inside collectives at N > 1, per phase: a synchronised host-side bracket round
    every calibration.gather_batch_table / torch.distributed.all_reduce / all_gather_into_tensor the package issues while
    the clock is installed (the bracket's own synchronisations are part of what is reported: the exchange is latency-
    bound, a few KB per call).  N = 1: nothing is patched and every figure is 0.0."""

    def __init__(self, world):
        self.world, self.total, self.calls, self._last, self.phases = world, 0.0, 0, 0.0, {}
        self._saved = []

    def _wrap(self, fn):
        def timed(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            self.total += time.perf_counter() - t
            self.calls += 1
            return r
        return timed

    def __enter__(self):
        if self.world > 1:
            import torch.distributed as dist
            from outlier_suppression_amd import calibration
            for mod, name in ((calibration, "gather_batch_table"), (dist, "all_reduce"), (dist, "all_gather_into_tensor")):
                self._saved.append((mod, name, getattr(mod, name)))
                setattr(mod, name, self._wrap(getattr(mod, name)))
        return self

    def __exit__(self, *exc):
        for mod, name, fn in self._saved:
            setattr(mod, name, fn)
        self._saved = []

    def mark(self, phase):
        """Close a phase: what the collectives took since the previous mark."""
        self.phases[phase] = round(self.phases.get(phase, 0.0) + self.total - self._last, 4)
        self._last = self.total

    def report(self):
        return {"collective_s": round(self.total, 4), "collective_calls": self.calls, "collective_phases_s": dict(self.phases)}


def quantizer_exchange_check(model, world, share, dev):
    """After a sharded calibration every rank must hold the same scale / zero_point bits for every quantizer (SURVEY 8e:
    the gathered tables are replayed in global batch order on every rank).  Gathers a checksum of all of them."""
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    acc, n = 0, 0
    for _, m in model.named_modules():
        if isinstance(m, QuantizeBase) and getattr(m, "scale", None) is not None:
            for t in (m.scale, m.zero_point):
                if t is None:
                    continue
                tt = t.detach().reshape(-1)
                bits = tt.view(torch.int32) if tt.dtype in (torch.float32, torch.int32) else tt.to(torch.float32).view(torch.int32)
                acc = (acc * 1000003 + int(bits.to(torch.int64).sum().item())) % (1 << 61)
                n += tt.numel()
    if world == 1:
        return {"ranks": 1, "parameters_compared": n, "same_bits_on_every_rank": True}
    import torch.distributed as dist
    mine = torch.tensor([acc], dtype=torch.int64, device="cpu" if share else dev)
    every = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(every, mine)
    same = bool((every == mine).all().item())
    if not same:
        raise SystemExit(f"bench.py: ranks ended a sharded calibration with different quantizer parameters: checksums {every.tolist()}")
    return {"ranks": world, "parameters_compared": n, "same_bits_on_every_rank": same}



def calibration_wall_clock(dev, rank, world, search="cached"):
    """BASELINE configs[1]: BERT-base (random init, HF default config), CoLA-shaped calibration set
    (256 samples = 8 batches of [32, 128], synthetic ids / lengths), twc_fine_gamma W6A6:
    gamma migration -> weight calibration -> token-wise-clipping grid (30 candidates, step 0.01) ->
    LSQ+ learn-scale (3 epochs, lr 1e-5).  Clock: batches resident on device -> every quantizer has
    its final scale / zero_point.  N > 1: the grid search is sharded (batch b on rank b mod N, one
    all-gather of statistics and one of losses per candidate); learn-scale is sequential Adam: every step
    is split inside the batch (32/N samples per rank, gradients averaged by one small all-reduce;
    DESIGN.md section 6)."""
    import logging
    from types import SimpleNamespace as NS
    import torch.distributed as dist
    from transformers import BertConfig, BertForSequenceClassification
    from outlier_suppression_amd import calibration, token_wise_clipping as TWC
    from outlier_suppression_amd.gamma_migration import delay_ln
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, disable_all
    from outlier_suppression_amd.quantization.state import set_observer_name

    logging.getLogger("transformer").setLevel(logging.WARNING)
    torch.manual_seed(0)
    cfg = BertConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    fp = BertForSequenceClassification(cfg).eval().to(dev)
    g = torch.Generator().manual_seed(42)
    n_batches, B, T = (2 if SHORT else 8), 32, 128
    batches = []
    for _ in range(n_batches):
        L = torch.randint(8, T + 1, (B,), generator=g)
        mask = (torch.arange(T)[None, :] < L[:, None]).long()
        ids = torch.randint(1000, 30000, (B, T), generator=g) * mask
        batches.append({"input_ids": ids.to(dev), "attention_mask": mask.to(dev),
                        "token_type_ids": torch.zeros_like(ids).to(dev)})
    a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    TWC.task_type, TWC.model_type = "glue", "bert"
    mine = calibration.shard_batches(n_batches, rank, world)
    model = None

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    share = os.environ.get("OSQ_BENCH_SHARE_GPU") in ("1", "check")
    clock = None

    def run(search, strict_learn=False):
        nonlocal model, clock
        model = quantize_model(fp, w_q, a_q).to(dev)      # deep copy of the FP model, as quant_model.py:44-48: fp stays pristine
        phases = {}
        with CollectiveClock(world) as clock:
            return _run(search, strict_learn, phases)

    def _run(search, strict_learn, phases):
        nonlocal model
        sync()
        t_start = t0 = time.perf_counter()
        with torch.no_grad():
            if world > 1:    # FP targets: each rank runs its own batches, the [batches, 32, 2] logits are all-gathered
                rows = (n_batches + world - 1) // world
                mine_out = [model(**batches[b])[0].detach() for b in mine]
                local = torch.zeros(rows, *mine_out[0].shape, device=dev)
                for j, o in enumerate(mine_out):
                    local[j] = o
                fp_output = list(calibration.gather_batch_table(local, n_batches).unbind(0))
            else:
                fp_output = [model(**b)[0].detach() for b in batches]
        sync(); phases["fp_outputs"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("fp_outputs")
        m = delay_ln(model, NS(a_qconfig=a_q, w_qconfig=w_q), NS(model_type="bert", task_type="glue"))
        sync(); phases["gamma_migration"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("gamma_migration")
        enable_calibration_woquantization(m, quantizer_type="weight_fake_quant")
        with torch.no_grad():
            m(**batches[0])
        disable_all(m)
        set_observer_name(m)
        sync(); phases["weight_calibration"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("weight_calibration")
        grid = {"iters": 3 if SHORT else 30, "step": 0.01}       # cac_step_iters(6 bit, bs 32, T 128), token_wise_clipping.py:118-129
        if search == "cached":
            ratio = TWC.find_ratio_cached(NS(model=m), [batches[b] for b in mine], [fp_output[b] for b in mine], grid,
                                          n_batches=n_batches)
        else:
            ratio = TWC.find_ratio(NS(model=m), batches, fp_output, grid)
        sync(); phases["twc_grid_search"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("twc_grid_search")
        # N > 1: every Adam step is split inside the batch (32/N samples per rank, averaged gradients)
        (TWC.learn_scale if strict_learn else TWC.learn_scale_sharded)(NS(model=m), batches, fp_output, {"lr": 1e-5, "epoch": 1 if SHORT else 3})
        sync(); phases["learn_scale"] = time.perf_counter() - t0; clock.mark("learn_scale")
        model = m
        return time.perf_counter() - t_start, phases, ratio

    if SHORT:
        wall, phases, ratio = run(search)
        return {"config": "configs[1] SHORT (profiling only)", "wall_s": round(wall, 3), "phases_s": {k: round(v, 3) for k, v in phases.items()}}

    # The whole calibration runs twice on fresh copies of the model: the first pass also pays the process's one-time
    # costs (rocBLAS / hipBLASLt kernel loading and heuristics for forward and backward shapes, allocator growth,
    # first RCCL collectives) and is reported separately; the second is the steady-state wall-clock.
    first_wall, first_phases, _ = run(search)
    wall, phases, ratio = run(search)
    out = {"config": "configs[1]: BERT-base CoLA twc_fine_gamma W6A6, 256 samples (8 x [32,128]), random-init weights, synthetic ids",
           "wall_s": round(wall, 3), "phases_s": {k: round(v, 3) for k, v in phases.items()}, "best_percentile": ratio,
           "first_run_wall_s": round(first_wall, 3), "first_run_phases_s": {k: round(v, 3) for k, v in first_phases.items()},
           "twc_candidates": 30, **clock.report(), "exchange_check": quantizer_exchange_check(model, world, share, dev),
           "search": ("cached per-token extrema + 1 re-threshold launch per candidate, sharded over ranks" if search == "cached"
                      else "literal reference order: 2 model passes per candidate"),
           "learn_scale": ("sequential Adam, one process" if world == 1 else
                           f"sequential Adam, every step data-parallel inside the batch ({B // world} samples per rank, "
                           "one all-reduce of the 196 gradients per step)" if B % world == 0 else "replicated on every rank"),
           "n_gpus": world}
    if world > 1:
        # bit-for-bit parity with the sequential reference needs learn-scale replicated on every rank; the line
        # above ran the rounding-close data-parallel variant, this is the strict one on the same warm process
        strict_wall, strict_phases, _ = run(search, strict_learn=True)
        out["strict_replicated_learn_scale"] = {"wall_s": round(strict_wall, 3), "learn_scale_s": round(strict_phases["learn_scale"], 3)}
    return out


def calibration_plain(dev, rank, world):
    """BASELINE configs[0]: BERT-base CoLA PTQ with the plain MinMax flow (exp/bert_ptq/minmax/cola/config.yaml: W6 per-channel
    MinMaxObserver, A6 AvgMinMaxObserver + FixedFakeQuantize, no token-wise clipping, no gamma migration), 256 samples = 8 x
    [32,128], through ptq.run (ptq_glue_quant.py:228-251).  N > 1: the observer pass is sharded (calibration.calibrate_sharded:
    batch b on rank b mod N, one all-gather of the per-batch statistics, replay in batch order -- bit-identical)."""
    import logging
    from types import SimpleNamespace as NS
    import torch.distributed as dist
    import transformers as T
    from outlier_suppression_amd import calibration, ptq
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization
    logging.getLogger("transformer").setLevel(logging.WARNING)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(100)
    fp = T.BertForSequenceClassification(T.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
    batches = []
    for _ in range(8):
        L = torch.randint(8, 129, (32,), generator=g)
        mask = (torch.arange(128)[None, :] < L[:, None]).long()
        ids = torch.randint(1000, 29000, (32, 128), generator=g) * mask
        batches.append({"input_ids": ids.to(dev), "attention_mask": mask.to(dev), "token_type_ids": torch.zeros_like(ids).to(dev)})
    section = ptq.SHIPPED_QUANT_SECTIONS["minmax"]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    res = {}
    for rep in range(2):                       # the second run is the steady state (allocator, library handles)
        model = quantize_model(fp, section.w_qconfig, section.a_qconfig).to(dev)
        clock = CollectiveClock(world).__enter__()
        sync()
        t0 = time.perf_counter()
        if world == 1:
            with torch.no_grad():
                fp_in, fp_out = ptq.prepare_input_output(model, batches)
                t1 = time.perf_counter()
                model = ptq.run(model, fp_in, fp_out, section, NS(model_type="bert", task_type="glue"))
        else:
            with torch.no_grad():
                t1 = time.perf_counter()
                enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
                model(**batches[0])
                enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
                mine = calibration.shard_batches(len(batches), rank, world)
                calibration.calibrate_sharded(model, [batches[b] for b in mine], lambda m, b: m(**b), n_batches=len(batches))
                enable_quantization(model)
        sync()
        res = {"wall_s": round(time.perf_counter() - t0, 4), "fp_outputs_s": round(t1 - t0, 4)}
        clock.mark("observer_pass")
        res.update(clock.report())
        clock.__exit__()
        res["exchange_check"] = quantizer_exchange_check(model, world, os.environ.get("OSQ_BENCH_SHARE_GPU") in ("1", "check"), dev)
    return {"config": "configs[0]: BERT-base CoLA PTQ, plain MinMax flow W6A6 (exp/bert_ptq/minmax), 256 samples (8 x [32,128]), "
                      "random-init weights, synthetic ids", **res, "n_gpus": world,
            "observer_pass": "every site of a forward reduced together (quantization/deferred.py)" if world == 1 else
                             "sharded over ranks, one all-gather of the per-batch statistics"}


def calibration_extra(dev, rank, world, which):
    """Calibration wall-clock (SURVEY.md section 8d Metric 2) of BASELINE configs[2], [3], [4] at the reference's sizes,
    random-init weights and synthetic ids; clock: batches resident on device -> every quantizer has its final
    scale / zero_point.  One run each (the process is warm from configs[1]); phases as section 8d lists them.

      2  BERT-base SQuAD-v1 twc_fine_gamma W6A6: T = 384, 256 features = 8 x [32, 384], 90 candidates (step 0.0033:
         cac_step_iters(6 bit, bs 32, T 384), token_wise_clipping.py:118-129), masked two-headed loss, learn-scale at
         batch 8 with re-prepared targets (ptq_qa_quant.py:235-277);
      3  RoBERTa-base MNLI W4A6: weights 4-bit symmetric per output channel with MSEFastObserver (one bounded-Brent search
         per row: 134 K rows, observer.py:496-517), activations 6-bit AvgMSEFastObserver, 8 x [32, 128];
      4  BART XSum twc_fine_gamma W6A6, encoder + decoder: bart-base dimensions as the reference's shipped config uses
         (exp/xsum/twc_fine_gamma/config.yaml:44; BASELINE names bart-large), 64 x ([4, 1024] source, [4, 62] target),
         30 candidates, learn-scale 3 epochs (ptq_summ_quant.py:124-154).
    N > 1: the grid search is sharded (batch b on rank b mod N); the statistics / loss tables are all-gathered per candidate
    (calibration.gather_batch_table); learn-scale runs data-parallel inside each batch when the batch divides over the ranks
    (TWC.learn_scale_sharded: configs 2 at batch 8; config 4's batches of 4 on 8 ranks: one sample on each of the first four); config 3's MSEFast observers keep
    state that the next batch's arithmetic depends on, so there the SITES are dealt over the ranks (calibration.calibrate_owned_sites)."""
    import logging
    from types import SimpleNamespace as NS
    import torch.distributed as dist
    import transformers as T
    from outlier_suppression_amd import calibration, token_wise_clipping as TWC
    from outlier_suppression_amd.gamma_migration import delay_ln
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, disable_all
    from outlier_suppression_amd.quantization.state import set_observer_name

    logging.getLogger("transformer").setLevel(logging.WARNING)
    torch.manual_seed(which)
    g = torch.Generator().manual_seed(100 + which)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    share = os.environ.get("OSQ_BENCH_SHARE_GPU") in ("1", "check")
    twc_a = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    twc_w = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    phases, out = {}, {}

    def masked_batches(n_batches, B, Tn, vocab, lo):
        res = []
        for _ in range(n_batches):
            L = torch.randint(lo, Tn + 1, (B,), generator=g)
            mask = (torch.arange(Tn)[None, :] < L[:, None]).long()
            ids = torch.randint(1000, vocab - 1000, (B, Tn), generator=g) * mask + (1 - mask)
            res.append({"input_ids": ids.to(dev), "attention_mask": mask.to(dev)})
        return res

    if which == 2:
        cfg = T.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        fp = T.BertForQuestionAnswering(cfg).eval().to(dev)
        batches = masked_batches(2 if SHORT else 8, 32, 384, 30522, 64)
        for b in batches:
            b["token_type_ids"] = torch.zeros_like(b["input_ids"])
        # OSQ_BENCH_SQUAD_CANDIDATES: test hook (tests/test_gpu_sharded.py runs eight ranks on one GPU); the measured config has 90
        task, mtype, grid = "squad", "bert", {"iters": int(os.environ.get("OSQ_BENCH_SQUAD_CANDIDATES", "3" if SHORT else "90")), "step": 0.0033}
        out["config"] = "configs[2]: BERT-base SQuAD-v1 twc_fine_gamma W6A6, 256 features (8 x [32,384]), 90 candidates, learn-scale at batch 8"
    elif which in (4, 5):
        # 4: bart-LARGE dimensions, what BASELINE.json's configs[4] names; 5: bart-base dimensions, what the reference's shipped
        # config points at (exp/xsum/twc_fine_gamma/config.yaml:44) -- both run by default
        d_model, layers, heads, ffn = (768, 6, 12, 3072) if which == 5 else (1024, 12, 16, 4096)
        cfg = T.BartConfig(d_model=d_model, encoder_layers=layers, decoder_layers=layers, encoder_attention_heads=heads,
                           decoder_attention_heads=heads, encoder_ffn_dim=ffn, decoder_ffn_dim=ffn, max_position_embeddings=1024,
                           dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
        fp = T.BartForConditionalGeneration(cfg).eval().to(dev)
        batches = masked_batches(4 if SHORT else 64, 4, 1024, 50265, 256)
        for b in batches:
            DL = torch.randint(16, 63, (4,), generator=g)
            DL[0] = 62
            dm = (torch.arange(62)[None, :] < DL[:, None]).long()
            b["decoder_input_ids"] = (torch.randint(1000, 49000, (4, 62), generator=g) * dm + (1 - dm)).to(dev)
            b["decoder_attention_mask"] = dm.to(dev)
        task, mtype, grid = "summ", "bart", {"iters": 3 if SHORT else 30, "step": 0.01}
        out["config"] = ("configs[4]: BART XSum twc_fine_gamma W6A6 encoder+decoder, " +
                         ("bart-base dimensions (the reference's shipped config)" if which == 5 else
                          "bart-LARGE dimensions (d 1024, 16 heads, 12 + 12 layers, ffn 4096: what BASELINE.json names)") +
                         ", 256 samples (64 x ([4,1024] source, [4,62] target)), 30 candidates, learn-scale 3 epochs")
    else:
        cfg = T.RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, num_labels=3,
                              hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        fp = T.RobertaForSequenceClassification(cfg).eval().to(dev)
        batches = masked_batches(2 if SHORT else 8, 32, 128, 50265, 8)
        w_q = NS(quantizer="FixedFakeQuantize", observer="MSEFastObserver", bit=4, symmetric=True, ch_axis=0)
        a_q = NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1)
        from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
        def run_once():
            model = quantize_model(fp, w_q, a_q).to(dev)
            phases = {}
            sync()
            t_start = t0 = time.perf_counter()
            fwd = lambda m, b: m(**b)
            # SITES are dealt over the ranks (calibration.calibrate_owned_sites): every rank runs every forward, an observer
            # is searched by its owner only -- all its batches in order, bit-identical to one process -- and one all-gather
            # hands every rank every site's final statistics / scale / zero_point
            enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
            info_w = calibration.calibrate_owned_sites(model, batches[:1], fwd, select=lambda n: "weight_fake_quant" in n)
            sync(); phases["weight_calibration_msefast_per_channel"] = time.perf_counter() - t0; t0 = time.perf_counter()
            enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
            info_a = calibration.calibrate_owned_sites(model, batches, fwd)
            sync(); phases["activation_calibration_msefast_per_tensor"] = time.perf_counter() - t0
            return time.perf_counter() - t_start, phases, info_w, info_a, model

        if SHORT:
            wall, phases, info_w, info_a, model = run_once()
            return {"config": "configs[3] SHORT (profiling only)", "wall_s": round(wall, 3), "phases_s": {k: round(v, 3) for k, v in phases.items()}}
        first_wall = run_once()[0]    # the second run is the steady state (code objects loaded, allocator grown, communicator built)
        wall, phases, info_w, info_a, model = run_once()
        # The default adds every per-tensor loss in the reference's one-thread order (outlier_suppression_amd.set_strict, ON by
        # default): rounds of one launch per loss evaluation of ALL the forward's searches.  The order-free tier
        # (set_strict(False): exact sums, one resident launch per group of searches) beside it: its wall-clock and how far
        # its results are from the default's.
        if os.environ.get("OSQ_BENCH_NO_STRICT") == "1":      # profiling runs of the DEFAULT flow (tools/collect_calibration_profiles.sh)
            return {"config": "configs[3] (default flow only)", "wall_s": round(wall, 3), "phases_s": {k: round(v, 3) for k, v in phases.items()}}
        import outlier_suppression_amd as osq
        prev_width = _ops_order()                             # 8 / 16: the tier this process runs in (0: OSQ_STRICT=0, then this run repeats it)
        osq.set_strict(False)
        try:
            free_wall, free_phases, _, _, free_model = run_once()
        finally:
            osq.set_strict(bool(prev_width), prev_width or 8)
        order_free = {"wall_s": round(free_wall, 3), "phases_s": {k: round(v, 3) for k, v in free_phases.items()},
                      "what": "set_strict(False): MSEFast losses as exact (order-free) sums, searches resident in one persistent launch per "
                              "group of sites; the default above adds them in ATen's one-thread order (bit-equal to the reference run on a "
                              "one-thread host, tests/test_gpu_strict_order.py); per-channel rows follow that order in either tier"}
        # how far the two tiers' results are apart: relative difference of every activation quantizer's scale
        d = [abs(a.scale.item() - b.scale.item()) / abs(b.scale.item())
             for (_, a), (_, b) in zip([(n, m) for n, m in free_model.named_modules() if isinstance(m, QuantizeBase) and "act" in n],
                                       [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n])]
        d.sort()
        order_free["activation_scale_rel_diff_vs_default"] = {"median": d[len(d) // 2], "max": d[-1], "equal": sum(1 for v in d if v == 0.0), "sites": len(d)}
        del free_model
        mine_w = [q for (n, q), r in zip([(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "weight_fake_quant" in n],
                                         info_w["owner"] or [0] * 10 ** 6) if r == rank]
        rows = sum(int(q.observer.min_val.numel()) for q in mine_w)
        evals = sum(int(q.observer.last_nfev.sum().item()) for q in mine_w if q.observer.last_nfev is not None)
        act_q = [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
        mine_a = [q for (n, q), r in zip(act_q, info_a["owner"] or [0] * 10 ** 6) if r == rank]
        act_evals = sum(int(q.observer.last_nfev.sum().item()) for q in mine_a if q.observer.last_nfev is not None)
        return {"config": "configs[3]: RoBERTa-base MNLI W4A6, per-channel weights + MSEFast, 256 samples (8 x [32,128])",
                "wall_s": round(wall, 3), "first_run_wall_s": round(first_wall, 3), "phases_s": {k: round(v, 3) for k, v in phases.items()},
                "collective_s": round(info_w["collective_s"] + info_a["collective_s"], 4),
                "collective_phases_s": {"weight_calibration_msefast_per_channel": round(info_w["collective_s"], 4),
                                        "activation_calibration_msefast_per_tensor": round(info_a["collective_s"], 4)},
                "exchange_check": quantizer_exchange_check(model, world, share, dev), "order_free": order_free,
                "weight_rows_searched_on_rank0": rows, "weight_loss_evaluations_on_rank0": evals,
                "activation_sites": len(act_q), "activation_sites_on_rank0": len(mine_a),
                "activation_loss_evaluations_last_batch_on_rank0": act_evals, "n_gpus": world,
                "sharding": ("one process" if world == 1 else
                             f"sites dealt over {world} ranks (calibration.calibrate_owned_sites: every rank runs every forward, each "
                             "observer is searched by its owner over all batches in order; one all-gather of the final states)")}

    TWC.task_type, TWC.model_type = task, mtype
    n_batches = len(batches)
    mine = calibration.shard_batches(n_batches, rank, world)
    model = quantize_model(fp, twc_w, twc_a).to(dev)
    clock = CollectiveClock(world).__enter__()
    try:
        def targets(bs):
            res = []
            with torch.no_grad():
                for b in bs:
                    o = model(**b)
                    if task == "squad":
                        keep = b["attention_mask"] == 1
                        res.append([o[0][keep].detach(), o[1][keep].detach()])
                    else:
                        res.append(o[0][b["decoder_attention_mask"] == 1, :].detach())
            return res
        sync()
        t_start = t0 = time.perf_counter()
        fp_output = targets(batches)           # every rank: FP targets of all batches (cheap next to the search)
        sync(); phases["fp_outputs"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("fp_outputs")
        m = delay_ln(model, NS(a_qconfig=twc_a, w_qconfig=twc_w), NS(model_type=mtype, task_type=task))
        sync(); phases["gamma_migration"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("gamma_migration")
        enable_calibration_woquantization(m, quantizer_type="weight_fake_quant")
        with torch.no_grad():
            m(**batches[0])
        disable_all(m)
        set_observer_name(m)
        sync(); phases["weight_calibration"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("weight_calibration")
        ratio = TWC.find_ratio_cached(NS(model=m), [batches[b] for b in mine], [fp_output[b] for b in mine], grid, n_batches=n_batches)
        sync(); phases["twc_grid_search"] = time.perf_counter() - t0; t0 = time.perf_counter(); clock.mark("twc_grid_search")
        if which == 2:        # ptq_qa_quant.py:262-267: smaller batches for the fine stage, targets recomputed with everything off
            disable_all(m)
            model = m
            small = []
            for b in batches:
                for i in range(0, 32, 8):
                    small.append({k: v[i:i + 8] for k, v in b.items()})
            learn_in, learn_out = small, targets(small)
        else:
            learn_in, learn_out = batches, fp_output
        TWC.learn_scale_sharded(NS(model=m), learn_in, learn_out, {"lr": 1e-5, "epoch": 1 if SHORT else 3})
        sync(); phases["learn_scale"] = time.perf_counter() - t0; clock.mark("learn_scale")
        wall = time.perf_counter() - t_start
        out.update({"wall_s": round(wall, 3), "phases_s": {k: round(v, 3) for k, v in phases.items()},
                    **clock.report(), "exchange_check": quantizer_exchange_check(m, world, share, dev),
                    "best_percentile": ratio, "twc_candidates": grid["iters"],
                    "search": "cached per-token extrema, one re-threshold launch per candidate and geometry group, sharded over ranks",
                    "learn_scale": ("sequential Adam, one process" if world == 1 else
                                    ("sequential Adam, every step data-parallel inside the batch (kept-token targets sliced per rank, gradients summed)"
                                     if all(next(iter(b.values())).shape[0] % world == 0 for b in learn_in)
                                     else ("sequential Adam, the batch's samples on the first ranks, one each, zero gradients from the others"
                                           if all(world % next(iter(b.values())).shape[0] == 0 for b in learn_in)
                                           else "sequential Adam, replicated on every rank (the batch does not divide over the ranks)"))),
                    "n_gpus": world})
        return out
    finally:
        clock.__exit__()
        TWC.task_type, TWC.model_type = "glue", "bert"


def quantized_forward_times(dev):
    """BERT-base, every weight and activation quantizer frozen and on (the PTQ evaluation state, ptq_glue_quant.py:251):
    one [32,128] forward with the weights fake-quantised per operator on every forward, as the reference does (77 launches,
    quantized_module.py:71-72,97-100), against the kept result (no launch) and the one-launch refresh."""
    from types import SimpleNamespace as NS
    from transformers import BertConfig, BertForSequenceClassification
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization, disable_all
    from outlier_suppression_amd.quantization import weight_cache as WC
    from outlier_suppression_amd import _hip
    torch.manual_seed(0)
    a_q = NS(quantizer="FixedFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    fp = BertForSequenceClassification(BertConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
    model = quantize_model(fp, w_q, a_q).to(dev)
    L = torch.randint(8, 129, (32,))
    mask = (torch.arange(128)[None, :] < L[:, None]).long()
    batch = {"input_ids": (torch.randint(1000, 30000, (32, 128)) * mask).to(dev), "attention_mask": mask.to(dev),
             "token_type_ids": torch.zeros(32, 128, dtype=torch.long, device=dev)}
    enable_calibration_woquantization(model)
    with torch.no_grad():
        model(**batch)
    disable_all(model)
    enable_quantization(model)

    def timed(n=20):
        with torch.no_grad():
            for _ in range(3):
                model(**batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                model(**batch)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    WC.enabled = False
    per_op = timed()
    WC.enabled = True
    kept = timed()
    for k in WC.stats:
        WC.stats[k] = 0
    WC.invalidate(model)
    torch.cuda.synchronize()
    with torch.no_grad():                  # the frozen-model state: nothing wants a gradient
        WC.prepare_weights(model)          # builds the pointer table (once per model)
        WC.invalidate(model)               # drops the results AND the table ...
        WC.prepare_weights(model)
        for m in model.modules():          # ... so stale results only: the table stays
            WC._CACHE.pop(m, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = WC.prepare_weights(model)
        torch.cuda.synchronize()
    refresh = (time.perf_counter() - t0) * 1e3
    # the launch alone (the wall-clock above is mostly the host walking 77 modules and comparing their keys)
    _, w_table, w_ends, w_views, w_rows = WC._PLAN[model]
    w_lib, w_ts = _hip.load(), []
    for _ in range(9):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        _hip.check(w_lib.osq_fake_quant_weights_multi(w_table.data_ptr(), w_ends.data_ptr(), len(w_views), w_rows, _hip.stream_ptr(dev)), "weights_multi")
        eb.record()
        torch.cuda.synchronize()
        w_ts.append(ea.elapsed_time(eb) * 1e3)
    w_us = sorted(w_ts)[len(w_ts) // 2]
    w_bytes = 8 * sum(v.numel() for v in w_views)
    # observer pass (token_wise_clipping.py:12-19, 29-47: observers on, fake-quant off) of one [32,128] batch: every
    # masked site its own two launches, against the sites of the forward recorded and reduced together
    from outlier_suppression_amd import token_wise_clipping as TWC
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    from outlier_suppression_amd.quantization.state import set_observer_name
    tw_a = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    obs_model = quantize_model(fp, w_q, tw_a).to(dev)
    set_observer_name(obs_model)
    TWC.set_ratio(obs_model, 0.95)

    def observer_pass(defer, n=20):
        info = {}
        with torch.no_grad():
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if defer:
                    with deferred_observation() as sites:
                        for _ in range(n):
                            obs_model(**batch)
                            sites.flush()
                    info = {"launches_per_forward": sites.launches / n, "sites_per_forward": sites.flushed_sites / n}
                else:
                    for _ in range(n):
                        obs_model(**batch)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
        return dt, info
    each_ms, _ = observer_pass(False)
    defer_ms, info = observer_pass(True)
    n_sites = info.get("sites_per_forward", 0) or 1
    return {"model": "BERT-base, W6A6, [32,128] batch, every quantizer frozen and on",
            "observer_pass_site_by_site_ms": round(each_ms, 3), "observer_launches_per_masked_site_then": 2,
            "observer_pass_deferred_ms": round(defer_ms, 3), "masked_sites_per_forward": n_sites,
            "observer_launches_per_masked_site_now": round(info.get("launches_per_forward", 0) / n_sites, 4),
            "weight_fake_quant_per_operator_every_forward_ms": round(per_op, 3), "weight_launches_per_forward_then": 77,
            "weights_kept_ms": round(kept, 3), "weight_launches_per_forward_now": 0,
            "one_launch_refresh_of_all_weights_ms": round(refresh, 3), "tensors_in_that_launch": n,
            "that_launch_us": round(w_us, 1), "that_launch_MB": round(w_bytes / 1e6, 1), "that_launch_frac_of_8TBps": round(w_bytes / w_us / 1e6 / 8.0, 3)}

