#!/usr/bin/env python3
"""Full-size golden: the REFERENCE's twc_fine_gamma pipeline (BASELINE configs[1]) on a seeded, randomly initialised
BERT-base (12 layers, hidden 768, 98 quantizers), CPU, build container only (imports /root/reference unmodified
through the shims of make_golden_model.py).

The weights are NOT stored (440 MB): the fixture holds the seeds, per-tensor checksums of the initialised state
dict (the GPU test re-creates the model from the same seed with the same torch build and checks them), the seeded
calibration batches, and what the reference produced: FP logits, per-candidate losses of find_ratio, the chosen
percentile, every quantizer's scale / zero_point after the coarse and after the fine stage, and the logits of the
quantized model.  ~6 minutes on 8 cores.
"""
import copy
import logging
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_model as M          # noqa: E402  (shims, Cfg, patch_hf_bert, quantizer_table)

SEED_MODEL, SEED_LN, SEED_DATA = 20260930, 17, 4242
B, T, NB = 32, 128, 4
ITERS, STEP = 10, 0.02
PROBE_RATIO = 0.9
LR, EPOCHS = 1e-4, 1


def build_fp():
    """Seeded BERT-base with LayerNorm gammas that carry outlier dimensions (paper Fig. 1).  Shared with the GPU test."""
    from transformers import BertConfig, BertForSequenceClassification
    torch.manual_seed(SEED_MODEL)
    cfg = BertConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    fp = BertForSequenceClassification(cfg).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(SEED_LN)
        for m in fp.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(768, generator=g) * 1.2 + 0.4)
                m.weight[[5, 308, 381]] = torch.tensor([4.0, 6.0, 3.0])
                m.bias.copy_(torch.randn(768, generator=g) * 0.2)
        for m in fp.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(2.0)
    return fp


def build_batches():
    g = torch.Generator().manual_seed(SEED_DATA)
    batches = []
    for b in range(NB):
        ids = torch.randint(1000, 30000, (B, T), generator=g)
        L = torch.randint(8, T + 1, (B,), generator=g)
        L[b % B] = T
        mask = (torch.arange(T)[None, :] < L[:, None]).long()
        batches.append({"input_ids": ids * mask, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)})
    return batches


def checksums(fp):
    return {k: np.array([v.double().sum().item(), v.double().abs().sum().item()]) for k, v in fp.state_dict().items()
            if v.dtype.is_floating_point}


def main():
    QB, GM, TWC, ST, QuantizeBase = M.import_reference()
    torch.set_num_threads(8)
    t0 = time.time()
    fp = M.patch_hf_bert(build_fp())
    out = {f"sum::{k}": v for k, v in checksums(fp).items()}
    batches = build_batches()
    out["input_ids"] = np.stack([b["input_ids"].numpy() for b in batches])
    out["attention_mask"] = np.stack([b["attention_mask"].numpy() for b in batches])
    out["seeds"] = np.array([SEED_MODEL, SEED_LN, SEED_DATA])
    out["twc_grid"] = np.array([ITERS, STEP])
    out["learn"] = np.array([LR, EPOCHS])

    a_q = M.Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = M.Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    model = QB.QuantizedBertForSequenceClassification(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic",
                                                      is_remove_padding=True).eval()
    del fp
    with torch.no_grad():
        fp_output = [model(**b)[0].detach() for b in batches]
    out["logits_wrapped_fp"] = np.stack([o.numpy() for o in fp_output])
    print("fp outputs", time.time() - t0, flush=True)
    model = GM.delay_ln(model, M.Cfg(a_qconfig=a_q, w_qconfig=w_q), M.Cfg(model_type="bert", task_type="glue"))
    with torch.no_grad():
        out["logits_after_gamma"] = np.stack([model(**b)[0].numpy() for b in batches])
    ST.enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    with torch.no_grad():
        model(**batches[0])
    ST.disable_all(model)
    ST.set_observer_name(model)
    print("weights calibrated", time.time() - t0, flush=True)

    TWC.task_type, TWC.model_type = "glue", "bert"
    losses = []

    class Grab(logging.Handler):
        def emit(self, record):
            msg = record.getMessage()
            if msg.startswith("the ratio is"):
                losses.append(float(msg.split("the loss is")[1]))
                print(msg, time.time() - t0, flush=True)
    h = Grab()
    TWC.logger.addHandler(h)
    TWC.logger.setLevel(logging.INFO)
    trainer = types.SimpleNamespace(model=model)
    TWC.find_ratio(trainer, batches, fp_output, {"iters": ITERS, "step": STEP})
    TWC.logger.removeHandler(h)
    out["twc_losses"] = np.array(losses, dtype=np.float64)
    names, scales, zps = M.quantizer_table(model, QuantizeBase)
    out["q_names"] = np.array(names)
    for i, (s, z) in enumerate(zip(scales, zps)):
        if s.size == 1:
            out[f"q_after_twc_scale::{i}"], out[f"q_after_twc_zp::{i}"] = s, z
        else:   # per-channel weights: keep the fixture small -- checksum + the first 8 channels
            out[f"q_after_twc_scale::{i}"] = np.concatenate([[s.sum(), np.abs(s).max()], s[:8]])
    out["best_ratio"] = np.array([m.observer.percentile for n, m in model.named_modules()
                                  if isinstance(m, QuantizeBase) and "act" in n][:1])
    # one more observer pass at a fixed percentile: statistics that do not depend on which candidate won the (noisy) search
    TWC.set_ratio(model, PROBE_RATIO)
    TWC.calibrate(model, batches)
    _, scales, zps = M.quantizer_table(model, QuantizeBase)
    for i, (s, z) in enumerate(zip(scales, zps)):
        if s.size == 1:
            out[f"q_at_probe_scale::{i}"], out[f"q_at_probe_zp::{i}"] = s, z
    out["probe_ratio"] = np.array([PROBE_RATIO])
    TWC.set_ratio(model, float(out["best_ratio"][0]))
    TWC.calibrate(model, batches)
    TWC.enable_quantization(model)
    with torch.no_grad():
        out["logits_act_quant"] = np.stack([model(**b)[0].numpy() for b in batches])
    print("coarse stage done", time.time() - t0, flush=True)

    TWC.learn_scale(trainer, batches, fp_output, {"lr": LR, "epoch": EPOCHS})
    names, scales, zps = M.quantizer_table(model, QuantizeBase)
    for i, (s, z) in enumerate(zip(scales, zps)):
        if s.size == 1:
            out[f"q_after_learn_scale::{i}"], out[f"q_after_learn_zp::{i}"] = s, z
    ST.enable_quantization(model)
    with torch.no_grad():
        out["logits_full_quant"] = np.stack([model(**b)[0].numpy() for b in batches])
    path = os.path.join(M.OUT, "bert_base_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(names), "quantizers; losses", losses, "best", out["best_ratio"],
          "in", time.time() - t0, "s")
    print("logit drift gamma:", np.abs(out["logits_after_gamma"] - out["logits_wrapped_fp"]).max(),
          "act-quant:", np.abs(out["logits_act_quant"] - out["logits_wrapped_fp"]).max(),
          "full:", np.abs(out["logits_full_quant"] - out["logits_wrapped_fp"]).max(),
          "fp scale:", np.abs(out["logits_wrapped_fp"]).max())


if __name__ == "__main__":
    main()
