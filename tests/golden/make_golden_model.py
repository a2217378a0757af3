#!/usr/bin/env python3
"""Model-level golden vectors: run the REFERENCE's quantized BERT pipeline on a tiny random BERT.

Build-container only (imports /root/reference unmodified, CPU).  Mirrors the order of
quant_transformer/solver/ptq_glue_quant.py:212-253 with the reference's own functions:
quantize (QuantizedBertForSequenceClassification) -> prepare fp input/output -> delay_ln
(gamma migration) -> weight calibration -> set_observer_name -> token_wise_clipping.find_ratio
-> learn_scale -> enable_quantization -> logits.  Stores DATA ONLY: the seeded FP weights, the
calibration batches, and what the reference produced at every stage (logits, per-ratio losses,
every quantizer's scale / zero_point).

Process-local shims (no reference file is touched): seaborn stub, identity .cuda(), and the
transformers-4.18 names the reference imports that transformers 5.x moved or dropped
(SURVEY.md 8c).
"""
import copy
import logging
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("OSQ_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


class Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def import_reference():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    import transformers.file_utils as fu
    import transformers.utils as tu
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    for name in ("add_code_sample_docstrings", "add_start_docstrings", "add_start_docstrings_to_model_forward",
                 "replace_return_docstrings", "ModelOutput"):
        if not hasattr(fu, name):
            setattr(fu, name, getattr(tu, name, lambda *a, **k: (lambda f: f)))
    # transformers 4.18 semantics of the two ModuleUtilsMixin helpers the reference calls (quant_bert.py:557-564)
    mu.ModuleUtilsMixin.get_extended_attention_mask = \
        lambda self, attention_mask, input_shape, device=None: (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
    mu.ModuleUtilsMixin.get_head_mask = lambda self, head_mask, num_hidden_layers, *a, **k: [None] * num_hidden_layers
    sys.path.insert(0, REF)
    from quant_transformer.model import quant_bert
    from quant_transformer.solver import gamma_migration, token_wise_clipping
    from quant_transformer.quantization import state
    from quant_transformer.quantization.fake_quant import QuantizeBase
    return quant_bert, gamma_migration, token_wise_clipping, state, QuantizeBase


def patch_hf_bert(model):
    """Attributes the 4.18-era wrappers read from the HF modules and 5.x no longer sets."""
    model.bert.embeddings.position_embedding_type = "absolute"
    model.bert.encoder.gradient_checkpointing = False
    for layer in model.bert.encoder.layer:
        layer.attention.pruned_heads = set()
        layer.attention.self.position_embedding_type = "absolute"
        if not hasattr(layer, "chunk_size_feed_forward"):
            layer.chunk_size_feed_forward = 0
    return model


def quantizer_table(model, QuantizeBase):
    names, scales, zps = [], [], []
    for n, m in model.named_modules():
        if isinstance(m, QuantizeBase):
            names.append(n)
            scales.append(m.scale.detach().reshape(-1).to(torch.float64).numpy())
            zps.append(m.zero_point.detach().reshape(-1).to(torch.float64).numpy())
    return names, scales, zps


def tiny_bert_cls():
    """The seeded tiny BERT classifier and its calibration batches (shared by main and main_plain)."""
    from transformers import BertConfig, BertForSequenceClassification
    torch.set_num_threads(1)
    torch.manual_seed(20240929)
    cfg = BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=40, num_labels=2, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, type_vocab_size=2)
    fp = patch_hf_bert(BertForSequenceClassification(cfg).eval())
    with torch.no_grad():   # make LayerNorm gammas non-trivial, with outliers, as in real checkpoints (paper Fig. 1)
        g = torch.Generator().manual_seed(7)
        for m in fp.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(32, generator=g) * 1.2 + 0.4)
                m.weight[5] = 4.0
                m.bias.copy_(torch.randn(32, generator=g) * 0.2)
        for m in fp.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0)      # livelier activations than init std 0.02 gives
    out = {f"sd::{k}": v.numpy() for k, v in fp.state_dict().items()}

    B, T, NB = 4, 16, 4
    g = torch.Generator().manual_seed(99)
    batches = []
    for b in range(NB):
        ids = torch.randint(1, 120, (B, T), generator=g)
        L = torch.randint(3, T + 1, (B,), generator=g)
        L[b % B] = T
        mask = (torch.arange(T)[None, :] < L[:, None]).long()
        ids = ids * mask
        batches.append({"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)})
    out["input_ids"] = np.stack([b["input_ids"].numpy() for b in batches])
    out["attention_mask"] = np.stack([b["attention_mask"].numpy() for b in batches])

    with torch.no_grad():
        out["logits_hf_fp"] = np.stack([fp(**b).logits.numpy() for b in batches])
    return fp, batches, out


def main():
    QB, GM, TWC, ST, QuantizeBase = import_reference()
    fp, batches, out = tiny_bert_cls()

    a_q = Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    config_quant = Cfg(a_qconfig=a_q, w_qconfig=w_q)
    config_model = Cfg(model_type="bert", task_type="glue")

    model = QB.QuantizedBertForSequenceClassification(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic",
                                                      is_remove_padding=True).eval()
    with torch.no_grad():
        fp_output = [model(**b)[0].detach() for b in batches]          # prepare_input_output, ptq_glue_quant.py:94-107
    out["logits_wrapped_fp"] = np.stack([o.numpy() for o in fp_output])

    model = GM.delay_ln(model, config_quant, config_model)             # ptq_glue_quant.py:230-232
    with torch.no_grad():
        out["logits_after_gamma"] = np.stack([model(**b)[0].numpy() for b in batches])
    out["module_names"] = np.array([n for n, _ in model.named_modules()])

    ST.enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")    # :234-235
    with torch.no_grad():
        model(**batches[0])
    ST.disable_all(model)                                              # :237-238
    ST.set_observer_name(model)

    # token_wise_clipping.find_ratio (token_wise_clipping.py:50-66) with a short grid
    TWC.task_type, TWC.model_type = "glue", "bert"
    losses = []

    class Grab(logging.Handler):
        def emit(self, record):
            msg = record.getMessage()
            if msg.startswith("the ratio is"):
                losses.append(float(msg.split("the loss is")[1]))
    h = Grab()
    TWC.logger.addHandler(h)
    TWC.logger.setLevel(logging.INFO)
    trainer = types.SimpleNamespace(model=model)
    iters, step = 6, 0.05
    TWC.find_ratio(trainer, batches, fp_output, {"iters": iters, "step": step})
    TWC.logger.removeHandler(h)
    out["twc_losses"] = np.array(losses, dtype=np.float64)
    out["twc_grid"] = np.array([iters, step])
    names, scales, zps = quantizer_table(model, QuantizeBase)
    out["q_names"] = np.array(names)
    for i, (s, z) in enumerate(zip(scales, zps)):
        out[f"q_after_twc_scale::{i}"], out[f"q_after_twc_zp::{i}"] = s, z
    out["best_ratio"] = np.array([m.observer.percentile for n, m in model.named_modules()
                                  if isinstance(m, QuantizeBase) and "act" in n][:1])

    TWC.enable_quantization(model)                                     # activations quantized, weights FP
    with torch.no_grad():
        out["logits_act_quant"] = np.stack([model(**b)[0].numpy() for b in batches])

    TWC.learn_scale(trainer, batches, fp_output, {"lr": 1e-3, "epoch": 2})   # :72-108 (larger lr so the step is visible)
    names, scales, zps = quantizer_table(model, QuantizeBase)
    for i, (s, z) in enumerate(zip(scales, zps)):
        out[f"q_after_learn_scale::{i}"], out[f"q_after_learn_zp::{i}"] = s, z

    ST.enable_quantization(model)                                      # ptq_glue_quant.py:251: weights + activations
    with torch.no_grad():
        out["logits_full_quant"] = np.stack([model(**b)[0].numpy() for b in batches])

    path = os.path.join(OUT, "bert_tiny_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(names), "quantizers; losses", losses)
    print("logit drift gamma:", np.abs(out["logits_after_gamma"] - out["logits_wrapped_fp"]).max(),
          "act-quant:", np.abs(out["logits_act_quant"] - out["logits_wrapped_fp"]).max(),
          "full:", np.abs(out["logits_full_quant"] - out["logits_wrapped_fp"]).max())


def tiny_bart():
    """Seeded tiny BART in the 4.18-era module layout the reference wrappers expect (plain nn.Embedding token
    tables + embed_scale on the stacks)."""
    from torch import nn
    from transformers import BartConfig, BartForConditionalGeneration
    torch.manual_seed(20240930)
    cfg = BartConfig(vocab_size=120, d_model=32, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
                     decoder_attention_heads=2, encoder_ffn_dim=64, decoder_ffn_dim=64, max_position_embeddings=40,
                     dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, pad_token_id=1, bos_token_id=0,
                     eos_token_id=2, decoder_start_token_id=2)
    fp = BartForConditionalGeneration(cfg).eval()

    def plain(e):
        p = nn.Embedding(e.num_embeddings, e.embedding_dim, padding_idx=e.padding_idx)
        p.weight.data = e.weight.data.clone()
        return p
    fp.model.shared = plain(fp.model.shared)
    for m in (fp.model.encoder, fp.model.decoder):
        m.embed_tokens = plain(m.embed_tokens)
        m.embed_scale = 1.0
        m.gradient_checkpointing = False
    fp.model.encoder.max_source_positions = 40
    fp.model.decoder.max_target_positions = 40
    with torch.no_grad():
        g = torch.Generator().manual_seed(11)
        for m in fp.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(32, generator=g) * 1.2 + 0.4)
                m.weight[7] = 3.5
                m.bias.copy_(torch.randn(32, generator=g) * 0.2)
        for m in fp.modules():
            if isinstance(m, torch.nn.Linear) and m is not fp.lm_head:
                m.weight.mul_(4.0)
    return cfg, fp


def main_bart():
    """BART: wrap -> gamma migration -> weight calibration -> one observer pass at percentile 0.9 (set_ratio +
    calibrate, token_wise_clipping.py:12-47) -> activation quantisation.  Exercises the reference's BART quirks:
    3-D attention probabilities against a length-B mask, cross-attention keys masked with decoder lengths."""
    QB, GM, TWC, ST, QuantizeBase = import_reference()
    gu = types.ModuleType("transformers.generation_utils")
    from transformers.generation import GenerationMixin
    gu.GenerationMixin = GenerationMixin
    sys.modules["transformers.generation_utils"] = gu
    from quant_transformer.model import quant_bart as RB
    torch.set_num_threads(1)
    cfg, fp = tiny_bart()
    out = {f"sd::{k}": v.numpy() for k, v in fp.state_dict().items()}
    B, S, Td, NB = 3, 14, 6, 3
    g = torch.Generator().manual_seed(5)
    batches = []
    for b in range(NB):
        L = torch.randint(4, S + 1, (B,), generator=g)
        L[b % B] = S
        mask = (torch.arange(S)[None, :] < L[:, None]).long()
        ids = torch.randint(3, 120, (B, S), generator=g) * mask + (1 - mask)
        DL = torch.randint(2, Td + 1, (B,), generator=g)
        DL[(b + 1) % B] = Td
        dmask = (torch.arange(Td)[None, :] < DL[:, None]).long()
        dids = torch.randint(3, 120, (B, Td), generator=g) * dmask + (1 - dmask)
        batches.append({"input_ids": ids, "attention_mask": mask, "decoder_input_ids": dids, "decoder_attention_mask": dmask})
    for k in ("input_ids", "attention_mask", "decoder_input_ids", "decoder_attention_mask"):
        out[k] = np.stack([b[k].numpy() for b in batches])
    a_q = Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    model = RB.QuantizedBartForConditionalGeneration(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic",
                                                     is_remove_padding=True).eval()
    kw = dict(use_cache=False, return_dict=False)
    with torch.no_grad():
        out["logits_wrapped_fp"] = np.stack([model(**b, **kw)[0].numpy() for b in batches])
    model = GM.delay_ln(model, Cfg(a_qconfig=a_q, w_qconfig=w_q), Cfg(model_type="bart", task_type="summ"))
    with torch.no_grad():
        out["logits_after_gamma"] = np.stack([model(**b, **kw)[0].numpy() for b in batches])
    out["module_names"] = np.array([n for n, _ in model.named_modules()])
    ST.enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    with torch.no_grad():
        model(**batches[0], **kw)
    ST.disable_all(model)
    ST.set_observer_name(model)
    TWC.set_ratio(model, 0.9)
    with torch.no_grad():
        for b in batches:
            model(**b, **kw)
    names, scales, zps = quantizer_table(model, QuantizeBase)
    out["q_names"] = np.array(names)
    for i, (s, z) in enumerate(zip(scales, zps)):
        out[f"q_scale::{i}"], out[f"q_zp::{i}"] = s, z
    obs = [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
    out["act_min"] = np.array([float(m.observer.min_val) for _, m in obs])
    out["act_max"] = np.array([float(m.observer.max_val) for _, m in obs])
    TWC.enable_quantization(model)
    with torch.no_grad():
        out["logits_act_quant"] = np.stack([model(**b, **kw)[0].numpy() for b in batches])
    path = os.path.join(OUT, "bart_tiny_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(names), "quantizers; drift gamma",
          np.abs(out["logits_after_gamma"] - out["logits_wrapped_fp"]).max(), "act-quant",
          np.abs(out["logits_act_quant"] - out["logits_wrapped_fp"]).max())


def _patch_stack(model, attr):
    m = getattr(model, attr)
    m.embeddings.position_embedding_type = "absolute"
    m.encoder.gradient_checkpointing = False
    for layer in m.encoder.layer:
        layer.attention.pruned_heads = set()
        layer.attention.self.position_embedding_type = "absolute"
        if not hasattr(layer, "chunk_size_feed_forward"):
            layer.chunk_size_feed_forward = 0
    return model


def main_variant(kind):
    """The same pipeline for the two flows VERDICT r01 found unpinned:

    bert-qa      BERT with the SQuAD head (model/quant_bert.py:690): start / end logits, the masked two-headed MSE of
                 token_wise_clipping.py:38-41 / 95-99, FP targets already masked (ptq_qa_quant.py:126-131), and the
                 batches RE-PREPARED at a smaller batch size before learn_scale (ptq_qa_quant.py:262-267);
    roberta-cls  RoBERTa with its two-layer classification head (model/quant_roberta.py:621-643).
    """
    QB, GM, TWC, ST, QuantizeBase = import_reference()
    gu = types.ModuleType("transformers.generation_utils")
    sys.modules.setdefault("transformers.generation_utils", gu)
    from quant_transformer.model import quant_roberta as RQ
    import transformers as T
    torch.set_num_threads(1)
    torch.manual_seed({"bert-qa": 20241001, "roberta-cls": 20241002}[kind])
    common = dict(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                  max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, type_vocab_size=2)
    if kind == "bert-qa":
        fp = _patch_stack(T.BertForQuestionAnswering(T.BertConfig(**common)).eval(), "bert")
        ref_cls, attr, task = QB.QuantizedBertForQuestionAnswering, "bert", "squad"
    else:
        fp = _patch_stack(T.RobertaForSequenceClassification(T.RobertaConfig(num_labels=3, pad_token_id=1, **common)).eval(), "roberta")
        ref_cls, attr, task = RQ.QuantizedRobertaForSequenceClassification, "roberta", "glue"
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for m in fp.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(32, generator=g) * 1.2 + 0.4)
                m.weight[5] = 4.0
                m.bias.copy_(torch.randn(32, generator=g) * 0.2)
        for m in fp.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0)
    out = {f"sd::{k}": v.numpy() for k, v in fp.state_dict().items()}
    out["kind"] = np.array(kind)

    B, Tn, NB = 4, 16, 4
    g = torch.Generator().manual_seed(99)
    samples = []
    for b in range(NB * B):
        L = int(torch.randint(3, Tn + 1, (1,), generator=g))
        if b % 5 == 0:
            L = Tn
        ids = torch.randint(3, 120, (Tn,), generator=g)
        mask = (torch.arange(Tn) < L).long()
        samples.append((ids * mask + (1 - mask), mask))        # pad id 1 (RoBERTa's padding_idx; harmless for BERT)

    def batches_of(bs):
        res = []
        for i in range(0, len(samples), bs):
            ids = torch.stack([s[0] for s in samples[i:i + bs]])
            mask = torch.stack([s[1] for s in samples[i:i + bs]])
            d = {"input_ids": ids, "attention_mask": mask}
            if kind == "bert-qa":
                d["token_type_ids"] = torch.zeros_like(ids)
            res.append(d)
        return res
    batches = batches_of(B)
    out["input_ids"] = np.stack([b["input_ids"].numpy() for b in batches])
    out["attention_mask"] = np.stack([b["attention_mask"].numpy() for b in batches])

    a_q = Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    model = ref_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic", is_remove_padding=True).eval()

    def heads(outputs):
        return [outputs[0], outputs[1]] if kind == "bert-qa" else [outputs[0]]

    def prepare(bs_batches):
        """prepare_input_output: ptq_qa_quant.py:114-134 (masked start / end logits) / ptq_glue_quant.py:94-107."""
        res = []
        with torch.no_grad():
            for b in bs_batches:
                o = model(**b)
                if kind == "bert-qa":
                    res.append([o[0][b["attention_mask"] == 1].detach(), o[1][b["attention_mask"] == 1].detach()])
                else:
                    res.append(o[0].detach())
        return res

    def logits(bs_batches):
        with torch.no_grad():
            return np.stack([np.stack([h.numpy() for h in heads(model(**b))]) for b in bs_batches])
    ST.disable_all(model)
    fp_output = prepare(batches)
    out["logits_wrapped_fp"] = logits(batches)
    model = GM.delay_ln(model, Cfg(a_qconfig=a_q, w_qconfig=w_q), Cfg(model_type=attr, task_type=task))
    out["logits_after_gamma"] = logits(batches)
    out["module_names"] = np.array([n for n, _ in model.named_modules()])
    ST.enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    with torch.no_grad():
        model(**batches[0])
    ST.disable_all(model)
    ST.set_observer_name(model)
    TWC.task_type, TWC.model_type = task, attr
    losses = []

    class Grab(logging.Handler):
        def emit(self, record):
            msg = record.getMessage()
            if msg.startswith("the ratio is"):
                losses.append(float(msg.split("the loss is")[1]))
    h = Grab()
    TWC.logger.addHandler(h)
    TWC.logger.setLevel(logging.INFO)
    trainer = types.SimpleNamespace(model=model)
    iters, step = 6, 0.05
    TWC.find_ratio(trainer, batches, fp_output, {"iters": iters, "step": step})
    TWC.logger.removeHandler(h)
    out["twc_losses"], out["twc_grid"] = np.array(losses, dtype=np.float64), np.array([iters, step])
    names, scales, zps = quantizer_table(model, QuantizeBase)
    out["q_names"] = np.array(names)
    for i, (s_, z_) in enumerate(zip(scales, zps)):
        out[f"q_after_twc_scale::{i}"], out[f"q_after_twc_zp::{i}"] = s_, z_
    out["best_ratio"] = np.array([m.observer.percentile for n, m in model.named_modules()
                                  if isinstance(m, QuantizeBase) and "act" in n][:1])
    TWC.enable_quantization(model)
    out["logits_act_quant"] = logits(batches)
    if kind == "bert-qa":
        # ptq_qa_quant.py:262-267: smaller batches for the fine stage, FP targets recomputed with every quantizer off
        learn_bs = 2
        ST.disable_all(model)
        learn_batches = batches_of(learn_bs)
        learn_output = prepare(learn_batches)
        out["learn_batch_size"] = np.array(learn_bs)
    else:
        learn_batches, learn_output = batches, fp_output
    TWC.learn_scale(trainer, learn_batches, learn_output, {"lr": 1e-3, "epoch": 2})
    names, scales, zps = quantizer_table(model, QuantizeBase)
    for i, (s_, z_) in enumerate(zip(scales, zps)):
        out[f"q_after_learn_scale::{i}"], out[f"q_after_learn_zp::{i}"] = s_, z_
    ST.enable_quantization(model)
    out["logits_full_quant"] = logits(batches)
    path = os.path.join(OUT, {"bert-qa": "bert_qa_tiny_pipeline.npz", "roberta-cls": "roberta_tiny_pipeline.npz"}[kind])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(names), "quantizers; losses", losses)
    print("logit drift gamma:", np.abs(out["logits_after_gamma"] - out["logits_wrapped_fp"]).max(),
          "act-quant:", np.abs(out["logits_act_quant"] - out["logits_wrapped_fp"]).max(),
          "full:", np.abs(out["logits_full_quant"] - out["logits_wrapped_fp"]).max())


def main_plain():
    """The plain calibration flows (ptq_glue_quant.py:234-246, the `else` branch: weight calibration on the first batch,
    activation calibration over all batches, enable_quantization) for the `quant:` sections the reference ships besides
    twc_fine_gamma -- exp/bert_ptq/{minmax,mse,quantile}/cola/config.yaml -- and for BASELINE configs[3] (4-bit per-channel
    MSEFast weights, 6-bit AvgMSEFast activations), on the tiny classifier of main() (same seeds: weights and batches are the
    ones bert_tiny_pipeline.npz already holds).  Stores what the reference produced: every quantizer's scale / zero_point and
    observer statistics after calibration, and the fully quantized logits."""
    QB, GM, TWC, ST, QuantizeBase = import_reference()
    fp, batches, base = tiny_bert_cls()
    out = {"input_ids": base["input_ids"], "attention_mask": base["attention_mask"], "logits_hf_fp": base["logits_hf_fp"]}
    w6 = Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)

    def act(observer):
        return Cfg(quantizer="FixedFakeQuantize", observer=observer, bit=6, symmetric=False, ch_axis=-1)
    variants = {"minmax": (act("AvgMinMaxObserver"), w6), "mse": (act("AvgMSEFastObserver"), w6),
                "quantile": (act("AvgQuantileObserver"), w6),
                "w4a6_msefast": (act("AvgMSEFastObserver"),
                                 Cfg(quantizer="FixedFakeQuantize", observer="MSEFastObserver", bit=4, symmetric=True, ch_axis=0))}
    out["variants"] = np.array(list(variants))
    for name, (a_q, w_q) in variants.items():
        model = QB.QuantizedBertForSequenceClassification(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic",
                                                          is_remove_padding=True).eval()
        with torch.no_grad():
            ST.enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
            model(**batches[0])
            ST.enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
            for b in batches:
                model(**b)
            names, scales, zps = quantizer_table(model, QuantizeBase)
            out[f"{name}::q_names"] = np.array(names)
            for i, (s_, z_) in enumerate(zip(scales, zps)):
                out[f"{name}::scale::{i}"], out[f"{name}::zp::{i}"] = s_, z_
            for i, (n, m) in enumerate((n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase)):
                out[f"{name}::min::{i}"] = m.observer.min_val.detach().reshape(-1).to(torch.float64).numpy()
                out[f"{name}::max::{i}"] = m.observer.max_val.detach().reshape(-1).to(torch.float64).numpy()
            ST.enable_quantization(model)
            out[f"{name}::logits_full_quant"] = np.stack([model(**b)[0].numpy() for b in batches])
        print(name, len(names), "quantizers; logit drift", np.abs(out[f"{name}::logits_full_quant"] - base["logits_hf_fp"]).max())
    path = os.path.join(OUT, "bert_tiny_plain_ptq.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bart":
        main_bart()
    elif len(sys.argv) > 1 and sys.argv[1] == "plain":
        main_plain()
    elif len(sys.argv) > 1 and sys.argv[1] in ("bert-qa", "roberta-cls"):
        main_variant(sys.argv[1])
    else:
        main()
