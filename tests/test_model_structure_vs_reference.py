"""CPU, build container only: this package's quantized models against the REFERENCE's model classes.

With every quantizer switched off both are plain FP models, so they can be compared on the CPU:
same quantizer names in the same order, same module tree (what Gamma Migration and the
name-substring switches rely on), identical logits.  Skipped where /root/reference is absent
(the GPU box); the GPU-side evidence is tests/test_gpu_model.py against committed goldens.
"""
import copy
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not available")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_model as M
    QB, GM, TWC, ST, QuantizeBase = M.import_reference()
    sys.modules.setdefault("transformers.generation_utils", types.ModuleType("transformers.generation_utils"))
    from quant_transformer.model import quant_roberta as RQ
    return M, QB, RQ, QuantizeBase


def _patch(model, attr):
    m = getattr(model, attr)
    m.embeddings.position_embedding_type = "absolute"
    m.encoder.gradient_checkpointing = False
    for layer in m.encoder.layer:
        layer.attention.pruned_heads = set()
        layer.attention.self.position_embedding_type = "absolute"
        if not hasattr(layer, "chunk_size_feed_forward"):
            layer.chunk_size_feed_forward = 0
    return model


CASES = ["bert-cls", "bert-qa", "roberta-cls", "roberta-qa"]


@pytest.mark.parametrize("backend", ["academic"])
@pytest.mark.parametrize("case", CASES)
def test_same_tree_names_and_fp_logits(ref, case, backend):
    M, QB, RQ, RefQuantizeBase = ref
    import transformers as T
    from outlier_suppression_amd.model import quant_bert as OB, quant_roberta as OR
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    common = dict(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                  max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    table = {
        "bert-cls": (T.BertForSequenceClassification, T.BertConfig(num_labels=3, **common),
                     QB.QuantizedBertForSequenceClassification, OB.QuantizedBertForSequenceClassification, "bert", 35),
        "bert-qa": (T.BertForQuestionAnswering, T.BertConfig(**common),
                    QB.QuantizedBertForQuestionAnswering, OB.QuantizedBertForQuestionAnswering, "bert", 33),
        "roberta-cls": (T.RobertaForSequenceClassification, T.RobertaConfig(num_labels=3, pad_token_id=1, **common),
                        RQ.QuantizedRobertaForSequenceClassification, OR.QuantizedRobertaForSequenceClassification, "roberta", 35),
        "roberta-qa": (T.RobertaForQuestionAnswering, T.RobertaConfig(pad_token_id=1, **common),
                       RQ.QuantizedRobertaForQuestionAnswering, OR.QuantizedRobertaForQuestionAnswering, "roberta", 33),
    }
    hf_cls, cfg, ref_cls, our_cls, attr, n_quant = table[case]
    torch.manual_seed(3)
    fp = _patch(hf_cls(cfg).eval(), attr)
    a_q = M.Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = M.Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    theirs = ref_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend=backend, is_remove_padding=True).eval()
    ours = our_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend=backend, is_remove_padding=True).eval()
    ref_q = [n for n, m in theirs.named_modules() if isinstance(m, RefQuantizeBase)]
    our_q = [n for n, m in ours.named_modules() if isinstance(m, QuantizeBase)]
    assert len(our_q) == n_quant and our_q == ref_q
    assert [n for n, _ in ours.named_modules()] == [n for n, _ in theirs.named_modules()]
    ids = torch.randint(3, 100, (3, 12))
    L = torch.tensor([12, 7, 4])
    mask = (torch.arange(12)[None] < L[:, None]).long()
    ids = ids * mask + (1 - mask)
    with torch.no_grad():
        r = theirs(input_ids=ids, attention_mask=mask)
        o = ours(input_ids=ids, attention_mask=mask)
    for a, b in zip(r[:len(o)], o):
        assert torch.equal(a, b)
    # with labels the tuple starts with the loss, computed as the reference does (model/losses.py)
    if case.endswith("qa"):
        lab = dict(start_positions=torch.tensor([1, 0, 30]), end_positions=torch.tensor([[3], [2], [5]]))
    else:
        lab = dict(labels=torch.randint(0, cfg.num_labels, (3,)))
    with torch.no_grad():
        rl = theirs(input_ids=ids, attention_mask=mask, return_dict=False, **lab)
        ol = ours(input_ids=ids, attention_mask=mask, **lab)
    assert rl[0].dim() == 0 and torch.equal(rl[0], ol[0]) and torch.equal(rl[1], ol[1])


@pytest.mark.parametrize("backend", ["academic"])
@pytest.mark.parametrize("task", ["summ", "cls", "qa"])
def test_bart_same_tree_names_and_fp_logits(ref, task, backend):
    M, QB, RQ, RefQuantizeBase = ref
    import transformers as T
    from torch import nn
    gu = types.ModuleType("transformers.generation_utils")
    from transformers.generation import GenerationMixin
    gu.GenerationMixin = GenerationMixin
    sys.modules["transformers.generation_utils"] = gu
    from quant_transformer.model import quant_bart as RB
    from outlier_suppression_amd.model import quant_bart as OB
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    torch.manual_seed(2)
    cfg = T.BartConfig(vocab_size=120, d_model=32, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
                       decoder_attention_heads=2, encoder_ffn_dim=64, decoder_ffn_dim=64, max_position_embeddings=40,
                       dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, pad_token_id=1, bos_token_id=0,
                       eos_token_id=2, decoder_start_token_id=2)
    hf_cls, name, n_head = {"summ": (T.BartForConditionalGeneration, "QuantizedBartForConditionalGeneration", 1),
                            "cls": (T.BartForSequenceClassification, "QuantizedBartForSequenceClassification", 4),
                            "qa": (T.BartForQuestionAnswering, "QuantizedBartForQuestionAnswering", 1)}[task]
    cfg.num_labels = 3 if task == "cls" else 2
    fp = hf_cls(cfg).eval()

    def plain(e):      # the 4.18-era layout the reference wrappers expect: plain nn.Embedding + embed_scale on the stack
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
    a_q = M.Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = M.Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    theirs = getattr(RB, name)(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend=backend, is_remove_padding=True).eval()
    ours = getattr(OB, name)(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend=backend, is_remove_padding=True).eval()
    ref_q = [n for n, m in theirs.named_modules() if isinstance(m, RefQuantizeBase)]
    our_q = [n for n, m in ours.named_modules() if isinstance(m, QuantizeBase)]
    body_q = 2 * 8 + 2 * 14 + 2 + 2 * 6 + 2 * 10 + 3 + 2
    assert our_q == ref_q and len(our_q) in (body_q + n_head, body_q + n_head - 1), len(our_q)
    assert [n for n, _ in ours.named_modules()] == [n for n, _ in theirs.named_modules()]
    ids = torch.randint(3, 100, (3, 12))
    L = torch.tensor([12, 7, 4])
    mask = (torch.arange(12)[None] < L[:, None]).long()
    ids = ids * mask + (1 - mask)
    dids = torch.randint(3, 100, (3, 6))
    DL = torch.tensor([6, 3, 5])
    dmask = (torch.arange(6)[None] < DL[:, None]).long()
    dids = dids * dmask + (1 - dmask)
    if task == "cls":      # the head reads the decoder state at the last <eos>; the decoder input is the shifted source
        ids[torch.arange(3), L - 1] = cfg.eos_token_id
        kw = dict(input_ids=ids, attention_mask=mask)
    else:
        kw = dict(input_ids=ids, attention_mask=mask, decoder_input_ids=dids, decoder_attention_mask=dmask)
    with torch.no_grad():
        r = theirs(use_cache=False, return_dict=False, **kw)
        o = ours(**kw)
    assert torch.equal(r[0], o[0])
    if task == "qa":
        assert torch.equal(r[1], o[1])
    # with labels the tuple starts with the loss, computed as the reference does (model/losses.py)
    if task == "summ":
        lab = dict(labels=torch.randint(3, 100, (3, 6)))
    elif task == "cls":
        lab = dict(labels=torch.tensor([0, 2, 1]))
    else:
        lab = dict(start_positions=torch.tensor([1, 0, 9]), end_positions=torch.tensor([3, 2, 40]))
    with torch.no_grad():
        rl = theirs(use_cache=False, return_dict=False, **kw, **lab)
        ol = ours(**kw, **lab)
    assert rl[0].dim() == 0 and torch.equal(rl[0], ol[0]) and torch.equal(rl[1], ol[1])


def test_other_backends_are_refused():
    """The reference's 'tensorrt' placement is out of scope: quantize_model refuses it instead of treating it
    as 'academic'."""
    import transformers as T
    from types import SimpleNamespace as NS
    from outlier_suppression_amd.quant_model import quantize_model
    fp = T.BertForSequenceClassification(T.BertConfig(vocab_size=50, hidden_size=16, num_hidden_layers=1, num_attention_heads=2,
                                                      intermediate_size=32, max_position_embeddings=16))
    a_q = NS(quantizer="FixedFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    with pytest.raises(NotImplementedError, match="academic"):
        quantize_model(fp, w_q, a_q, backend="tensorrt")
