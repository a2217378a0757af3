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


@pytest.mark.parametrize("case", CASES)
def test_same_tree_names_and_fp_logits(ref, case):
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
    theirs = ref_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic", is_remove_padding=True).eval()
    ours = our_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic", is_remove_padding=True).eval()
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
