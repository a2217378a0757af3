#!/usr/bin/env python3
"""Helper of tests/test_reference_shim.py (build container only): builds the REFERENCE's quantized model classes

  pure      on the reference's own quant_transformer.quantization,
  shim      with quant_transformer.quantization[.state|.fake_quant|.observer|.quantized_module|.util_quant] aliased to
            outlier_suppression_amd.quantization -- the reference's model files,
            util_layernorm.py and gamma_migration.py run unchanged on this package's quantizers,
  shim+ln   additionally quant_transformer.model.util_layernorm / quant_transformer.solver.gamma_migration aliased to
            this package's (construction only: this package's gamma fold needs a HIP device),

and prints one JSON object: per model its module tree, quantizer names and classes, state-dict keys / dtypes / shapes,
the tree after the reference's delay_ln, a state-dict round trip and the FP logits with every quantizer off.
"""
import copy
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "golden"))
mode = sys.argv[1]

if mode != "pure":
    import outlier_suppression_amd.quantization as Q
    from outlier_suppression_amd.quantization import fake_quant, observer, quantized_module, state, util_quant
    sys.modules["quant_transformer.quantization"] = Q
    for name, mod in (("state", state), ("fake_quant", fake_quant), ("observer", observer),
                      ("quantized_module", quantized_module), ("util_quant", util_quant)):
        sys.modules["quant_transformer.quantization." + name] = mod
    if mode == "shim+ln":
        import outlier_suppression_amd.gamma_migration as our_gm
        import outlier_suppression_amd.util_layernorm as our_ln
        sys.modules["quant_transformer.model.util_layernorm"] = our_ln
        sys.modules["quant_transformer.solver.gamma_migration"] = our_gm

import make_golden_model as M    # noqa: E402

QB, GM, TWC, ST, QuantizeBase = M.import_reference()
gu = types.ModuleType("transformers.generation_utils")
from transformers.generation import GenerationMixin    # noqa: E402
gu.GenerationMixin = GenerationMixin
sys.modules["transformers.generation_utils"] = gu
from quant_transformer.model import quant_bart as RB, quant_roberta as RQ    # noqa: E402
import transformers as T    # noqa: E402

torch.set_num_threads(1)
a_q = M.Cfg(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
w_q = M.Cfg(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
common = dict(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
              max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def patch(model, attr):
    m = getattr(model, attr)
    m.embeddings.position_embedding_type = "absolute"
    m.encoder.gradient_checkpointing = False
    for layer in m.encoder.layer:
        layer.attention.pruned_heads = set()
        layer.attention.self.position_embedding_type = "absolute"
        if not hasattr(layer, "chunk_size_feed_forward"):
            layer.chunk_size_feed_forward = 0
    return model


def describe(model):
    qs = [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase)]
    sd = model.state_dict()
    return {"modules": [n for n, _ in model.named_modules()],
            "quantizers": [n for n, _ in qs],
            "quantizer_classes": sorted({type(m).__name__ + "/" + type(m.observer).__name__ for _, m in qs}),
            "quantizer_package": sorted({type(m).__module__.split(".")[0] for _, m in qs}),
            "state_dict": [[k, str(v.dtype), list(v.shape)] for k, v in sd.items()]}


out = {"mode": mode, "Quantizer_from": QB.Quantizer.__module__, "QuantizedModule_from": QB.QuantizedModule.__module__}
torch.manual_seed(3)
cases = {
    "bert-cls": (T.BertForSequenceClassification, T.BertConfig(num_labels=3, **common), QB.QuantizedBertForSequenceClassification, "bert", "glue"),
    "bert-qa": (T.BertForQuestionAnswering, T.BertConfig(**common), QB.QuantizedBertForQuestionAnswering, "bert", "qa"),
    "roberta-cls": (T.RobertaForSequenceClassification, T.RobertaConfig(num_labels=3, pad_token_id=1, **common),
                    RQ.QuantizedRobertaForSequenceClassification, "roberta", "glue"),
    "roberta-qa": (T.RobertaForQuestionAnswering, T.RobertaConfig(pad_token_id=1, **common), RQ.QuantizedRobertaForQuestionAnswering,
                   "roberta", "qa"),
}
ids = torch.randint(3, 100, (3, 12), generator=torch.Generator().manual_seed(1))
L = torch.tensor([12, 7, 4])
mask = (torch.arange(12)[None] < L[:, None]).long()
ids = ids * mask + (1 - mask)
for k, (name, (hf_cls, cfg, ref_cls, attr, task)) in enumerate(cases.items()):
    torch.manual_seed(100 + k)           # the same FP weights in every mode, whatever the earlier models drew
    fp = patch(hf_cls(cfg).eval(), attr)
    model = ref_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic", is_remove_padding=True).eval()
    d = describe(model)
    with torch.no_grad():
        d["fp_logits"] = [t.tolist() for t in model(input_ids=ids, attention_mask=mask)[:2] if torch.is_tensor(t)]
    if mode != "shim+ln":
        model = GM.delay_ln(model, M.Cfg(a_qconfig=a_q, w_qconfig=w_q), M.Cfg(model_type=attr, task_type=task))
        d["after_delay_ln"] = describe(model)
        with torch.no_grad():
            d["fp_logits_after_delay_ln"] = [t.tolist() for t in model(input_ids=ids, attention_mask=mask)[:2] if torch.is_tensor(t)]
        # state-dict round trip into a freshly built model of the same kind
        again = ref_cls(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic", is_remove_padding=True).eval()
        again = GM.delay_ln(again, M.Cfg(a_qconfig=a_q, w_qconfig=w_q), M.Cfg(model_type=attr, task_type=task))
        res = again.load_state_dict(model.state_dict())
        d["round_trip"] = [list(res.missing_keys), list(res.unexpected_keys)]
    out[name] = d

cfg, fp = M.tiny_bart()
model = RB.QuantizedBartForConditionalGeneration(copy.deepcopy(fp), w_q, a_q, qoutput=False, backend="academic",
                                                 is_remove_padding=True).eval()
d = describe(model)
if mode != "shim+ln":
    model = GM.delay_ln(model, M.Cfg(a_qconfig=a_q, w_qconfig=w_q), M.Cfg(model_type="bart", task_type="summ"))
    d["after_delay_ln"] = describe(model)
out["bart"] = d
print("SHIM_JSON " + json.dumps(out))
