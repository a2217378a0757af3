"""Deferred observation (quantization/deferred.py): the masked activation observers of a forward recorded and reduced in a
handful of launches -- bit-identical statistics and parameters to the site-by-site path, on raw quantizer calls over
every layout the models produce and on a whole tiny BERT observer pass."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, observer, quantizer="LSQPlusFakeQuantize", name="layer.x_post_act_fake_quantize", pct=0.9, sym=False):
    from outlier_suppression_amd.quantization import Quantizer
    q = Quantizer(None, NS(quantizer=quantizer, observer=observer, bit=6, symmetric=sym, ch_axis=-1)).to(dev)
    q.observer.set_name(name + ".observer")
    if hasattr(q.observer, "set_percentile"):
        q.observer.set_percentile(pct)
    q.enable_observer()
    q.disable_fake_quant()
    return q


def test_deferred_sites_equal_site_by_site():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(4)
    B, T = 6, 20
    mem = torch.randn(B, T, 4, 16, generator=gen).to(dev)
    L = torch.randint(1, T + 1, (B,), generator=gen).to(dev)
    L2 = torch.randint(1, T + 1, (B,), generator=gen).to(dev)
    specs = [("AvgPruneMinMaxObserver", "LSQPlusFakeQuantize", "a.q", lambda i: (torch.randn(B, T, 96, generator=gen).to(dev) * (i + 1), L, 1)),
             ("AvgPruneMinMaxObserver", "LSQPlusFakeQuantize", "a.attention_probs", lambda i: (torch.rand(B, 3, T, T, generator=gen).to(dev), L, 2)),
             ("AvgPruneMinMaxObserver", "FixedFakeQuantize", "a.v", lambda i: (mem.permute(0, 2, 1, 3) * (i + 1), L, 2)),       # [B,h,T,d] view
             ("AvgPruneMinMaxObserver", "LSQPlusFakeQuantize", "a.k", lambda i: (mem.permute(0, 2, 3, 1) * (i + 2), L2, 3)),    # [B,h,d,T] view, own mask
             ("AvgMinMaxObserver", "FixedFakeQuantize", "a.m", lambda i: (torch.randn(B, T, 33, generator=gen).to(dev), L, 1)),  # odd feature count
             ("MinMaxObserver", "FixedFakeQuantize", "a.r", lambda i: (torch.randn(B, T, 64, generator=gen).to(dev), L2, 1)),
             ("AvgPruneMinMaxObserver", "LSQPlusFakeQuantize", "a.nomask", lambda i: (torch.randn(B, T, 64, generator=gen).to(dev), None, 1)),
             ("AvgPruneMinMaxObserver", "LSQPlusFakeQuantize", "a.pooler", lambda i: (torch.randn(B, 64, generator=gen).to(dev), None, -1))]
    inputs = [[f(i) for (_, _, _, f) in specs] for i in range(3)]
    results = []
    for deferred in (False, True):
        qs = [_mk(dev, o, qz, n, pct=0.8 if "k" in n else 0.9, sym=(n == "a.m")) for (o, qz, n, _) in specs]
        if deferred:
            with deferred_observation() as sites:
                for i in range(3):
                    for q, (x, m, sp) in zip(qs, inputs[i]):
                        assert q(x, m, sp) is x
                    n = sites.flush()
                    assert n == len(specs) - 1             # the flat pooler site runs at once
            assert sites.launches <= 3 * (1 + 2 * 4)
        else:
            for i in range(3):
                for q, (x, m, sp) in zip(qs, inputs[i]):
                    q(x, m, sp)
        torch.cuda.synchronize()
        results.append([(q.observer.min_val.clone(), q.observer.max_val.clone(), q.scale.detach().clone(),
                         q.zero_point.detach().clone(), getattr(q.observer, "cnt", None)) for q in qs])
    for a, b in zip(*results):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert a[4] == b[4]


def test_deferred_calibrate_on_tiny_bert(golden):
    """token_wise_clipping.calibrate over a tiny BERT (observer pass at percentile 0.9): deferred == site by site."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from transformers import BertConfig, BertForSequenceClassification
    from outlier_suppression_amd import token_wise_clipping as TWC
    from outlier_suppression_amd.gamma_migration import delay_ln
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization.state import set_observer_name
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    g = golden("bert_tiny_pipeline")
    dev = torch.device("cuda:0")
    cfg = BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=40, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, type_vocab_size=2)
    fp = BertForSequenceClassification(cfg).eval()
    fp.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}, strict=False)
    fp = fp.to(dev)
    batches = [{"input_ids": torch.from_numpy(g["input_ids"][b]).to(dev), "attention_mask": torch.from_numpy(g["attention_mask"][b]).to(dev),
                "token_type_ids": torch.zeros_like(torch.from_numpy(g["input_ids"][b])).to(dev)} for b in range(4)]
    a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    out = []
    for defer in (False, True):
        model = quantize_model(fp, w_q, a_q).to(dev)
        model = delay_ln(model, NS(a_qconfig=a_q, w_qconfig=w_q), NS(model_type="bert", task_type="glue"))
        set_observer_name(model)
        TWC.DEFER_OBSERVATION = defer
        try:
            TWC.set_ratio(model, 0.9)
            TWC.calibrate(model, batches)
        finally:
            TWC.DEFER_OBSERVATION = True
        qs = [m for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
        out.append([(q.observer.min_val.clone(), q.observer.max_val.clone(), q.scale.detach().clone(), q.zero_point.detach().clone(),
                     q.observer.cnt) for q in qs])
    assert len(out[0]) >= 17
    for a, b in zip(*out):
        assert all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4] == 4
