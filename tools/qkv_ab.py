#!/usr/bin/env python3
"""The query / key / value head-split sites as one launch (util_layernorm.FUSE_QKV) against the three per-site launches:
fully quantized BERT-base forward on [32,128] and [32,384], every quantizer frozen -- milliseconds per forward (CUDA
events over 100 forwards) and whether the logits are the same bits."""
import os
import sys
import time
from types import SimpleNamespace as NS

import torch
import transformers as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from outlier_suppression_amd import util_layernorm as UL  # noqa: E402
from outlier_suppression_amd.quant_model import quantize_model  # noqa: E402
from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
fp = T.BertForSequenceClassification(T.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
a = NS(quantizer="LSQPlusFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
w = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
for tokens in (128, 384):
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 29000, (32, tokens), generator=g).to(dev)
    mask = torch.ones(32, tokens, dtype=torch.long, device=dev)
    batch = {"input_ids": ids, "attention_mask": mask}
    model = quantize_model(fp, w, a).to(dev)
    with torch.no_grad():
        enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
        model(**batch)
        enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
        model(**batch)
        enable_quantization(model)
        outs = {}
        for fuse in (False, True, False, True):
            UL.FUSE_QKV = fuse
            for _ in range(10):
                y = model(**batch)[0]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(100):
                y = model(**batch)[0]
            e1.record()
            torch.cuda.synchronize()
            outs[fuse] = y.clone()
            print(f"[32,{tokens}] FUSE_QKV={fuse}: {e0.elapsed_time(e1) / 100:.3f} ms per forward (GPU), {(time.perf_counter() - t0) * 10:.3f} ms wall")
        print(f"[32,{tokens}] logits identical: {torch.equal(outs[True], outs[False])}")
UL.FUSE_QKV = True
