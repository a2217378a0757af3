#!/usr/bin/env python3
"""How well do the rounds of an MSEFast observer pass and the NEXT batch's forward share the GPU?  (Would pipelining the calibration
loop -- forward(b + 1) under the rounds of batch b -- pay?)  RoBERTa-base fp32 forward [32,128] on one stream, rounds of 12 open
[32,128,3072] float64 searches (memo off: every round streams) on another: each alone, then together."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformers as T  # noqa: E402
from outlier_suppression_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = T.RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, num_labels=3,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
model = T.RobertaForSequenceClassification(cfg).eval().to(dev)
ids = torch.randint(1000, 50000, (32, 128), device=dev)
mask = torch.ones(32, 128, dtype=torch.long, device=dev)
g = torch.Generator().manual_seed(1)
ops.set_tuning("mse_memo", 0)
sites = [(torch.randn(32, 128, 3072, generator=g) * (1 + i % 3)).to(dev) for i in range(12)]
curs = [torch.stack([x.min(), x.max()]).to(torch.float32) for x in sites]
s_f, s_r = torch.cuda.Stream(), torch.cuda.Stream()
N_F, N_R = 3, 300           # 300 rounds: fewer than a search's evaluations, so every round streams every site
ctxs = []


def fresh_group():
    group = [ops.msefast_tensor_begin(x, c, None, 1, 0, 63, False, "no", True, float64_input=True) for x, c in zip(sites, curs)]
    ctxs.append((ops._ordered_group_prepare(group), group))
    torch.cuda.synchronize()


def forwards():
    with torch.cuda.stream(s_f), torch.no_grad():
        for _ in range(N_F):
            model(input_ids=ids, attention_mask=mask)


def rounds():
    with torch.cuda.stream(s_r):
        ops._ordered_group_rounds(ctxs[-1][0], N_R)


def timed(*fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


fresh_group(); forwards(); rounds()
for rep in range(2):
    fresh_group()
    tf, tr = timed(forwards), timed(rounds)
    fresh_group()
    both = timed(rounds, forwards)            # the rounds are enqueued first (~2 ms of host time), the forwards' launches follow
    print(f"{N_F} forwards alone {tf:7.2f} ms   {N_R} rounds alone {tr:7.2f} ms   together {both:7.2f} ms   (sum {tf + tr:7.2f}; overlap saves {100 * (tf + tr - both) / (tf + tr):.1f} %)")
ops.set_tuning("mse_memo", 1)
