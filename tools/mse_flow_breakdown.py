#!/usr/bin/env python3
"""Where a batch of configs[3]'s activation calibration (RoBERTa-base, 98 AvgMSEFast sites, [32,128]) spends its wall time:
forward, the flush's set-up (gathers of masked sites, tables), the rounds, the rest.  Synchronising timers round the pieces
(they break the overlap of the two streams' set-up, nothing else)."""
import os
import sys
import time
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformers as T  # noqa: E402
from outlier_suppression_amd import calibration, ops  # noqa: E402
from outlier_suppression_amd.quant_model import quantize_model  # noqa: E402
from outlier_suppression_amd.quantization import enable_calibration_woquantization  # noqa: E402
from outlier_suppression_amd.quantization import deferred  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
g = torch.Generator().manual_seed(42)
cfg = T.RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, num_labels=3,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
fp = T.RobertaForSequenceClassification(cfg).eval().to(dev)
batches = []
for _ in range(4):
    L = torch.randint(8, 129, (32,), generator=g)
    mask = (torch.arange(128)[None, :] < L[:, None]).long()
    ids = torch.randint(1000, 50000, (32, 128), generator=g) * mask + (1 - mask)
    batches.append({"input_ids": ids.to(dev), "attention_mask": mask.to(dev)})
w_q = NS(quantizer="FixedFakeQuantize", observer="MSEFastObserver", bit=4, symmetric=True, ch_axis=0)
a_q = NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1)
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrapper


ops._ordered_group_prepare = timed("prepare (gathers + tables)", ops._ordered_group_prepare)
ops.msefast_tensor_run_ordered_groups = timed("run_ordered_groups (prepare + rounds)", ops.msefast_tensor_run_ordered_groups)
deferred.DeferredSites._flush_mse = timed("flush_mse (begin + groups + commit)", deferred.DeferredSites._flush_mse)
deferred.DeferredSites.flush = timed("flush (all)", deferred.DeferredSites.flush)
fwd = timed("forward", lambda m, b: m(**b))
for rep in range(2):
    model = quantize_model(fp, w_q, a_q).to(dev)
    enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    calibration.calibrate_owned_sites(model, batches[:1], lambda m, b: m(**b), select=lambda n: "weight_fake_quant" in n)
    enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
    acc.clear()
    stats = deferred.MSE_STATS_SINK = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    calibration.calibrate_owned_sites(model, batches, fwd)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"run {rep}: {len(batches)} batches, {total * 1e3:.1f} ms = {total / len(batches) * 1e3:.1f} ms per batch")
    for k, v in acc.items():
        print(f"    {k:42s} {v / len(batches) * 1e3:7.2f} ms per batch")
    if rep == 1 and stats:
        # the loss memo, search by search: what streamed and what was answered (last run, all batches)
        import collections
        rows = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
        rounds_per_batch = []
        for fwd_stats in stats:
            most = 0
            for elems, nested, t in fwd_stats:
                nfev, pairs, hits, done = (int(v) for v in t.cpu())
                r = rows[(elems, nested)]
                r[0] += 1; r[1] += nfev; r[2] += nfev - hits; r[3] += hits; r[4] += elems * 4 * (nfev - hits)
                most = max(most, nfev - hits)
            rounds_per_batch.append(most)
        print("    loss memo (elements of the site as recorded, nested search?): searches, nfev, streamed, answered, GB streamed")
        tot_b, tot_all = 0, 0
        for (elems, nested), r in sorted(rows.items()):
            print(f"      {elems:9d} {str(nested):5s}  {r[0]:4d} searches  nfev {r[1] / r[0]:6.1f}  streamed {r[2] / r[0]:6.1f}  answered {r[3] / r[0]:6.1f}  {r[4] / 1e9:7.2f} GB")
            tot_b += r[4]; tot_all += elems * 4 * r[1]
        print(f"      bytes streamed {tot_b / 1e9:.1f} GB of {tot_all / 1e9:.1f} GB without the memo (slots, padding included); most streamed evaluations of a search per batch: {rounds_per_batch}")
