#!/usr/bin/env python3
"""Microseconds per ROUND of the strict MSEFast rounds kernel (one loss evaluation of K open float64 searches), against
osq_set_tuning("mse_round_groups", n) when the tunable build is loaded (OSQ_HIP_LIBRARY=.../libosq_hip_dbg.so); the memo is
switched off so that every round streams every site."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from outlier_suppression_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
ROUNDS = 12
knobs = [int(a) for a in sys.argv[1:]] or [8]
modes = [int(m) for m in os.environ.get("MSE_DBG_MODES", "0").split()]      # -DOSQ_MSE_DBG builds: 1 = no loads, 2 = trivial term, 3 = both
ops.set_tuning("mse_memo", 0)
for kv in os.environ.get("MSE_TUNING", "").split():
    k, v = kv.split("=")
    ops.set_tuning(k, int(v))
CASES = (((32, 128, 768), 48), ((32, 128, 3072), 12), ((32, 128, 3072), 6), ((32, 128, 768), 8))
if os.environ.get("MSE_PROBE_CASES"):
    CASES = tuple(CASES[int(i)] for i in os.environ["MSE_PROBE_CASES"].split())
for shape, k in CASES:
    xs = [(torch.randn(*shape, generator=g) * (1 + i % 3)).to(dev) for i in range(k)]
    for groups, mode in [(a, b) for a in knobs for b in modes]:
        try:
            ops.set_tuning("mse_round_groups", groups)
        except Exception:
            if groups != knobs[0]:
                continue
        if mode or len(modes) > 1:
            ops.set_tuning("mse_dbg", mode)
        group = []
        for x in xs:
            cur = torch.stack([x.min(), x.max()]).to(torch.float32)
            group.append(ops.msefast_tensor_begin(x, cur, None, 1, 0, 63, False, "no", True, float64_input=True))
        ctx = ops._ordered_group_prepare(group)
        ops._ordered_group_rounds(ctx, 2)           # warm
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops._ordered_group_rounds(ctx, ROUNDS)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / ROUNDS)
        mb = sum(x.numel() for x in xs) * 4 / 1e6
        print(f"{k:3d} x {list(shape)} ({mb:6.1f} MB)  round_groups/slots {groups:4d} mse_dbg {mode}: {best:8.2f} us per round  ({mb / best:6.3f} TB/s of x)  blocks {ctx['blocks']}", flush=True)
ops.set_tuning("mse_memo", 1)
