#!/usr/bin/env python3
"""Where a round of the MSEFast rounds kernel spends its time (development build: make CXXFLAGS="... -DOSQ_MSE_DBG"): K
unmasked float64 searches in one table, the first rounds (every search still open) timed by stream events, with
osq_set_tuning("mse_dbg", d): 0 = the kernel as shipped, 1 = data loads replaced by generated values, 2 = the float64 term
replaced by a conversion, 3 = both (the cascade's additions, LDS tile, barriers and publishing alone).  Modes 1-3 give WRONG
losses; only the durations mean anything."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from outlier_suppression_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
ROUNDS = 12
for shape, k in (((32, 128, 768), 24), ((32, 128, 768), 48), ((32, 128, 3072), 12)):
    xs = [(torch.randn(*shape, generator=g) * (1 + i % 3)).to(dev) for i in range(k)]
    for groups, mode in ((8, 0), (8, 1), (8, 2), (8, 3), (16, 0), (4, 0)):
        ops.set_tuning("mse_round_groups", groups)
        try:
            ops.set_tuning("mse_dbg", mode)
        except Exception:                        # the shipped build has no such switch
            if mode:
                continue
        group = []
        for x in xs:
            cur = torch.stack([x.min(), x.max()]).to(torch.float32)
            group.append(ops.msefast_tensor_begin(x, cur, None, 1, 0, 63, False, "no", True, float64_input=True))
        ctx = ops._ordered_group_prepare(group)
        ops._ordered_group_rounds(ctx, 2)           # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops._ordered_group_rounds(ctx, ROUNDS)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / ROUNDS
        mb = sum(x.numel() for x in xs) * 4 / 1e6
        try:
            import ctypes
            from outlier_suppression_amd import _hip
            ph = (ctypes.c_ulonglong * 8)()
            if _hip.load().osq_mse_dbg_read(ph) == 0 and ph[4]:
                wg, fin = ph[4], max(ph[5], 1)
                print(f"      per workgroup (us, stamps drain the memory counters): entry->state {ph[0] / wg / 100:.2f}  groups+drain {ph[1] / wg / 100:.2f}  "
                      f"ticket {ph[2] / wg / 100:.2f}   per finisher: finish+step {ph[3] / fin / 100:.2f}   ({wg} workgroups, {fin} finishers)")
        except AttributeError:
            pass
        print(f"{k:3d} x {list(shape)} ({mb:6.1f} MB)  round_groups {groups:2d} mse_dbg {mode}: {us:8.2f} us per round  ({mb / us * 1e3 / 1e3:6.2f} TB/s of x)  all done: {int(ctx['done'].item())}")
ops.set_tuning("mse_round_groups", 8)
try:
    ops.set_tuning("mse_dbg", 0)
except Exception:
    pass
