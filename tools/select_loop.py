#!/usr/bin/env python3
"""A loop of stand-alone token-wise-clipping selections at the BASELINE token count (256 x 128 = 32768 slots), for the
PMC passes of tools/pmc_kernel.sh: token_minmax once, then N x token_range_finalize (token_select_kernel<8>) on its output.
    python tools/select_loop.py [--n 200] [--lengths bench|full] [--events]
--events prints the launch duration by dispatch events (median of the N launches)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.common import make_inputs, PERCENTILE, SHAPE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--lengths", default="bench", choices=["bench", "full"])
    ap.add_argument("--events", action="store_true")
    ap.add_argument("--state", action="store_true", help="an averaging observer's call from its second batch on: the running statistic predicts the threshold (pre-listed selection)")
    a = ap.parse_args()
    from outlier_suppression_amd import _hip, ops
    dev = torch.device("cuda:0")
    lib = _hip.load()
    xs, lengths = make_inputs(dev, 1, 1234)
    lengths = lengths.to(dev) if a.lengths == "bench" else torch.full((SHAPE[0],), SHAPE[1], device=dev)
    tok = ops.token_minmax(xs[0], 1, lengths)
    cur = torch.empty(2, device=dev)
    rule, cnt, mn, mx = ops.UPDATE_NONE, 0, None, None
    if a.state:
        rule, mn, mx = ops.UPDATE_AVERAGE, torch.tensor(float("inf"), device=dev), torch.tensor(float("-inf"), device=dev)
        ops.token_range_finalize(tok[0], tok[1], tok[2], tok[3], tok[4], True, PERCENTILE, rule, 0, mn, mx, 0, 63, False, None, cur)   # first batch
        cnt = 3
    us_all = []
    for i in range(a.n):
        if a.events:
            ea, eb = ctypes.c_void_p(), ctypes.c_void_p()
            _hip.check(lib.osq_timing_events_create(ctypes.byref(ea), ctypes.byref(eb)), "events")
            lib.osq_time_next_launch(_hip.TIME_TOKEN_SELECT, ea, eb)
        ops.token_range_finalize(tok[0], tok[1], tok[2], tok[3], tok[4], True, PERCENTILE, rule, cnt, mn, mx, 0, 63, False, None, cur)
        if a.events:
            us = ctypes.c_float()
            _hip.check(lib.osq_timing_elapsed_us(ea, eb, ctypes.byref(us)), "elapsed")
            lib.osq_timing_events_destroy(ea, eb)
            us_all.append(us.value)
    torch.cuda.synchronize()
    if us_all:
        us_all.sort()
        print(f"token_select_kernel, {a.lengths} lengths{', running state (hint)' if a.state else ''}, {SHAPE[0] * SHAPE[1]} slots: median {us_all[len(us_all) // 2]:.2f} us, min {us_all[0]:.2f}, max {us_all[-1]:.2f} over {len(us_all)}")
    print("cur (min, max):", cur.tolist())


if __name__ == "__main__":
    main()
