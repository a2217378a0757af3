#!/usr/bin/env python3
"""The masked observation alone (AvgPruneMinMaxObserver / AvgMinMaxObserver, fake-quant off): round 4's two plain launches
against round 5's bucketed form (token_observe.h: the streaming launch files every token's extrema into a window
histogram, the selecting launch ranks from the buckets).  Per shape and form: each launch's own duration (HIP events on
its dispatch packet, what rocprofv3 --kernel-trace reports) and the stream time per call of 200 back-to-back calls (kernel
boundaries included).  Results are compared bit for bit across the forms on the way."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from outlier_suppression_amd import _hip, ops  # noqa: E402

lib = _hip.load()
dev = torch.device("cuda:0")
FORMS = (("two plain launches (round 4)", 0), ("bucketed (round 5)", 1), ("buckets filled, not used", 3), ("bucketed launches, no hint", 2))


def kernel_us(which, fn, reps=30):
    out = []
    for i in range(reps + 3):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "events")
        lib.osq_time_next_launch(which, a, b)
        fn(i)
        us = ctypes.c_float()
        _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "elapsed")
        lib.osq_timing_events_destroy(a, b)
        if i >= 3:
            out.append(us.value)
    out.sort()
    return out[len(out) // 2]


def run(shape, lengths, prune, tag):
    g = torch.Generator().manual_seed(0)
    xs = []
    nbuf = max(2, int(400e6 // (4 * shape[0] * shape[1] * shape[2])) + 1)
    for i in range(min(nbuf, 5)):
        x = torch.randn(*shape, generator=g)
        idx = torch.randperm(shape[-1], generator=g)[:6]
        x[..., idx] *= 20
        xs.append(x.to(dev))
    L = None if lengths is None else lengths.to(dev)
    print(f"== {tag}: {list(shape)}, {'all tokens' if L is None else str(int(lengths.sum())) + ' valid tokens'}, prune={prune}")
    ref = None
    for name, mode in FORMS:
        ops.set_tuning("observe_hist", mode)
        mn, mx = torch.tensor(float("inf"), device=dev), torch.tensor(float("-inf"), device=dev)
        state = {"cnt": 0}

        def call(i):
            ops.observe_tokens(xs[i % len(xs)], 1, L, prune, 0.95, ops.UPDATE_AVERAGE, state["cnt"], mn, mx, 0, 63, False)
            state["cnt"] += 1
        for i in range(3):
            call(i)
        stat = (mn.item(), mx.item())
        if ref is None:
            ref = stat
        assert stat == ref, (name, stat, ref)
        k1 = kernel_us(_hip.TIME_TOKEN_MINMAX, call)
        k2 = kernel_us(_hip.TIME_TOKEN_SELECT, call)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(20):
            call(i)
        e0.record()
        for i in range(200):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        print(f"   {name:32s} per-token launch {k1:6.2f} us   select launch {k2:6.2f} us   sum {k1 + k2:6.2f} us   stream {e0.elapsed_time(e1) * 5:6.2f} us/call")
    ops.set_tuning("observe_hist", 1)


g = torch.Generator().manual_seed(0)
bench_len = torch.randint(8, 129, (256,), generator=g)
run((256, 128, 768), bench_len, True, "bench tensor, bench lengths")
run((256, 128, 768), None, True, "bench tensor, all tokens")
run((256, 128, 768), None, False, "bench tensor, all tokens, plain extrema (AvgMinMax)")
run((32, 128, 768), torch.randint(8, 129, (32,), generator=g), True, "site size")
run((32, 384, 768), torch.randint(8, 385, (32,), generator=g), True, "SQuAD site size")
run((32, 128, 3072), None, True, "GELU site")
