"""A/B of the masked observation: ONE launch (csrc/observe_onelaunch.h) against the two launches it replaces
(token_minmax + token_select), dispatch-event clock, per shape.  Usage: python tools/onelaunch_ab.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
lib = _hip.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)


def timed(which, fn, reps=30):
    out = []
    for i in range(reps + 5):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "create")
        lib.osq_time_next_launch(which, a, b)
        fn(i)
        us = ctypes.c_float()
        _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "elapsed")
        lib.osq_timing_events_destroy(a, b)
        if i >= 5:
            out.append(us.value)
    out.sort()
    return out[len(out) // 2]


for shape, seq_pos in (((32, 128, 768), 1), ((32, 12, 128, 128), 2), ((32, 128, 3072), 1), ((32, 384, 768), 1), ((64, 128, 768), 1),
                       ((128, 128, 768), 1), ((256, 128, 768), 1)):
    xs = [torch.randn(*shape, generator=g).to(dev) for _ in range(3)]
    for x in xs:
        x[..., [5, 40, 61]] *= 20
    B, T = shape[0], shape[seq_pos]
    for name, L in (("lengths 8..T", torch.randint(8, T + 1, (B,), generator=g).to(dev)), ("all valid", torch.full((B,), T, dtype=torch.int64, device=dev))):
        res = {}
        for mode in (2, 1, 0):
            ops.set_tuning("observe_onelaunch", mode)
            mn, mx = torch.tensor(float("inf"), device=dev), torch.tensor(float("-inf"), device=dev)
            st = {"c": 0}

            def call(i):
                ops.observe_tokens(xs[i % 3], seq_pos, L, True, 0.95, ops.UPDATE_AVERAGE, st["c"], mn, mx, 0, 63, False)
                st["c"] += 1
            if mode == 2:
                res["split_a"] = timed(_hip.TIME_OBSERVE_TOKENS, call)
                res["split_b"] = timed(_hip.TIME_TOKEN_SELECT, call)
            elif mode:
                res["one"] = timed(_hip.TIME_OBSERVE_TOKENS, call)
            else:
                res["minmax"] = timed(_hip.TIME_TOKEN_MINMAX, call)
                res["select"] = timed(_hip.TIME_TOKEN_SELECT, call)
        ops.set_tuning("observe_onelaunch", 0)
        print(f"{str(shape):>20s} {name:13s}: one launch {res['one']:6.2f} us | two launches {res['minmax']:6.2f} + {res['select']:6.2f} = {res['minmax'] + res['select']:6.2f} us | records + small selectors {res['split_a']:6.2f} + {res['split_b']:6.2f} = {res['split_a'] + res['split_b']:6.2f} us", flush=True)
