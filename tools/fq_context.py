import ctypes, os, sys, time
import torch
sys.path.insert(0, '/root/repo')
import bench
from outlier_suppression_amd import _hip, ops
dev = torch.device("cuda:0"); lib = _hip.load()
q = bench.make_quantizer(dev)
xs, lengths = bench.make_inputs(dev, 4, 1234); lengths = lengths.to(dev)
s = torch.tensor([0.7], device=dev); zf = torch.tensor([31.0], device=dev)
tok = ops.token_minmax(xs[0], 1, lengths); cur = torch.empty(2, device=dev)
def timed_fq(pre, reps=40, sync_each=True, ybuf=None):
    out = []
    with torch.no_grad():
        for i in range(reps + 3):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b))
            pre(i)
            lib.osq_time_next_launch(_hip.TIME_FAKE_QUANT, a, b)
            ops.fake_quant_per_tensor(xs[i % 4], s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
            us = ctypes.c_float(); lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)); lib.osq_timing_events_destroy(a, b)
            if i >= 3: out.append(us.value)
    return sum(out) / len(out)
print("fq alone (sync between)            ", round(timed_fq(lambda i: None), 2))
print("fq after select                    ", round(timed_fq(lambda i: ops.token_range_finalize(tok[0], tok[1], tok[2], tok[3], tok[4], True, 0.95, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur)), 2))
print("fq after token_minmax              ", round(timed_fq(lambda i: ops.token_minmax(xs[i % 4], 1, lengths)), 2))
print("fq after token_minmax + select     ", round(timed_fq(lambda i: (ops.token_minmax(xs[i % 4], 1, lengths), ops.token_range_finalize(tok[0], tok[1], tok[2], tok[3], tok[4], True, 0.95, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur))), 2))
print("fq after q._observe (scale written)", round(timed_fq(lambda i: q._observe(xs[i % 4], lengths, 1)), 2))
