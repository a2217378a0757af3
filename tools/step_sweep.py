"""Development aid: does the fake-quant's re-read of the tensor that the observer just read come from the
Infinity Cache?  Times the bench step for every combination of load/store cache policies."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from outlier_suppression_amd import _hip, ops
dev = torch.device("cuda:0")
lib = _hip.load()
q = bench.make_quantizer(dev)
xs, lengths = bench.make_inputs(dev, 4, 1234)
lengths = lengths.to(dev)
def run(steps=200):
    with torch.no_grad():
        for i in range(20): q(xs[i % 4], lengths, 1)
        pairs = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            q._observe(xs[i % 4], lengths, 1)
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)); lib.osq_time_next_launch(_hip.TIME_FAKE_QUANT, a, b)
            q._quantize(xs[i % 4]); pairs.append((a, b))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e6
    tot = 0.0
    for a, b in pairs:
        us = ctypes.c_float(); lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)); lib.osq_timing_events_destroy(a, b); tot += us.value
    return dt, tot / steps
for tok_nt in (1, 0):
    for fq_nt in (3, 2, 1, 0):
        ops.set_tuning("tok_nt", tok_nt); ops.set_tuning("fq_nt", fq_nt)
        dt, fq = run()
        print(f"tok_nt={tok_nt} fq_nt={fq_nt} (bit0 loads, bit1 stores): step {dt:6.2f} us, fq kernel {fq:6.2f} us", flush=True)
