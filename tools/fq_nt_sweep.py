"""Development aid: cache policy of the dense fake-quant kernel's loads / stores (osq_set_tuning("fq_nt", 0..5)), every launch
timed by the events on its own dispatch packet AND as a graph of 50 dependent launches (the kernel boundary counts)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
lib = _hip.load()
dev = torch.device("cuda:0")
shape = tuple(int(v) for v in os.environ.get("AB_SHAPE", "256x128x768").split("x"))
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(*shape, device=dev, generator=g) for _ in range(4)]
s = torch.tensor([0.7], device=dev); z = torch.tensor([31.3], device=dev)
n = xs[0].numel()


def timed(fn, reps=30):
    out = []
    for i in range(reps + 3):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "events")
        lib.osq_time_next_launch(_hip.TIME_FAKE_QUANT, a, b)
        fn(i)
        us = ctypes.c_float()
        _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "elapsed")
        lib.osq_timing_events_destroy(a, b)
        if i >= 3:
            out.append(us.value)
    out.sort()
    return out[len(out) // 2], out[0]


def graph_step(steps=50):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            y = ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for i in range(steps):
            y = ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    reps = []
    for _ in range(5):
        t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / steps * 1e6)
    return min(reps)


for rnd in range(3):
    for nt in (3, 0, 1, 2, 4, 5):
        lib.osq_set_tuning(b"fq_nt", nt)
        med, mn = timed(lambda i: ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
        gs = graph_step()
        print(f"fq_nt={nt}: kernel median {med:6.2f} us min {mn:6.2f} us -> {8 * n / med / 1e3:6.0f} GB/s | graph {gs:6.2f} us per launch -> {8 * n / gs / 1e3:6.0f} GB/s", flush=True)
