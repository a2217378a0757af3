"""Development aid: write-through against nt stores (osq_set_tuning("stream_wt", 0|1), "fq_nt" 3|5) for the streaming kernels,
A/B inside one process, every launch timed by the events on its own dispatch packet."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
lib = _hip.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
s = torch.tensor([0.05], device=dev); z = torch.tensor([31.3], device=dev)


def timed(which, fn, reps=30):
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


for shape in ((256, 128, 768), (32, 128, 768), (32, 128, 3072)):
    xs = [torch.randn(*shape, device=dev, generator=g) for _ in range(4)]
    gy = torch.randn(*shape, device=dev, generator=g)
    H = shape[-1]
    gamma, w, b = (torch.randn(H, device=dev, generator=g) for _ in range(3))
    quant = (s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
    rows = {"lsq_backward": (_hip.TIME_LSQ_BACKWARD, lambda i: ops.lsq_backward_per_tensor(xs[i % 4], gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)),
            "layernorm site": (_hip.TIME_LAYERNORM, lambda i: ops.residual_layernorm_fake_quant(xs[i % 4], gy, gamma, w, b, 1e-5, quant)),
            "fake_quant": (_hip.TIME_FAKE_QUANT, lambda i: ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4))}
    for name, (which, fn) in rows.items():
        res = {0: [], 1: []}
        for rnd in range(3):
            for wt in (0, 1):
                lib.osq_set_tuning(b"stream_wt", wt); lib.osq_set_tuning(b"fq_nt", 5 if wt else 3)
                res[wt].append(timed(which, fn))
        print(f"{str(shape):18s} {name:16s} nt stores {sorted(res[0])[1]:6.2f} us | write-through {sorted(res[1])[1]:6.2f} us", flush=True)
