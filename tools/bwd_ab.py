"""Development aid: grid caps of the three-stream kernels (dense LSQ+ backward "bwd_blocks", LayerNorm site "ln_blocks")
A/B inside one process, every launch timed by the events on its own dispatch packet, three interleaved rounds."""
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


shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(256, 128, 768), (32, 128, 768), (32, 128, 3072), (32, 384, 768)]
for shape in shapes:
    xs = [torch.randn(*shape, device=dev, generator=g) for _ in range(4)]
    gy = torch.randn(*shape, device=dev, generator=g)
    H = shape[-1]
    gamma, w, b = (torch.randn(H, device=dev, generator=g) for _ in range(3))
    quant = (s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
    rows = {"lsq_backward": (b"bwd_blocks", (512, 768, 1024, 1152, 1280, 1408, 1536, 1664, 1792, 2048), _hip.TIME_LSQ_BACKWARD,
                             lambda i: ops.lsq_backward_per_tensor(xs[i % 4], gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)),
            "layernorm site": (b"ln_blocks", (512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192), _hip.TIME_LAYERNORM,
                               lambda i: ops.residual_layernorm_fake_quant(xs[i % 4], gy, gamma, w, b, 1e-5, quant))}
    for name, (knob, values, which, fn) in rows.items():
        res = {v: [] for v in values}
        for rnd in range(3):
            for v in values:
                assert lib.osq_set_tuning(knob, v) == 0
                res[v].append(timed(which, fn))
        print(f"{str(shape):18s} {name:16s} " + "  ".join(f"{v}: {sorted(r)[1]:.2f}" for v, r in res.items()), flush=True)
