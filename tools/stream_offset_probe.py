"""Development aid: does the relative placement of the three streams of the LSQ+ backward (x, grad_out -> grad_x) matter?
All three live in one buffer; grad_out and grad_x are shifted by `off` / 2*`off` bytes against a 2 MiB-aligned start."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
lib = _hip.load()
dev = torch.device("cuda:0")
n = 256 * 128 * 768
seg = n * 4 + (8 << 20)
big = torch.empty(3 * seg + (4 << 20), dtype=torch.uint8, device=dev)
base = (big.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - big.data_ptr()
s = torch.tensor([0.05], device=dev); z = torch.tensor([31.3], device=dev)
ds = torch.empty(1, device=dev); dz = torch.empty(1, device=dev)
ws = _hip.workspace(dev)


def view(off):
    return big[base + off: base + off + n * 4].view(torch.float32)


def timed(x, g, dx, reps=30):
    out = []
    for i in range(reps + 3):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "events")
        lib.osq_time_next_launch(_hip.TIME_LSQ_BACKWARD, a, b)
        _hip.check(lib.osq_lsq_backward_per_tensor(_hip.ptr(x), _hip.ptr(g), _hip.ptr(dx), n, _hip.ptr(s), _hip.ptr(z), ops._zp_type(z), ops.PARAM_LSQPLUS,
                                                   1e-4, 0, 63, _hip.ptr(ds), _hip.ptr(dz), _hip.ptr(ws), _hip.stream_ptr(dev)), "bwd")
        us = ctypes.c_float()
        _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "elapsed")
        lib.osq_timing_events_destroy(a, b)
        if i >= 3:
            out.append(us.value)
    out.sort()
    return out[len(out) // 2], out[0]


offs = [0, 256, 1024, 4096, 4096 + 1024, 65536 + 4096, (1 << 20) + 12288, 3 * 1024]
for rnd in range(2):
    for off in offs:
        x, g, dx = view(0), view(seg - seg % (2 << 20) + off), view(2 * (seg - seg % (2 << 20)) + 2 * off)
        x.normal_(); g.normal_()
        med, mn = timed(x, g, dx)
        print(f"offset {off:8d} B: median {med:6.2f} us  min {mn:6.2f} us  -> {12 * n / med / 1e3:6.0f} GB/s", flush=True)
