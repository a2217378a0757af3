"""Development aid: what the dispatch-event clock of the kernel table reads for kernels that move (almost) nothing -- the
floor under the site-size rows of bench.py's `kernels` table (a launch-bound kernel cannot show a bandwidth fraction)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from outlier_suppression_amd import _hip, ops
lib = _hip.load()
dev = torch.device("cuda:0")
s = torch.tensor([0.05], device=dev)
z = torch.tensor([31.0], device=dev)


def timed(fn, which, iters=60, warm=10):
    out = []
    for i in range(iters + warm):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b))
        lib.osq_time_next_launch(which, a, b)
        fn()
        us = ctypes.c_float()
        lib.osq_timing_elapsed_us(a, b, ctypes.byref(us))
        lib.osq_timing_events_destroy(a, b)
        if i >= warm:
            out.append(us.value)
    out.sort()
    return out[len(out) // 2], out[0]


for n in (1024, 64 * 1024, 1024 * 1024, 6 * 1024 * 1024):
    x = torch.randn(n, device=dev)
    med, mn = timed(lambda: ops.fake_quant_per_tensor(x, s, z, 0, 63, ops.PARAM_LSQPLUS, 0.01), _hip.TIME_FAKE_QUANT)
    print(f"fake_quant of {n * 4 / 1024:9.0f} KiB (x2 traffic): median {med:5.2f} us  min {mn:5.2f} us", flush=True)
for B, T, H in ((1, 16, 768), (32, 128, 768), (32, 128, 3072)):
    x = torch.randn(B, T, H, device=dev)
    L = torch.full((B,), T, device=dev, dtype=torch.int64)
    med, mn = timed(lambda: ops.token_minmax(x, 1, L), _hip.TIME_TOKEN_MINMAX)
    print(f"token_minmax [{B},{T},{H}] all valid ({B * T * H * 4 / 1e6:6.2f} MB): median {med:5.2f} us  min {mn:5.2f} us", flush=True)
