"""Development aid: grid-cap sweeps of the streaming kernels on the BASELINE tensor (dispatch-attached event timing).
Power-of-two grids put a thread's strided loads on the same memory channels; this finds the caps that do not."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
dev = torch.device("cuda:0")
lib = _hip.load()
xs = [torch.randn(256, 128, 768, device=dev) for _ in range(4)]
gy = torch.randn(256, 128, 768, device=dev)
mn = torch.tensor(float("inf"), device=dev); mx = torch.tensor(float("-inf"), device=dev)
s = torch.tensor([0.7], device=dev); z = torch.tensor([31.3], device=dev)
def timed(which, fn, reps=24):
    out = []
    for i in range(reps + 3):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b))
        lib.osq_time_next_launch(which, a, b)
        fn(i)
        us = ctypes.c_float(); lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)); lib.osq_timing_events_destroy(a, b)
        if i >= 3: out.append(us.value)
    out.sort()
    return sum(out) / len(out), out[0]
what = sys.argv[1] if len(sys.argv) > 1 else "obs"
if what == "obs":
    for blocks in (256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048):
        ops.set_tuning("obs_blocks", blocks)
        avg, lo = timed(_hip.TIME_OBSERVE_FLAT, lambda i: ops.observe_flat(xs[i % 4], ops.UPDATE_RUNNING, 0, mn, mx, 0, 63, False))
        print(f"obs_blocks={blocks:5d}  avg {avg:6.2f} us  min {lo:6.2f} us  {100.66 / avg:5.2f} TB/s", flush=True)
elif what == "bwd":
    for blocks in (512, 768, 1024, 1280, 1536, 1792, 2048):
        ops.set_tuning("bwd_blocks", blocks)
        avg, lo = timed(_hip.TIME_LSQ_BACKWARD, lambda i: ops.lsq_backward_per_tensor(xs[i % 4], gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
        print(f"bwd_blocks={blocks:5d}  avg {avg:6.2f} us  min {lo:6.2f} us  {301.99 / avg:5.2f} TB/s", flush=True)
elif what == "fq":
    import itertools
    for unroll, blocks in itertools.product((2, 4), (1536, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288)):
        ops.set_tuning("fq_unroll", unroll); ops.set_tuning("fq_max_blocks", blocks)
        avg, lo = timed(_hip.TIME_FAKE_QUANT, lambda i: ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
        print(f"unroll={unroll} fq_max_blocks={blocks:6d}  avg {avg:6.2f} us  min {lo:6.2f} us  {201.33 / avg:5.2f} TB/s", flush=True)
