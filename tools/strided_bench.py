"""Development aid: fake-quant of the head-split attention views ([B,h,T,d] and the key's [B,h,d,T] seen through
[B,T,h*d] memory) against the dense form of the same bytes; checks bit-equality with the dense kernel's result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from outlier_suppression_amd import _hip
if os.environ.get("AB_LIB"):
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), os.environ["AB_LIB"])
from outlier_suppression_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
scale = torch.tensor([0.05], device=dev)
zp = torch.tensor([31.0], device=dev)


import ctypes
lib = _hip.load()


def timed(fn, which, iters=40, warm=6):
    """the kernel's own run time: events on its dispatch packet (osq_time_next_launch)"""
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


for B, T, h, d in ((32, 128, 12, 64), (32, 384, 12, 64), (4, 1024, 16, 64), (256, 128, 12, 64)):
    mems = [torch.randn(B, T, h * d, device=dev, generator=g) for _ in range(6)]
    nbytes = mems[0].numel() * 8
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) % len(mems)
        return mems[k[0]]
    dense = lambda: ops.fake_quant_per_tensor(nxt(), scale, zp, 0, 63, ops.PARAM_LSQPLUS, 0.01)
    qv = lambda: ops.fake_quant_per_tensor(nxt().view(B, T, h, d).permute(0, 2, 1, 3), scale, zp, 0, 63, ops.PARAM_LSQPLUS, 0.01)
    key = lambda: ops.fake_quant_per_tensor(nxt().view(B, T, h, d).permute(0, 2, 3, 1), scale, zp, 0, 63, ops.PARAM_LSQPLUS, 0.01)
    ref = ops.fake_quant_per_tensor(mems[0], scale, zp, 0, 63, ops.PARAM_LSQPLUS, 0.01).view(B, T, h, d)
    y1 = ops.fake_quant_per_tensor(mems[0].view(B, T, h, d).permute(0, 2, 1, 3), scale, zp, 0, 63, ops.PARAM_LSQPLUS, 0.01)
    y2 = ops.fake_quant_per_tensor(mems[0].view(B, T, h, d).permute(0, 2, 3, 1), scale, zp, 0, 63, ops.PARAM_LSQPLUS, 0.01)
    ok = torch.equal(y1, ref.permute(0, 2, 1, 3)) and torch.equal(y2, ref.permute(0, 2, 3, 1)) and y1.is_contiguous()
    rows = [("dense", dense, _hip.TIME_FAKE_QUANT, None)]
    rows += [("q/v view", qv, _hip.TIME_FAKE_QUANT_STRIDED, None), ("key view", key, _hip.TIME_FAKE_QUANT_STRIDED, None)]
    rows += [("q/v view, generic strided kernel", qv, _hip.TIME_FAKE_QUANT_STRIDED, 0), ("key view, generic strided kernel", key, _hip.TIME_FAKE_QUANT_STRIDED, 0)]
    for name, fn, which, knob in rows:
        ops.set_tuning("fq_headsplit", 1 if knob is None else knob)
        med, mn = timed(fn, which)
        ops.set_tuning("fq_headsplit", 1)
        print(f"[{B},{T},{h},{d}] {name:32s}: median {med:7.2f} us  min {mn:7.2f} us -> {nbytes / med / 1e3:7.0f} GB/s   bit-equal {ok}", flush=True)
