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


def timed(fn, iters=60, warm=8):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


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
    for name, fn in (("dense", dense), ("q/v view", qv), ("key view", key)):
        med, mn = timed(fn)
        print(f"[{B},{T},{h},{d}] {name:9s}: median {med:7.2f} us  min {mn:7.2f} us -> {nbytes / med / 1e3:7.0f} GB/s   bit-equal {ok}", flush=True)
