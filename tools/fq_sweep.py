"""Development aid: sweep the dense fake-quant kernel's tuning knobs (rocprofv3 gives the kernel times)."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
lib = _hip.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(256, 128, 768, device=dev, generator=g) for _ in range(4)]
s = torch.tensor([0.7], device=dev); z = torch.tensor([31.3], device=dev)
n = xs[0].numel()
for unroll, blocks, nt in itertools.product((2, 4, 8), (1024, 2048, 4096, 8192), (0, 1, 2, 3)):
    lib.osq_set_tuning(b"fq_unroll", unroll); lib.osq_set_tuning(b"fq_max_blocks", blocks); lib.osq_set_tuning(b"fq_nt", nt)
    for i in range(3):
        ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    ys = []
    torch.cuda.synchronize()
    for i in range(12):
        ev[i][0].record(); y = ops.fake_quant_per_tensor(xs[i % 4], s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4); ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print(f"unroll={unroll} blocks={blocks} nt={nt}: median {ts[6]:.2f} us min {ts[0]:.2f} us -> {8*n/ts[6]/1e3:.0f} GB/s", flush=True)
