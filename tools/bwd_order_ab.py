"""LSQ+ backward in the reference's order: chunks per workgroup (osq_set_tuning("bwd_order_chunks", n)) against the order-free kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import outlier_suppression_amd as osq
from outlier_suppression_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for shape in ((32, 128, 768), (32, 384, 768), (32, 128, 3072), (256, 128, 768)):
    x = torch.randn(*shape, generator=g).to(dev)
    gy = torch.randn(*shape, generator=g).to(dev)
    s = torch.tensor([0.05], device=dev); z = torch.tensor([31.0], device=dev)
    f = lambda: ops.lsq_backward_per_tensor(x, gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
    osq.set_strict(False)
    row = [f"order-free {timed(f):6.1f}"]
    osq.set_strict(True)
    ref = None
    for c in (1, 2, 4, 8, 16):
        ops.set_tuning("bwd_order_chunks", c)
        out = f()
        if ref is None:
            ref = out
        assert all(torch.equal(a, b) for a, b in zip(ref, out))
        row.append(f"chunks {c}: {timed(f):6.1f}")
    ops.set_tuning("bwd_order_chunks", 4)
    print(f"{str(shape):>16} us per call (back to back) | " + " | ".join(row), flush=True)
