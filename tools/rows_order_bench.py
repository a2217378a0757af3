import sys, time, torch
sys.path.insert(0, '/root/repo')
"""Development aid: per-channel MSEFast with the rows summed order-free (0) and in the reference's order (8, the default)."""
from outlier_suppression_amd import ops
from outlier_suppression_amd.quantization.observer import MSEFastObserver
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for order in (0, 8):
    ops.set_tuning("mse_rows_order", order)
    for rows, cols in ((768, 768), (3072, 768), (768, 3072), (50265, 768)):
        w = torch.randn(rows, cols, device=dev, generator=g) * 0.05
        ob = MSEFastObserver(bit=4, symmetric=True, ch_axis=0).to(dev)
        ob(w); torch.cuda.synchronize()
        t0 = time.perf_counter(); ob(w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"order {order}: MSEFast per-channel 4-bit sym [{rows},{cols}]: {dt*1e3:.2f} ms, nfev mean {ob.last_nfev.float().mean().item():.1f}", flush=True)
