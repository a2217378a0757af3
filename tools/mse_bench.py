import sys, time, torch
sys.path.insert(0, '/root/repo')
from outlier_suppression_amd.quantization.observer import MSEFastObserver, AvgMSEFastObserver
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for rows, cols in ((768, 768), (3072, 768), (768, 3072), (50265, 768)):
    w = torch.randn(rows, cols, device=dev, generator=g) * 0.05
    ob = MSEFastObserver(bit=4, symmetric=True, ch_axis=0).to(dev)
    ob(w); torch.cuda.synchronize()
    t0 = time.perf_counter(); ob(w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"MSEFast per-channel 4-bit sym [{rows},{cols}]: {dt*1e3:.2f} ms, nfev mean {ob.last_nfev.float().mean().item():.1f}")
x = torch.randn(32, 128, 768, device=dev, generator=g); x[..., 5] *= 20
L = torch.randint(8, 129, (32,), device=dev, generator=g)
for sym in (True, False):
    ob = AvgMSEFastObserver(bit=6, symmetric=sym).to(dev)
    ob(x, L, 1); torch.cuda.synchronize()
    t0 = time.perf_counter(); ob(x, L, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"AvgMSEFast per-tensor 6-bit sym={sym} [32,128,768] masked: {dt*1e3:.2f} ms, nfev {int(ob.last_nfev.sum().item())}")
from outlier_suppression_amd.quantization.quantized_module import ObserverDict
for name, kw in (("AvgQuantileObserver", dict(bit=6, symmetric=True)), ("AvgMSEObserver", dict(bit=6, symmetric=False)),
                 ("MSEObserver", dict(bit=6, symmetric=True)), ("LSQPlusObserver", dict(bit=6, symmetric=True))):
    ob = ObserverDict[name](**kw).to(dev)
    args = (x, L, 1) if name != "LSQPlusObserver" else (x,)
    try:
        ob(*args); torch.cuda.synchronize()
        t0 = time.perf_counter(); ob(*args); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} [32,128,768]: {dt*1e3:.2f} ms")
    except Exception as e:
        print(name, "failed:", type(e).__name__, e)
