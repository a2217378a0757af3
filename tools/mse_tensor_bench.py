"""Development aid: AvgMSEFastObserver per-tensor search on the site shapes of a RoBERTa/BERT-base layer (configs[3]):
wall time per call and per loss evaluation, first call (float32 arithmetic) and later calls (float64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from outlier_suppression_amd import _hip
if os.environ.get("OSQ_DBG_LIB"): _hip.LIB_PATH = _hip.LIB_PATH.replace("libosq_hip.so", "libosq_hip_dbg.so")
from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
L = torch.randint(8, 129, (32,), device=dev, generator=g)
sites = [("hidden [32,128,768]", (32, 128, 768), 1, False), ("fc1 [32,128,3072]", (32, 128, 3072), 1, False),
         ("q view [32,12,128,64]", (32, 12, 128, 64), 2, False), ("probs [32,12,128,128]", (32, 12, 128, 128), 2, True)]
for name, shape, seq_pos, probs in sites:
    x = torch.rand(*shape, device=dev, generator=g) if probs else torch.randn(*shape, device=dev, generator=g)
    if not probs:
        x.select(-1, 5).mul_(20)
    ob = AvgMSEFastObserver(bit=6, symmetric=False).to(dev)
    for call in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ob(x, L, seq_pos); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n = int(ob.last_nfev.sum().item())
        print(f"{name:24s} call {call} ({'f32' if call == 0 else 'f64'}): {dt * 1e3:7.2f} ms, {n:4d} evaluations, {dt / max(n, 1) * 1e6:6.2f} us each", flush=True)
