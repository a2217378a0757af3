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

# ---- several sites of one forward: immediate (one persistent launch per search) against deferred (shared launches)
from outlier_suppression_amd.quantization.deferred import deferred_observation
from outlier_suppression_amd.quantization import Quantizer
from types import SimpleNamespace as NS
for label, shape, n_sites in (("5 x hidden [32,128,768]", (32, 128, 768), 5), ("10 x hidden", (32, 128, 768), 10), ("2 x probs [32,12,128,128]", (32, 12, 128, 128), 2)):
    seq_pos = 1 if len(shape) == 3 else 2
    xs = [torch.randn(*shape, device=dev, generator=g) * (1 + 0.1 * i) for i in range(n_sites)]
    for x in xs:
        x.select(-1, 5).mul_(20)
    for deferred in (False, True):
        qs = [Quantizer(None, NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1)).to(dev) for _ in xs]
        for q in qs:
            q.enable_observer(); q.disable_fake_quant()
        for call in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if deferred:
                with deferred_observation() as sites:
                    for q, x in zip(qs, xs):
                        q(x, L, seq_pos)
                    sites.flush()
            else:
                for q, x in zip(qs, xs):
                    q(x, L, seq_pos)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        n = sum(int(q.observer.last_nfev.sum().item()) for q in qs)
        print(f"{label:28s} {'deferred ' if deferred else 'immediate'}: {dt * 1e3:7.2f} ms, {n:5d} evaluations, {dt / n * 1e6:6.2f} us each", flush=True)
