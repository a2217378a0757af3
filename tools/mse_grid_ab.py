"""A/B of the per-tensor MSE grid (MSEObserver / AvgMSEObserver, observer.py:285-409): all candidates in ONE launch
(osq_set_tuning("mse_grid_all", 1), default) against one launch per 32 candidates; results must be identical."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
if os.environ.get("OSQ_LIB_VARIANT"):
    _hip.LIB_PATH = _hip.LIB_PATH.replace("libosq_hip.so", "libosq_hip_%s.so" % os.environ["OSQ_LIB_VARIANT"])
from outlier_suppression_amd.quantization.quantized_module import ObserverDict
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for shape in ((32, 128, 768), (32, 128, 3072), (8, 32, 96))[:int(os.environ.get("OSQ_AB_SHAPES", "3"))]:
    x = torch.randn(*shape, generator=g)
    x[..., 5] *= 20
    x = x.to(dev)
    L = torch.randint(8, shape[1] + 1, (shape[0],), generator=g).to(dev)
    for name, sym in (("AvgMSEObserver", False), ("MSEObserver", True)):
        res = {}
        for mode in (1, 0):
            ops.set_tuning("mse_grid_all", mode)
            ob = ObserverDict[name](bit=6, symmetric=sym).to(dev)
            ob(x, L, 1); torch.cuda.synchronize()
            x2 = x * 1.1
            ts = []
            for rep in range(3):
                ob = ObserverDict[name](bit=6, symmetric=sym).to(dev)
                ob(x, L, 1); torch.cuda.synchronize()
                t0 = time.perf_counter(); ob(x2, L, 1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            dt = sorted(ts)[1]
            res[mode] = (dt, ob.min_val.clone(), ob.max_val.clone(), ts)
        ops.set_tuning("mse_grid_all", 1)
        same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        print(f"{name} sym={sym} {shape}: one launch {res[1][0] * 1e3:7.2f} ms | launch per 32 candidates {res[0][0] * 1e3:7.2f} ms | identical {same} | samples {[round(t * 1e3, 1) for t in res[1][3]]}", flush=True)
