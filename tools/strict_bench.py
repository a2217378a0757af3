"""What the strict switch costs per call (outlier_suppression_amd.set_strict): per-tensor AvgMSEFast searches and the LSQ+
backward, default against strict, at BERT-base site shapes and the BASELINE tensor."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import outlier_suppression_amd as osq
from outlier_suppression_amd import ops
from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


for shape, seq_pos, kind in (((32, 128, 768), 1, "hidden"), ((32, 12, 128, 128), 2, "probs"), ((32, 128, 3072), 1, "hidden")):
    x = torch.randn(*shape, generator=g)
    if kind == "probs":
        x = torch.softmax(x * 2, -1)
    else:
        x[..., 5] *= 20
    x = x.to(dev)
    L = torch.randint(8, shape[seq_pos] + 1, (shape[0],), generator=g).to(dev)
    for strict in (False, True):
        osq.set_strict(strict)
        ob = AvgMSEFastObserver(bit=6, symmetric=False).to(dev)
        ob(x, L, seq_pos)                                            # first call: fp32
        t = wall(lambda: ob(x, L, seq_pos))                          # later calls: float64 where the reference switches
        n = int(ob.last_nfev.sum().item())
        print(f"AvgMSEFast {shape} strict={strict}: {t * 1e3:7.2f} ms per call, {n} evaluations -> {t / n * 1e6:6.2f} us per evaluation", flush=True)
    osq.set_strict(False)
for shape in ((32, 128, 768), (32, 128, 3072), (256, 128, 768)):
    x = torch.randn(*shape, generator=g).to(dev)
    gy = torch.randn(*shape, generator=g).to(dev)
    s = torch.tensor([0.05], device=dev); z = torch.tensor([31.0], device=dev)
    for strict in (False, True):
        osq.set_strict(strict)
        t = wall(lambda: ops.lsq_backward_per_tensor(x, gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4), reps=9)
        print(f"LSQ+ backward {shape} strict={strict}: {t * 1e6:7.1f} us per call (wall, one call per synchronisation)", flush=True)
    osq.set_strict(False)
