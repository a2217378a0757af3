"""One strict AvgMSEFast call per shape given on the command line (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import outlier_suppression_amd as osq
from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shape = tuple(int(v) for v in sys.argv[1].split(","))
masked = len(sys.argv) < 3 or sys.argv[2] != "full"
x = torch.randn(*shape, generator=g)
x[..., 5] *= 20
x = x.to(dev)
L = (torch.randint(8, shape[1] + 1, (shape[0],), generator=g) if masked else torch.full((shape[0],), shape[1])).to(dev)
osq.set_strict(True)
ob = AvgMSEFastObserver(bit=6, symmetric=False).to(dev)
ob(x, L, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
ob(x, L, 1)
torch.cuda.synchronize()
t = time.perf_counter() - t0
n = int(ob.last_nfev.sum().item())
print(f"{shape} masked={masked}: {t * 1e3:.2f} ms, {n} evaluations, {t / n * 1e6:.2f} us per evaluation")
