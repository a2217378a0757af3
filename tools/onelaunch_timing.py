"""Development aid: phase stamps of the one-launch masked observation (csrc/observe_onelaunch.h) printed by the
-DOSQ_FINAL_TIMING build (`make -C outlier_suppression_amd/csrc dbg`): selector phases and a few streaming workgroups."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
_hip.LIB_PATH = _hip.LIB_PATH.replace("libosq_hip.so", "libosq_hip_dbg.so")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = torch.randn(256, 128, 768, generator=g)
x[..., [5, 77, 130, 400, 555, 700]] *= 20
x = x.to(dev)
for name, L in (("bench lengths", torch.randint(8, 129, (256,), generator=g).to(dev)), ("all valid", torch.full((256,), 128, dtype=torch.int64, device=dev))):
    mn, mx = torch.tensor(float("inf"), device=dev), torch.tensor(float("-inf"), device=dev)
    for it in range(4):
        print(f"--- {name}, call {it}", flush=True)
        ops.observe_tokens(x, 1, L, True, 0.95, ops.UPDATE_AVERAGE, it, mn, mx, 0, 63, False)
        torch.cuda.synchronize()
