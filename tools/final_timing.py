"""Development aid: phase timestamps (100 MHz wall clock) of the token finaliser, taken by thread 0 of the
side-0 workgroup of token_select_kernel: [issue, loads + pass 0, histogram + scan, list + ranking, threshold,
exchange + finish].  Needs the -DOSQ_FINAL_TIMING build: `make -C outlier_suppression_amd/csrc dbg`."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
_hip.LIB_PATH = _hip.LIB_PATH.replace("libosq_hip.so", "libosq_hip_dbg.so")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for B in (32, 256):
    x = torch.randn(B, 128, 768, device=dev, generator=g)
    x[..., 5] *= 20
    L = torch.randint(8, 129, (B,), device=dev, generator=g)
    tmin, tmax, b, t, LL = ops.token_minmax(x, 1, L)
    cur = torch.zeros(64, device=dev)
    for prune in (True, False):
        for rep in range(3):
            ops.token_range_finalize(tmin, tmax, b, t, LL, prune, 0.95, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur)
            torch.cuda.synchronize()
        st = cur[2:2 + 14].view(torch.int64).cpu().tolist()
        d = [(st[i + 1] - st[i]) for i in range(6) if st[i + 1] and st[i]]
        print(f"B={B} prune={prune} phase us:", [x / 100 for x in d], "total", (st[6] - st[0]) / 100, "us")
