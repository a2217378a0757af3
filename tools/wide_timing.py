"""Development aid: per-workgroup phase stamps of wide_hist_kernel (needs `make -C csrc dbg`)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
_hip.LIB_PATH = _hip.LIB_PATH.replace("libosq_hip.so", "libosq_hip_dbg.so")
lib = _hip.load()
lib.osq_debug_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(256, 128, 768, device=dev, generator=g); x[..., 5] *= 20
L = torch.randint(8, 129, (256,), device=dev, generator=g)
dbg = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
lib.osq_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
cur = torch.zeros(64, device=dev)
for rep in range(3):
    tmin, tmax, b, t, LL = ops.token_minmax(x, 1, L)
    ops.token_range_finalize(tmin, tmax, b, t, LL, True, 0.95, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur)
    torch.cuda.synchronize()
d = dbg.view(64, 8).cpu()
for k in range(3):
    ph = d[:, k + 1] - d[:, k]
    print("phase", k, "->", k + 1, "min", ph.min().item(), "median", ph.median().item(), "max", ph.max().item())
print("list sizes", d[0, 6].item(), d[0, 7].item())
