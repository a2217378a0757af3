"""Does the per-tensor fake-quant kernel run as fast on COLD tensors as on the one tensor a benchmark loop keeps re-reading?

A loop over one [256,128,768] tensor finds its 100 MB input and 100 MB output in the 256 MB Infinity Cache; in a calibration
flow every site tensor is new.  This times osq_fake_quant_per_tensor (C-ABI, pre-allocated buffers) on ONE buffer pair and on
a ROTATION of pairs whose total exceeds the cache several times, next to torch's device copy on the same buffers."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
dev = torch.device("cuda:0")
lib = _hip.load()
st = _hip.stream_ptr(dev)
scale = torch.tensor([0.05], device=dev)
zp = torch.tensor([31], device=dev, dtype=torch.int32)


def timed(fn, n):
    for i in range(min(n, 8)):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n          # us per launch


TOTAL = int(os.environ.get("OSQ_COLD_TOTAL_MB", "3072")) << 20
for shape in ((32, 128, 768), (32, 384, 768), (32, 128, 3072), (256, 128, 768), (32, 384, 3072)):
    n = 1
    for s in shape:
        n *= s
    pair = 8 * n
    K = max(2, TOTAL // pair)
    xs = [torch.randn(n, device=dev) for _ in range(K)]
    ys = [torch.empty(n, device=dev) for _ in range(K)]

    def fq(i, rot):
        j = i % K if rot else 0
        _hip.check(lib.osq_fake_quant_per_tensor(xs[j].data_ptr(), ys[j].data_ptr(), None, n, scale.data_ptr(), zp.data_ptr(),
                                                 ops._zp_type(zp), ops.PARAM_FIXED, 1.0, 0, 63, st), "fq")

    def cp(i, rot):
        j = i % K if rot else 0
        ys[j].copy_(xs[j])

    reps = max(40, 3 * K)
    row = []
    for name, f in (("fake-quant", fq), ("torch copy", cp)):
        hot = min(timed(lambda i: f(i, False), reps) for _ in range(3))
        cold = min(timed(lambda i: f(i, True), reps) for _ in range(3))
        row.append(f"{name}: one pair {hot:6.1f} us ({pair / hot / 1e6:5.2f} TB/s), rotation of {K} pairs {cold:6.1f} us ({pair / cold / 1e6:5.2f} TB/s)")
    print(f"{str(shape):>16} {pair / 1e6:6.1f} MB | " + " | ".join(row), flush=True)
    del xs, ys
    torch.cuda.empty_cache()
