"""Development aid: wall-clock stamps (100 MHz) of the fused observe + fake-quant launch.  `make -C outlier_suppression_amd/csrc dbg` first.
Streaming workgroups: 0 start, 1 prefix sums + row map done, 2 valid tokens reduced + published (wave 0), 3 arrived,
4 scale seen by the poller, 5 after the barrier, 6 end.  Selectors: 0 start, 1 prefix sums done, 2 side selected, 3 end;
within the selection: start, LDS ready, every chunk gathered, histogram levels + scan, list + rank, threshold."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import _hip, ops
_hip.LIB_PATH = _hip.LIB_PATH.replace("libosq_hip.so", "libosq_hip_dbg.so")
from benchlib.common import make_quantizer
dev = torch.device("cuda:0")
mk = lambda: make_quantizer(dev)
lib = _hip.load()
lib.osq_debug_buffer.argtypes = [ctypes.c_void_p]
shape = (256, 128, 768)
g = torch.Generator().manual_seed(1234)
full = "full" in sys.argv[1:]
for a in sys.argv[1:]:
    if a.startswith("hint="):
        ops.set_tuning("select_hint", int(a[5:]))
    if a.startswith("gate="):
        ops.set_tuning("fused_gate", int(a[5:]))
lengths = (torch.full((shape[0],), shape[1]) if full else torch.randint(8, 129, (shape[0],), generator=g)).to(dev)
xs = [torch.randn(*shape, device=dev) for _ in range(4)]
dbg = torch.zeros(260 * 8 + 128, dtype=torch.int64, device=dev)
assert lib.osq_debug_buffer(dbg.data_ptr()) == 0
q = mk()
with torch.no_grad():
    for i in range(10):
        q(xs[i % 4], lengths, 1)
    torch.cuda.synchronize()
    dbg.zero_()
    q(xs[2], lengths, 1)
    torch.cuda.synchronize()
waves = dbg[2080:].view(2, 16, 4).cpu()
d = dbg[:2080].view(260, 8).cpu()
sel = d[256:].reshape(2, 16)
d = d[:256]
t0 = int(d[:, 0][d[:, 0] > 0].min())
us = lambda v: (v.double() - t0) / 100.0
print("selectors (us since first start): ", [[round(float(x), 2) for x in us(d[b, :4])] for b in range(2)])
for side in range(2):
    print(f"select side {side}: prehist/n_below/k_lo/bin count/window", sel[side, 11:16].tolist())
# the same sequence of calls through the three-launch path: must leave the same statistics
ops.set_tuning("fused_step", 0)
q3 = mk()
with torch.no_grad():
    for i in range(10):
        q3(xs[i % 4], lengths, 1)
    q3(xs[2], lengths, 1)
torch.cuda.synchronize()
ops.set_tuning("fused_step", 1)
print("one launch  min/max:", q.observer.min_val.item(), q.observer.max_val.item(), "| three launches:", q3.observer.min_val.item(), q3.observer.max_val.item(),
      "| EQUAL" if (q.observer.min_val.item(), q.observer.max_val.item()) == (q3.observer.min_val.item(), q3.observer.max_val.item()) else "| DIFFERENT")
for side in range(2):
    names = [(0, "start"), (1, "LDS ready"), (2, "gathered+folded"), (7, "level set up"), (8, "level-0 scan done"), (3, "levels done"), (9, "list compacted"), (4, "ranked"), (5, "threshold"), (6, "granule out")]
    print(f"side {side} selection (LDS stamps):", ", ".join(f"{n} {float(us(sel[side, k])):.2f}" for k, n in names))
    print(f"side {side} waves: gathered at", [round(float(x), 1) for x in us(waves[side, :, 0])], "first chunk at", [round(float(x), 1) for x in us(waves[side, :, 3])],
          "rounds", waves[side, :, 1].tolist(), "empty polls", waves[side, :, 2].tolist())
s = d[2:]
for k, name in [(0, "start"), (7, "prefix sums"), (1, "mapped"), (2, "A1 done (wave 0)"), (3, "arrived"), (4, "scale seen"), (5, "after barrier"), (6, "end")]:
    v = us(s[:, k])
    print(f"streaming {name:>18}: min {float(v.min()):7.2f}  median {float(v.median()):7.2f}  max {float(v.max()):7.2f}")
xcd = (torch.arange(254) + 2) % 8
for k, name in [(0, "start"), (1, "mapped"), (2, "A1 done (wave 0)"), (3, "arrived"), (6, "end")]:
    v = us(s[:, k])
    print(f"by XCD {name:>18} (median):", [round(float(v[xcd == x].median()), 2) for x in range(8)])
arr = us(s[:, 3])
print("arrival by XCD (median/max):", [(round(float(arr[(torch.arange(254) + 2) % 8 == x].median()), 1), round(float(arr[(torch.arange(254) + 2) % 8 == x].max()), 1)) for x in range(8)])
order = arr.argsort(descending=True)[:12]
print("latest arrivals: workgroup (arrival, A1-done-wave0, start):", [(int(o) + 2, round(float(arr[o]), 1), round(float(us(s[o, 2])), 1), round(float(us(s[o, 0])), 1)) for o in order])
