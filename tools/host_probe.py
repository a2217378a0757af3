"""Where does a 20-step timed region lose time?  Per-call host times right after a synchronize, with and without
a busy-wait that keeps the core awake, and the same 20 steps replayed from a captured graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.fused_check import mk, dev
g = torch.Generator().manual_seed(1234)
shape = (256, 128, 768)
lengths = torch.randint(8, 129, (shape[0],), generator=g).to(dev)
xs = [torch.randn(*shape, device=dev) for _ in range(4)]
q = mk()
K = 20
with torch.no_grad():
    for i in range(300):
        q(xs[i % 4], lengths, 1)
    torch.cuda.synchronize()
    for prime in (0.0, 0.002, 0.02):
        for rep in range(3):
            torch.cuda.synchronize()
            tp = time.perf_counter()
            while time.perf_counter() - tp < prime:
                pass
            ts = [time.perf_counter()]
            for i in range(K):
                q(xs[i % 4], lengths, 1)
                ts.append(time.perf_counter())
            torch.cuda.synchronize()
            te = time.perf_counter()
            per = [(b - a) * 1e6 for a, b in zip(ts, ts[1:])]
            print(f"prime={prime * 1e3:.0f}ms rep{rep}: total {(te - ts[0]) / K * 1e6:.1f} us/step, host calls us: first {per[0]:.0f} {per[1]:.0f} {per[2]:.0f} {per[3]:.0f} ... median {sorted(per)[K // 2]:.1f} max {max(per):.0f}")
    # graph
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3):
                q(xs[i % 4], lengths, 1)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(K):
                y = q(xs[i % 4], lengths, 1)
        torch.cuda.synchronize()
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gr.replay()
            torch.cuda.synchronize()
            print(f"graph replay of {K} steps: {(time.perf_counter() - t0) / K * 1e6:.1f} us/step")
    except Exception as e:
        print("graph capture failed:", repr(e))
