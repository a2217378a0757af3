"""Development aid: where the fixed cost of a short timed region goes (graph launch, GPU time, synchronisation wake-up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.fused_check import mk, dev
shape = (256, 128, 768)
g = torch.Generator().manual_seed(1234)
lengths = torch.randint(8, 129, (shape[0],), generator=g).to(dev)
xs = [torch.randn(*shape, device=dev) for _ in range(4)]
for x in xs:
    x[..., 7] *= 20
q = mk()
for steps in (20, 200):
    with torch.no_grad():
        for i in range(20):
            y = q(xs[i % 4], lengths, 1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(3):
                y = q(xs[i % 4], lengths, 1)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for i in range(steps):
                y = q(xs[i % 4], lengths, 1)
        torch.cuda.synchronize()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        for mode in ("sync", "spin"):
            rows = []
            for _ in range(7):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0.record(); graph.replay(); e1.record()
                t1 = time.perf_counter()
                if mode == "spin":
                    while not e1.query():
                        pass
                t2 = time.perf_counter()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                rows.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
            rows.sort(key=lambda r: r[2])
            r = rows[len(rows) // 2]
            print(f"steps {steps:4d} {mode}: launch returns {r[0]:7.1f} us, spin sees the end {r[1]:7.1f}, synchronize returns {r[2]:7.1f} "
                  f"({r[2] / steps:6.2f} per step), GPU events {r[3]:7.1f} ({r[3] / steps:6.2f} per step)", flush=True)
