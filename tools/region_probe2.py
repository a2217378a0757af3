"""Development aid: how long one K-step region (one graph replay between two synchronisations) takes as a function of what
the GPU did directly before it: nothing (idle gap), or N replays issued back to back.  Decides bench.py's pre-roll."""
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
K = int(os.environ.get("K", "20"))
with torch.no_grad():
    for i in range(50):
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
        for i in range(K):
            y = q(xs[i % 4], lengths, 1)
    torch.cuda.synchronize()
    graph.replay(); torch.cuda.synchronize()

    def poll():
        e = torch.cuda.Event(); e.record()
        while not e.query():
            pass
        torch.cuda.synchronize()

    import gc
    gc.collect(); gc.disable()
    for rnd in range(2):
        for pre, gap_ms in ((0, 0.0), (0, 5.0), (3, 0.0), (20, 0.0), (75, 0.0), (300, 0.0)):
            ts = []
            for rep in range(9):
                if gap_ms:
                    time.sleep(gap_ms / 1e3)
                for _ in range(pre):
                    graph.replay()
                poll()
                t0 = time.perf_counter()
                graph.replay()
                poll()
                ts.append((time.perf_counter() - t0) * 1e6 / K)
            ts.sort()
            print(f"round {rnd}: {pre:3d} replays before, idle gap {gap_ms} ms: median {ts[4]:6.2f} us per step, min {ts[0]:6.2f}, max {ts[-1]:6.2f}", flush=True)
