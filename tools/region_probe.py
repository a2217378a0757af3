"""Development aid: a 20-step timed region (synchronize -> K module calls -> synchronize) issued three ways:
one graph of K calls, a short graph followed by the rest (the first kernel starts while the host still enqueues),
and the eager loop.  Wall-clock per step as bench.py measures it."""
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


def capture(first, count):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            y = q(xs[i % 4], lengths, 1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(first, first + count):
            y = q(xs[i % 4], lengths, 1)
    torch.cuda.synchronize()
    graph.replay(); graph.replay()
    torch.cuda.synchronize()
    return graph


with torch.no_grad():
    for i in range(50):
        y = q(xs[i % 4], lengths, 1)
    whole = capture(0, K)
    variants = {"one graph": [whole]}
    for head in (1, 2, 4):
        variants[f"{head} + {K - head}"] = [capture(0, head), capture(head, K - head)]

    def region(graphs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graphs is None:
            for i in range(K):
                y = q(xs[i % 4], lengths, 1)
        else:
            for gr in graphs:
                gr.replay()
        done = torch.cuda.Event()
        done.record()
        while not done.query():
            pass
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    variants["eager"] = None
    import gc
    gc.collect(); gc.disable()
    for rnd in range(3):
        for name, graphs in variants.items():
            for _ in range(3):
                region(graphs)
            ts = sorted(region(graphs) for _ in range(9))
            print(f"round {rnd} {name:10s}: median {ts[4] / K:6.2f} us per step, min {ts[0] / K:6.2f}", flush=True)
