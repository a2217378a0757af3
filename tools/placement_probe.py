"""Development aid: does the step time depend on WHERE the tensors were allocated?  One process, several allocation
rounds (the caching allocator is emptied in between, junk blocks of varying size shift the next addresses)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.fused_check import mk, dev

shape = (256, 128, 768)
g = torch.Generator().manual_seed(1234)
lengths = torch.randint(8, 129, (shape[0],), generator=g).to(dev)
steps = 100
junk = []
for rnd in range(8):
    torch.cuda.synchronize()
    if rnd % 2 == 1:
        junk.append(torch.empty((37 + 11 * rnd) * (1 << 20), dtype=torch.uint8, device=dev))   # shift what comes next
    xs = [torch.randn(*shape, device=dev) for _ in range(4)]
    for x in xs:
        x[..., 7] *= 20
    q = mk()
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
        graph.replay(); torch.cuda.synchronize()
        reps = []
        for _ in range(5):
            t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / steps * 1e6)
    print(f"round {rnd}: {min(reps):6.2f} us/step  x at {[hex(x.data_ptr()) for x in xs]}", flush=True)
    del graph, xs, y, q
    torch.cuda.empty_cache()
