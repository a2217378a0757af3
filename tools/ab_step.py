"""Development aid: A/B of several builds of libosq_hip.so on ONE box (boxes differ by a few percent, so variants are
only comparable inside one gpurun call).  python tools/ab_step.py libA.so libB.so ...  (paths relative to the package
directory); every library is measured in its own process, three interleaved rounds, on the bench step
([256,128,768], bench lengths): graph replay of 200 captured module calls and the kernel's own launch duration."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib_name, full):
    import ctypes
    import time
    import torch
    from outlier_suppression_amd import _hip
    label = lib_name
    lib_name, _, knobs = lib_name.partition(":")          # libosq_hip.so:fused_deal=2,fused_gate=1
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), lib_name)
    from benchlib.common import make_quantizer
    dev = torch.device("cuda:0")
    mk = lambda: make_quantizer(dev)
    lib = _hip.load()

    def status():
        st = ctypes.c_int(-1)
        _hip.check(lib.osq_fused_step_status(_hip.ptr(_hip.workspace(dev)), ctypes.byref(st), _hip.stream_ptr(dev)), "status")
        return st.value
    for kv in filter(None, knobs.split(",")):
        key, _, val = kv.partition("=")
        assert lib.osq_set_tuning(key.encode(), int(val)) == 0, kv
    shape = tuple(int(v) for v in os.environ.get("AB_SHAPE", "256x128x768").split("x"))
    g = torch.Generator().manual_seed(1234)
    lengths = (torch.full((shape[0],), shape[1]) if full else torch.randint(8, shape[1] + 1, (shape[0],), generator=g)).to(dev)
    xs = [torch.randn(*shape, device=dev) for _ in range(4)]
    for x in xs:
        x[..., 7] *= 20
    q = mk()
    steps = 200
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
            for i in range(steps):
                y = q(xs[i % 4], lengths, 1)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        reps = []
        for _ in range(7):
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / steps * 1e6)
        pairs = []
        for _ in range(100):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "events")
            pairs.append((a, b))
        for i in range(100):
            lib.osq_time_next_launch(_hip.TIME_FUSED_STEP, *pairs[i])
            y = q(xs[i % 4], lengths, 1)
        torch.cuda.synchronize()
        ks = []
        for a, b in pairs:
            us = ctypes.c_float()
            _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "elapsed")
            ks.append(us.value)
    ks.sort()
    reps.sort()
    print(f"{label:28s} graph us/step min {reps[0]:6.2f} med {reps[len(reps) // 2]:6.2f} | kernel us med {ks[len(ks) // 2]:6.2f} min {ks[0]:6.2f} | status {status()}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], len(sys.argv) > 3 and sys.argv[3] == "full")
    else:
        libs = [a for a in sys.argv[1:] if a != "full"]
        full = ["full"] if "full" in sys.argv else []
        for rnd in range(3):
            for name in libs:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name] + full, check=False)
