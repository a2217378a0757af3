"""Fused one-launch observe+fake-quant vs the three-launch path: bit-equality and timing (run on the GPU box)."""
import ctypes
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from outlier_suppression_amd import _hip, ops
from outlier_suppression_amd.quantization import Quantizer

dev = torch.device("cuda:0")
lib = _hip.load()


def mk(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", sym=False):
    cfg = NS(quantizer=quantizer, observer=observer, bit=6, symmetric=sym, ch_axis=-1)
    q = Quantizer(None, cfg).to(dev)
    q.observer.set_name("bert.encoder.layer.0.output.LayerNorm.layernorm_post_act_fake_quantize.observer")
    if hasattr(q.observer, "set_percentile"):
        q.observer.set_percentile(0.95)
    q.enable_observer()
    q.enable_fake_quant()
    return q


def status():
    st = ctypes.c_int(-1)
    _hip.check(lib.osq_fused_step_status(_hip.ptr(_hip.workspace(dev)), ctypes.byref(st), _hip.stream_ptr(dev)), "status")
    return st.value


def compare(shape, lengths, quantizer, observer, sym, reps=3):
    g = torch.Generator(device=dev).manual_seed(hash(shape) % 1000)
    xs = [torch.randn(*shape, device=dev, generator=g) * (1 + i) for i in range(reps)]
    for x in xs:
        x[..., 5] *= 20
    out = {}
    for fused in (1, 0):
        ops.set_tuning("fused_step", fused)
        q = mk(quantizer, observer, sym)
        ys = []
        with torch.no_grad():
            for x in xs:
                ys.append(q(x, lengths, 1).clone())
        torch.cuda.synchronize()
        out[fused] = (ys, q.observer.min_val.clone(), q.observer.max_val.clone(), q.scale.detach().clone(), q.zero_point.detach().clone())
    ops.set_tuning("fused_step", 1)
    ok = True
    for a, b in zip(out[1][0], out[0][0]):
        ok &= bool(torch.equal(a, b)) or bool(torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))
    for k in range(1, 5):
        ok &= bool(torch.equal(out[1][k].float(), out[0][k].float()))
    st = status()
    print(f"{'OK ' if ok and st == 0 else 'BAD'} shape={shape} {quantizer}/{observer} sym={sym} lens={None if lengths is None else lengths[:6].tolist()} "
          f"scale={out[1][3].item():.6g}/{out[0][3].item():.6g} zp={out[1][4].item()}/{out[0][4].item()} status={st}", flush=True)
    return ok and st == 0


def timing(shape=(256, 128, 768), steps=400):
    g = torch.Generator().manual_seed(1234)
    lengths = torch.randint(8, 129, (shape[0],), generator=g).to(dev)
    xs = [torch.randn(*shape, device=dev) for _ in range(4)]
    for x in xs:
        x[..., 7] *= 20
    valid = int(lengths.sum().item()) * shape[2]
    nbytes = 4 * valid + 8 * xs[0].numel()
    for fused, gate in ((0, 1), (1, 1), (1, 2), (1, 0), (1, 1), (1, 2), (1, 0)):
        ops.set_tuning("fused_step", fused)
        ops.set_tuning("fused_gate", gate)
        q = mk()
        with torch.no_grad():
            for i in range(50):
                q(xs[i % 4], lengths, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                q(xs[i % 4], lengths, 1)
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"fused={fused} gate={gate}: {dt / steps * 1e6:.2f} us/step (host enqueue {th / steps * 1e6:.2f}) -> {nbytes / (dt / steps) / 1e9:.1f} GB/s algorithmic "
              f"= {nbytes / (dt / steps) / 8e12 * 100:.1f} % of 8 TB/s; status={status()}", flush=True)
    ops.set_tuning("fused_step", 1)
    ops.set_tuning("fused_gate", 1)


if __name__ == "__main__":
    allok = True
    g = torch.Generator().manual_seed(0)
    cases = [((8, 32, 768), torch.randint(1, 33, (8,), generator=g)),
             ((32, 128, 768), torch.randint(8, 129, (32,), generator=g)),
             ((256, 128, 768), torch.randint(8, 129, (256,), generator=g)),
             ((256, 128, 768), torch.full((256,), 128)),
             ((8, 384, 768), torch.randint(1, 385, (8,), generator=g)),
             ((4, 16, 1024), torch.tensor([16, 0, 3, 9])),
             ((32, 128, 3072), torch.randint(8, 129, (32,), generator=g)),
             ((2, 8, 4096), torch.tensor([8, 5])),
             ((256, 128, 1024), torch.randint(100, 129, (256,), generator=g)),    # more tokens than the waves can keep: streamed tail
             ((4, 16, 768), torch.zeros(4, dtype=torch.long))]                     # nothing observed
    for shape, lens in cases:
        for quantizer, observer, sym in (("LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False),
                                         ("FixedFakeQuantize", "AvgMinMaxObserver", True),
                                         ("FixedFakeQuantize", "MinMaxObserver", False)):
            if int(lens.sum()) == 0 and observer != "AvgPruneMinMaxObserver":
                continue
            allok &= compare(shape, lens.to(dev), quantizer, observer, sym)
    print("ALL OK" if allok else "FAILURES", flush=True)
    timing()
    sys.exit(0 if allok else 1)
