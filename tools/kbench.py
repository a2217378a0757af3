#!/usr/bin/env python3
"""Per-kernel micro-benchmark on MI355X (development aid; bench.py is the graded harness).

Times every C-ABI entry point on the BASELINE shapes with HIP events, cycling through enough
distinct buffers to defeat the 256 MiB Infinity Cache ("cold"), and prints algorithmic GB/s
and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outlier_suppression_amd import ops  # noqa: E402

PEAK = 8000.0


def timed(fn, iters, warm=5):
    for i in range(warm):
        fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        ev[i][0].record()
        fn(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []

    def report(name, us_med, us_min, nbytes):
        gbs = nbytes / (us_med * 1e-6) / 1e9
        rows.append({"kernel": name, "us_median": round(us_med, 2), "us_min": round(us_min, 2),
                     "alg_MB": round(nbytes / 1e6, 1), "GBps": round(gbs, 1), "frac_peak": round(gbs / PEAK, 3)})
        print(f"{name:52s} {us_med:9.2f} us (min {us_min:8.2f})  {nbytes / 1e6:8.1f} MB  {gbs:8.1f} GB/s  {100 * gbs / PEAK:5.1f}% peak",
              flush=True)

    def want(name):
        return not args.only or args.only in name

    shape = (256, 128, 768)
    nbuf = 4
    xs = [torch.randn(*shape, device=dev, generator=g) for _ in range(nbuf)]
    for x in xs:
        x[..., [7, 300, 511]] *= 20
    n = xs[0].numel()
    L = torch.randint(8, 129, (shape[0],), device=dev, generator=g)
    Lfull = torch.full((shape[0],), 128, device=dev, dtype=torch.int64)
    valid = int(L.sum().item()) * shape[2]
    s = torch.tensor([0.7], device=dev)
    zi = torch.tensor([31], dtype=torch.int32, device=dev)
    zf = torch.tensor([31.3], device=dev)
    ys = [torch.empty_like(x) for x in xs]
    mn = torch.tensor(float("inf"), device=dev)
    mx = torch.tensor(float("-inf"), device=dev)
    cur = torch.empty(2, device=dev)

    if want("fq_fixed"):
        report("fq per-tensor Fixed [256,128,768]", *timed(lambda i: ops.fake_quant_per_tensor(xs[i % nbuf], s, zi, 0, 63), args.iters), 8 * n)
    if want("fq_lsqplus"):
        report("fq per-tensor LSQ+ [256,128,768]", *timed(lambda i: ops.fake_quant_per_tensor(xs[i % nbuf], s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4), args.iters), 8 * n)
    if want("fq_warm"):
        report("fq per-tensor Fixed, same buffer (cache-warm)", *timed(lambda i: ops.fake_quant_per_tensor(xs[0], s, zi, 0, 63), args.iters), 8 * n)
    if want("observe_flat"):
        report("observe_flat (MinMax, unmasked) [256,128,768]", *timed(lambda i: ops.observe_flat(xs[i % nbuf], ops.UPDATE_RUNNING, 0, mn, mx, 0, 63, False), args.iters), 4 * n)
    if want("token_minmax_masked"):
        report("token_minmax masked (54% valid)", *timed(lambda i: ops.token_minmax(xs[i % nbuf], 1, L), args.iters), 4 * valid)
    if want("token_minmax_full"):
        report("token_minmax all tokens", *timed(lambda i: ops.token_minmax(xs[i % nbuf], 1, Lfull), args.iters), 4 * n)
    if want("finalize"):
        tmin, tmax, B, T, LL = ops.token_minmax(xs[0], 1, L)
        report("token_range_finalize prune p=0.95 (32768 slots)", *timed(lambda i: ops.token_range_finalize(tmin, tmax, B, T, LL, True, 0.95, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur), args.iters), 8 * int(L.sum().item()))
        report("token_range_finalize no prune (32768 slots)", *timed(lambda i: ops.token_range_finalize(tmin, tmax, B, T, LL, False, 1.0, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur), args.iters), 8 * int(L.sum().item()))
        x32 = xs[0][:32]
        tmin2, tmax2, B2, T2, LL2 = ops.token_minmax(x32, 1, L[:32], out=(torch.empty(4096, device=dev), torch.empty(4096, device=dev)))
        report("token_range_finalize prune p=0.95 (4096 slots)", *timed(lambda i: ops.token_range_finalize(tmin2, tmax2, B2, T2, LL2, True, 0.95, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur), args.iters), 8 * int(L[:32].sum().item()))
    if want("lsq_bwd"):
        gy = torch.randn(*shape, device=dev, generator=g)
        report("LSQ+ backward [256,128,768]", *timed(lambda i: ops.lsq_backward_per_tensor(xs[i % nbuf], gy, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4), args.iters), 12 * n)
        del gy
    if want("site"):
        # per-site calibration shapes (launch-bound): [32,128,768] and friends
        for shp, sp in (((32, 128, 768), 1), ((32, 128, 3072), 1), ((32, 12, 128, 128), 2), ((32, 384, 768), 1)):
            x = torch.randn(*shp, device=dev, generator=g)
            Ls = torch.randint(8, shp[sp] + 1, (shp[0],), device=dev, generator=g)
            v = int(Ls.sum().item()) * (x.numel() // shp[0] // shp[sp])
            report(f"site fq Fixed {list(shp)}", *timed(lambda i: ops.fake_quant_per_tensor(x, s, zi, 0, 63), args.iters), 8 * x.numel())
            report(f"site token_minmax {list(shp)} sp={sp}", *timed(lambda i: ops.token_minmax(x, sp, Ls), args.iters), 4 * v)
        xv = torch.randn(32, 128, 768, device=dev, generator=g).view(32, 128, 12, 64).permute(0, 2, 1, 3)
        Ls = torch.randint(8, 129, (32,), device=dev, generator=g)
        v = int(Ls.sum().item()) * 768
        report("site token_minmax [32,12,128,64] view sp=2", *timed(lambda i: ops.token_minmax(xv, 2, Ls), args.iters), 4 * v)
        report("site token_minmax [32,12,64,128] view sp=3", *timed(lambda i: ops.token_minmax(xv.transpose(-1, -2), 3, Ls), args.iters), 4 * v)
    if want("weights"):
        for rows_, cols in ((768, 768), (3072, 768), (30522, 768)):
            w = torch.randn(rows_, cols, device=dev, generator=g) * 0.05
            sc = torch.rand(rows_, device=dev) * 0.01 + 0.001
            zp = torch.zeros(rows_, dtype=torch.int32, device=dev)
            wmn = torch.full((rows_,), float("inf"), device=dev)
            wmx = torch.full((rows_,), float("-inf"), device=dev)
            report(f"fq per-channel weight [{rows_},{cols}]", *timed(lambda i: ops.fake_quant_per_channel(w, sc, zp, 0, -32, 31), args.iters), 8 * w.numel())
            report(f"observe_channels weight [{rows_},{cols}]", *timed(lambda i: ops.observe_channels(w, 0, ops.UPDATE_RUNNING, 0, wmn, wmx, -32, 31, True, ops.QParamSink(sc, zp)), args.iters), 4 * w.numel())
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kbench.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
