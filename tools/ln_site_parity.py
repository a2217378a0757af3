#!/usr/bin/env python3
"""Which LayerNorm is closer to the reference's CPU run -- torch-ROCm's (the eager site) or csrc/layernorm.hip (the
one-launch site)?  Run on the GPU box; reads tests/golden/ln_site.npz (made by running the reference).  Prints, per case
and form: max / mean |LayerNorm output - reference| on the stored samples, the relative scale error, the number of integer
entries that differ from the reference's integer tensor (of 3.1 M), and the site's time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ln_site import CASES, FLOAT_SAMPLES, ln_site_inputs  # noqa: E402
from test_gpu_ln_site import build_site, run_site  # noqa: E402
from outlier_suppression_amd import util_layernorm as UL  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "ln_site.npz"))
print("%-20s %-11s %12s %12s %11s %9s %9s" % ("case", "form", "max|d ln|", "mean|d ln|", "d scale rel", "xq diffs", "us/site"))
for name, cls, eps, with_gamma, quantizer, observer, pct, seed in CASES:
    x, hidden, gamma, beta, L = ln_site_inputs(seed)
    xd, hd, Ld = x.to(dev), hidden.to(dev), L.to(dev)
    for form in ("eager", "one-launch"):
        UL.FUSE_LAYERNORM = form == "one-launch"
        mod, res, q = build_site(cls, eps, with_gamma, quantizer, observer, pct, gamma, beta, dev)
        with torch.no_grad():
            q.enable_observer(); q.disable_fake_quant()
            y = run_site(mod, res, with_gamma, xd, hd, Ld)[:FLOAT_SAMPLES].cpu().numpy()
            d = np.abs(y.astype(np.float64) - g[name + "_ln"])
            ds = abs(float(q.scale.item()) - float(g[name + "_scale"][0])) / float(g[name + "_scale"][0])
            q.disable_observer(); q.enable_fake_quant()
            rs, rz = float(g[name + "_scale"][0]), float(g[name + "_zp"][0])
            q.scale.data.fill_(rs)
            q.zero_point.data.fill_(int(rz) if q.zero_point.dtype == torch.int32 else rz)
            yq = run_site(mod, res, with_gamma, xd, hd, Ld)
            ref_y = (g[name + "_xq"].astype(np.float32) - np.float32(rz)) * np.float32(rs)
            ndiff = int((np.rint((yq.cpu().numpy() - ref_y) / np.float32(rs)) != 0).sum())
            for _ in range(3):
                run_site(mod, res, with_gamma, xd, hd, Ld)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run_site(mod, res, with_gamma, xd, hd, Ld)
            e1.record()
            torch.cuda.synchronize()
        print("%-20s %-11s %12.3e %12.3e %11.2e %9d %9.1f" % (name, form, d.max(), d.mean(), ds, ndiff, e0.elapsed_time(e1) * 20))
UL.FUSE_LAYERNORM = False
