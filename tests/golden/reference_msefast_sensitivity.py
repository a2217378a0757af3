#!/usr/bin/env python3
"""Build-container only: how far does the REFERENCE's own AvgMSEFastObserver result move when half of its inputs move by
one ulp?  (The per-tensor search is a bounded Brent iteration on a piecewise objective: one loss that differs in its last
bit can send it down another iterate sequence.)  Measured here: typically 1e-5 relative on (min_val, max_val), 1.7e-2 in 2
of 30 trials -- the bar tests/test_gpu_model.py::test_plain_ptq_flows_match_reference uses for activation quantizers that
sit behind GPU matmuls / LayerNorms (inputs equal to the reference's only to rounding)."""
import torch

import make_golden_model as M

M.import_reference()
from quant_transformer.quantization.observer import AvgMSEFastObserver  # noqa: E402

g = torch.Generator().manual_seed(3)
worst = []
for trial in range(6):
    x = torch.randn(4, 16, 32, generator=g) * 1.3
    x[..., 5] *= 6
    lengths = torch.randint(3, 17, (4,), generator=g)

    def run(xx):
        ob = AvgMSEFastObserver(bit=6, symmetric=False, ch_axis=-1)
        ob(xx, lengths, 1)
        return float(ob.min_val), float(ob.max_val)
    base = run(x)
    devs = []
    for _ in range(5):
        flip = torch.rand(x.shape, generator=g) < 0.5
        r = run(torch.where(flip, torch.nextafter(x, torch.full_like(x, float("inf"))), x))
        devs.append(max(abs(r[0] - base[0]) / abs(base[0]), abs(r[1] - base[1]) / abs(base[1])))
    worst.append(max(devs))
    print(trial, base, ["%.1e" % d for d in devs])
print("worst relative movement:", max(worst))
