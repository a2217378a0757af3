"""Shared by the CPU and GPU MSEFast row tests: the seeded weights of tests/golden/msefast_rows.npz and the bounds on
how far exact-sum searches may be from the reference's (whose loss is an fp32 torch sum in the build machine's order)."""
import numpy as np

from oracle import observer_oracle as OB

MSEFAST_ROW_BOUNDS = {   # measured (512 rows each): median 2e-5 / 1.4e-5 / 4e-7, p99 7e-5 / 2e-4 / 3e-5, max 1.8e-3, x_quant 1.5e-5 .. 3.8e-5
    "median_rel": 1e-4, "p99_rel": 1e-3, "max_rel": 5e-2, "xquant_mismatch": 2e-4}   # max over all 2048 rows of w3072: 1.3e-2 (one row, a neighbouring step of its staircase)


def msefast_row_weights(seed, rows, cols):
    """Same recipe as tests/golden/make_golden.py::msefast_row_weights (torch's CPU generator is reproducible)."""
    import torch
    return (torch.randn(rows, cols, generator=torch.Generator().manual_seed(seed)) * 0.05).numpy()


def msefast_row_deviation(w, mn, mx, ref_min, ref_max, quant_min, quant_max):
    """How far per-row ranges are from the reference's: relative range error and the fraction of x_quant entries that
    differ when each row is quantised with the scale either range implies."""
    rel = np.abs(mx - ref_max) / ref_max
    s_a, _ = OB.calculate_qparams(mn, mx, quant_min, quant_max, True)
    s_b, _ = OB.calculate_qparams(ref_min, ref_max, quant_min, quant_max, True)
    xa = np.clip(np.round(w / s_a[:, None]), quant_min, quant_max)
    xb = np.clip(np.round(w / s_b[:, None]), quant_min, quant_max)
    return rel, float((xa != xb).mean()), float((s_a != s_b).mean())


