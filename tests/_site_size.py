"""Seeded inputs of the site-size fixtures (tests/golden/site_size.npz): shared by the generator that runs the reference
(tests/golden/make_golden_site_size.py) and by the tests that re-draw the same tensors.  The tensors are not stored (12-50 MB
each); torch's CPU generator is deterministic for a given build, and the fixture keeps each tensor's float64 sum."""
import torch

MSE_CASES = (
    # name, class, shape, seq_pos, kind, bit, symmetric, batches, seed
    ("hidden768", "AvgMSEFastObserver", (32, 128, 768), 1, "hidden", 6, False, 3, 4101),
    ("probs128", "AvgMSEFastObserver", (32, 12, 128, 128), 2, "probs", 6, False, 3, 4102),
    ("gelu3072", "AvgMSEFastObserver", (32, 128, 3072), 1, "gelu", 6, False, 2, 4103),
    ("hidden768_running", "MSEFastObserver", (32, 128, 768), 1, "hidden", 4, True, 2, 4104),
)
BWD_CASES = (
    # name, shape, kind, seed
    ("bwd768", (32, 128, 768), "hidden", 4201),
    ("bwd3072", (32, 128, 3072), "gelu", 4202),
)


def site_input(gen, shape, kind, r):
    """One calibration batch of a site: hidden states with six outlier channels (the paper's setting), attention-
    probability-like values (non-negative, piled up near zero: a one-sided search), or GELU-like outputs (two-sided,
    minimum about -0.1).  Built from the generator's normal draws with +, *, / and abs only: torch's vectorised exp / erf
    -- and, measured, even sqrt (Xeon build host vs the GPU box's EPYC, tools/site_input_probe.py) -- differ in the last bit
    between hosts, and the fixture's inputs must be the same tensors everywhere."""
    x = torch.randn(*shape, generator=gen)
    if kind == "probs":
        p = x * x
        return p / (p + (6.0 + 2.0 * r))
    x = x * (1.0 + 0.3 * r)
    idx = torch.randperm(shape[-1], generator=gen)[:6]
    x[..., idx] *= 20.0
    if kind == "gelu":
        x = x * 0.25
        t = x * 1.25
        x = x * (0.5 + 0.5 * (t / (1.0 + t.abs())))                # x * Phi(x) with an algebraic sigmoid
    return x


def site_lengths(gen, shape, seq_pos):
    T = shape[seq_pos]
    L = torch.randint(8, T + 1, (shape[0],), generator=gen)
    L[int(torch.randint(0, shape[0], (1,), generator=gen))] = T
    return L


def bwd_case(shape, kind, seed):
    """(x, grad_out, scale, zero_point, grad_factor) of one LSQ+ backward case: 6-bit asymmetric, 30 % of the range clipped."""
    gen = torch.Generator().manual_seed(seed)
    x = site_input(gen, shape, kind, 0)
    gy = torch.randn(*shape, generator=gen)
    lo, hi = float(x.min()), float(x.max())
    scale = torch.tensor([(hi - lo) / 63.0 * 0.7])
    zp = torch.tensor([float(round(-lo * 0.7 / float(scale)))])
    g = 1.0 / (x.numel() * 63) ** 0.5
    return x, gy, scale, zp, g


def checksum(t):
    """Order-free checksum of an fp32 tensor: the int64 sum of its bit patterns (a float sum would depend on the host's
    thread count -- the very thing these fixtures are about)."""
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())
