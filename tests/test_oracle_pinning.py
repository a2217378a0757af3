"""Pin the oracle's restatements of third-party arithmetic against the installed libraries.

torch.quantile (linear) and scipy's bounded minimiser are not in the reference
tree; the oracle restates them (oracle/observer_oracle.py, oracle/brent.py).
"""
import numpy as np
import pytest
import torch
from scipy.optimize import minimize_scalar

from oracle.observer_oracle import torch_quantile_linear, fma_f32
from oracle.brent import BoundedBrent, minimize_bounded


def test_quantile_matches_torch_bit_for_bit():
    rng = np.random.default_rng(0)
    qs = [1.0, 0.99, 0.97, 0.9967, 0.71, 0.95, 0.9, 0.85, 0.7, 1 - 0.0033 * 7, 0.5, 0.0, 0.123]
    for trial in range(1500):
        n = int(rng.integers(1, 6000))
        v = (np.abs(rng.standard_normal(n)) * rng.choice([1, 20, 0.01])).astype(np.float32)
        q = float(rng.choice(qs))
        want = np.float32(torch.quantile(torch.from_numpy(v), q).item())
        assert torch_quantile_linear(v, q) == want, (n, q)


def test_fma_is_single_rounded():
    import fractions
    rng = np.random.default_rng(1)
    for _ in range(2000):
        a, b, c = (np.float32(v) for v in rng.standard_normal(3) * rng.choice([1e-3, 1, 1e3]))
        exact = fractions.Fraction(float(a)) * fractions.Fraction(float(b)) + fractions.Fraction(float(c))
        r = fma_f32(a, b, c)
        lo, hi = np.nextafter(r, np.float32(-np.inf)), np.nextafter(r, np.float32(np.inf))
        err = abs(fractions.Fraction(float(r)) - exact)
        assert err <= abs(fractions.Fraction(float(lo)) - exact) and err <= abs(fractions.Fraction(float(hi)) - exact)


FUNCS = [
    (lambda x: (x - 2.0) ** 2, (0.0, 5.0)),
    (lambda x: np.sin(x) + 0.1 * x, (0.0, 10.0)),
    (lambda x: abs(x - 0.3) + 0.01 * x * x, (-1.0, 1.0)),
    (lambda x: np.cosh(x - 1.5), (0.1, 4.0)),
    (lambda x: -x, (0.0, 1.0)),            # minimum at the upper bound
    (lambda x: x, (0.25, 3.0)),            # minimum at the lower bound
    (lambda x: np.floor(x * 7) / 7.0 + (x - 1) ** 2 * 0.05, (0.0, 2.0)),   # staircase, like a quantisation loss
]


@pytest.mark.parametrize("idx", range(len(FUNCS)))
def test_bounded_brent_matches_scipy_iterates(idx):
    f, (lo, hi) = FUNCS[idx]
    seen = []

    def wrapped(x):
        seen.append(float(x))
        return float(f(x))
    res = minimize_scalar(wrapped, bounds=(lo, hi), method="Bounded")
    mine = []

    def wrapped2(x):
        mine.append(float(x))
        return float(f(x))
    x, fx, nfev = minimize_bounded(wrapped2, lo, hi)
    assert nfev == res.nfev
    assert mine == seen            # identical iterate sequence
    assert x == float(res.x) and fx == float(res.fun)


@pytest.mark.parametrize("idx", range(len(FUNCS)))
def test_bounded_brent_matches_scipy_with_float32_values(idx):
    """The reference's objective returns np.float32 (observer.py:431-432): scipy then subtracts function values in
    float32.  Offsets spread the values over more than a factor of two so that those subtractions do round."""
    f, (lo, hi) = FUNCS[idx]
    for offset in (0.0, 3.0, -7.5):
        seen, mine = [], []

        def wrapped(x):
            seen.append(float(x))
            return np.float32(f(x) + offset)
        res = minimize_scalar(wrapped, bounds=(lo, hi), method="Bounded")

        def wrapped2(x):
            mine.append(float(x))
            return np.float32(f(x) + offset)
        x, fx, nfev = minimize_bounded(wrapped2, lo, hi, f32_values=True)
        assert nfev == res.nfev
        assert mine == seen
        assert x == float(res.x) and fx == float(res.fun)


def test_bounded_brent_ask_tell_and_maxiter():
    st = BoundedBrent(0.0, 1.0, maxiter=5)
    x = st.start()
    n = 0
    while x is not None:
        n += 1
        x = st.tell((x - 0.77) ** 2)
    assert n == 5 and st.done


def test_aten_sum_order():
    """oracle/aten_sum.py restates the ORDER in which torch's CPU kernel adds a contiguous fp32 vector (cascade_sum,
    dispatched at AVX2 width: 8 lanes, also on AVX-512 machines) -- the one machine-dependent step of the reference's
    MSEFast loss (observer.py:420-432).  Pinned against torch.sum / torch.mean themselves, bit for bit, over the row
    lengths of the BASELINE weights and every awkward size around the vector / ILP / cascade boundaries."""
    import torch
    from oracle.aten_sum import aten_mean_f32, aten_sum_f32
    if torch.backends.cpu.get_cpu_capability() not in ("AVX2", "AVX512"):
        pytest.skip("torch's sum kernel runs at another vector width on this host")
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 100, 127, 128, 129, 200, 255, 256, 257, 511, 512, 513, 768, 769,
             1000, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 3100, 4095, 4096, 4097, 8191, 8192, 10000, 16384, 20000, 32767]
    for n in sizes:
        for rep in range(4):
            x = (rng.standard_normal(n) ** 2 * rng.choice([1e-6, 1e-3, 1.0, 1e3])).astype(np.float32)
            t = torch.from_numpy(x)
            assert np.float32(t.sum().item()) == aten_sum_f32(x, 8), (n, rep)
            assert np.float32(t.mean().item()) == aten_mean_f32(x, 8), (n, rep)
    # float64 (a per-tensor MSEFast observer's second call on, observer.py:524,549): 256-bit vectors hold FOUR doubles; and with
    # a one-thread pool (how tests/golden/make_golden.py runs the reference) the serial order holds beyond 32768 elements
    from oracle.aten_sum import aten_mean, aten_sum
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for n in [1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 512, 1000, 2048, 4097, 12345, 32767, 32768, 50000, 100001]:
            x = (rng.standard_normal(n) ** 2 * rng.choice([1e-6, 1.0, 1e3])).astype(np.float64)
            t = torch.from_numpy(x)
            assert np.float64(t.sum().item()) == aten_sum(x, 4, np.float64, serial_only=False), n
            assert np.float64(t.mean().item()) == aten_mean(x, 4, np.float64, serial_only=False), n
            xf = x.astype(np.float32)
            assert np.float32(torch.from_numpy(xf).sum().item()) == aten_sum(xf, 8, np.float32, serial_only=False), n
    finally:
        torch.set_num_threads(threads)
    x = (rng.standard_normal((9, 3072)) ** 2).astype(np.float32)           # leading axes are independent rows
    assert np.array_equal(aten_sum_f32(x, 8), np.array([torch.from_numpy(r).sum().item() for r in x], dtype=np.float32))


def test_aten_sum_flat_any_length():
    """oracle/aten_sum.py::aten_sum_flat walks the cascade block-wise (what the strict kernels do, csrc/aten_order.h) and
    holds for any length on a one-thread host: equal to the row-by-row restatement and to torch.sum itself, fp32 in 8 / 16
    lanes and float64 in 4, at lengths round every boundary of the cascade -- including the one where the level step
    doubles (beyond 2^19 rows per column: 16.8 M fp32 elements)."""
    import torch
    from oracle.aten_sum import aten_sum, aten_sum_flat
    if torch.backends.cpu.get_cpu_capability() not in ("AVX2", "AVX512"):
        pytest.skip("torch's sum kernel runs at another vector width on this host")
    rng = np.random.default_rng(3)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for n in [1, 5, 7, 8, 9, 31, 32, 33, 100, 511, 512, 513, 8191, 8192, 8193, 16384 + 515, 65536, 100001, 131072 + 77,
                  2 * 131072 + 8192 + 512 + 33, 1 << 20, 3145728 + 13, (1 << 24) + 32 * 1024 + 37]:
            x = (rng.standard_normal(n) ** 2 * rng.choice([1e-3, 1.0, 1e3])).astype(np.float32)
            got = aten_sum_flat(x, 8, np.float32)
            assert got == np.float32(torch.from_numpy(x).sum().item()), n
            xd = x.astype(np.float64) ** 1.5
            assert aten_sum_flat(xd, 4, np.float64) == np.float64(torch.from_numpy(xd).sum().item()), n
            if n <= 300000:
                assert got == aten_sum(x, 8, np.float32, serial_only=False), n
                assert aten_sum_flat(x, 16, np.float32) == aten_sum(x, 16, np.float32, serial_only=False), n
                assert aten_sum_flat(xd, 8, np.float64) == aten_sum(xd, 8, np.float64, serial_only=False), n
    finally:
        torch.set_num_threads(threads)


def test_exact_sum():
    """oracle.observer_oracle.exact_sum beyond 200 000 elements (a pairwise tree of error-free two-sums) against
    math.fsum, on squared-error-like data: non-negative, six decades of dynamic range, fp32-valued and full float64."""
    import math
    from oracle.observer_oracle import exact_sum
    rng = np.random.default_rng(17)
    for n in [200001, 262144, 300007, 1 << 20, 1700003]:
        for kind in range(3):
            a = rng.standard_normal(n) ** 2 * 10.0 ** rng.integers(-6, 1, size=n)
            if kind == 1:
                a = a.astype(np.float32).astype(np.float64)
            if kind == 2:
                a[rng.integers(0, n, size=n // 2)] = 0.0
            assert exact_sum(a) == math.fsum(a.tolist()), (n, kind)


def test_torch_adds_a_dense_permuted_view_in_memory_order():
    """What an UNMASKED observer call of the reference sums over: `x_orig.clone()` keeps the strides of a dense permuted
    view (the key layer [B,h,d,T] of [B,T,h,d] memory), and torch's CPU reduction adds such a tensor in memory order --
    the order the kernels read it in, and the order the oracle must be handed (observer_oracle._prepare)."""
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(0)
    differs_from_logical = 0
    for trial in range(100):
        B, T, h, d = 8, 3 + trial % 5, 1 + trial % 3, 8
        mem = torch.randn(B, T, h, d, generator=g).abs()
        x = mem.permute(0, 2, 3, 1).clone()                   # strides preserved
        assert x.stride() == mem.permute(0, 2, 3, 1).stride()
        sq = x.abs().pow(2)
        got = sq.mean().item()
        assert got == sq.permute(0, 3, 1, 2).contiguous().view(-1).mean().item()
        differs_from_logical += int(got != sq.contiguous().view(-1).mean().item())
    assert differs_from_logical > 10


def test_autograd_adds_a_dense_permuted_input_in_memory_order():
    """scale.grad / zero_point.grad of an LSQ+ quantizer whose input is a dense permuted view (the key layer [B,h,d,T] of
    [B,T,h,d] memory): autograd's sum_to_size reductions run over tensors with the input's strides, in MEMORY order -- the
    oracle's reference-order backward reproduces torch bit for bit when handed the memory image, not the logical one.
    (The kernels read the saved input in memory order and lay grad_out out like it: ops._like_layout.)"""
    from oracle import fake_quant_oracle as FQ
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1)
    differs_from_logical = 0
    for trial in range(12):
        B, T, h, d = 4, 20 + trial, 3, 16
        mem = torch.randn(B, T, h, d, generator=g)
        gmem = torch.randn(B, T, h, d, generator=g)
        x, gy = mem.permute(0, 2, 3, 1), gmem.permute(0, 2, 3, 1)
        s = torch.tensor([0.07], requires_grad=True)
        z = torch.tensor([29.0], requires_grad=True)
        gf = 1e-3
        scaled = lambda t: (t - t * gf).detach() + t * gf              # grad_scale: value kept, gradient times gf
        zz = scaled((z.round() - z).detach() + z)
        ss = scaled(s)
        u = x / ss + zz
        u = torch.clamp((u.round() - u).detach() + u, 0, 63)
        ((u - zz) * ss).backward(gy)
        in_memory = FQ.lsqplus_backward_per_tensor_reference_order(mem.numpy().reshape(-1), gmem.numpy().reshape(-1), np.float32(0.07),
                                                                   np.float32(29.0), 0, 63, gf, vec=8)
        logical = FQ.lsqplus_backward_per_tensor_reference_order(x.contiguous().numpy().reshape(-1), gy.contiguous().numpy().reshape(-1),
                                                                 np.float32(0.07), np.float32(29.0), 0, 63, gf, vec=8)
        assert np.float32(s.grad.item()) == in_memory[1] and np.float32(z.grad.item()) == in_memory[2], trial
        differs_from_logical += int(np.float32(s.grad.item()) != logical[1] or np.float32(z.grad.item()) != logical[2])
    assert differs_from_logical > 0
