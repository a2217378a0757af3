"""The STRICT switch on the GPU: sums in the reference's own order at any length (csrc/aten_order.h).

The reference's MSEFast loss (`.pow(2).mean()`, quantization/observer.py:420-432) and its LSQ / LSQ+ parameter gradients
(autograd's `sum_to_size`, quantization/util_quant.py:29-67) are torch.sum on the CPU, whose order -- ATen's cascade_sum --
is defined for every length once torch runs on one thread.  With `outlier_suppression_amd.set_strict(True)`
(osq_set_tuning "mse_sum_order" / "bwd_sum_order" = 8) the kernels add in that order on the whole chip, and every number
below is compared BIT FOR BIT:

  * against the oracle (oracle/aten_sum.py::aten_sum_flat, pinned against torch.sum) at lengths round every boundary of
    the cascade -- below one SIMD vector, one level-0 block, one level-1 chunk, one level-2 unit, the point where the
    level step doubles (16.8 M fp32 elements), odd tails;
  * against the REFERENCE's own one-thread run at the site sizes of BASELINE configs[3] (tests/golden/site_size.npz,
    made by tests/golden/make_golden_site_size.py: [32,128,768] masked, [32,12,128,128] probabilities, [32,128,3072] GELU
    outputs; three batches, so the float64 calls are covered; statistics after every call and the evaluation counts).
"""
import numpy as np
from conftest import bits_equal
import pytest
import torch

from _site_size import BWD_CASES, MSE_CASES, bwd_case, checksum, site_input, site_lengths

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()
    return torch.device("cuda:0")


@pytest.fixture()
def strict():
    import outlier_suppression_amd as osq
    osq.set_strict(True)
    yield
    osq.reset_tier()


def N(t):
    return t.detach().cpu().numpy()


# lengths round the boundaries of the cascade: W = 8 lanes, 32 columns, level-0 block 512 elements, level-1 chunk 8192,
# level-2 unit 131072, level 3 beyond; 2^24 + ... : the level step becomes 32
LENGTHS = [1, 5, 7, 8, 9, 31, 32, 33, 100, 511, 512, 513, 8191, 8192, 8193, 8200 + 16 * 512 + 3, 65536, 100001,
           131072, 131072 + 77, 2 * 131072 + 8192 + 512 + 33, 1 << 20, 3145728 + 13, (1 << 24) + 32 * 1024 + 37]


@pytest.mark.parametrize("width", [8, 16])
def test_ordered_backward_equals_oracle_at_any_length(dev, width):
    """scale.grad / zero_point.grad of the LSQ+ backward in autograd's order (four fp32 reductions in ATen's one-thread
    order) against the oracle's restatement, dx on the way, at lengths round every boundary of the cascade."""
    from outlier_suppression_amd import ops
    from oracle import fake_quant_oracle as FQ
    rng = np.random.default_rng(11)
    ops.set_tuning("bwd_sum_order", width)
    try:
        for n in LENGTHS:
            x = (rng.standard_normal(n) * 1.5).astype(np.float32)
            gy = rng.standard_normal(n).astype(np.float32)
            scale, zp, gf = np.float32(0.07), np.float32(29.0), FQ.lsqplus_grad_factor(n, 63)
            xd, gd = torch.from_numpy(x).to(dev), torch.from_numpy(gy).to(dev)
            s = torch.tensor([scale], device=dev)
            z = torch.tensor([zp], device=dev)
            dx, ds, dz = ops.lsq_backward_per_tensor(xd, gd, s, z, 0, 63, ops.PARAM_LSQPLUS, gf)
            rdx, rds, rdz = FQ.lsqplus_backward_per_tensor_reference_order(x, gy, scale, zp, 0, 63, gf, vec=width)
            assert bits_equal(N(dx), rdx), n
            assert np.float32(ds.item()) == rds and np.float32(dz.item()) == rdz, (n, width, ds.item(), rds, dz.item(), rdz)
    finally:
        ops.set_tuning("bwd_sum_order", 0)      # the default tier


def _oracle_order_mean(width):
    from oracle.aten_sum import aten_mean_flat

    def mean(sq):
        sq = np.asarray(sq)
        if sq.dtype == np.float64:
            return aten_mean_flat(sq.reshape(-1), width // 2, np.float64)
        return aten_mean_flat(sq.reshape(-1), width, np.float32)
    return mean


@pytest.mark.parametrize("width", [8, 16])
def test_ordered_msefast_equals_oracle_at_any_length(dev, width):
    """Per-tensor searches (symmetric 1-D: ~25 evaluations; two calls, the second in float64) on flat tensors whose
    lengths sit round the boundaries of the cascade, and on masked 3-D / 4-D sites that are gathered first: ranges and
    evaluation counts equal the oracle's with the loss summed in the same order."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import observer as OBS
    from oracle import observer_oracle as OB
    rng = np.random.default_rng(5)
    ops.set_tuning("mse_sum_order", width)
    OB.MEAN_LIKE_TORCH = _oracle_order_mean(width)
    try:
        # the last length (width 8 only: 20 s of NumPy) puts the float64 call beyond 2^19 rows per column: level step 32
        for n in [100, 8193, 3 * 8192 + 5, 70001, 131072 + 77, 262144 + 8192 + 33, (1 << 20) + 7] + ([(1 << 23) + 4096 * 3 + 21] if width == 8 else []):
            ob = OBS.MSEFastObserver(bit=4, symmetric=True, ch_axis=-1).to(dev)
            st = OB.ObserverState(bit=4, symmetric=True, ch_axis=-1)
            counter = [0]
            evals = 0
            for r in range(2):
                x = (rng.standard_normal(n) * (1.0 + r)).astype(np.float32)
                ob(torch.from_numpy(x).to(dev))
                evals += int(ob.last_nfev.sum().item())
                OB.observe_msefast(st, x, average=False, counter=counter)
                assert bits_equal(N(ob.min_val).astype(np.float64), np.asarray(st.min_val, dtype=np.float64)) and \
                    bits_equal(N(ob.max_val).astype(np.float64), np.asarray(st.max_val, dtype=np.float64)), (n, r, N(ob.max_val), st.max_val)
            assert evals == counter[0], (n, evals, counter[0])
        # (shape, seq_pos, symmetric, view): "split" = [B,h,T,d] seen through [B,T,h,d] memory (quant_bert.py:128-150);
        # "zip" = BART's 3-D probabilities [B*h, T, S] masked with B lengths: remove_padding's zip keeps the first B rows (observer.py:82)
        for shape, seq_pos, sym, view in [((16, 96, 200), 1, False, None), ((4, 6, 50, 40), 2, False, None), ((5, 3, 24, 70), 3, True, None),
                                          ((6, 4, 40, 32), 2, False, "split"), ((12, 20, 36), 1, False, "zip")]:
            ob = OBS.AvgMSEFastObserver(bit=6, symmetric=sym, ch_axis=-1).to(dev)
            st = OB.ObserverState(bit=6, symmetric=sym, ch_axis=-1)
            counter = [0]
            evals = 0
            for r in range(2):
                if view == "split":
                    x = np.transpose((rng.standard_normal((shape[0], shape[2], shape[1], shape[3])) * (1.0 + 0.5 * r)).astype(np.float32), (0, 2, 1, 3))
                else:
                    x = (rng.standard_normal(shape) * (1.0 + 0.5 * r)).astype(np.float32)
                x[..., 3] *= 12
                lengths = rng.integers(1, shape[seq_pos] + 1, size=3 if view == "zip" else shape[0])
                ob(torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev), seq_pos)
                evals += int(ob.last_nfev.sum().item())
                OB.observe_msefast(st, x, lengths, seq_pos, average=True, counter=counter)
                assert bits_equal(N(ob.min_val).astype(np.float64), np.asarray(st.min_val, dtype=np.float64)) and \
                    bits_equal(N(ob.max_val).astype(np.float64), np.asarray(st.max_val, dtype=np.float64)), (shape, r, N(ob.min_val), st.min_val)
            assert evals == counter[0], (shape, evals, counter[0])
    finally:
        OB.MEAN_LIKE_TORCH = None
        ops.set_tuning("mse_sum_order", 8)      # the default


@pytest.mark.parametrize("case", [c[0] for c in MSE_CASES])
def test_msefast_site_size_equals_reference_in_its_summation_order(golden, dev, strict, case):
    """BASELINE configs[3]'s per-tensor activation searches at their real site shapes against the REFERENCE's own run
    (one thread): min_val / max_val after every call -- float64 arithmetic from the second call on where the reference
    switches -- and the cumulative number of loss evaluations, bit for bit."""
    from outlier_suppression_amd.quantization import observer as OBS
    g = golden("site_size")
    name, cls, shape, seq_pos, kind, bit, sym, batches, seed = next(c for c in MSE_CASES if c[0] == case)
    gen = torch.Generator().manual_seed(seed)
    ob = getattr(OBS, cls)(bit=bit, symmetric=sym, ch_axis=-1).to(dev)
    evals = 0
    for r in range(batches):
        x = site_input(gen, shape, kind, r)
        L = site_lengths(gen, shape, seq_pos)
        if checksum(x) != int(g[f"{name}_xsum"][r]):      # another torch build / CPU draws other tensors: the fixture does not apply
            pytest.skip("the seeded input differs from the one the fixture was made with (torch's CPU generator on this host)")
        ob(x.to(dev), L.to(dev), seq_pos)
        evals += int(ob.last_nfev.sum().item())
        got = (float(N(ob.min_val).reshape(-1)[0]), float(N(ob.max_val).reshape(-1)[0]))
        want = (float(g[f"{name}_min"][r]), float(g[f"{name}_max"][r]))
        assert got == want, (name, r, got, want)
        assert evals == int(g[f"{name}_nfev"][r]), (name, r, evals, int(g[f"{name}_nfev"][r]))
    assert ob.one_side_dist == str(g[f"{name}_side"])


@pytest.mark.parametrize("case", [c[0] for c in BWD_CASES])
def test_lsqplus_site_size_gradients_equal_reference_in_its_summation_order(golden, dev, strict, case):
    """LSQ+ forward + backward at BERT-base site sizes against the reference's own autograd run on one thread:
    scale.grad and zero_point.grad bit for bit; y and x.grad through the sums of their bit patterns (they are bit-exact element-wise
    in every mode: tests/test_gpu_parity.py)."""
    from outlier_suppression_amd.quantization import util_quant as U
    g = golden("site_size")
    name, shape, kind, seed = next(c for c in BWD_CASES if c[0] == case)
    x, gy, scale, zp, gf = bwd_case(shape, kind, seed)
    if [checksum(x), checksum(gy)] != [int(v) for v in g[f"{name}_xsum"]]:
        pytest.skip("the seeded input differs from the fixture's (torch's CPU generator on this host)")
    assert bits_equal(scale.numpy(), g[f"{name}_scale"]) and bits_equal(zp.numpy(), g[f"{name}_zp"])
    xd = x.to(dev).requires_grad_(True)
    s = scale.to(dev).requires_grad_(True)
    z = zp.to(dev).requires_grad_(True)
    y = U.fake_quantize_learnableplus_per_tensor_affine_training(xd, s, z, 0, 63, gf)
    y.backward(gy.to(dev))
    assert checksum(y) == int(g[f"{name}_ysum"][0])
    assert checksum(xd.grad) == int(g[f"{name}_dxsum"][0])
    assert bits_equal(N(s.grad), g[f"{name}_dscale"]) and bits_equal(N(z.grad), g[f"{name}_dzp"]), \
        (name, N(s.grad), g[f"{name}_dscale"], N(z.grad), g[f"{name}_dzp"])


@pytest.fixture(params=[8, 16])
def simd_width(request):
    import outlier_suppression_amd as osq
    osq.set_strict(True, simd_width=request.param)
    yield request.param
    osq.reset_tier()                                                # the default tier and width


def test_ordered_rounds_equal_search_by_search(dev, simd_width):
    """osq_msefast_ordered_multi_*: the strict evaluations of several searches as rounds (one launch = one evaluation of
    every unfinished search) against the same searches run one by one (one launch per evaluation): converged ranges,
    evaluation counts and the running statistics after two batches (fp32 call, then the float64 one) equal bit for bit."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver, MSEFastObserver
    g = torch.Generator().manual_seed(5)
    cases = []
    for shape, seq_pos, masked in (((8, 64, 96), 1, True), ((4, 6, 32, 32), 2, True), ((16, 128, 768), 1, True),
                                   ((3, 50, 37), 1, False), ((32, 128, 768), 1, True), ((2, 16, 8), 1, True)):
        xs = []
        for _ in range(2):
            x = torch.randn(*shape, generator=g)
            if len(shape) == 4:
                x = torch.softmax(x * 2, -1)
            else:
                x[..., 3] *= 12
            xs.append(x.to(dev))
        L = (torch.randint(1, shape[seq_pos] + 1, (shape[0],), generator=g) if masked else torch.full((shape[0],), shape[seq_pos])).to(dev)
        cases.append((xs, L, seq_pos))

    def observers():
        return [(AvgMSEFastObserver if i % 2 == 0 else MSEFastObserver)(bit=6 if i % 3 else 4, symmetric=False).to(dev) for i in range(len(cases))]

    one_by_one = observers()
    for b in range(2):
        for ob, (xs, L, sp) in zip(one_by_one, cases):
            ob(xs[b], L, sp)
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    rounds = observers()
    for ob in rounds:
        object.__setattr__(ob, "_defer_ok", True)                  # what the quantizer of an observer pass grants
    with deferred_observation() as sites:
        for b in range(2):
            for ob, (xs, L, sp) in zip(rounds, cases):
                ob(xs[b], L, sp)
            assert len(sites.mse) == len(cases)
            sites.flush()
    for a, b in zip(one_by_one, rounds):
        assert bits_equal(N(a.min_val), N(b.min_val)) and bits_equal(N(a.max_val), N(b.max_val))
        assert bits_equal(N(a.last_nfev), N(b.last_nfev))


@pytest.mark.parametrize("round_groups", [1, 3, 40, 64])
def test_rounds_do_not_depend_on_the_workgroups_share_of_chunk_groups(dev, strict, round_groups):
    """The chunk pipeline holds a batch of up to 16 groups' block sums in LDS and publishes them in one burst
    (csrc/aten_order.h, cascade_chunks_pipelined): a workgroup with 1, 3, 40 or 64 groups (several batches, a partial last one,
    groups past the last chunk) adds the same numbers -- the rounds' results equal the default share's bit for bit, fp32 and
    float64 call, a site with an odd number of chunks and an S = 32 site (beyond 8.4 M elements) included."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver
    if not ops.tunable_build():
        pytest.skip("'mse_round_groups' is a compile-time constant in the release library; tests/test_gpu_tunable_build.py runs this "
                    "test against the -DOSQ_TUNABLE build (libosq_hip_dbg.so) in a subprocess")
    g = torch.Generator().manual_seed(11)
    shapes = [(32, 128, 768), (5, 77, 211), (3, 4097, 31), (36, 128, 2048)]                # 3.1 M; 81 K (odd chunks, open rows); 381 K; 9.4 M (S = 32)
    batches = [[(torch.randn(*sh, generator=g) * (1 + 0.3 * b)).to(dev) for sh in shapes] for b in range(2)]

    def run(groups):
        ops.set_tuning("mse_round_groups", groups)
        try:
            obs = [AvgMSEFastObserver(bit=6, symmetric=False).to(dev) for _ in shapes]
            for ob in obs:
                object.__setattr__(ob, "_defer_ok", True)
            with deferred_observation() as sites:
                for b in range(2):
                    for ob, x in zip(obs, batches[b]):
                        ob(x, None, 1)
                    sites.flush()
            return [(N(ob.min_val).copy(), N(ob.max_val).copy(), N(ob.last_nfev).copy()) for ob in obs]
        finally:
            ops.set_tuning("mse_round_groups", 8)

    want, got = run(8), run(round_groups)
    for sh, a, b in zip(shapes, want, got):
        assert all(bits_equal(u, v) for u, v in zip(a, b)), (sh, round_groups, a, b)


def test_tensors_beyond_the_ordered_capacity_keep_order_free_sums(dev):
    """The reference-order kernels stop at a cascade step of 32 (2^23 rows: 268 M fp32 / 134 M float64 elements, aten_order.h);
    a larger tensor takes the order-free sums for that call -- same numbers as set_strict(False) -- and the default tier is
    back afterwards."""
    import outlier_suppression_amd as osq
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MSEFastObserver
    osq.set_strict(True)
    try:
        assert ops.reference_sum_order("mse") == 8 and ops.reference_sum_order("bwd") == 8
        n = (1 << 28) + 64
        assert ops.ordered_sum_fits(1 << 28, 8) and not ops.ordered_sum_fits(n, 8)
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn(n, device=dev, generator=g)
        gy = torch.randn(n, device=dev, generator=g)
        s = torch.tensor([0.05], device=dev)
        z = torch.tensor([31.0], device=dev)
        got = ops.lsq_backward_per_tensor(x, gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
        assert ops.reference_sum_order("bwd") == 8
        osq.set_strict(False)
        want = ops.lsq_backward_per_tensor(x, gy, s, z, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
        osq.set_strict(True)
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        del gy, got, want
        m = (1 << 27) + 32                                   # float64 calls: half the lanes
        xs = x[:m]
        a = MSEFastObserver(bit=4, symmetric=True, ch_axis=-1).to(dev)
        a(xs); a(xs)
        assert ops.reference_sum_order("mse") == 8
        osq.set_strict(False)
        b = MSEFastObserver(bit=4, symmetric=True, ch_axis=-1).to(dev)
        b(xs); b(xs)
    finally:
        osq.reset_tier()
    assert torch.equal(a.min_val, b.min_val) and torch.equal(a.max_val, b.max_val) and torch.equal(a.last_nfev, b.last_nfev)


def test_small_sites_in_every_layout_equal_oracle_lone_and_in_rounds(dev):
    """Small sites (a few hundred elements: below one level-1 chunk; lengths that may be zero, or no mask at all; one-sided
    data) in the layouts the attention blocks hand over -- [B,h,T,d] and [B,h,d,T] views of [B,T,h,d] memory, dense tensors,
    slices that start 4 bytes off a 16-byte boundary -- searched alone and as the rounds of a deferred forward of several
    sites: both equal the oracle with the loss summed in the reference's order, call after call."""
    from outlier_suppression_amd.quantization import observer as OBS
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    from oracle import observer_oracle as OB
    from conftest import aten_order_mean
    rng = np.random.default_rng(77)
    old = OB.MEAN_LIKE_TORCH
    OB.MEAN_LIKE_TORCH = aten_order_mean
    try:
        for case in range(60):
            sites = []
            for k in range(int(rng.integers(1, 4))):
                B, h, d, T = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.choice([4, 8, 24])), int(rng.integers(2, 40))
                kind = str(rng.choice(["bhtd", "bhdt"]))
                shape, seq_pos = ((B, h, T, d), 2) if kind == "bhtd" else ((B, h, d, T), 3)
                sym = bool(rng.integers(0, 2))
                lone = OBS.AvgMSEFastObserver(bit=6, symmetric=sym, ch_axis=-1).to(dev)
                inround = OBS.AvgMSEFastObserver(bit=6, symmetric=sym, ch_axis=-1).to(dev)
                object.__setattr__(inround, "_defer_ok", True)
                sites.append((kind, shape, seq_pos, sym, rng.random() < 0.5, rng.random() < 0.6, str(rng.choice(["dense", "permuted", "offset"])),
                              lone, inround, OB.ObserverState(bit=6, symmetric=sym, ch_axis=-1)))
            for r in range(2):
                fed = []
                with deferred_observation() as rec:
                    for kind, shape, seq_pos, sym, positive, masked, how, lone, inround, st in sites:
                        B, T = shape[0], shape[seq_pos]
                        x_np = rng.standard_normal(shape).astype(np.float32)
                        if positive:
                            x_np = np.abs(x_np)
                        L_np = None
                        if masked:
                            L_np = rng.integers(0, T + 1, (B,)).astype(np.int64)
                            L_np[int(rng.integers(0, B))] = T
                        if how == "permuted":
                            perm = (0, 2, 1, 3) if kind == "bhtd" else (0, 3, 1, 2)
                            back = (0, 2, 1, 3) if kind == "bhtd" else (0, 2, 3, 1)
                            x = torch.from_numpy(x_np).permute(*perm).contiguous().to(dev).permute(*back)      # [B,T,h,d] memory
                        elif how == "offset":
                            buf = torch.zeros(x_np.size + 1, device=dev)
                            buf[1:] = torch.from_numpy(x_np).reshape(-1).to(dev)
                            x = buf[1:].view(shape)
                        else:
                            x = torch.from_numpy(x_np).to(dev)
                        L = None if L_np is None else torch.from_numpy(L_np).to(dev)
                        lone(x, L, seq_pos)
                        inround(x, L, seq_pos)
                        if L_np is None and how == "permuted":
                            # no mask: the reference searches on x_orig.clone() -- strides preserved -- and torch adds a dense
                            # permuted tensor in MEMORY order (checked against torch on this host: 0 of 200 sums differ from the
                            # memory-order sum, 97 from the logical-order one); the oracle gets the memory image of the view
                            x_np = np.ascontiguousarray(np.transpose(x_np, (0, 2, 1, 3) if kind == "bhtd" else (0, 3, 1, 2)))
                        fed.append((x_np, L_np))
                    assert len(rec.mse) == len(sites)
                    rec.flush()
                for (kind, shape, seq_pos, sym, positive, masked, how, lone, inround, st), (x_np, L_np) in zip(sites, fed):
                    OB.observe_msefast(st, x_np, L_np, seq_pos, average=True)
                    want = (np.asarray(st.min_val, dtype=np.float64), np.asarray(st.max_val, dtype=np.float64))
                    tag = (case, r, kind, shape, sym, positive, how, None if L_np is None else L_np.tolist())
                    assert bits_equal(N(lone.min_val).astype(np.float64), want[0]) and bits_equal(N(lone.max_val).astype(np.float64), want[1]), \
                        str(("lone", tag, N(lone.min_val), N(lone.max_val), want))
                    assert bits_equal(N(inround.min_val).astype(np.float64), want[0]) and bits_equal(N(inround.max_val).astype(np.float64), want[1]), \
                        str(("rounds", tag, N(inround.min_val), N(inround.max_val), want))
    finally:
        OB.MEAN_LIKE_TORCH = old


def test_backward_on_a_dense_permuted_input_follows_memory_order(dev, strict):
    """The reference-order LSQ+ backward (set_strict(True)) on the key layer's layout -- [B,h,d,T] view of [B,T,h,d] memory, grad_out
    handed over contiguous in the LOGICAL layout as autograd may do -- equals the oracle fed the memory image: what torch's
    own autograd computes for such an input (tests/test_oracle_pinning.py::test_autograd_adds_a_dense_permuted_input_in_memory_order)."""
    from outlier_suppression_amd import ops
    from oracle import fake_quant_oracle as FQ
    rng = np.random.default_rng(19)
    for B, T, h, d in ((4, 33, 3, 16), (32, 128, 12, 64), (2, 7, 1, 8)):
        mem = rng.standard_normal((B, T, h, d)).astype(np.float32)
        gmem = rng.standard_normal((B, T, h, d)).astype(np.float32)
        x = torch.from_numpy(mem).to(dev).permute(0, 2, 3, 1)
        gy_logical = torch.from_numpy(gmem).to(dev).permute(0, 2, 3, 1).contiguous()
        scale, zp, gf = np.float32(0.07), np.float32(29.0), FQ.lsqplus_grad_factor(mem.size, 63)
        s = torch.tensor([scale], device=dev)
        z = torch.tensor([zp], device=dev)
        dx, ds, dz = ops.lsq_backward_per_tensor(x, gy_logical, s, z, 0, 63, ops.PARAM_LSQPLUS, gf)
        rdx, rds, rdz = FQ.lsqplus_backward_per_tensor_reference_order(mem.reshape(-1), gmem.reshape(-1), scale, zp, 0, 63, gf, vec=8)
        assert dx.stride() == x.stride()
        assert bits_equal(N(dx.permute(0, 3, 1, 2).contiguous()).reshape(-1), rdx.reshape(-1)), (B, T, h, d)
        assert np.float32(ds.item()) == rds and np.float32(dz.item()) == rdz, (B, T, h, d, ds.item(), rds, dz.item(), rdz)


def _lean_cases():
    g = torch.Generator().manual_seed(505)
    cases = []
    base = torch.randn(8, 64, 96, generator=g)
    base[..., 5] *= 15
    cases.append([base, base * 1.1 + 0.01, base * 0.9])
    # values on a lattice of half-steps of scales the search is likely to visit, with offsets of a few ulps to either side
    lattice = (torch.arange(-40, 41, dtype=torch.float32)[None, :] + 0.5) * torch.tensor([0.05, 0.0625, 0.1, 0.3])[:, None]
    ulps = torch.tensor([0.0, 1e-7, -1e-7, 2e-6, -2e-6, 5e-6, -5e-6, 1e-5, -1e-5, 3e-4, -3e-4, 6e-4, -6e-4])     # round the guard (1e-4 absolute of a tie) on both sides
    tie = (lattice.reshape(-1, 1) * (1.0 + ulps[None, :])).reshape(-1)
    tie = torch.cat([tie, torch.tensor([0.0, -0.0, 1e-30, -1e-30, 3e4, -3e4, 1e-3, 77.7])])
    pad = torch.randn(4 * 32 * 64 - tie.numel(), generator=g) * 2
    adv = torch.cat([tie, pad])[torch.randperm(4 * 32 * 64, generator=g)].reshape(4, 32, 64)
    cases.append([adv, adv.flip(0), adv * 1.003])
    cases.append([torch.randn(2, 16, 8, generator=g), torch.randn(2, 16, 8, generator=g) * 5, torch.randn(2, 16, 8, generator=g)])
    return cases


def test_lean_float64_term_equals_the_oracles_full_chain(dev, strict):
    """csrc/msefast.hip, sq_err_f64_lean: from a site's second call on the reference-order evaluations take the integer
    level from a guarded fp32 quotient (6 fp32 + 5 float64 operations per element instead of 15 float64 ones); an element
    within 1e-4 of a rounding tie takes the exact float64 chain.  The ORACLE knows only the full chain (observer.py:420-432
    in float64 once min_val has turned float64): statistics and evaluation counts of every batch must be EQUAL -- on ordinary
    activations and on data built to sit on and next to the ties of plausible candidate scales (both sides of the guard), far
    outside the clamp range, tiny, and of mixed sign.  Runs against the release library (the lean term is what ships)."""
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver, MSEFastObserver
    from oracle import observer_oracle as OB
    old = OB.MEAN_LIKE_TORCH
    OB.MEAN_LIKE_TORCH = _oracle_order_mean(8)
    try:
        for ci, batches in enumerate(_lean_cases()):
            for cls, bit, sym, avg in ((AvgMSEFastObserver, 6, False, True), (MSEFastObserver, 4, True, False), (AvgMSEFastObserver, 8, False, True)):
                ob = cls(bit=bit, symmetric=sym).to(dev)
                st = OB.ObserverState(bit=bit, symmetric=sym, ch_axis=-1)
                L = torch.full((batches[0].shape[0],), batches[0].shape[1], dtype=torch.int64)
                L[0] = max(1, batches[0].shape[1] // 2)
                counter, evals = [0], 0
                for r, x in enumerate(batches):
                    ob(x.to(dev), L.to(dev), 1)
                    evals += int(ob.last_nfev.sum().item())
                    OB.observe_msefast(st, x.numpy(), L.numpy(), 1, average=avg, counter=counter)
                    assert bits_equal(N(ob.min_val).astype(np.float64), np.asarray(st.min_val, dtype=np.float64)) and \
                        bits_equal(N(ob.max_val).astype(np.float64), np.asarray(st.max_val, dtype=np.float64)), (ci, cls.__name__, bit, r, N(ob.min_val), st.min_val)
                assert evals == counter[0], (ci, cls.__name__, bit, evals, counter[0])
    finally:
        OB.MEAN_LIKE_TORCH = old


def test_lean_float64_term_equals_the_full_chain(dev, strict):
    """The same property inside the library: osq_set_tuning("mse_lean", 0) runs the full float64 chain everywhere, and
    statistics and evaluation counts of every batch must equal the lean form's.  'mse_lean' is a compile-time constant (1) in
    the release library: this form of the test runs against the -DOSQ_TUNABLE build (tests/test_gpu_tunable_build.py)."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver, MSEFastObserver
    if not ops.tunable_build():
        pytest.skip("'mse_lean' is a compile-time constant in the release library (see test_lean_float64_term_equals_the_oracles_full_chain); "
                    "tests/test_gpu_tunable_build.py runs this test against libosq_hip_dbg.so in a subprocess")
    cases = _lean_cases()
    results = {}
    for lean in (1, 0):
        ops.set_tuning("mse_lean", lean)
        try:
            out = []
            for ci, batches in enumerate(cases):
                for cls, bit, sym in ((AvgMSEFastObserver, 6, False), (MSEFastObserver, 4, True), (AvgMSEFastObserver, 8, False)):
                    ob = cls(bit=bit, symmetric=sym).to(dev)
                    L = torch.full((batches[0].shape[0],), batches[0].shape[1], device=dev)
                    L[0] = max(1, batches[0].shape[1] // 2)
                    for x in batches:
                        ob(x.to(dev), L, 1)
                        out.append((N(ob.min_val).copy(), N(ob.max_val).copy(), N(ob.last_nfev).copy()))
            results[lean] = out
        finally:
            ops.set_tuning("mse_lean", 1)
    assert len(results[1]) == len(results[0])
    for i, (a, b) in enumerate(zip(results[1], results[0])):
        assert all(bits_equal(u, v) for u, v in zip(a, b)), (i, a, b)
