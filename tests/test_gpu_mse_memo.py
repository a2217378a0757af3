"""The LOSS MEMO of the per-tensor MSEFast searches (csrc/msefast.hip, tensor_search_advance; include/osq_hip.h).

loss_fx (quantization/observer.py:423-432) is a pure function of the tensor and of the pair it hands to the fake-quant, and scipy's
bounded search asks for the same pair many times (observer.py:434-446: the shift of a FIXED range moves, the scale stays, the
integer zero point changes once per quantisation step; observer.py:469-475 repeats an earlier inner search entirely).  A search
answers those from the pairs it has already streamed.  What must hold, and is checked here on every entry point that uses the
memo: converged range, running statistics and nfev are THE SAME BITS with the memo and with osq_set_tuning("mse_memo", 0), the
memo does answer evaluations (hits > 0 on two-sided data), and nfev = streamed + answered."""
import numpy as np
import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()
    return torch.device("cuda:0")


def N(t):
    return t.detach().cpu().numpy()


def _tensors(dev):
    g = torch.Generator().manual_seed(17)
    two_sided = torch.randn(8, 64, 96, generator=g)
    two_sided[..., 5] *= 9
    clipped = torch.clamp(torch.randn(8, 64, 96, generator=g) * 3, min=-0.17)         # GELU-like: the lower bound clips nearly every shift
    shifted = torch.randn(4, 32, 768, generator=g) * 2 + 3
    big = torch.randn(16, 128, 768, generator=g)
    big[..., 11] *= 20
    L8 = torch.randint(1, 65, (8,), generator=g)
    return [("two_sided", two_sided.to(dev), None, 1), ("two_sided_masked", two_sided.to(dev), L8.to(dev), 1),
            ("clipped", clipped.to(dev), None, 1), ("shifted", shifted.to(dev), None, 1), ("big", big.to(dev), None, 1)]


def _search(ops, x, lengths, seq_pos, float64_input):
    cur = ops.batch_minmax(x, lengths, seq_pos)
    r = ops.msefast_tensor_begin(x, cur, lengths, seq_pos, 0, 63, False, "no", True, float64_input)
    ops.msefast_tensor_run(r, None, True)
    stats = N(ops.msefast_tensor_stats(r))
    mn = torch.full((1,), float("inf"), dtype=torch.float64, device=x.device)
    mx = torch.full((1,), float("-inf"), dtype=torch.float64, device=x.device)
    nfev = ops.msefast_tensor_commit(r, ops.UPDATE_RUNNING, 0, mn, mx)
    return N(mn), N(mx), int(nfev.item()), stats


@pytest.mark.parametrize("tier", ["strict8", "strict16", "order_free_streaming"])
def test_memo_on_equals_memo_off(dev, tier):
    import outlier_suppression_amd as osq
    from outlier_suppression_amd import ops
    try:
        if tier.startswith("strict"):
            osq.set_strict(True, simd_width=int(tier[6:]))
        else:
            osq.set_strict(False)
            ops.set_tuning("mse_resident", 0)                   # one streaming launch per evaluation: the entry points with the memo
        total_hits = 0
        for name, x, L, sp in _tensors(dev):
            for f64 in (False, True):
                ops.set_tuning("mse_memo", 1)
                a = _search(ops, x, L, sp, f64)
                ops.set_tuning("mse_memo", 0)
                b = _search(ops, x, L, sp, f64)
                assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1]) and a[2] == b[2], (tier, name, f64, a, b)
                nfev, pairs, hits, done = (int(v) for v in a[3])
                assert done == 1 and nfev == a[2]
                assert tuple(int(v) for v in b[3][1:3]) == (0, 0), (tier, name, f64, b[3])
                if pairs < 512:
                    assert pairs + hits == nfev, (tier, name, f64, a[3])          # every evaluation either streamed (and was kept) or was answered
                if name != "clipped":
                    assert hits > nfev // 2, (tier, name, f64, a[3])              # two-sided data: most evaluations repeat a pair
                else:
                    # the range's lower end sits at the data's minimum: a new scale per shift, more distinct pairs than the table
                    # holds (512) -- it fills, the rest stream without being kept, and the results still equal the memo-less run
                    assert pairs > 400, (tier, name, f64, a[3])
                    if tier.startswith("strict"):
                        assert pairs == 512 and nfev - hits > 512, (tier, name, f64, a[3])
                total_hits += hits
        assert total_hits > 0
    finally:
        ops.set_tuning("mse_memo", 1)
        ops.set_tuning("mse_resident", 1)
        osq.reset_tier()


def test_memo_in_rounds_equals_memo_off(dev):
    """The rounds of an observer pass (osq_msefast_ordered_multi_*): observers of several sites, two batches (fp32 call, then the
    float64 one), with and without the memo -- statistics and evaluation counts equal bit for bit."""
    import outlier_suppression_amd as osq
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver, MSEFastObserver
    osq.set_strict(True)
    try:
        cases = _tensors(dev)

        def run():
            obs = [(AvgMSEFastObserver if i % 2 == 0 else MSEFastObserver)(bit=6 if i % 3 else 4, symmetric=False).to(dev) for i in range(len(cases))]
            for ob in obs:
                object.__setattr__(ob, "_defer_ok", True)
            with deferred_observation() as sites:
                for b in range(2):
                    for ob, (_, x, L, sp) in zip(obs, cases):
                        ob(x * (1.0 + 0.25 * b), L, sp)
                    sites.flush()
            return [(N(o.min_val), N(o.max_val), N(o.last_nfev)) for o in obs]

        ops.set_tuning("mse_memo", 1)
        a = run()
        ops.set_tuning("mse_memo", 0)
        b = run()
        for (amin, amax, an), (bmin, bmax, bn) in zip(a, b):
            assert bits_equal(amin, bmin) and bits_equal(amax, bmax) and bits_equal(an, bn)
    finally:
        ops.set_tuning("mse_memo", 1)
        osq.reset_tier()


def test_one_dimensional_searches_have_nothing_to_repeat(dev):
    """The 1-D search (observer.py:483-494: one-sided data) changes the scale with every candidate: the memo keeps every pair, answers
    none, and the result equals the memo-less run."""
    import outlier_suppression_amd as osq
    from outlier_suppression_amd import ops
    osq.set_strict(True)
    try:
        g = torch.Generator().manual_seed(23)
        x = torch.softmax(torch.randn(4, 6, 32, 32, generator=g) * 2, -1).to(dev)
        out = []
        for memo in (1, 0):
            ops.set_tuning("mse_memo", memo)
            cur = ops.batch_minmax(x, None, 2)
            r = ops.msefast_tensor_begin(x, cur, None, 2, 0, 63, False, "pos", False, False)
            ops.msefast_tensor_run(r, None, False)
            stats = N(ops.msefast_tensor_stats(r))
            mn = torch.full((1,), float("inf"), dtype=torch.float64, device=dev)
            mx = torch.full((1,), float("-inf"), dtype=torch.float64, device=dev)
            nfev = int(ops.msefast_tensor_commit(r, ops.UPDATE_RUNNING, 0, mn, mx).item())
            out.append((N(mn), N(mx), nfev, stats))
        (amin, amax, an, astats), (bmin, bmax, bn, bstats) = out
        assert bits_equal(amin, bmin) and bits_equal(amax, bmax) and an == bn
        assert int(astats[1]) == an and int(astats[2]) == 0 and tuple(int(v) for v in bstats[1:3]) == (0, 0)
    finally:
        ops.set_tuning("mse_memo", 1)
        osq.reset_tier()


def test_memo_at_baseline_size(dev):
    """BASELINE's [256,128,768] activation, masked (lengths randint(8,129)), both calls of an AvgMSEFast observer (fp32 then float64
    arithmetic): a size-independent property instead of an oracle pass -- the memo changes no bit of range, statistic or nfev."""
    import outlier_suppression_amd as osq
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver
    osq.set_strict(True)
    try:
        gen = torch.Generator(device=dev).manual_seed(0)
        xs = []
        for _ in range(2):
            x = torch.randn(256, 128, 768, device=dev, generator=gen)
            x[..., [7, 300, 511]] *= 20
            xs.append(x)
        L = torch.randint(8, 129, (256,), device=dev, generator=gen)
        out = []
        for memo in (1, 0):
            ops.set_tuning("mse_memo", memo)
            ob = AvgMSEFastObserver(bit=6, symmetric=False).to(dev)
            per_call = []
            for x in xs:
                ob(x, L, 1)
                per_call.append((N(ob.min_val), N(ob.max_val), N(ob.last_nfev)))
            out.append(per_call)
        for (amin, amax, an), (bmin, bmax, bn) in zip(*out):
            assert bits_equal(amin, bmin) and bits_equal(amax, bmax) and bits_equal(an, bn)
    finally:
        ops.set_tuning("mse_memo", 1)
        osq.reset_tier()
