"""The one-launch observe + fake-quant step (csrc/fused_step.h) through the module API and the C ABI:
bit-exact against the oracle at oracle-sized inputs, bit-equal to the three-launch path at BASELINE sizes,
launch-to-launch state (epoch, arrival counters, time-out flags) clean after hundreds of launches on two streams."""
import ctypes
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()
    return torch.device("cuda:0")


def make(dev, quantizer, observer, sym, percentile=0.9, bit=6):
    from outlier_suppression_amd.quantization import Quantizer
    q = Quantizer(None, NS(quantizer=quantizer, observer=observer, bit=bit, symmetric=sym, ch_axis=-1)).to(dev)
    q.observer.set_name("encoder.layer.0.output.LayerNorm.layernorm_post_act_fake_quantize.observer")
    if hasattr(q.observer, "set_percentile"):
        q.observer.set_percentile(percentile)
    q.enable_observer()
    q.enable_fake_quant()
    return q


def fused_status(dev):
    from outlier_suppression_amd import _hip
    st = ctypes.c_int(-1)
    _hip.check(_hip.load().osq_fused_step_status(_hip.ptr(_hip.workspace(dev)), ctypes.byref(st), _hip.stream_ptr(dev)), "status")
    return st.value


CONFIGS = (("LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False), ("FixedFakeQuantize", "AvgMinMaxObserver", True),
           ("FixedFakeQuantize", "MinMaxObserver", False), ("LSQFakeQuantize", "AvgPruneMinMaxObserver", False))


@pytest.mark.parametrize("shape", [(8, 32, 768), (4, 16, 1024), (3, 8, 3072), (2, 6, 4096), (5, 12, 768)])
def test_fused_step_vs_oracle(shape, eq32, dev):
    """Three batches per configuration; lengths with zeros and full rows; every statistic, scale, zero_point and the
    dequantised tensor against the oracle (observer.py:50-70,184-237 + util_quant.py:11-55 restated in oracle/)."""
    from oracle import observer_oracle as OB, fake_quant_oracle as FQ
    gen = torch.Generator().manual_seed(sum(shape))
    B, T, H = shape
    for quantizer, observer, sym in CONFIGS:
        q = make(dev, quantizer, observer, sym)
        st = OB.ObserverState(bit=6, symmetric=sym, name=q.observer.name)
        st.percentile = 0.9
        fn = {"AvgPruneMinMaxObserver": OB.observe_avg_prune_minmax, "AvgMinMaxObserver": OB.observe_avg_minmax,
              "MinMaxObserver": OB.observe_minmax}[observer]
        for it in range(3):
            x = torch.randn(*shape, generator=gen) * (1.0 + it)
            x[..., 3] *= 12.0
            L = torch.randint(0, T + 1, (B,), generator=gen)
            L[it % B] = T
            with torch.no_grad():
                y = q(x.to(dev), L.to(dev), 1)
            fn(st, x.numpy(), L.numpy(), 1)
            scale, zp = st.qparams()
            assert eq32(q.observer.min_val.cpu().numpy(), st.min_val) and eq32(q.observer.max_val.cpu().numpy(), st.max_val), (quantizer, it)
            assert np.float32(q.scale.item()) == np.float32(scale) and np.float32(q.zero_point.item()) == np.float32(zp), (quantizer, it)
            qmin, qmax = q.quant_min, q.quant_max
            if quantizer == "LSQPlusFakeQuantize":
                _, ref = FQ.fake_quantize_learnableplus_per_tensor(x.numpy(), scale, zp, qmin, qmax, FQ.lsqplus_grad_factor(x.numel(), qmax))
            elif quantizer == "LSQFakeQuantize":
                _, ref = FQ.fake_quantize_learnable_per_tensor(x.numpy(), scale, zp, qmin, qmax, FQ.lsqplus_grad_factor(x.numel(), qmax))
            else:
                _, ref = FQ.fake_quantize_per_tensor_affine(x.numpy(), scale, zp, qmin, qmax)
            assert eq32(y.cpu().numpy(), ref), (quantizer, observer, it)
    assert fused_status(dev) == 0


def test_per_call_escape_from_the_persistent_launch(eq32, dev):
    """forward(..., persistent=False) (and the module attribute) keeps ONE call off the whole-GPU one-launch form
    (OSQ_PARAM_NO_PERSISTENT, include/osq_hip.h): the dispatch events armed for the fused kernel family stay unrecorded, the
    results -- output, statistics, parameters -- are the one-launch form's bit for bit, and the next plain call is
    persistent again."""
    import ctypes
    from outlier_suppression_amd import _hip
    lib = _hip.load()
    gen = torch.Generator().manual_seed(77)
    x = [torch.randn(8, 64, 768, generator=gen).to(dev) * (1 + i) for i in range(3)]
    L = torch.randint(1, 65, (8,), generator=gen).to(dev)

    def fused_launch_seen(call):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "events")
        lib.osq_time_next_launch(_hip.TIME_FUSED_STEP, a, b)
        y = call()
        torch.cuda.synchronize()
        us = ctypes.c_float()
        seen = lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)) == 0        # unrecorded events: error, nothing of that family ran
        lib.osq_time_next_launch(0, None, None)
        lib.osq_timing_events_destroy(a, b)
        return seen, y

    qa, qb, qc = (make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False) for _ in range(3))
    qc.persistent = False
    with torch.no_grad():
        for i in range(3):
            seen_a, ya = fused_launch_seen(lambda: qa(x[i], L, 1))
            seen_b, yb = fused_launch_seen(lambda: qb(x[i], L, 1, persistent=False))
            seen_c, yc = fused_launch_seen(lambda: qc(x[i], L, 1))
            assert seen_a and not seen_b and not seen_c, (i, seen_a, seen_b, seen_c)
            for q, y in ((qb, yb), (qc, yc)):
                assert eq32(y.cpu().numpy(), ya.cpu().numpy()), i
                assert eq32(q.scale.detach().cpu().numpy(), qa.scale.detach().cpu().numpy())
                assert eq32(q.zero_point.detach().cpu().numpy(), qa.zero_point.detach().cpu().numpy())
                assert eq32(q.observer.min_val.cpu().numpy(), qa.observer.min_val.cpu().numpy())
                assert eq32(q.observer.max_val.cpu().numpy(), qa.observer.max_val.cpu().numpy())
        seen, _ = fused_launch_seen(lambda: qb(x[0], L, 1))
        assert seen, "a plain call after an escaped one takes the one-launch form again"
    assert fused_status(dev) == 0


@pytest.mark.parametrize("shape,lengths", [((256, 128, 768), "bench"), ((256, 128, 768), "full"), ((256, 128, 768), "zeros"),
                                           ((256, 128, 1024), "bench")])
def test_fused_step_vs_oracle_at_the_headline_shape(shape, lengths, eq32, dev):
    """The bench kernel at the bench shape, DIRECTLY against the oracle: BERT-base [256,128,768]
    with the bench's length distribution, with every token valid, and with a lengths vector that holds zeros and full
    rows; plus [256,128,1024], whose 32768 tokens of 1024 floats exceed what the waves keep (streamed tail).  Six
    batches each whose magnitude moves (x1, x1, x1.03, x2, x0.4, x1), so the running mean is exercised and the selectors'
    hinted window (token_select.h) is absent, hit, missed from below and missed from above; min_val / max_val / scale / zero_point and every element of y
    bit-equal to observer.py:50-70,206-237 + util_quant.py:48-55 as restated in oracle/."""
    from oracle import observer_oracle as OB, fake_quant_oracle as FQ
    B, T, H = shape
    gen = torch.Generator().manual_seed(1234)
    outliers = torch.randperm(H, generator=gen)[:6]
    q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False, percentile=0.95)
    st = OB.ObserverState(bit=6, symmetric=False, name=q.observer.name)
    st.percentile = 0.95
    for it, mult in enumerate((1.0, 1.0, 1.03, 2.0, 0.4, 1.0)):
        if lengths == "bench":
            L = torch.randint(8, T + 1, (B,), generator=gen)
        elif lengths == "full":
            L = torch.full((B,), T)
        else:
            L = torch.randint(0, T + 1, (B,), generator=gen)
            L[::7] = 0
            L[3::11] = T
        x = torch.randn(*shape, generator=gen) * mult
        x[..., outliers] *= 20.0
        with torch.no_grad():
            y = q(x.to(dev), L.to(dev), 1)
        OB.observe_avg_prune_minmax(st, x.numpy(), L.numpy(), 1)
        scale, zp = st.qparams()
        assert eq32(q.observer.min_val.cpu().numpy(), st.min_val) and eq32(q.observer.max_val.cpu().numpy(), st.max_val), it
        assert np.float32(q.scale.item()) == np.float32(scale) and np.float32(q.zero_point.item()) == np.float32(zp), it
        _, ref = FQ.fake_quantize_learnableplus_per_tensor(x.numpy(), scale, zp, 0, 63, FQ.lsqplus_grad_factor(x.numel(), 63))
        assert eq32(y.cpu().numpy(), ref), (shape, lengths, it)
        del y, ref
    assert fused_status(dev) == 0


def test_fused_step_time_out_is_loud_and_recoverable(eq32, dev):
    """A persistent launch whose workgroups do not meet (here: the test knob shortens every cross-workgroup wait to one
    poll, so the selectors give up before the streaming workgroups arrive) must not go unnoticed: y and the statistics
    are NaN, the sticky flag is up, ops.check_persistent raises at the next synchronisation point (here: state_dict())
    and resets the launch state -- the next launch is correct again, also after the late arrivals of the failed one."""
    from outlier_suppression_amd import ops
    x = torch.randn(64, 128, 768)
    L = torch.randint(8, 129, (64,))
    ops.check_persistent()
    ref_q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
    with torch.no_grad():
        ref = ref_q(x.to(dev), L.to(dev), 1).cpu()
    ops.check_persistent()
    ops.set_tuning("fused_spin_limit", 1)
    try:
        q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
        with torch.no_grad():
            y = q(x.to(dev), L.to(dev), 1)
            y2 = q(x.to(dev), L.to(dev), 1)           # a second launch on the poisoned state, before anybody looked
        torch.cuda.synchronize()
        assert torch.isnan(y).all() and torch.isnan(y2).all()
        with pytest.raises(ops.PersistentLaunchTimeout):
            q.state_dict()
    finally:
        ops.set_tuning("fused_spin_limit", 0)
    ops.check_persistent()                             # the flag was consumed and the state reset: nothing left to report
    q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
    with torch.no_grad():
        for _ in range(3):
            y = q(x.to(dev), L.to(dev), 1)
    q2 = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
    with torch.no_grad():
        y2 = q2(x.to(dev), L.to(dev), 1)
    assert eq32(y2.cpu().numpy(), ref.numpy())
    q.state_dict()
    assert fused_status(dev) == 0


def test_fused_step_time_out_without_reset_heals(eq32, dev):
    """The launch state itself survives a time-out (round 2 zeroed the arrival counters in use, which a workgroup arriving
    after the time-out then left at 1 for every later launch): WITHOUT the host's reset -- only the flag is read through
    the C ABI -- the launches after a timed-out one are correct."""
    from outlier_suppression_amd import ops
    x = torch.randn(64, 128, 768)
    L = torch.randint(8, 129, (64,))
    ops.check_persistent()
    ref_q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
    with torch.no_grad():
        ref = ref_q(x.to(dev), L.to(dev), 1).cpu()
    ops.set_tuning("fused_spin_limit", 1)
    try:
        q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
        with torch.no_grad():
            for _ in range(3):
                q(x.to(dev), L.to(dev), 1)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("fused_spin_limit", 0)
    assert fused_status(dev) != 0                      # reads and clears the flag only (no reset of counters / epoch)
    ops._persistent_dirty.clear()
    for _ in range(4):
        q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
        with torch.no_grad():
            y = q(x.to(dev), L.to(dev), 1)
        assert eq32(y.cpu().numpy(), ref.numpy())
    assert fused_status(dev) == 0
    ops.check_persistent()


def test_fused_step_special_values(eq32, dev):
    """NaN among the valid tokens poisons the statistics (torch.max / quantile propagate it) and therefore y; NaN and inf
    in PADDED tokens do not touch the statistics and quantise like everywhere else (inf -> NaN, util_quant.py:8)."""
    from outlier_suppression_amd import ops
    x = torch.randn(4, 16, 768)
    L = torch.tensor([16, 3, 0, 9])
    x[1, 10, 5] = float("nan")
    x[2, 0, 0] = float("inf")
    out = {}
    for fused in (1, 0):
        ops.set_tuning("fused_step", fused)
        q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
        with torch.no_grad():
            out[fused] = (q(x.to(dev), L.to(dev), 1).cpu(), q.scale.item(), q.observer.min_val.item())
    ops.set_tuning("fused_step", 1)
    assert np.isfinite(out[1][1]) and out[1][1] == out[0][1] and out[1][2] == out[0][2]
    assert eq32(out[1][0].numpy(), out[0][0].numpy())
    assert torch.isnan(out[1][0][1, 10, 5]) and torch.isnan(out[1][0][2, 0, 0]) and torch.isfinite(out[1][0][0]).all()
    x[0, 2, 7] = float("nan")                   # a valid token
    q = make(dev, "LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", False)
    with torch.no_grad():
        y = q(x.to(dev), L.to(dev), 1)
    assert torch.isnan(q.observer.min_val).all() and torch.isnan(q.observer.max_val).all() and torch.isnan(y).all()
    assert fused_status(dev) == 0


@pytest.mark.parametrize("shape,lengths", [((256, 128, 768), "bench"), ((256, 128, 768), "full"), ((256, 128, 1024), "long"),
                                           ((32, 128, 3072), "bench"), ((32, 384, 768), "bench"), ((16, 128, 4096), "bench"),
                                           ((1024, 32, 768), "bench"), ((4, 16, 768), "zero")])
def test_fused_step_equals_three_launches(shape, lengths, dev):
    """BASELINE-sized tensors (more tokens than the waves can keep for 1024 / 3072 / 4096 features: streamed tail):
    statistics, parameters and every output element equal the three-launch path's, which is pinned to the oracle."""
    from outlier_suppression_amd import ops
    B, T, H = shape
    gen = torch.Generator().manual_seed(B + T + H)
    L = {"bench": torch.randint(8 if T >= 8 else 0, T + 1, (B,), generator=gen), "full": torch.full((B,), T),
         "long": torch.randint(T - T // 8, T + 1, (B,), generator=gen), "zero": torch.zeros(B, dtype=torch.long)}[lengths].to(dev)
    gd = torch.Generator(device=dev).manual_seed(7)
    xs = [torch.randn(*shape, device=dev, generator=gd) * (1 + i) for i in range(3)]
    for x in xs:
        x[..., 5] *= 20
    for quantizer, observer, sym in CONFIGS[:3]:
        out = {}
        for fused in (1, 0):
            ops.set_tuning("fused_step", fused)
            q = make(dev, quantizer, observer, sym, percentile=0.95)
            with torch.no_grad():
                ys = [q(x, L, 1) for x in xs]
            out[fused] = (ys, q.observer.min_val.clone(), q.observer.max_val.clone(), q.scale.detach().clone(), q.zero_point.detach().clone())
        ops.set_tuning("fused_step", 1)
        for k in range(1, 5):
            assert torch.equal(out[1][k], out[0][k]), (quantizer, k)
        for a, b in zip(out[1][0], out[0][0]):
            assert torch.equal(a, b), quantizer
        del out
    assert fused_status(dev) == 0


def test_fused_step_many_launches_two_streams(dev):
    """Launch-to-launch state: 300 launches with changing masks alternate between two streams (each has its own
    workspace; the host orders launches of different streams, two persistent grids must never share the device);
    every 25th result is compared with stock torch reductions + the three-launch fake-quant."""
    from outlier_suppression_amd import ops
    B, T, H = 64, 128, 768
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, H, device=dev)
    x[..., 11] *= 9
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    qs = [make(dev, "LSQPlusFakeQuantize", "MinMaxObserver", False) for _ in streams]
    torch.cuda.synchronize()
    checks = []
    for it in range(300):
        L = torch.randint(1, T + 1, (B,), generator=gen).to(dev)
        k = it % 2
        streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[k]), torch.no_grad():
            y = qs[k](x, L, 1)
            if it % 25 == 0:
                checks.append((k, L, y, qs[k].observer.min_val.clone(), qs[k].observer.max_val.clone(), qs[k].scale.detach().clone(),
                               qs[k].zero_point.detach().clone()))
    torch.cuda.synchronize()
    run_min, run_max = [float("inf")] * 2, [float("-inf")] * 2
    for k, L, y, mn, mx, s, z in checks:
        valid = (torch.arange(T, device=dev)[None, :] < L[:, None])
        v = x[valid]
        # MinMaxObserver: running extrema; the check points see a prefix of the sequence, so only bounds can be asserted
        assert mn.item() <= v.min().item() and mx.item() >= v.max().item()
        ref = ops.fake_quant_per_tensor(x, s, z, 0, 63, ops.PARAM_LSQPLUS, 1.0 / (x.numel() * 63) ** 0.5)
        assert torch.equal(y, ref)
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            assert fused_status(dev) == 0


def test_two_persistent_kernel_families_share_the_device(dev):
    """The fused observe + fake-quant step and the resident MSEFast search are both persistent grids that want every CU:
    launches of the two families alternate between two streams (the library orders them at every stream switch); every
    search must return what it returns alone, every fused step what the three launches return, no time-out."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MSEFastObserver
    B, T, H = 32, 128, 768
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, H, device=dev)
    x[..., 11] *= 9
    L = torch.randint(1, T + 1, (B,), generator=gen).to(dev)
    alone = MSEFastObserver(bit=6, symmetric=False).to(dev)
    alone(x, L, 1)
    want = (alone.min_val.clone(), alone.max_val.clone(), int(alone.last_nfev.sum().item()))
    q = make(dev, "LSQPlusFakeQuantize", "MinMaxObserver", False)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs, searches = [], []
    for it in range(40):
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[0]), torch.no_grad():
            y = q(x, L, 1)
            if it % 10 == 0:
                outs.append((y, q.scale.detach().clone(), q.zero_point.detach().clone()))
        with torch.cuda.stream(streams[1]), torch.no_grad():
            ob = MSEFastObserver(bit=6, symmetric=False).to(dev)
            ob(x, L, 1)
            searches.append(ob)
    torch.cuda.synchronize()
    for ob in searches:
        assert torch.equal(ob.min_val, want[0]) and torch.equal(ob.max_val, want[1]) and int(ob.last_nfev.sum().item()) == want[2]
    for y, s, z in outs:
        assert torch.equal(y, ops.fake_quant_per_tensor(x, s, z, 0, 63, ops.PARAM_LSQPLUS, 1.0 / (x.numel() * 63) ** 0.5))
    for s in streams:
        with torch.cuda.stream(s):
            assert fused_status(dev) == 0
