"""The ONE-LAUNCH masked observation (csrc/observe_onelaunch.h: the selectors of token-wise clipping work while the
tokens stream, candidates above a pivot taken from the observer's running statistic) against the oracle, bit for bit,
and against the two-launch form it replaces.  The pivot is only ever a hint, so every way it can be wrong is driven:
no running statistic (first batch), a batch whose thresholds dropped far below it (the rank lies among the
non-candidates: the search falls back to the token arrays), one that rose far above it (the rank lies above the
histogrammed octave: full-range levels over the candidate list), more candidates than the list holds, percentiles
from the median to 1.0, observers that do not prune, NaN / inf in valid and in padded tokens, empty samples.
Reference: quantization/observer.py:50-70 (prune_token, cac_thres, quantile_range), 176-237 (the observers)."""
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


@pytest.fixture(autouse=True)
def one_launch_on(dev):
    """The one-launch form is opt-in (it measured slower than the two launches it was meant to replace,
    profiles/r04_onelaunch_ab.txt): the tests switch it on."""
    from outlier_suppression_amd import ops
    ops.set_tuning("observe_onelaunch", 1)
    yield
    ops.set_tuning("observe_onelaunch", 0)


def N(t):
    return t.detach().cpu().numpy()


def _observers(kind, pct, dev):
    from outlier_suppression_amd.quantization import observer as OBS
    from oracle import observer_oracle as OB
    name = "encoder.layer.0.output.LayerNorm.layernorm_post_act_fake_quantize.observer"
    ob = getattr(OBS, kind)(bit=6, symmetric=False, ch_axis=-1).to(dev)
    ob.set_name(name)
    st = OB.ObserverState(bit=6, symmetric=False, name=name)
    if kind == "AvgPruneMinMaxObserver":
        ob.set_percentile(pct)
        st.percentile = pct
    fn = {"AvgPruneMinMaxObserver": OB.observe_avg_prune_minmax, "AvgMinMaxObserver": OB.observe_avg_minmax,
          "MinMaxObserver": OB.observe_minmax}[kind]
    return ob, st, fn


def _same(ob, st, what):
    got = (np.float32(N(ob.min_val).reshape(-1)[0]), np.float32(N(ob.max_val).reshape(-1)[0]))
    want = (np.float32(st.min_val), np.float32(st.max_val))
    both_nan = all(np.isnan(v) for v in got + want)
    assert both_nan or got == want, (what, got, want)


SHAPES = [
    # shape, seq_pos, permutation that builds the view (None: dense)
    ((32, 128, 768), 1, None),
    ((256, 128, 768), 1, None),
    ((8, 32, 96), 1, None),
    ((32, 12, 128, 64), 2, (0, 2, 1, 3)),          # [B,h,T,d] seen through [B,T,h,d] memory (quant_bert.py:128-150)
    ((4, 1024, 64), 1, None),
    ((16, 12, 64, 64), 2, None),                   # dense 4-D, tokens on axis 2
]
SCALES = [1.0, 1.03, 0.35, 0.36, 3.0, 2.9, 1.0]      # batch-to-batch drift: hint hit, far below (fall back to memory), far above


@pytest.mark.parametrize("kind,pct", [("AvgPruneMinMaxObserver", 0.95), ("AvgPruneMinMaxObserver", 0.5), ("AvgPruneMinMaxObserver", 1.0),
                                      ("AvgPruneMinMaxObserver", 0.999), ("AvgMinMaxObserver", None), ("MinMaxObserver", None)])
def test_one_launch_observation_equals_oracle(dev, kind, pct):
    gen = torch.Generator().manual_seed(91)
    for shape, seq_pos, perm in SHAPES:
        ob, st, fn = _observers(kind, pct, dev)
        B, T = shape[0], shape[seq_pos]
        for r, scale in enumerate(SCALES[:4] if shape[0] == 256 else SCALES):
            if perm is None:
                x = torch.randn(*shape, generator=gen)
            else:
                mem = [shape[i] for i in np.argsort(perm)]
                x = torch.randn(*mem, generator=gen).permute(*perm)
            x = x * scale
            x[..., 5] *= 20
            L = torch.randint(0 if r == 2 else 1, T + 1, (B,), generator=gen)
            L[int(torch.randint(0, B, (1,), generator=gen))] = T
            if r == 3:
                L[:] = T                                            # every token valid
            ob(x.to(dev), L.to(dev), seq_pos)
            fn(st, x.numpy(), L.numpy(), seq_pos)
            _same(ob, st, (kind, pct, shape, r))
        assert getattr(ob, "cnt", len(SCALES)) in (4, len(SCALES))


def test_one_launch_equals_two_launches_and_handles_special_values(dev):
    """Same inputs through osq_set_tuning("observe_onelaunch", 1 | 0) and with the pivot switched off: identical statistics;
    NaN in a valid token poisons, NaN / inf in padded tokens do not, +inf in a valid token is an ordinary large value."""
    from outlier_suppression_amd import ops
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(5)
    B, T, H = 32, 128, 768
    base = torch.randn(B, T, H, generator=gen)
    base[..., 9] *= 15
    L = torch.randint(1, T + 1, (B,), generator=gen)
    L[3] = T
    L[4] = 0
    variants = {"plain": base}
    v = base.clone(); v[7, int(L[7]):, :] = float("nan"); v[9, int(L[9]):, 3] = float("inf"); variants["padding holds NaN / inf"] = v
    v = base.clone(); v[3, 5, 100] = float("inf"); variants["+inf in a valid token"] = v
    v = base.clone(); v[3, 6, 7] = float("-inf"); variants["-inf in a valid token"] = v
    v = base.clone(); v[3, 2, 1] = float("nan"); variants["NaN in a valid token"] = v
    for name, x in variants.items():
        stats = {}
        for mode, (one, hint) in {"one launch": (1, 1), "one launch, no pivot": (1, 0), "two launches": (0, 1)}.items():
            ops.set_tuning("observe_onelaunch", one)
            ops.set_tuning("observe_hint", hint)
            try:
                ob, st, fn = _observers("AvgPruneMinMaxObserver", 0.9, dev)
                for r in range(3):
                    xi = x * (1.0 + 0.05 * r)
                    ob(xi.to(dev), L.to(dev), 1)
                    if name == "NaN in a valid token":              # the reference raises there (max of an empty selection); the kernels poison
                        assert np.isnan(N(ob.min_val)).all() and np.isnan(N(ob.max_val)).all(), (name, mode, r)
                        continue
                    fn(st, xi.numpy(), L.numpy(), 1)
                    _same(ob, st, (name, mode, r))
                stats[mode] = (N(ob.min_val).tobytes(), N(ob.max_val).tobytes())
            finally:
                ops.set_tuning("observe_onelaunch", 1)      # the module's fixture state
                ops.set_tuning("observe_hint", 1)
        assert len(set(stats.values())) == 1, (name, stats)
    ops.check_persistent("one-launch observation")


def test_one_launch_quantizer_state_and_capture(dev):
    """Through the module API: the observer-only state of a quantizer writes scale / zero_point in the same launch, and
    capture mode (sharded calibration: no running statistic, hence no pivot) records the batch's own row."""
    from types import SimpleNamespace as NS
    from outlier_suppression_amd.quantization import Quantizer
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(12)
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    q = Quantizer(None, cfg).to(dev)
    q.observer.set_name("x.act_fake_quant.observer")
    q.observer.set_percentile(0.97)
    q.enable_observer()
    q.disable_fake_quant()
    st = OB.ObserverState(bit=6, symmetric=False, name="x")
    st.percentile = 0.97
    for r in range(3):
        x = torch.randn(32, 128, 768, generator=gen) * (1 + 0.1 * r)
        x[..., 2] *= 30
        L = torch.randint(8, 129, (32,), generator=gen)
        assert q(x.to(dev), L.to(dev), 1).data_ptr() != 0
        OB.observe_avg_prune_minmax(st, x.numpy(), L.numpy(), 1)
        scale, zp = st.qparams()
        assert np.float32(q.scale.item()) == np.float32(scale) and np.float32(q.zero_point.item()) == np.float32(zp), r
    slot = torch.zeros(2, device=dev)
    q.observer._capture = slot
    x = torch.randn(32, 128, 768, generator=gen)
    L = torch.randint(8, 129, (32,), generator=gen)
    before = (q.observer.min_val.clone(), q.observer.max_val.clone())
    q(x.to(dev), L.to(dev), 1)
    q.observer._capture = None
    one = OB.ObserverState(bit=6, symmetric=False, name="x")
    one.percentile = 0.97
    OB.observe_avg_prune_minmax(one, x.numpy(), L.numpy(), 1)
    assert np.float32(slot[0].item()) == np.float32(one.min_val) and np.float32(slot[1].item()) == np.float32(one.max_val)
    assert torch.equal(before[0], q.observer.min_val) and torch.equal(before[1], q.observer.max_val)


def test_one_launch_time_out_is_loud(dev):
    """A selector that gives up waiting (the test knob makes every wait expire at once) poisons the statistic and raises at
    the next synchronisation point of the host; the launch after that is clean."""
    from outlier_suppression_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(256, 128, 768, generator=gen).to(dev)
    L = torch.randint(8, 129, (256,), generator=gen).to(dev)
    ob, st, fn = _observers("AvgPruneMinMaxObserver", 0.9, dev)
    ops.check_persistent("before")
    ops.set_tuning("fused_spin_limit", 1)
    try:
        ob(x, L, 1)
        with pytest.raises(ops.PersistentLaunchTimeout):
            ops.check_persistent("one-launch observation")
    finally:
        ops.set_tuning("fused_spin_limit", 0)
    assert np.isnan(N(ob.min_val)).all() and np.isnan(N(ob.max_val)).all()
    ob2, st2, fn2 = _observers("AvgPruneMinMaxObserver", 0.9, dev)
    ob2(x, L, 1)
    fn2(st2, N(x), N(L), 1)
    _same(ob2, st2, "after the time-out")
    ops.check_persistent("after")
