"""World-size-2 gloo tests (CPU) of the sharded-calibration exchange: plumbing only.

The per-batch statistics themselves come from HIP kernels on the GPU; here the table is
filled by the oracle so that the exchange + ordering + sequential replay can be checked
against the single-process reference result, bit for bit.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, n_batches, n_q, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from outlier_suppression_amd import calibration
    rng = np.random.default_rng(123)
    full = rng.standard_normal((n_batches, n_q, 2)).astype(np.float32)      # what a single process would see
    mine = calibration.shard_batches(n_batches, rank, world)
    rows = (n_batches + world - 1) // world
    local = torch.full((rows, n_q, 2), float("nan"))
    for j, b in enumerate(mine):
        local[j] = torch.from_numpy(full[b])
    ordered = calibration.gather_batch_table(local, n_batches)
    np.save(os.path.join(out_dir, f"ordered_{rank}.npy"), ordered.numpy())
    # per-batch TWC losses travel the same way (token_wise_clipping.py:58: summed in batch order)
    losses = torch.full((rows, 1), float("nan"))
    for j, b in enumerate(mine):
        losses[j, 0] = float(b) * 0.25 + 1.0
    tot = calibration.gather_batch_table(losses, n_batches)
    np.save(os.path.join(out_dir, f"loss_{rank}.npy"), tot.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_batches", [8, 5])
def test_gather_in_global_batch_order_world2(tmp_path, n_batches):
    world, n_q = 2, 7
    port = 29600 + (os.getpid() % 200) + n_batches
    mp.spawn(_worker, args=(world, port, n_batches, n_q, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    full = rng.standard_normal((n_batches, n_q, 2)).astype(np.float32)
    o0, o1 = np.load(tmp_path / "ordered_0.npy"), np.load(tmp_path / "ordered_1.npy")
    assert np.array_equal(o0, full) and np.array_equal(o1, full)       # identical on every rank, global order
    l0 = np.load(tmp_path / "loss_0.npy")
    assert np.array_equal(l0[:, 0], np.arange(n_batches) * 0.25 + 1.0)

    # replaying the gathered table sequentially == the single-process running mean (observer.py:194-202)
    from oracle.observer_oracle import ObserverState
    for qi in range(n_q):
        seq, rep = ObserverState(bit=6), ObserverState(bit=6)
        for b in range(n_batches):
            seq._avg_update(full[b, qi, 0], full[b, qi, 1])
            rep._avg_update(o0[b, qi, 0], o0[b, qi, 1])
        assert seq.min_val == rep.min_val and seq.max_val == rep.max_val and rep.cnt == n_batches
    # ... and a SUM all-reduce / n would NOT be bit-identical in general (why we gather + replay)
    diffs = 0
    for qi in range(n_q):
        st = ObserverState(bit=6)
        for b in range(n_batches):
            st._avg_update(full[b, qi, 0], full[b, qi, 1])
        diffs += int(np.float32(full[:, qi, 0].sum(dtype=np.float32) / np.float32(n_batches)) != st.min_val)
    assert diffs > 0


def test_shard_batches_round_robin():
    from outlier_suppression_amd.calibration import shard_batches
    assert shard_batches(8, 0, 2) == [0, 2, 4, 6] and shard_batches(8, 1, 2) == [1, 3, 5, 7]
    assert shard_batches(5, 1, 4) == [1] and shard_batches(5, 0, 4) == [0, 4]
    assert sorted(sum((shard_batches(11, r, 4) for r in range(4)), [])) == list(range(11))


def test_single_process_passthrough():
    from outlier_suppression_amd.calibration import gather_batch_table
    t = torch.arange(24.0).reshape(4, 3, 2)
    assert torch.equal(gather_batch_table(t, 3), t[:3])


# ---- site-sharded passes (calibration.calibrate_owned_sites): the deal and the exchange of final states

def _site_quantizers():
    """Quantizers as a site-sharded pass leaves them on their OWNER: a per-tensor AvgMSEFast activation quantizer
    (float64 statistics, dtype flags, counter, sidedness), a per-channel MSEFast weight quantizer ([C] fp32 statistics,
    int32 zero-points) and a learnable per-tensor one (fp32 Parameters)."""
    from types import SimpleNamespace as NS
    from outlier_suppression_amd.quantization import Quantizer
    torch.manual_seed(5)
    a = Quantizer(None, NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1))
    w = Quantizer(None, NS(quantizer="FixedFakeQuantize", observer="MSEFastObserver", bit=4, symmetric=True, ch_axis=0))
    l = Quantizer(None, NS(quantizer="LSQPlusFakeQuantize", observer="AvgQuantileObserver", bit=6, symmetric=False, ch_axis=-1))
    return [("layer.act_fake_quant_a", a), ("layer.weight_fake_quant", w), ("layer.act_fake_quant_l", l)]


def _fill_owned(q, i):
    obs = q.observer
    if i == 0:
        obs.min_val = torch.tensor(-1.2345678901234567, dtype=torch.float64)
        obs.max_val = torch.tensor(float("nan"), dtype=torch.float64)           # NaN travels as NaN
        obs.cnt, obs.one_side_dist = 8, "no"
        obs._ref_flags(torch.device("cpu")).copy_(torch.tensor([1, 0], dtype=torch.int32))
        q.scale.copy_(torch.tensor([0.0371]))
        q.zero_point.copy_(torch.tensor([31], dtype=torch.int32))
    elif i == 1:
        obs.min_val = -torch.rand(5) - 0.1
        obs.max_val = torch.rand(5) + 0.1
        obs.one_side_dist = "no"
        q.scale = torch.rand(5) + 0.01
        q.zero_point = torch.zeros(5, dtype=torch.int32)
        obs.last_nfev = torch.tensor([15, 14, 16, 15, 17], dtype=torch.int32)
    else:
        obs.min_val = torch.tensor(-0.5)
        obs.max_val = torch.tensor(7.25)
        obs.cnt = 3
        q.scale.data = torch.tensor([0.123])
        q.zero_point.data = torch.tensor([4.0])


def _site_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from outlier_suppression_amd import calibration
    qs = _site_quantizers()
    channels = [1, 5, 1]
    owner = [0, 1, 0]
    for i, (_, q) in enumerate(qs):
        if owner[i] == rank:
            _fill_owned(q, i)
    calibration.exchange_site_states(qs, channels, owner, rank, world, torch.device("cpu"))
    out = {}
    for i, (_, q) in enumerate(qs):
        obs = q.observer
        out[f"mn{i}"], out[f"mx{i}"] = obs.min_val.numpy(), obs.max_val.numpy()
        out[f"s{i}"], out[f"z{i}"] = q.scale.detach().numpy(), q.zero_point.detach().numpy()
        out[f"cnt{i}"] = np.array(getattr(obs, "cnt", -1))
        out[f"side{i}"] = np.array(str(getattr(obs, "one_side_dist", None)))
    out["flags0"] = qs[0][1].observer._ref_flags(torch.device("cpu")).numpy()
    np.savez(os.path.join(out_dir, f"sites_{rank}.npz"), **out)
    dist.destroy_process_group()


def test_site_states_reach_every_rank_world2(tmp_path):
    """calibration.exchange_site_states: after the one all-gather every rank holds every site's final statistics, scale,
    zero_point, counters and dtype flags exactly as its owner computed them -- values, dtypes and shapes."""
    port = 29850 + (os.getpid() % 100)
    mp.spawn(_site_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = _site_quantizers()
    for i, (_, q) in enumerate(want):
        _fill_owned(q, i)
    r0, r1 = np.load(tmp_path / "sites_0.npz"), np.load(tmp_path / "sites_1.npz")
    for got in (r0, r1):
        for i, (_, q) in enumerate(want):
            obs = q.observer
            for key, ref in ((f"mn{i}", obs.min_val), (f"mx{i}", obs.max_val), (f"s{i}", q.scale.detach()), (f"z{i}", q.zero_point.detach())):
                a, b = got[key], ref.numpy()
                assert a.dtype == b.dtype and a.shape == b.shape, (key, a.dtype, b.dtype, a.shape, b.shape)
                assert np.array_equal(a, b, equal_nan=True), key
            assert int(got[f"cnt{i}"]) == getattr(obs, "cnt", -1)
            assert str(got[f"side{i}"]) == str(getattr(obs, "one_side_dist", None))
        assert list(got["flags0"]) == [1, 0]


def test_deal_sites_is_balanced_and_deterministic():
    from outlier_suppression_amd.calibration import deal_sites
    # a BERT-base layer's sites (elements x search weight): 7 hidden-sized, attention probabilities (1-D), GELU output
    layer = [3.1e6 * 350] * 7 + [6.3e6 * 20] + [12.6e6 * 350]
    costs = layer * 12 + [24.6e3 * 350] * 2
    for world in (2, 4, 8):
        owner = deal_sites(costs, world)
        assert owner == deal_sites(list(costs), world)
        load = [sum(c for c, r in zip(costs, owner) if r == k) for k in range(world)]
        assert max(load) <= 1.08 * (sum(costs) / world), (world, load)
    assert deal_sites([], 4) == [] and deal_sites([5.0, 1.0], 1) == [0, 0]


def _timeout_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from outlier_suppression_amd import calibration, ops

    def fake_check(where=""):                      # rank 1's persistent launch "timed out"; rank 0's did not
        if rank == 1:
            raise ops.PersistentLaunchTimeout("rank 1 timed out")
    real = ops.check_persistent
    ops.check_persistent = fake_check
    try:
        try:
            calibration.check_persistent_collectively("test")
            verdict = "no error"
        except ops.PersistentLaunchTimeout as e:
            verdict = "raised: " + str(e)[:40]
        # nobody is left behind in a collective: both ranks reach this gather
        t = torch.tensor([float(rank)])
        out = torch.empty(world)
        dist.all_gather_into_tensor(out, t)
    finally:
        ops.check_persistent = real
    open(os.path.join(out_dir, f"verdict_{rank}.txt"), "w").write(verdict + f" | gathered {out.tolist()}")
    dist.destroy_process_group()


def test_persistent_time_out_is_raised_on_every_rank(tmp_path):
    """calibration.check_persistent_collectively: a time-out on ONE rank raises on every rank before the
    table exchange, so no rank waits in the gather alone and none carries on with the other's NaN-poisoned rows."""
    port = 29850 + (os.getpid() % 100)
    mp.spawn(_timeout_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    v0, v1 = (open(tmp_path / f"verdict_{r}.txt").read() for r in (0, 1))
    assert v0.startswith("raised"), v0                                   # rank 0 had no time-out of its own: it raises for rank 1's
    assert v1.startswith("raised: rank 1 timed out"), v1
    assert v0.endswith("gathered [0.0, 1.0]") and v1.endswith("gathered [0.0, 1.0]")


def test_probe_sites_switches_every_quantizer_off():
    """calibration.probe_sites: the geometry probe must not let ANY observer see its batch -- not the
    selected ones, not the others -- and must restore every flag."""
    from outlier_suppression_amd import calibration
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase

    class Spy(QuantizeBase):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.observer_enabled, self.fake_quant_enabled, self.ch_axis, self.calls = 1, 1, -1, []

        def forward(self, X, observation_mask=None, seq_pos=-1):
            self.calls.append((self.observer_enabled, self.fake_quant_enabled))
            return X

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a_act_fake_quant, self.w_weight_fake_quant = Spy(), Spy()

        def forward(self, x):
            return self.w_weight_fake_quant(self.a_act_fake_quant(x))
    net = Net()
    selected = [("a_act_fake_quant", net.a_act_fake_quant)]
    geo = calibration.probe_sites(net, torch.zeros(3, 5), lambda m, b: m(b), selected)
    assert geo == [(15, 1)]
    assert net.a_act_fake_quant.calls == [(0, 0)] and net.w_weight_fake_quant.calls == [(0, 0)]      # nobody observed, nobody quantised
    assert (net.a_act_fake_quant.observer_enabled, net.a_act_fake_quant.fake_quant_enabled) == (1, 1)
    assert (net.w_weight_fake_quant.observer_enabled, net.w_weight_fake_quant.fake_quant_enabled) == (1, 1)
