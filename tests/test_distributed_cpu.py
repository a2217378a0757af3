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
