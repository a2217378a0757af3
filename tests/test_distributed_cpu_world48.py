"""Rehearsal of the first 8-GPU run on the CPU: world-4 and world-8 gloo groups through the package's own
`calibration.calibrate_sharded` and `token_wise_clipping.learn_scale_sharded`, with a batch count the ranks do NOT divide
(9 batches: SQuAD's 256 examples give >= 256 features, i.e. 9+ batches of 32; ptq_qa_quant.py:237-248).

There is no GPU here, so the two places that launch HIP kernels are stood in for -- everything between them is the
package's code, run as the ranks of an 8-GPU job run it:

  * a quantizer's forward (`_TorchLSQPlus`): observer on -> this batch's (min, max) goes into the armed capture slot, the
    way the observer kernels leave it (CaptureTable); fake-quant on -> util_quant.py:48-55 spelt in torch ops with autograd
    (round_ste, grad_scale), `numel_multiplier` honoured in the gradient factor;
  * `calibration.replay` (one HIP launch, osq_replay_statistics): the same fp32 loop on the host (observer.py:194-202).

Checked: the replayed statistics on EVERY rank equal the one-process sequential loop BIT FOR BIT at world 4 and 8 (ranks
with one batch, ranks with two, trailing rows of the table unused); the data-parallel learn-scale ends with the same bits on
every rank and within float rounding of the one-process loop (its documented bar)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_BATCHES, BATCH, SEQ, HID = 9, 32, 12, 24


def _build():
    from types import SimpleNamespace as NS
    from outlier_suppression_amd.quantization import Quantizer
    from outlier_suppression_amd.quantization.fake_quant import LSQPlusFakeQuantize

    class _TorchLSQPlus(LSQPlusFakeQuantize):
        """Stand-in for the two HIP launches of a quantizer call (see the module docstring)."""

        def forward(self, X, observation_mask=None, seq_pos=-1, persistent=None):
            if self.observer_enabled == 1:
                slot = self.observer.__dict__.get("_capture")
                assert slot is not None, "the sharded pass arms a capture slot before every forward"
                slot.copy_(torch.stack([X.detach().min(), X.detach().max()]))
            if self.fake_quant_enabled != 1:
                return X
            with torch.no_grad():                                    # fake_quant.py:188-191
                self.scale.abs_().clamp_(min=float(torch.finfo(torch.float32).eps))
                self.zero_point.clamp_(self.quant_min, self.quant_max)
            g = 1.0 / (X.numel() * getattr(self, "numel_multiplier", 1) * self.quant_max) ** 0.5
            gs = lambda t: (t - t * g).detach() + t * g              # util_quant.py:70-71
            ste = lambda t: (t.round() - t).detach() + t             # util_quant.py:4-8
            zp, s = gs(ste(self.zero_point)), gs(self.scale)
            xq = torch.clamp(ste(X / s) + zp, self.quant_min, self.quant_max)
            return (xq - zp) * s

        def _sanitize(self):
            with torch.no_grad():
                self.scale.abs_().clamp_(min=float(torch.finfo(torch.float32).eps))
                self.zero_point.clamp_(self.quant_min, self.quant_max)

    def quantizer():
        q = Quantizer(None, NS(quantizer="LSQPlusFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1))
        q.__class__ = _TorchLSQPlus
        return q

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(100, HID)
            self.fc1, self.fc2, self.head = torch.nn.Linear(HID, HID), torch.nn.Linear(HID, HID), torch.nn.Linear(HID, 3)
            self.a_post_act_fake_quantize, self.b_post_act_fake_quantize, self.c_post_act_fake_quantize = quantizer(), quantizer(), quantizer()

        def forward(self, input_ids=None, attention_mask=None):
            h = self.a_post_act_fake_quantize(self.emb(input_ids) * 3.0)
            h = self.b_post_act_fake_quantize(torch.tanh(self.fc1(h)) * 2.0)
            h = self.c_post_act_fake_quantize(self.fc2(h))
            return (self.head(h[:, 0]),)

    torch.manual_seed(7)
    net = Net().eval()
    g = torch.Generator().manual_seed(8)
    batches = [{"input_ids": torch.randint(0, 100, (BATCH, SEQ), generator=g), "attention_mask": torch.ones(BATCH, SEQ, dtype=torch.long)}
               for _ in range(N_BATCHES)]
    return net, batches


def _host_replay(ordered, quantizers, fresh=False, plan=None):
    """calibration.replay on the host: observer.py:194-202 in fp32, batch by batch, then calculate_qparams (oracle)."""
    from oracle.observer_oracle import ObserverState
    table = ordered.detach().cpu().numpy().astype(np.float32)
    for i, (_, q) in enumerate(quantizers):
        st = ObserverState(bit=q.bit, symmetric=q.symmetric)
        for b in range(table.shape[0]):
            st._avg_update(table[b, i, 0], table[b, i, 1])
        q.observer.min_val = torch.tensor(np.float32(st.min_val))
        q.observer.max_val = torch.tensor(np.float32(st.max_val))
        q.observer.cnt = table.shape[0]
        scale, zp = st.qparams()
        with torch.no_grad():
            q.scale.copy_(torch.tensor([np.float32(scale)]))
            q.zero_point.copy_(torch.tensor([np.float32(zp)]))
    return plan


def _run(rank, world, port, out_dir):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from types import SimpleNamespace as NS
    from outlier_suppression_amd import calibration, ops, token_wise_clipping as TWC
    calibration.replay = _host_replay
    ops.check_persistent = lambda where="": None
    TWC.task_type, TWC.model_type = "glue", "bert"
    net, batches = _build()
    with torch.no_grad():
        fp_out = [net(**b)[0].detach() for b in batches]
    qs = calibration.act_quantizers(net)
    assert len(qs) == 3
    for _, q in qs:
        q.enable_observer()
        q.disable_fake_quant()
    mine = calibration.shard_batches(N_BATCHES, rank, world)
    ordered = calibration.calibrate_sharded(net, [batches[b] for b in mine], lambda m, b: m(**b), n_batches=N_BATCHES)
    out = {"ordered": ordered.numpy(), "mine": np.array(mine)}
    for i, (_, q) in enumerate(qs):
        out[f"min{i}"], out[f"max{i}"] = q.observer.min_val.numpy(), q.observer.max_val.numpy()
        out[f"scale{i}"], out[f"zp{i}"] = q.scale.detach().numpy().copy(), q.zero_point.detach().numpy().copy()
    TWC.learn_scale_sharded(NS(model=net), batches, fp_out, {"lr": 1e-3, "epoch": 2})
    for i, (_, q) in enumerate(qs):
        out[f"lscale{i}"], out[f"lzp{i}"] = q.scale.detach().numpy().copy(), q.zero_point.detach().numpy().copy()
        assert getattr(q, "numel_multiplier", 1) == 1          # restored after the loop
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), **out)
    if world > 1:
        dist.destroy_process_group()


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_calibration_and_learn_scale_equal_one_process(tmp_path, world):
    port = 29400 + (os.getpid() % 300) + world
    _run(0, 1, 0, str(tmp_path))                                     # the one-process loop, same code path with world 1
    one = np.load(tmp_path / "w1_r0.npz")
    mp.spawn(_run, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"w{world}_r{r}.npz") for r in range(world)]
    # the deal: 9 batches over the ranks, round robin; every batch exactly once
    assert sorted(int(b) for r in ranks for b in r["mine"]) == list(range(N_BATCHES))
    assert [len(r["mine"]) for r in ranks] == [len(range(k, N_BATCHES, world)) for k in range(world)]
    for r, got in enumerate(ranks):
        # the gathered table is the one-process table, in global batch order, on every rank
        assert np.array_equal(_bits(got["ordered"]), _bits(one["ordered"])), r
        for i in range(3):
            for key in (f"min{i}", f"max{i}", f"scale{i}", f"zp{i}"):                 # replayed statistics: BIT for bit
                assert np.array_equal(_bits(got[key]), _bits(one[key])), (world, r, key, got[key], one[key])
            for key in (f"lscale{i}", f"lzp{i}"):
                assert np.array_equal(_bits(got[key]), _bits(ranks[0][key])), (world, r, key)   # every rank ends with the same parameter bits
                np.testing.assert_allclose(got[key], one[key], rtol=2e-5, atol=1e-6)            # and within rounding of one process
    # the learn-scale really moved the parameters (the comparison above is not of untouched values)
    assert any(not np.array_equal(one[f"lscale{i}"], one[f"scale{i}"]) for i in range(3))
