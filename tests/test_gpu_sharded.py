"""Sharded calibration end to end with TWO and with FOUR ranks sharing the one test GPU (gloo rendezvous, tables
staged through the host): each rank observes half of the batches, the per-batch statistics and
losses are all-gathered and replayed in global batch order -- percentile, per-candidate losses and
every scale / zero-point must equal the single-process run bit for bit.  On the 8-GPU node the same
code runs with backend "nccl" (RCCL) and device tensors."""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
from conftest import bits_equal
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, out_dir, backend="gloo"):
    import logging
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    tag = f"w{world}" if backend == "gloo" else f"{backend}{world}"
    if world > 1 or backend != "gloo":
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if backend == "nccl":      # RCCL: device tensors go into the collectives as they are
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers import BertConfig, BertForSequenceClassification
    from outlier_suppression_amd import calibration, token_wise_clipping as TWC
    from outlier_suppression_amd.gamma_migration import delay_ln
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, disable_all
    from outlier_suppression_amd.quantization.state import set_observer_name
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    g = np.load(os.path.join(ROOT, "tests", "golden", "bert_tiny_pipeline.npz"))
    cfg = BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=40, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     type_vocab_size=2)
    fp = BertForSequenceClassification(cfg).eval()
    fp.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}, strict=False)
    dev = torch.device("cuda:0")
    fp = fp.to(dev)
    batches = [{"input_ids": torch.from_numpy(g["input_ids"][b]).to(dev),
                "attention_mask": torch.from_numpy(g["attention_mask"][b]).to(dev),
                "token_type_ids": torch.zeros_like(torch.from_numpy(g["input_ids"][b])).to(dev)} for b in range(4)]
    a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    model = quantize_model(fp, w_q, a_q).to(dev)
    with torch.no_grad():
        fp_output = [model(**b)[0].detach() for b in batches]
    model = delay_ln(model, NS(a_qconfig=a_q, w_qconfig=w_q), NS(model_type="bert", task_type="glue"))
    enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    with torch.no_grad():
        model(**batches[0])
    disable_all(model)
    set_observer_name(model)
    TWC.task_type = "glue"
    losses = []

    class Grab(logging.Handler):
        def emit(self, record):
            m = record.getMessage()
            if m.startswith("the ratio is"):
                losses.append(float(m.split("the loss is")[1]))
    TWC.logger.addHandler(Grab())
    TWC.logger.setLevel(logging.INFO)
    mine = calibration.shard_batches(len(batches), rank, world)
    if backend == "nccl":
        # calibrate_sharded end to end on device tensors: capture -> all_gather_into_tensor (RCCL) -> ordered replay,
        # against the plain sequential observer pass (token_wise_clipping.py:12-47) on a copy of the model
        import copy
        twin = copy.deepcopy(model)
        TWC.set_ratio(model, 0.9)
        TWC.set_ratio(twin, 0.9)
        assert dist.get_backend() == "nccl"
        ordered = calibration.calibrate_sharded(model, [batches[b] for b in mine], lambda m, b: m(**b), n_batches=len(batches))
        assert ordered.is_cuda and ordered.shape[0] == len(batches)
        with torch.no_grad():
            for b in batches:
                twin(**b)
        qa = [m for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
        qb = [m for n, m in twin.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
        for x, y in zip(qa, qb):
            assert torch.equal(x.observer.min_val, y.observer.min_val) and torch.equal(x.observer.max_val, y.observer.max_val)
            assert torch.equal(x.scale, y.scale) and torch.equal(x.zero_point, y.zero_point) and x.observer.cnt == y.observer.cnt
        disable_all(model)
    ratio = TWC.find_ratio_cached(NS(model=model), [batches[b] for b in mine], [fp_output[b] for b in mine],
                                  {"iters": 5, "step": 0.05}, n_batches=len(batches))
    qs = [m for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
    np.savez(os.path.join(out_dir, f"{tag}_r{rank}.npz"), ratio=ratio, losses=np.array(losses),
             scale=np.stack([q.scale.detach().cpu().numpy() for q in qs]),
             zp=np.stack([q.zero_point.detach().cpu().numpy() for q in qs]),
             mn=np.stack([q.observer.min_val.cpu().numpy() for q in qs]),
             cnt=np.array([q.observer.cnt for q in qs]))
    # fine stage: sequential Adam on (scale, zero_point).  One process: the reference's loop; two ranks:
    # every step split inside the batch (half the samples each, averaged gradients)
    TWC.learn_scale_sharded(NS(model=model), batches, fp_output, {"lr": 1e-3, "epoch": 2})
    np.savez(os.path.join(out_dir, f"learn_{tag}_r{rank}.npz"),
             scale=np.stack([q.scale.detach().cpu().numpy() for q in qs]),
             zp=np.stack([q.zero_point.detach().cpu().numpy() for q in qs]))
    if world > 1 or backend != "gloo":
        dist.destroy_process_group()


def test_one_rank_rccl_group_equals_no_group(tmp_path):
    """backend="nccl" IS RCCL on ROCm.  A one-rank group on the test GPU executes init_process_group(device_id=...),
    the device-tensor all_gather_into_tensor of the statistics / loss tables and the all_reduce of the learn-scale
    gradients -- the code path the 8-GPU node runs -- and must change nothing."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_run, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run, args=(1, 29741, str(tmp_path), "nccl"), nprocs=1, join=True)
    one, rccl = np.load(tmp_path / "w1_r0.npz"), np.load(tmp_path / "nccl1_r0.npz")
    for k in ("ratio", "losses", "scale", "zp", "mn", "cnt"):
        assert bits_equal(one[k], rccl[k]), k
    l1, lr = np.load(tmp_path / "learn_w1_r0.npz"), np.load(tmp_path / "learn_nccl1_r0.npz")
    assert bits_equal(l1["scale"], lr["scale"]) and bits_equal(l1["zp"], lr["zp"])


def test_two_ranks_equal_one(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_run, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run, args=(2, 29731, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_run, args=(4, 29733, str(tmp_path)), nprocs=4, join=True)     # one batch, one sample of every batch per rank
    one = np.load(tmp_path / "w1_r0.npz")
    for world, r in ((2, 0), (2, 1), (4, 0), (4, 3)):
        two = np.load(tmp_path / f"w{world}_r{r}.npz")
        assert float(two["ratio"]) == float(one["ratio"])
        assert bits_equal(two["losses"], one["losses"])
        assert bits_equal(two["scale"], one["scale"]) and bits_equal(two["zp"], one["zp"])
        assert bits_equal(two["mn"], one["mn"]) and bits_equal(two["cnt"], one["cnt"])
    assert int(one["cnt"][0]) == 4
    # learn-scale: same mathematics, per-rank partial sums -> float-rounding agreement; ranks identical to each other
    base = np.load(tmp_path / "w1_r0.npz")
    l1 = np.load(tmp_path / "learn_w1_r0.npz")
    l2 = [np.load(tmp_path / f"learn_w2_r{r}.npz") for r in (0, 1)]
    assert bits_equal(l2[0]["scale"], l2[1]["scale"]) and bits_equal(l2[0]["zp"], l2[1]["zp"])
    l4 = [np.load(tmp_path / f"learn_w4_r{r}.npz") for r in range(4)]
    assert all(bits_equal(l4[0]["scale"], l["scale"]) and bits_equal(l4[0]["zp"], l["zp"]) for l in l4[1:])
    np.testing.assert_allclose(l4[0]["scale"], l1["scale"], rtol=5e-5, atol=0)
    assert not bits_equal(l1["scale"], base["scale"])          # the parameters did move
    np.testing.assert_allclose(l2[0]["scale"], l1["scale"], rtol=2e-5, atol=0)
    np.testing.assert_allclose(l2[0]["zp"], l1["zp"], rtol=2e-5, atol=2e-5)
    moved = np.abs(l1["scale"] - base["scale"]).max()
    assert np.abs(l2[0]["scale"] - l1["scale"]).max() < 1e-2 * moved   # far below the size of the learned update


def _run_masked(rank, world, port, out_dir, task):
    """learn_scale_sharded on the masked tasks: tiny BERT-QA (two heads over the attended tokens) / tiny BART (valid decoder
    tokens), every Adam step split inside the batch over `world` ranks sharing the test GPU."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import transformers as T
    from outlier_suppression_amd import token_wise_clipping as TWC
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, disable_all
    from outlier_suppression_amd.quantization.state import set_observer_name
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(5)
    dev = torch.device("cuda:0")
    B, Tn = 4, 12
    if task == "squad":
        fp = T.BertForQuestionAnswering(T.BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                                     intermediate_size=64, max_position_embeddings=40, hidden_dropout_prob=0.0,
                                                     attention_probs_dropout_prob=0.0, type_vocab_size=2)).eval().to(dev)
    else:
        fp = T.BartForConditionalGeneration(T.BartConfig(vocab_size=120, d_model=32, encoder_layers=2, decoder_layers=2,
                                                         encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=64,
                                                         decoder_ffn_dim=64, max_position_embeddings=40, dropout=0.0,
                                                         attention_dropout=0.0, activation_dropout=0.0)).eval().to(dev)
    batches = []
    for _ in range(3):
        L = torch.randint(3, Tn + 1, (B,), generator=gen)
        L[0] = Tn
        mask = (torch.arange(Tn)[None, :] < L[:, None]).long()
        ids = torch.randint(5, 115, (B, Tn), generator=gen) * mask + (1 - mask)
        b = {"input_ids": ids.to(dev), "attention_mask": mask.to(dev)}
        if task == "squad":
            b["token_type_ids"] = torch.zeros_like(ids).to(dev)
        else:
            DL = torch.randint(2, 8, (B,), generator=gen)
            DL[1] = 7
            dm = (torch.arange(7)[None, :] < DL[:, None]).long()
            b["decoder_input_ids"] = (torch.randint(5, 115, (B, 7), generator=gen) * dm + (1 - dm)).to(dev)
            b["decoder_attention_mask"] = dm.to(dev)
        batches.append(b)
    a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    model = quantize_model(fp, w_q, a_q).to(dev)
    TWC.task_type, TWC.model_type = task, ("bert" if task == "squad" else "bart")
    with torch.no_grad():
        if task == "squad":
            fp_output = []
            for b in batches:
                o = model(**b)
                keep = b["attention_mask"] == 1
                fp_output.append([o[0][keep].detach(), o[1][keep].detach()])
        else:
            fp_output = [model(**b)[0][b["decoder_attention_mask"] == 1, :].detach() for b in batches]
    enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    with torch.no_grad():
        model(**batches[0])
    disable_all(model)
    set_observer_name(model)
    TWC.set_ratio(model, 0.9)
    TWC.calibrate(model, batches)
    qs = [m for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
    before = np.stack([q.scale.detach().cpu().numpy() for q in qs])
    TWC.learn_scale_sharded(NS(model=model), batches, fp_output, {"lr": 1e-3, "epoch": 2})
    np.savez(os.path.join(out_dir, f"{task}_w{world}_r{rank}.npz"), before=before,
             scale=np.stack([q.scale.detach().cpu().numpy() for q in qs]),
             zp=np.stack([q.zero_point.detach().cpu().numpy() for q in qs]))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("task,port", [("squad", 29751), ("summ", 29761)])
def test_masked_learn_scale_two_ranks_equal_one(tmp_path, task, port):
    """The fine stage of the QA and summarisation entry points (ptq_qa_quant.py:262-277, ptq_summ_quant.py) data-parallel
    inside the batch: kept-token targets sliced by per-sample counts, losses over the full batch's denominator, gradients
    summed -- the learned parameters agree with the sequential loop to float rounding; the ranks agree bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_run_masked, args=(1, 0, str(tmp_path), task), nprocs=1, join=True)
    mp.spawn(_run_masked, args=(2, port, str(tmp_path), task), nprocs=2, join=True)
    mp.spawn(_run_masked, args=(4, port + 2, str(tmp_path), task), nprocs=4, join=True)
    worlds = (2, 4)
    if task == "summ":          # batches of 4 on 8 ranks: one sample on each of the first four, zeros from the others
        mp.spawn(_run_masked, args=(8, port + 4, str(tmp_path), task), nprocs=8, join=True)
        worlds = (2, 4, 8)
    one = np.load(tmp_path / f"{task}_w1_r0.npz")
    assert not bits_equal(one["scale"], one["before"])
    moved = np.abs(one["scale"] - one["before"]).max()
    for world in worlds:
        rs = [np.load(tmp_path / f"{task}_w{world}_r{r}.npz") for r in range(world)]
        assert all(bits_equal(rs[0]["scale"], r["scale"]) and bits_equal(rs[0]["zp"], r["zp"]) for r in rs[1:])
        assert bits_equal(rs[0]["before"], one["before"])
        assert np.abs(rs[0]["scale"] - one["scale"]).max() <= 0.02 * moved, (world, np.abs(rs[0]["scale"] - one["scale"]).max(), moved)
        # Adam normalises every gradient by its own running magnitude: a parameter whose gradient is rounding noise moves by
        # a fraction of lr either way, so the bound is in steps (6 steps of lr = 1e-3), not relative to the value
        assert np.abs(rs[0]["zp"] - one["zp"]).max() <= 0.05 * 6 * 1e-3, (world, np.abs(rs[0]["zp"] - one["zp"]).max())


# ---- site-sharded passes: the observers that cannot be recorded per batch (MSEFast and friends)

def _run_sites(rank, world, port, out_dir):
    """RoBERTa-like W4A6 flow of BASELINE configs[3] on the tiny golden BERT: per-channel MSEFast weight observers (one
    search per row), per-tensor AvgMSEFast activation observers over four batches (float32 search on the first batch,
    float64 from the second on, where the reference's dtype accident says so), sites dealt over the ranks."""
    import torch.distributed as dist
    os.environ["OSQ_FUSED_STEP"] = "0"        # several processes share the test GPU: no persistent grids (same path in the 1-rank run)
    sys.path.insert(0, ROOT)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers import BertConfig, BertForSequenceClassification
    from outlier_suppression_amd import calibration
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    g = np.load(os.path.join(ROOT, "tests", "golden", "bert_tiny_pipeline.npz"))
    cfg = BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=40, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     type_vocab_size=2)
    fp = BertForSequenceClassification(cfg).eval()
    fp.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}, strict=False)
    dev = torch.device("cuda:0")
    fp = fp.to(dev)
    batches = [{"input_ids": torch.from_numpy(g["input_ids"][b]).to(dev),
                "attention_mask": torch.from_numpy(g["attention_mask"][b]).to(dev),
                "token_type_ids": torch.zeros_like(torch.from_numpy(g["input_ids"][b])).to(dev)} for b in range(4)]
    w_q = NS(quantizer="FixedFakeQuantize", observer="MSEFastObserver", bit=4, symmetric=True, ch_axis=0)
    a_q = NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1)
    model = quantize_model(fp, w_q, a_q).to(dev)
    fwd = lambda m, b: m(**b)
    enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    info_w = calibration.calibrate_owned_sites(model, batches[:1], fwd, select=lambda n: "weight_fake_quant" in n)
    enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
    info_a = calibration.calibrate_owned_sites(model, batches, fwd)
    if world > 1:
        assert len(set(info_w["owner"])) == world and len(set(info_a["owner"])) == world     # every rank had work
    out = {}
    for i, (n, q) in enumerate((n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase)):
        obs = q.observer
        out[f"{i}:mn"], out[f"{i}:mx"] = obs.min_val.cpu().numpy(), obs.max_val.cpu().numpy()
        out[f"{i}:s"], out[f"{i}:z"] = q.scale.detach().cpu().numpy(), q.zero_point.detach().cpu().numpy()
        out[f"{i}:cnt"] = np.array(getattr(obs, "cnt", -1))
        out[f"{i}:side"] = np.array(str(getattr(obs, "one_side_dist", None)))
        if "_ref_f64" in obs.__dict__:
            out[f"{i}:flags"] = obs._ref_f64.cpu().numpy()
    # the model every rank ends with quantises identically: logits of the fully quantised model
    from outlier_suppression_amd.quantization import enable_quantization
    enable_quantization(model)
    with torch.no_grad():
        out["logits"] = model(**batches[1])[0].cpu().numpy()
    np.savez(os.path.join(out_dir, f"sites_w{world}_r{rank}.npz"), **out)
    if world > 1:
        dist.destroy_process_group()


def test_site_sharded_msefast_equals_one_process(tmp_path):
    """calibration.calibrate_owned_sites with 2 and 4 ranks: every statistic (value, dtype, shape), scale, zero_point,
    counter, sidedness and reference-dtype flag of every weight and activation quantizer, and the quantised model's
    logits, equal the one-process pass bit for bit -- on every rank."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_run_sites, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run_sites, args=(2, 29761, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_run_sites, args=(4, 29763, str(tmp_path)), nprocs=4, join=True)
    one = np.load(tmp_path / "sites_w1_r0.npz")
    assert len(one.files) > 100
    for world, r in ((2, 0), (2, 1), (4, 0), (4, 2), (4, 3)):
        got = np.load(tmp_path / f"sites_w{world}_r{r}.npz")
        assert sorted(got.files) == sorted(one.files)
        for k in one.files:
            a, b = got[k], one[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (world, r, k, a.dtype, b.dtype, a.shape, b.shape)
            assert bits_equal(a, b, equal_nan=(a.dtype.kind == "f")), (world, r, k)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it (the form the driver used for N = 1) re-launches itself
    under torch.distributed.run and prints ONE JSON line.  On this one-GPU box the two ranks share the device through the
    OSQ_BENCH_SHARE_GPU=1 test hook (gloo, three-launch path); with N visible devices the same command runs on RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    env = dict(os.environ, OSQ_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--settle", "0.1",
                        "--no-calib", "--no-kernel-table", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0], r.stdout[-2000:]      # ONE line, and it is the last
    from benchlib import line as BL
    BL.check_line(lines[0])                                  # under the size cap, contract keys present
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["value"] > 0
    assert out["config"]["timed_regions"] == 5 and out["config"]["ms_per_step_min"] <= out["ms_per_step"] <= out["config"]["ms_per_step_max"]
    assert out["collective"]["ranks_seen"] == 2 and out["collective"]["world_size"] == 2
    # the exchange moved the rows the timed steps recorded, and they replay to the one-process statistic on every rank
    chk = out["collective"]["exchange_check"]
    assert chk["rows_gathered"] == 10 and chk["replay_equals_one_process_loop"] and chk["same_bits_on_every_rank"] and chk["own_rows_intact"], chk
    # without the hook and without a second device the command must refuse, loudly, instead of measuring something else
    if torch.cuda.device_count() < 2:
        env.pop("OSQ_BENCH_SHARE_GPU")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "visible HIP devices" in r.stderr


def test_bench_self_launch_eight_ranks_with_sharded_calibration():
    """The command the driver runs on an 8-GPU node, on this one-GPU box through the shared-GPU hook (gloo; eight processes
    on one device): `bench.py --gpus 8` with BASELINE configs[2]'s calibration (BERT-base SQuAD, T = 384) sharded over the
    eight ranks.  Inside the run every rank asserts that the rows exchanged after the timed steps replay to the
    one-process running mean (bench.py raises otherwise); here: rc 0, one JSON line, eight distinct ranks seen, the
    check's verdicts, and the calibration section carrying wall_s / collective_s at the launched N."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    env = dict(os.environ, OSQ_BENCH_SHARE_GPU="1", OSQ_BENCH_SQUAD_CANDIDATES="3")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--settle", "0.1",
                        "--calib-configs", "2", "--no-kernel-table", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0], r.stdout[-2000:]
    from benchlib import line as BL
    BL.check_line(lines[0])
    out, detail = BL.parse_stdout(r.stdout)                  # the compact line; the sections printed before it
    assert out["n_gpus"] == 8 and out["collective"]["ranks_seen"] == 8
    chk = out["collective"]["exchange_check"]
    assert chk["rows_gathered"] == 32 and chk["replay_equals_one_process_loop"] and chk["same_bits_on_every_rank"] and chk["own_rows_intact"], chk
    assert out["calibration_summary"]["calibration_config2"]["wall_s"] == detail["calibration_config2"]["wall_s"]
    cal = detail["calibration_config2"]
    assert "error" not in cal, cal
    assert cal["n_gpus"] == 8 and cal["wall_s"] > 0 and cal["collective_s"] >= 0, cal
    # round 5: the collective's share per phase, and every rank ending the calibration with the same parameter bits
    assert set(cal["collective_phases_s"]) >= {"twc_grid_search", "learn_scale"} and cal["collective_calls"] > 0, cal
    assert cal["exchange_check"]["ranks"] == 8 and cal["exchange_check"]["same_bits_on_every_rank"], cal["exchange_check"]
    summary = detail["calibration_summary"]["configs"]["calibration_config2"]
    assert summary["same_parameters_on_every_rank"] is True and summary["collective_s"] == cal["collective_s"], summary
    # the probe regions that chose how the timed region is issued, as numbers
    cfg = detail["config"]
    assert cfg["launch_picked"] in ("graph", "eager") and isinstance(cfg["probe_regions_us_per_step"], dict), cfg
    assert out["config"]["launch_picked"] == cfg["launch_picked"]


def test_bench_refuses_two_ranks_on_one_device():
    """`bench.py --gpus N` all-gathers every rank's device identity (UUID / PCI address) and refuses to measure when N
    ranks do not sit on N different GPUs -- the failure a mis-set LOCAL_RANK or visibility mask would otherwise turn into a
    silently wrong scaling point.  Here: two gloo ranks on the one GPU of the box with the assertion left armed
    (OSQ_BENCH_SHARE_GPU=check)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    env = dict(os.environ, OSQ_BENCH_SHARE_GPU="check")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--settle", "0.1",
                        "--no-calib", "--no-kernel-table", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, r.stdout[-2000:]
    assert "one process per GPU is the contract" in r.stderr, r.stderr[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no JSON line may come out of a refused run"
