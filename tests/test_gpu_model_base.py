"""BASELINE configs[1] at FULL model size on the GPU against the reference's own run: a seeded, randomly initialised
BERT-base (12 layers, hidden 768, 98 activation + 77 weight quantizers) through wrap -> gamma migration -> weight
calibration -> token-wise-clipping search -> learn-scale -> quantized logits, compared with
tests/golden/bert_base_pipeline.npz (tests/golden/make_golden_bert_base.py ran quant_transformer's own functions on
the CPU in the build container).

The weights are not in the fixture: the model is re-created from the same seeds (same torch build) and its per-tensor
checksums are checked first.  What can be compared how tightly:

  * everything computed with fake-quant OFF -- FP logits, gamma migration, weight scales, and every activation
    quantizer's scale / zero_point after an observer pass (token-wise clipping runs its observer passes on the FP
    model) -- differs only by fp32 GEMM rounding (rocBLAS here, MKL there): tight bars;
  * everything downstream of a 6-bit fake-quant of a 12-layer random network is chaotic in the GEMM rounding (one
    activation on the other side of a rounding boundary moves everything after it): the per-candidate losses and the
    quantized logits are compared as distributions, against the size of the quantization noise itself.
"""
import importlib.util
import logging
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

A_Q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
W_Q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
REPORT = os.environ.get("OSQ_REPORT_BASE") == "1"


def _generator_module():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_bert_base", os.path.join(here, "make_golden_bert_base.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Grab(logging.Handler):
    def __init__(self):
        super().__init__()
        self.losses = []

    def emit(self, record):
        m = record.getMessage()
        if m.startswith("the ratio is"):
            self.losses.append(float(m.split("the loss is")[1]))


def test_bert_base_pipeline_matches_reference(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import token_wise_clipping as TWC, util_layernorm as UL
    from outlier_suppression_amd.gamma_migration import delay_ln
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization, disable_all
    from outlier_suppression_amd.quantization.state import set_observer_name
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    g = golden("bert_base_pipeline")
    gen = _generator_module()
    assert [gen.SEED_MODEL, gen.SEED_LN, gen.SEED_DATA] == [int(v) for v in g["seeds"]]
    dev = torch.device("cuda:0")
    torch.set_num_threads(8)
    fp = gen.build_fp()
    for k, v in gen.checksums(fp).items():     # the seeded initialisation is the one the reference ran on
        if not np.allclose(v, g[f"sum::{k}"], rtol=1e-9, atol=1e-9):
            pytest.skip(f"seeded BERT-base initialisation differs on this host ({k}): fixture not applicable")
    fp = fp.to(dev)
    batches = [{"input_ids": torch.from_numpy(i).to(dev), "attention_mask": torch.from_numpy(a).to(dev),
                "token_type_ids": torch.zeros_like(torch.from_numpy(i)).to(dev)} for i, a in zip(g["input_ids"], g["attention_mask"])]
    report = {}

    model = quantize_model(fp, W_Q, A_Q).to(dev)

    def logits():
        with torch.no_grad():
            return np.stack([model(**b)[0].float().cpu().numpy() for b in batches])
    fp_logits = logits()
    logit_scale = float(np.abs(g["logits_wrapped_fp"]).max())
    report["fp logits"] = float(np.abs(fp_logits - g["logits_wrapped_fp"]).max())
    assert report["fp logits"] < 2e-5 * max(1.0, logit_scale)
    with torch.no_grad():
        fp_output = [model(**b)[0].detach() for b in batches]

    model = delay_ln(model, NS(a_qconfig=A_Q, w_qconfig=W_Q), NS(model_type="bert", task_type="glue"))
    report["after gamma"] = float(np.abs(logits() - g["logits_after_gamma"]).max())
    assert report["after gamma"] < 2e-5 * max(1.0, logit_scale)

    enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    with torch.no_grad():
        model(**batches[0])
    disable_all(model)
    set_observer_name(model)
    names = [n for n, m in model.named_modules() if isinstance(m, QuantizeBase)]
    assert names == [str(s) for s in g["q_names"]]
    quantizers = dict((n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase))
    acts = [i for i, n in enumerate(names) if quantizers[n].scale.numel() == 1]
    assert len(names) == 175 and len(acts) == 98

    def scale_of(i):
        return quantizers[names[i]].scale.detach().reshape(-1).double().cpu().numpy()

    def zp_of(i):
        return quantizers[names[i]].zero_point.detach().reshape(-1).double().cpu().numpy()
    # per-channel weight scales: MinMax over identical weights (gamma-folded ones to GEMM-free rounding) -- checksum + first 8
    for i, n in enumerate(names):
        if i in acts:
            continue
        s = scale_of(i)
        np.testing.assert_allclose(np.concatenate([[s.sum(), np.abs(s).max()], s[:8]]), g[f"q_after_twc_scale::{i}"], rtol=2e-6, err_msg=n)

    # ---- the percentile search: observer passes on the FP model, quantized passes for the losses
    TWC.task_type, TWC.model_type = "glue", "bert"
    h = _Grab()
    TWC.logger.addHandler(h)
    TWC.logger.setLevel(logging.INFO)
    iters, step = int(g["twc_grid"][0]), float(g["twc_grid"][1])
    ratio = TWC.find_ratio_cached(NS(model=model), batches, fp_output, {"iters": iters, "step": step})
    TWC.logger.removeHandler(h)
    ref_losses = g["twc_losses"]
    rel = np.abs(np.array(h.losses) - ref_losses) / ref_losses
    report["loss rel dev (max, median)"] = (float(rel.max()), float(np.median(rel)))
    report["losses"] = [round(x, 5) for x in h.losses]
    report["ref losses"] = [round(float(x), 5) for x in ref_losses]
    report["ratio (ours, ref)"] = (ratio, float(g["best_ratio"][0]))
    # the same search once more with every LayerNorm site as ONE fused launch (opt-in: values differ from the eager
    # sequence at the 1e-6 level) and in the literal order: how far this package is from ITSELF under a rounding-level
    # change is the yardstick
    h2 = _Grab()
    TWC.logger.addHandler(h2)
    fuse_before = UL.FUSE_LAYERNORM
    UL.FUSE_LAYERNORM = not fuse_before
    try:
        TWC.find_ratio(NS(model=model), batches, fp_output, {"iters": iters, "step": step})
    finally:
        UL.FUSE_LAYERNORM = fuse_before
        TWC.logger.removeHandler(h2)
    rel_self = np.abs(np.array(h2.losses) - np.array(h.losses)) / np.array(h.losses)
    report["loss rel dev ours(fused LayerNorm sites) vs ours(default) (max, median)"] = (float(rel_self.max()), float(np.median(rel_self)))

    def compare_table(prefix, tag):
        worst_s, worst_z = 0.0, 0.0
        for i in acts:
            s_ref, z_ref = g[f"{prefix}_scale::{i}"], g[f"{prefix}_zp::{i}"]
            worst_s = max(worst_s, float(np.abs(scale_of(i) - s_ref).max() / np.abs(s_ref).max()))
            worst_z = max(worst_z, float(np.abs(zp_of(i) - z_ref).max()))
        report[tag] = (worst_s, worst_z)
        return worst_s, worst_z
    # a fixed percentile: does not depend on which candidate won
    TWC.set_ratio(model, float(g["probe_ratio"][0]))
    TWC.calibrate(model, batches)
    ws, wz = compare_table("q_at_probe", "scales at probe ratio (rel, zp)")
    assert ws < 1e-4 and wz <= 1.0
    # ---- the INTEGER tensors at full model size (the north star's bar is stated on x_quant): every quantizer's input of
    # one forward, quantised with this run's (scale, zero_point) and with the reference run's -- activations: the
    # fixture's parameters at the probe percentile; weights: the reference's MinMax rule (observer.py:122-145,
    # calculate_qparams 101-119) restated by the oracle on the very weight tensor, rows reduced in NumPy
    from oracle import observer_oracle as OB
    from test_gpu_model import format_integer_report, integer_tensor_report

    def weight_reference(name, w):
        st = OB.ObserverState(bit=W_Q.bit, symmetric=True, ch_axis=0)
        OB.observe_minmax(st, w.detach().cpu().numpy())
        return st.qparams()
    ref_s = [g[f"q_at_probe_scale::{i}"] if i in acts else None for i in range(len(names))]
    ref_z = [g[f"q_at_probe_zp::{i}"] if i in acts else None for i in range(len(names))]
    rows = integer_tensor_report(model, batches[0], names, ref_s, ref_z, reference_fn=weight_reference)
    act_set = {names[i] for i in acts}
    w_frac = np.array([r[1] for r in rows if r[0] not in act_set])
    a_frac = np.array([r[1] for r in rows if r[0] in act_set])
    report["integer tensors: weights differing (max), activations differing (median, max)"] = (float(w_frac.max()), float(np.median(a_frac)), float(a_frac.max()))
    if os.environ.get("OSQ_PARITY_REPORT_DIR"):
        with open(os.path.join(os.environ["OSQ_PARITY_REPORT_DIR"], "bert_base_integer_tensors.md"), "w") as f:
            f.write(format_integer_report("BERT-base (12 layers, 98 activation + 77 weight quantizers), observer pass at the probe percentile: "
                                          "x_quant with this run's parameters vs the reference run's", rows) + "\n")
    assert len(w_frac) == 77 and len(a_frac) >= 96
    assert w_frac.max() == 0.0, report
    # measured on the MI355X (profiles/r04_bert_base_integer_tensors.md): median 2.4e-7 (one entry of a 3-4 M element tensor), worst 3.8e-6
    assert float(np.median(a_frac)) <= 1e-6 and float(a_frac.max()) <= 1e-4, report
    # the reference's winner
    TWC.set_ratio(model, float(g["best_ratio"][0]))
    TWC.calibrate(model, batches)
    ws, wz = compare_table("q_after_twc", "scales at reference's ratio (rel, zp)")
    assert ws < 1e-4 and wz <= 1.0
    TWC.enable_quantization(model)
    aq = logits()
    noise = float(np.abs(g["logits_act_quant"] - g["logits_wrapped_fp"]).mean())       # what 6-bit activations do to the logits
    report["act-quant logits: mean |ours - ref|, mean |ref - fp|"] = (float(np.abs(aq - g["logits_act_quant"]).mean()), noise)

    lr, epochs = float(g["learn"][0]), int(g["learn"][1])
    before = {i: float(scale_of(i)[0]) for i in acts}
    TWC.learn_scale(NS(model=model), batches, fp_output, {"lr": lr, "epoch": epochs})
    off = [abs(float(scale_of(i)[0]) - float(g[f"q_after_learn_scale::{i}"][0])) / lr for i in acts]
    moved = [abs(float(scale_of(i)[0]) - before[i]) / lr for i in acts]
    moved_ref = [abs(float(g[f"q_after_learn_scale::{i}"][0]) - float(g[f"q_after_twc_scale::{i}"][0])) / lr for i in acts]
    report["learn-scale: |ours - ref| in lr steps (max, median)"] = (max(off), float(np.median(off)))
    report["learn-scale: movement in lr steps ours / ref (median)"] = (float(np.median(moved)), float(np.median(moved_ref)))
    enable_quantization(model)
    fq = logits()
    noise_f = float(np.abs(g["logits_full_quant"] - g["logits_wrapped_fp"]).mean())
    report["full-quant logits: mean |ours - ref|, mean |ref - fp|"] = (float(np.abs(fq - g["logits_full_quant"]).mean()), noise_f)
    if REPORT:
        for k, v in report.items():
            print(f"[bert-base] {k}: {v}")
    # ---- bars on the chaotic quantities (set from the measurement recorded in DESIGN.md section 2)
    assert float(np.median(rel)) < 0.25 and float(rel.max()) < 0.5, report
    assert float(np.mean(rel)) < 3.0 * float(np.mean(rel_self)) + 0.02, report      # no further from the reference than from itself
    assert report["act-quant logits: mean |ours - ref|, mean |ref - fp|"][0] < 1.0 * noise, report
    assert report["full-quant logits: mean |ours - ref|, mean |ref - fp|"][0] < 1.0 * noise_f, report
    n_steps = epochs * len(batches)
    assert max(off) <= n_steps + 0.5 and float(np.median(off)) <= 0.75 * n_steps, report
