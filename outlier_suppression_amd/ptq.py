"""The calibration block of the reference's PTQ entry points, as one function.

``solver/ptq_glue_quant.py:226-251`` (and the same block in ``ptq_qa_quant.py:235-277``, ``ptq_summ_quant.py``) is the direct
caller of the hot path: FP targets -> optional gamma migration -> weight calibration on the first batch -> either
token-wise clipping (+ learn-scale) or a plain activation calibration -> every quantizer on.  Dataset / tokenizer /
HF ``Trainer`` handling stays with the reference's solvers (out of scope); what they hand to the block --
calibration batches resident on the device and the ``quant:`` section of a shipped ``exp/**/config.yaml`` -- is what ``run``
takes.  The four distinct ``quant:`` sections the reference ships (exp/bert_ptq/{minmax,mse,quantile,twc_fine_gamma}) are
``SHIPPED_QUANT_SECTIONS``.
"""
import logging
from types import SimpleNamespace as NS

import torch

from . import token_wise_clipping as TWC
from .gamma_migration import delay_ln
from .quantization import disable_all, enable_calibration_woquantization, enable_quantization
from .quantization.state import set_observer_name

logger = logging.getLogger("transformer")

_W6 = dict(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)


def _section(a_quantizer, a_observer, delay):
    return NS(is_remove_padding=True, ln=NS(delay=delay), calibrate=256, w_qconfig=NS(**_W6),
              a_qconfig=NS(quantizer=a_quantizer, observer=a_observer, bit=6, symmetric=False, ch_axis=-1))


# the `quant:` sections of exp/bert_ptq/<name>/cola/config.yaml (twc_fine_gamma: also every other task, cnn_dailymail, xsum)
SHIPPED_QUANT_SECTIONS = {
    "minmax": _section("FixedFakeQuantize", "AvgMinMaxObserver", False),
    "mse": _section("FixedFakeQuantize", "AvgMSEFastObserver", False),
    "quantile": _section("FixedFakeQuantize", "AvgQuantileObserver", False),
    "twc_fine_gamma": _section("LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", True),
}


def namespace(section):
    """A parsed YAML ``quant:`` section (nested dicts, EasyDict or namespaces) as attribute-style namespaces."""
    if isinstance(section, dict):
        return NS(**{k: namespace(v) for k, v in section.items()})
    return section


def prepare_input_output(model, batches, keep=None):
    """ptq_glue_quant.py:94-107: FP targets of the wrapped model.  ``batches``: dicts of device tensors without labels.
    ``keep(batch, outputs)`` post-processes the outputs (the QA and summarisation entry points mask them,
    ptq_qa_quant.py:126-131, ptq_summ_quant.py:148-149); default: the first output."""
    fp_output = []
    with torch.no_grad():
        for batch in batches:
            out = model(**batch)
            fp_output.append(keep(batch, out) if keep is not None else out[0].detach())
    return list(batches), fp_output


def calibrate(model, fp_input):
    """ptq_glue_quant.py:110-114.  Observers whose fake-quantizer is off are reduced together after every forward
    (quantization/deferred.py: same statistics, a handful of launches per forward)."""
    logger.info("*** Calibrate ***")
    TWC.calibrate(model, fp_input)


def run(model, fp_input, fp_output, config_quant, config_model, per_device_eval_batch_size=32, config_data=None,
        search="cached", learn_input=None, learn_output=None, full_quantization=True):
    """ptq_glue_quant.py:228-251.  ``model``: the wrapped model (``quant_model.quantize_model``), every quantizer off.
    ``config_quant``: a ``quant:`` section; ``config_model``: ``model_type`` / ``task_type`` (ptq_glue_quant.py:230-232).
    ``search``: "cached" (per-token extrema kept, one re-threshold launch per candidate; same result) or "literal".
    ``learn_input`` / ``learn_output``: the re-prepared smaller batches of the QA entry point (ptq_qa_quant.py:262-267).
    Returns the calibrated model (gamma migration replaces sub-modules)."""
    config_quant = namespace(config_quant)
    config_model = namespace(config_model)
    if getattr(getattr(config_quant, "ln", None), "delay", False):
        model = delay_ln(model, config_quant, config_model)
    # calibrate the weight
    enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
    calibrate(model, [fp_input[0]])
    if "PruneMinMaxObserver" in config_quant.a_qconfig.observer:
        disable_all(model)
        set_observer_name(model)
        TWC.model_type, TWC.task_type = config_model.model_type, config_model.task_type
        # token_wise_clipping.py:140-146: step / iters from cac_step_iters, EACH overridden on its own by the section
        if hasattr(config_quant, "iters") and hasattr(config_quant, "step"):
            step, iters = config_quant.step, config_quant.iters          # nothing left to derive: config_data not needed
        else:
            if config_data is None:
                raise ValueError("ptq.run: the quant section gives no `iters` + `step`, so the grid comes from "
                                 "cac_step_iters(a_bit, batch size, config.data) (token_wise_clipping.py:118-129): pass "
                                 "config_data with max_seq_length (GLUE / SQuAD) or max_source_length (summarisation)")
            step, iters = TWC.cac_step_iters(config_quant.a_qconfig.bit, per_device_eval_batch_size, namespace(config_data))
        grid = {"iters": getattr(config_quant, "iters", iters), "step": getattr(config_quant, "step", step)}
        trainer = NS(model=model)
        (TWC.find_ratio_cached if search == "cached" else TWC.find_ratio)(trainer, fp_input, fp_output, grid)
        if "LSQ" in config_quant.a_qconfig.quantizer:
            learn = getattr(config_quant, "learn", {"lr": 1e-5, "epoch": 3})
            learn = learn if isinstance(learn, dict) else vars(learn)
            TWC.learn_scale(trainer, learn_input or fp_input, learn_output or fp_output, learn)
    else:
        # calibrate the activation
        enable_calibration_woquantization(model, quantizer_type="act_fake_quant")
        calibrate(model, fp_input)
    if full_quantization:
        enable_quantization(model)         # ptq_glue_quant.py:249-251
    from . import ops
    ops.check_persistent("ptq.run")
    return model
