"""Token-wise clipping driver: coarse percentile search + fine LSQ+ scale learning.

Reference: quant_transformer/solver/token_wise_clipping.py (same function names and order of
operations).  It is the direct caller of the HIP hot path: every ``calibrate`` pass runs every
activation quantizer once per batch.  ``find_ratio_cached`` is the MI355X-first variant of the
coarse search (same results, see its docstring); ``find_ratio`` is the literal one.
"""
import logging

import torch
from torch.nn import MSELoss

from .quantization.fake_quant import LSQFakeQuantize, LSQPlusFakeQuantize, QuantizeBase
from .quantization.state import disable_all

logger = logging.getLogger("transformer")
task_type = None
model_type = None
loss_fct = MSELoss()

a_bit_iters = {8: 0.1, 6: 0.3, 4: 0.9}


def _act_quantizers(model):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]


def set_ratio(model, ratio):
    """token_wise_clipping.py:12-19."""
    for _, q in _act_quantizers(model):
        q.observer.set_percentile(ratio)
        q.observer.cnt = 0
        q.disable_fake_quant()
        q.enable_observer()


def enable_quantization(model):
    """token_wise_clipping.py:22-26: activations quantized with frozen statistics, weights untouched."""
    for _, q in _act_quantizers(model):
        q.disable_observer()
        q.enable_fake_quant()


def batch_loss(outputs, batch, target, task=None):
    """token_wise_clipping.py:35-45 / 93-104: MSE against the FP model's outputs."""
    task = task or task_type
    if task == "glue":
        return loss_fct(outputs[0], target)
    if task in ("squad", "squad_v2"):
        keep = batch["attention_mask"] == 1
        return loss_fct(outputs[0][keep], target[0]) + loss_fct(outputs[1][keep], target[1])
    if task == "summ":
        return loss_fct(outputs[0][batch["decoder_attention_mask"] == 1, :], target)
    raise NotImplementedError(task)


def calibrate(model, fp_input, fp_output=None):
    """token_wise_clipping.py:29-47: forward over the cached calibration batches; optional loss sum."""
    loss = 0
    with torch.no_grad():
        for i, batch in enumerate(fp_input):
            outputs = model(**batch)
            if fp_output is not None:
                loss += batch_loss(outputs, batch, fp_output[i])
    return loss


def find_ratio(trainer, fp_input, fp_output, param):
    """token_wise_clipping.py:50-66: grid over the percentile; first best wins (strict '>')."""
    model = trainer.model
    best, best_loss = 0, 10000000
    for i in range(param["iters"]):
        set_ratio(model, 1.0 - param["step"] * i)
        calibrate(model, fp_input)
        enable_quantization(model)
        cur = calibrate(model, fp_input, fp_output)
        logger.info("the ratio is {}, the loss is {}".format(1.0 - param["step"] * i, cur))
        if best_loss > cur:
            best_loss, best = cur, i
    ratio = 1.0 - param["step"] * best
    logger.info("the best percentile is {}".format(ratio))
    set_ratio(model, ratio)
    calibrate(model, fp_input)
    return ratio


def learn_scale(trainer, fp_input, fp_output, config_quant_learn):
    """token_wise_clipping.py:72-108: Adam on (scale, zero_point) of every activation quantizer."""
    model = trainer.model
    disable_all(model)
    logger.info("*** begin learn the scale now! ***")
    params = []
    for _, q in _act_quantizers(model):
        q.enable_fake_quant()
        q.disable_observer()
        if isinstance(q, LSQPlusFakeQuantize):
            params += [q.scale, q.zero_point]
        elif isinstance(q, LSQFakeQuantize):
            params.append(q.scale)
    opt = torch.optim.Adam(params, lr=config_quant_learn["lr"])
    steps = config_quant_learn["epoch"] * len(fp_input)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=0.0)
    for _ in range(config_quant_learn["epoch"]):
        for i, batch in enumerate(fp_input):
            opt.zero_grad()
            loss = batch_loss(model(**batch), batch, fp_output[i])
            loss.backward()
            opt.step()
            sched.step()


def cac_step_iters(a_bit, bs, config_data):
    """token_wise_clipping.py:118-129."""
    seq = config_data.max_seq_length if hasattr(config_data, "max_seq_length") else config_data.max_source_length
    step = min(float(format(128 * 32 * 0.01 / bs / seq, ".2g")), 0.01)
    iters = int(a_bit_iters[a_bit] / step)
    logger.info("the step is {}, the iters is {}".format(step, iters))
    return step, iters


def token_wise_clipping(trainer, fp_input, fp_output, config):
    """token_wise_clipping.py:132-146."""
    global model_type, task_type
    model_type, task_type = config.model.model_type, config.model.task_type
    logger.info("*** Evaluate Token Percentile ***")
    step, iters = cac_step_iters(config.quant.a_qconfig.bit, trainer.args.per_device_eval_batch_size, config.data)
    return find_ratio(trainer, fp_input, fp_output, {"iters": getattr(config.quant, "iters", iters),
                                                     "step": getattr(config.quant, "step", step)})
