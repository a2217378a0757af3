"""Losses of the task-model wrappers (part of their forward contract: SURVEY 8f N2).

The reference's task models return the loss in front of their outputs when ``labels`` (or ``start_positions`` /
``end_positions``) are given (model/quant_bert.py:656-680, 744-765, quant_bart.py:1104-1111, 1248-1274, 1376-1392); an
evaluation loop (HF ``Trainer.evaluate``) reads ``outputs[0]`` as the loss then.  The wrappers of this package return plain
tuples (no ``ModelOutput``, ``return_dict`` is ignored); with labels the tuple starts with the loss, computed exactly as the
reference does."""
import torch
from torch.nn import functional as F


def classification_loss(config, num_labels, logits, labels):
    """quant_bert.py:656-680 (the same block in quant_roberta.py and quant_bart.py:1248-1270); sets config.problem_type
    on first use as the reference does."""
    if labels is None:
        return None
    if config.problem_type is None:
        if num_labels == 1:
            config.problem_type = "regression"
        elif num_labels > 1 and labels.dtype in (torch.long, torch.int):
            config.problem_type = "single_label_classification"
        else:
            config.problem_type = "multi_label_classification"
    if config.problem_type == "regression":
        return F.mse_loss(logits.squeeze(), labels.squeeze()) if num_labels == 1 else F.mse_loss(logits, labels)
    if config.problem_type == "single_label_classification":
        return F.cross_entropy(logits.view(-1, num_labels), labels.view(-1))
    return F.binary_cross_entropy_with_logits(logits, labels)


def span_loss(start_logits, end_logits, start_positions, end_positions):
    """quant_bert.py:748-765 (quant_roberta.py, quant_bart.py:1376-1392)."""
    if start_positions is None or end_positions is None:
        return None
    if start_positions.dim() > 1:
        start_positions = start_positions.squeeze(-1)
    if end_positions.dim() > 1:
        end_positions = end_positions.squeeze(-1)
    ignored = start_logits.size(1)      # positions outside the inputs are ignored
    start_positions, end_positions = start_positions.clamp(0, ignored), end_positions.clamp(0, ignored)
    return (F.cross_entropy(start_logits, start_positions, ignore_index=ignored) +
            F.cross_entropy(end_logits, end_positions, ignore_index=ignored)) / 2


def lm_loss(logits, labels, vocab_size):
    """quant_bart.py:1104-1107."""
    if labels is None:
        return None
    return F.cross_entropy(logits.view(-1, vocab_size), labels.view(-1))


def with_loss(loss, outputs):
    return outputs if loss is None else (loss,) + tuple(outputs)
