"""Wrap an FP HuggingFace model into its quantized counterpart (reference: solver/quant_model.py:31-50)."""
import copy

from .model import quant_bart, quant_bert, quant_roberta

_WRAPPERS = {
    "BertForSequenceClassification": quant_bert.QuantizedBertForSequenceClassification,
    "BertForQuestionAnswering": quant_bert.QuantizedBertForQuestionAnswering,
    "RobertaForSequenceClassification": quant_roberta.QuantizedRobertaForSequenceClassification,
    "RobertaForQuestionAnswering": quant_roberta.QuantizedRobertaForQuestionAnswering,
    "BartForConditionalGeneration": quant_bart.QuantizedBartForConditionalGeneration,
}


def model_type_of(model_name):
    """quant_model.py:20-27."""
    name = model_name.lower()
    for key in ("roberta", "bert", "bart"):
        if key in name:
            return key
    raise NotImplementedError(model_name)


def quantize_model(fp_model, w_qconfig, a_qconfig, backend="academic", is_remove_padding=True):
    """deepcopy the FP model and wrap it with qoutput=False (quant_model.py:43-49)."""
    from .quant_model_checks import require_academic
    require_academic(backend)
    cls = type(fp_model).__name__
    if cls not in _WRAPPERS:
        raise NotImplementedError(f"no quantized counterpart for {cls} yet")
    fp_model.eval()
    model = _WRAPPERS[cls](copy.deepcopy(fp_model), w_qconfig, a_qconfig, qoutput=False, backend=backend,
                           is_remove_padding=is_remove_padding)
    from .quantization.weight_cache import adopt
    return adopt(model.eval())
