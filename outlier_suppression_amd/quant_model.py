"""Wrap an FP HuggingFace model into its quantized counterpart (reference: solver/quant_model.py:31-50)."""
import copy

from .model import quant_bart, quant_bert, quant_roberta

_WRAPPERS = {
    "BertForSequenceClassification": quant_bert.QuantizedBertForSequenceClassification,
    "BertForQuestionAnswering": quant_bert.QuantizedBertForQuestionAnswering,
    "RobertaForSequenceClassification": quant_roberta.QuantizedRobertaForSequenceClassification,
    "RobertaForQuestionAnswering": quant_roberta.QuantizedRobertaForQuestionAnswering,
    "BartForConditionalGeneration": quant_bart.QuantizedBartForConditionalGeneration,
    "BartForSequenceClassification": quant_bart.QuantizedBartForSequenceClassification,
    "BartForQuestionAnswering": quant_bart.QuantizedBartForQuestionAnswering,
}


def model_type_of(model_name):
    """quant_model.py:20-27."""
    name = model_name.lower()
    for key in ("roberta", "bert", "bart"):
        if key in name:
            return key
    raise NotImplementedError(model_name)


def get_model_task_type(model_name, config_data):
    """quant_model.py:11-28."""
    name = config_data.dataset_name
    if name in ("cola", "mnli", "mrpc", "qnli", "qqp", "sst2", "rte", "stsb"):
        task_type = "glue"
    elif name in ("squad", "squad_v2"):
        task_type = name
    elif name in ("cnn_dailymail", "xsum"):
        task_type = "summ"
    else:
        raise NotImplementedError(name)
    return task_type, model_type_of(model_name)


def _get(section, key, default):
    if isinstance(section, dict):
        return section.get(key, default)
    return getattr(section, key, default)


def _set(section, key, value):
    if isinstance(section, dict):
        section[key] = value
    else:
        setattr(section, key, value)


def quantize_model(fp_model, w_qconfig, a_qconfig=None, backend="academic", is_remove_padding=True):
    """deepcopy the FP model and wrap it with qoutput=False (quant_model.py:43-49).  Two call forms:
    ``quantize_model(fp_model, w_qconfig, a_qconfig, backend, is_remove_padding)`` and the reference's
    ``quantize_model(fp_model, config)`` (quant_model.py:31-50: the parsed config with ``quant`` / ``model`` / ``data``
    sections; defaults are filled in and ``config.model.model_type`` / ``task_type`` set, as the reference does)."""
    if a_qconfig is None:
        config = w_qconfig
        quant, model_section = _get(config, "quant", None), _get(config, "model", None)
        _set(quant, "backend", _get(quant, "backend", "academic"))
        _set(quant, "is_remove_padding", _get(quant, "is_remove_padding", True))
        ln = _get(quant, "ln", None)
        if ln is None:
            ln = type(quant)() if isinstance(quant, dict) else type("ln", (), {})()
            _set(quant, "ln", ln)
        _set(ln, "delay", _get(ln, "delay", False))
        task_type, model_type = get_model_task_type(type(fp_model).__name__.lower(), _get(config, "data", None))
        _set(model_section, "model_type", model_type)
        _set(model_section, "task_type", task_type)
        w_qconfig, a_qconfig = _get(quant, "w_qconfig", None), _get(quant, "a_qconfig", None)
        backend, is_remove_padding = _get(quant, "backend", "academic"), _get(quant, "is_remove_padding", True)
        from .ptq import namespace
        w_qconfig, a_qconfig = namespace(w_qconfig), namespace(a_qconfig)
    if backend != "academic":
        # the reference's 'tensorrt' placement (extra residual-branch sites, quant_bert.py:204-216, quant_bart.py:305-307,
        # 401-404) is out of this package's scope (no shipped config enables it) -- refused, never silently
        # treated as 'academic'
        raise NotImplementedError(f"backend={backend!r}: the quantizer placement of this package is 'academic'")
    cls = type(fp_model).__name__
    if cls not in _WRAPPERS:
        raise NotImplementedError(f"no quantized counterpart for {cls} yet")
    fp_model.eval()
    model = _WRAPPERS[cls](copy.deepcopy(fp_model), w_qconfig, a_qconfig, qoutput=False, backend=backend,
                           is_remove_padding=is_remove_padding)
    from .quantization.weight_cache import adopt
    return adopt(model.eval())
