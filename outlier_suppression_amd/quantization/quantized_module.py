"""Quantizer factory and weight-quantized operators with the reference's names.

Reference: quant_transformer/quantization/quantized_module.py.  ``Quantizer(None, cfg)`` builds
an activation quantizer; ``Quantizer(nn.Linear | nn.Embedding | nn.Conv2d, cfg)`` returns a
new module of the Q-type with cloned parameters and a child named ``weight_fake_quant``
that fake-quantises the weight on every forward (quantized_module.py:56,72,98); any other
module is returned unchanged.
"""
import torch.nn.functional as F
from torch import nn

from .fake_quant import FixedFakeQuantize, LSQFakeQuantize, LSQPlusFakeQuantize
from .observer import AvgMinMaxObserver, AvgPruneMinMaxObserver, MinMaxObserver

ObserverDict = {
    "MinMaxObserver": MinMaxObserver,
    "AvgMinMaxObserver": AvgMinMaxObserver,
    "AvgPruneMinMaxObserver": AvgPruneMinMaxObserver,
}

FakeQuantizeDict = {
    "FixedFakeQuantize": FixedFakeQuantize,
    "LSQFakeQuantize": LSQFakeQuantize,
    "LSQPlusFakeQuantize": LSQPlusFakeQuantize,
}


def _register_optional_observers():
    from . import observer as _obs
    for name in ("MSEFastObserver", "AvgMSEFastObserver", "MSEObserver", "AvgMSEObserver",
                 "AvgQuantileObserver", "LSQPlusObserver"):
        if hasattr(_obs, name):
            ObserverDict[name] = getattr(_obs, name)


_register_optional_observers()


class QuantizedModule(nn.Module):
    def __init__(self, backend="academic"):
        super().__init__()
        self.backend = backend


class QuantizedOperator:
    """Marker base of the weight-quantized operators.  ``_quantized_weight`` is ``self.weight_fake_quant(self.weight)``
    (quantized_module.py:72,98) served from quantization/weight_cache.py while weight and parameters are unchanged."""

    def _quantized_weight(self):
        from .weight_cache import quantized_weight
        return quantized_weight(self)


def _build_quantizer(cfg):
    return FakeQuantizeDict[cfg.quantizer](ObserverDict[cfg.observer], bit=cfg.bit, symmetric=cfg.symmetric,
                                           ch_axis=cfg.ch_axis)


def ActivationQuantizer(a_qconfig):
    return _build_quantizer(a_qconfig)


def WeightQuantizer(w_qconfig):
    return _build_quantizer(w_qconfig)


class QLinear(QuantizedOperator, nn.Linear):
    def __init__(self, in_features, out_features, bias, w_qconfig):
        super().__init__(in_features=in_features, out_features=out_features, bias=bias)
        self.weight_fake_quant = WeightQuantizer(w_qconfig)

    def forward(self, input):
        return F.linear(input, self._quantized_weight(), self.bias)


class QConv2d(QuantizedOperator, nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                 padding_mode, w_qconfig):
        super().__init__(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                         stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias,
                         padding_mode=padding_mode)
        self.weight_fake_quant = WeightQuantizer(w_qconfig)

    def forward(self, input):
        return self._conv_forward(input, self._quantized_weight(), self.bias)


class QEmbedding(QuantizedOperator, nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, padding_idx, max_norm, norm_type, scale_grad_by_freq,
                 sparse, _weight, w_qconfig):
        super().__init__(num_embeddings=num_embeddings, embedding_dim=embedding_dim, padding_idx=padding_idx,
                         max_norm=max_norm, norm_type=norm_type, scale_grad_by_freq=scale_grad_by_freq,
                         sparse=sparse, _weight=_weight)
        self.weight_fake_quant = WeightQuantizer(w_qconfig)

    def forward(self, input):
        return F.embedding(input, self._quantized_weight(), self.padding_idx, self.max_norm,
                           self.norm_type, self.scale_grad_by_freq, self.sparse)


module_type_to_quant_weight = {nn.Linear: QLinear, nn.Conv2d: QConv2d, nn.Embedding: QEmbedding}

_CTOR_FIELDS = {
    nn.Linear: ("in_features", "out_features"),
    nn.Conv2d: ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups",
                "padding_mode"),
    nn.Embedding: ("num_embeddings", "embedding_dim", "padding_idx", "max_norm", "norm_type",
                   "scale_grad_by_freq", "sparse"),
}


def get_module_args(module):
    """Constructor arguments that rebuild ``module`` as its Q-type (quantized_module.py:110-141)."""
    for base, fields in _CTOR_FIELDS.items():
        if isinstance(module, base):
            kwargs = {f: getattr(module, f) for f in fields}
            if base is nn.Embedding:
                kwargs["_weight"] = None
            else:
                kwargs["bias"] = module.bias is not None
            return kwargs
    raise NotImplementedError(type(module))


def Quantizer(module, config):
    if module is None:
        return ActivationQuantizer(a_qconfig=config)
    qtype = module_type_to_quant_weight.get(type(module))
    if qtype is None:
        return module
    qmodule = qtype(**get_module_args(module), w_qconfig=config)
    qmodule.weight.data = module.weight.data.clone()
    if getattr(module, "bias", None) is not None:
        qmodule.bias.data = module.bias.data.clone()
    return qmodule
