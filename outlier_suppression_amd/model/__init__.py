"""Quantizer placement for transformer encoders (harness that drives the HIP hot path).

Same class names, sub-module names and quantizer call sites as the reference's
``quant_transformer/model`` so that module-name based switches, Gamma Migration and state-dict
keys line up; the transformer maths itself is stock PyTorch-ROCm.
"""
from .quant_bert import (QuantizedBertForSequenceClassification, QuantizedBertForQuestionAnswering,  # noqa: F401
                         QuantizedBertModel)
from .quant_roberta import (QuantizedRobertaForSequenceClassification, QuantizedRobertaForQuestionAnswering,  # noqa: F401
                            QuantizedRobertaModel)
from .quant_bart import QuantizedBartForConditionalGeneration, QuantizedBartModel  # noqa: F401
