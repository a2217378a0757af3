"""LayerNorm wrappers used by Gamma Migration, with the reference's class names.

Reference: quant_transformer/model/util_layernorm.py.  The normalisation itself stays stock
PyTorch-ROCm (model maths is out of scope); the quantizer on the output is
the HIP path, beta/gamma is computed by the HIP split-bias kernel.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .quantization import QuantizedModule, Quantizer


class QuantizedLayerNorm(QuantizedModule):
    """util_layernorm.py:6-18: LayerNorm followed by an activation quantizer (seq axis 1)."""

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.layernorm = org_module
        if qoutput:
            self.layernorm_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, hidden_states, observation_mask=None):
        hidden_states = self.layernorm(hidden_states)
        if self.qoutput:
            hidden_states = self.layernorm_post_act_fake_quantize(hidden_states, observation_mask, 1)
        return hidden_states


class QuantizedSplitLayerNorm(QuantizedModule):
    """util_layernorm.py:21-37: the non-scaling LayerNorm  X' = (x - mu)/sigma + beta/gamma.

    Two reference quirks are kept on purpose: the inner LayerNorm is built
    with PyTorch's default eps = 1e-5 (not the model's), and the output quantizer is a NEW one.
    """

    def __init__(self, org_module, w_qconfig, a_qconfig, qoutput=True, backend="academic"):
        super().__init__(backend=backend)
        self.qoutput = qoutput
        self.layernorm = nn.LayerNorm(org_module.normalized_shape, elementwise_affine=False)
        beta, gamma = org_module.bias.data.detach(), org_module.weight.data.detach()
        if beta.is_cuda:
            shifted = ops.gamma_split_bias(beta, gamma)
        else:
            raise RuntimeError("QuantizedSplitLayerNorm: LayerNorm parameters must be on a HIP device "
                               "(gamma migration runs after model.cuda(), ptq_glue_quant.py:212-232)")
        self.bias = nn.Parameter(shifted)
        if qoutput:
            self.layernorm_post_act_fake_quantize = Quantizer(None, a_qconfig)

    def forward(self, hidden_states, observation_mask=None):
        hidden_states = F.layer_norm(hidden_states, self.layernorm.normalized_shape, None, None, self.layernorm.eps)
        hidden_states += self.bias
        if self.qoutput:
            hidden_states = self.layernorm_post_act_fake_quantize(hidden_states, observation_mask, 1)
        return hidden_states


class GammaResidual(nn.Module):
    """util_layernorm.py:40-52: shortcut that re-applies gamma after migration: input*gamma + hidden."""

    def __init__(self):
        super().__init__()
        self.mul_gamma = False

    def set_gamma(self, gamma):
        self.mul_gamma = True
        self.gamma = nn.Parameter(gamma.data.detach().clone())

    def forward(self, input, hidden_states):
        needs_graph = torch.is_grad_enabled() and (input.requires_grad or hidden_states.requires_grad or
                                                   (self.mul_gamma and self.gamma.requires_grad))
        if (not needs_graph and input.is_cuda and input.dtype == torch.float32 and input.shape == hidden_states.shape):
            return ops.gamma_residual(input, hidden_states, self.gamma.data if self.mul_gamma else None)
        if self.mul_gamma:
            input = input * self.gamma
        return input + hidden_states
