"""Gamma Migration: move LayerNorm's gamma into the next linear layers and the shortcut.

Reference: quant_transformer/solver/gamma_migration.py.  Same walk over ``named_modules()`` in
registration order (a QuantizedLayerNorm followed by a GammaResidual marks one migration
site), same list of weights per site; the fold ``W[:, j] *= gamma[j]`` is a HIP kernel.
Works on any model that uses the class names / attribute paths below -- this package's own
model counterparts and the reference's ``quant_transformer/model`` classes alike.
"""
import logging
from collections import OrderedDict

import torch

from . import ops
from .util_layernorm import GammaResidual, QuantizedLayerNorm, QuantizedSplitLayerNorm

logger = logging.getLogger("transformer")


def get_weight_modules(model, config_model):
    """gamma_migration.py:8-42: per migration site, the linears whose input is that LayerNorm's output."""
    n = model.config.num_hidden_layers
    kind = config_model.model_type
    sites = []
    if kind in ("bert", "roberta"):
        for layer in getattr(model, kind).encoder.layer[:n]:
            att = layer.attention.self
            sites.append([att.query, att.key, att.value])
            sites.append([layer.intermediate.dense])
    elif kind == "bart":
        for layer in model.model.encoder.layers[:n]:
            a = layer.self_attn
            sites.append([a.q_proj, a.k_proj, a.v_proj])
            sites.append([layer.fc1])
        for layer in model.model.decoder.layers[:n]:
            a = layer.self_attn
            sites.append([a.q_proj, a.k_proj, a.v_proj])
            sites.append([layer.encoder_attn.q_proj])
            sites.append([layer.fc1])
    return sites


def _set_submodule(root, dotted, new):
    *parents, leaf = dotted.split(".")
    for p in parents:
        root = getattr(root, p)
    setattr(root, leaf, new)


@torch.no_grad()
def gamma_migration(model, config_quant, config_model):
    """gamma_migration.py:46-76."""
    sites = get_weight_modules(model, config_model)
    site = 0
    pending = None      # (name, QuantizedLayerNorm) waiting for its GammaResidual
    for name, module in OrderedDict(model.named_modules()).items():
        if isinstance(module, GammaResidual) and pending is not None:
            ln_name, old = pending
            gamma = old.layernorm.weight.data
            split = QuantizedSplitLayerNorm(old.layernorm, config_quant.w_qconfig, config_quant.a_qconfig,
                                            old.qoutput, old.backend).to(gamma.device).eval()
            _set_submodule(model, ln_name, split)
            module.set_gamma(gamma)
            for lin in sites[site]:
                ops.gamma_fold_(lin.weight.data, gamma)
            site += 1
            pending = None
        if isinstance(module, QuantizedLayerNorm):
            pending = (name, module)
    logger.info("gamma migration: %d LayerNorms split", site)
    return model


def delay_ln(model, config_quant, config_model):
    """gamma_migration.py:79-81."""
    return gamma_migration(model, config_quant, config_model)
