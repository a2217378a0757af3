"""Switches for observers / fake-quantizers, selected by module-name substring.

Reference: quant_transformer/quantization/state.py.  Same function names, same selection
rule (``quantizer_type in name``; names in ``except_quantizer`` are switched fully off),
same logger name.
"""
import logging

from .. import ops
from .fake_quant import LSQFakeQuantize, LSQPlusFakeQuantize, QuantizeBase
from .observer import ObserverBase

logger = logging.getLogger("transformer")


def _quantizers(model):
    return ((n, m) for n, m in model.named_modules() if isinstance(m, QuantizeBase))


def _transition(model):
    """Every state change is a synchronisation point of the calibration flow: surface a time-out of the persistent
    launches issued in the state that ends here (ops.check_persistent: free when none ran), and drop the kept
    fake-quantised weights -- callers edit weights between states through paths the cache key cannot see (`.data`
    arithmetic as in the reference's gamma_migration.py:70-71, foreign kernels)."""
    ops.check_persistent("quantizer state change")
    from . import weight_cache
    weight_cache.invalidate(model)


def _switch(model, quantizer_type, except_quantizer, observe, quantize, learnable_observe=None):
    """Apply (observer on/off, fake-quant on/off) to selected quantizers; everything else off."""
    _transition(model)
    for name, q in _quantizers(model):
        selected = quantizer_type in name and not (except_quantizer is not None and name in except_quantizer)
        want_obs, want_fq = (observe, quantize) if selected else (False, False)
        if selected and learnable_observe is not None and isinstance(q, (LSQFakeQuantize, LSQPlusFakeQuantize)):
            want_obs = learnable_observe
            logger.info("Extrally disable observer for LSQ/LSQPlusFakeQuantize during training!")
        (q.enable_observer if want_obs else q.disable_observer)()
        (q.enable_fake_quant if want_fq else q.disable_fake_quant)()
        logger.debug("%s: observer=%d fake_quant=%d", name, int(want_obs), int(want_fq))


def enable_calibration_woquantization(model, quantizer_type="fake_quant", except_quantizer=None):
    """state.py:7-19: observe, do not quantize."""
    logger.info("Enable observer and Disable quantize for %s", quantizer_type)
    _switch(model, quantizer_type, except_quantizer, observe=True, quantize=False)


def enable_calibration_quantization(model, quantizer_type="fake_quant", except_quantizer=None):
    """state.py:22-38: observe and quantize; learnable quantizers keep their observer off."""
    logger.info("Enable observer and Enable quantize for %s", quantizer_type)
    _switch(model, quantizer_type, except_quantizer, observe=True, quantize=True, learnable_observe=False)


def enable_quantization(model, quantizer_type="fake_quant", except_quantizer=None):
    """state.py:41-53: quantize with frozen statistics."""
    logger.info("Disable observer and Enable quantize.")
    _switch(model, quantizer_type, except_quantizer, observe=False, quantize=True)


def disable_all(model):
    """state.py:56-62."""
    logger.info("Disable observer and disable quantize.")
    _transition(model)
    for _, q in _quantizers(model):
        q.disable_observer()
        q.disable_fake_quant()


def set_observer_name(model):
    """state.py:65-69: every observer learns its dotted module path (pruning checks it for
    'attention_probs', observer.py:62)."""
    logger.info("set name for obsever")
    for name, m in model.named_modules():
        if isinstance(m, ObserverBase):
            m.set_name(name)
