"""What this package's own task-model wrappers do NOT do, said loudly.

The reference's task models return HF ``ModelOutput``s with a loss when ``labels`` are given.  The wrappers here exist for
calibration and quantised inference of the hot path: they return plain tuples of logits.  A training / evaluation loop
that passes ``labels`` would otherwise read the first logits tensor as its loss -- refuse instead."""


def _no_labels(kwargs):
    if kwargs.get("labels") is not None or kwargs.get("start_positions") is not None or kwargs.get("end_positions") is not None:
        raise NotImplementedError(
            "outlier_suppression_amd task models return logits only; compute the loss outside, or use the reference's own "
            "model files on this package through the sys.modules shim (INTEGRATION.md section 1)")


def require_academic(backend):
    """'academic' and 'tensorrt' (the reference's extra residual-branch sites, quant_bert.py:204-216, quant_bart.py:305-307,
    401-404) are the placements the reference's model files know; anything else is refused rather than silently treated
    as 'academic'."""
    if backend not in ("academic", "tensorrt"):
        raise NotImplementedError(f"backend={backend!r}: the quantizer placements are 'academic' and 'tensorrt'")
