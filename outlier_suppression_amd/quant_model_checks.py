"""What this package's own task-model wrappers do NOT do, said loudly.

The reference's task models return HF ``ModelOutput``s with a loss when ``labels`` are given and build extra
``output_post_act_fake_quantize`` sites for ``backend='tensorrt'`` (model/quant_bert.py).  The wrappers here exist for
calibration and quantised inference of the hot path: they return plain tuples of logits.  A training / evaluation loop
that passes ``labels`` would otherwise read the first logits tensor as its loss -- refuse instead."""


def _no_labels(kwargs):
    if kwargs.get("labels") is not None or kwargs.get("start_positions") is not None or kwargs.get("end_positions") is not None:
        raise NotImplementedError(
            "outlier_suppression_amd task models return logits only; compute the loss outside, or use the reference's own "
            "model files on this package through the sys.modules shim (INTEGRATION.md section 1)")


def require_academic(backend):
    if backend != "academic":
        raise NotImplementedError(
            f"backend={backend!r}: only the 'academic' quantizer placement is built by this package's model classes; the "
            "reference's model files (which add output_post_act_fake_quantize sites for 'tensorrt') run on this package "
            "through the sys.modules shim (INTEGRATION.md section 1)")
