"""MI355X-native fake-quant / observer hot path of Outlier Suppression.

``outlier_suppression_amd.quantization`` mirrors the reference's
``quant_transformer.quantization`` import surface (same class and function names,
constructor arguments, buffers and state-dict keys); underneath, every reduction and
every scale-clip-round-dequant pass is a hand-written gfx950 HIP kernel reached
through the C ABI in ``include/osq_hip.h``.  There is no CPU path.
"""
__version__ = "0.1.0"


def set_strict(on=True, simd_width=8, backward=None, _lib=None):
    """Which rounding of the path's two whole-tensor sums is returned: the REFERENCE's own order, or the correctly rounded sum.

    Two numbers of the path are sums over a whole activation: the loss of a per-tensor MSEFast search
    (`.pow(2).mean()`, quantization/observer.py:420-432) and the LSQ / LSQ+ parameter gradients (autograd's `sum_to_size`,
    quantization/util_quant.py:29-67).  The reference adds them with torch.sum on the CPU, whose order (ATen's
    cascade_sum) depends on the host's SIMD width and -- beyond 32768 elements -- on its thread count.  The order this
    package can pin is torch's on a ONE-thread host with ``simd_width`` fp32 lanes per vector (8: x86 torch; float64 sums
    use half as many), at any length (csrc/aten_order.h).  Note what that target is: the upstream reference hard-codes
    `.cuda()` on this path (observer.py:81,95,425), so no upstream run ever had this order -- it is the order of the
    reference's code run on a CPU with torch.set_num_threads(1), which is what the golden fixtures are
    (tests/golden/make_golden*.py); a multi-threaded CPU run differs from it beyond 32768 elements.

    ``on`` -- the MSEFast losses.  ON by default: min_val / max_val / scale / zero_point of every MSEFast observer equal
    that reference run bit for bit, and it is also the FASTER form of an observer pass (the searches of a forward run as
    rounds, quantization/deferred.py: BASELINE configs[3] 0.47-0.50 s against 2.2 s since round 6).  A lone search pays one launch per
    evaluation (15-24 us against 6-9 for the resident order-free form).  Per-channel (row) searches follow the
    reference's order in either mode.

    ``backward`` -- scale.grad / zero_point.grad of the learnable quantizers (default: follows ``on`` when set_strict is
    called; OFF in the package's default tier since round 5).  ON: bit-equal to the reference's one-thread autograd run
    (tests/test_gpu_strict_order.py; per-channel weights with ch_axis = 0 and rows of up to 3072 columns included).  OFF:
    float64 accumulation rounded once -- within 2e-5 of autograd's fp32 sums, 1.3x faster (53 against 68 us on
    [256,128,768]); BASELINE.json asks 1e-5 on the dequantised tensor and names no bar for gradients.

    Default tier when the library loads: ``set_strict(True, backward=False)``.  OSQ_STRICT=1: everything in the
    reference's order; OSQ_STRICT=0: everything order-free; OSQ_STRICT_SIMD=16 for a 16-lane reference host.
    ``reset_tier()`` returns to what the environment says."""
    from . import ops
    if simd_width not in (8, 16):
        raise ValueError("simd_width must be 8 or 16")
    backward = on if backward is None else backward
    ops.set_tuning("mse_sum_order", simd_width if on else 0, _lib)
    ops.set_tuning("bwd_sum_order", simd_width if backward else 0, _lib)


def set_fast(on=True):
    """The one-launch LayerNorm site (util_layernorm.FUSE_LAYERNORM; default ON since round 5 -- see there for how it
    compares with torch-ROCm's LayerNorm against the reference's CPU run).  ``set_fast(False)`` keeps the eager sequence."""
    from . import util_layernorm
    util_layernorm.FUSE_LAYERNORM = bool(on)


def reset_tier(_lib=None):
    """The package's default tier, as the environment states it (applied when the library is first loaded): MSEFast sums in
    the reference's one-thread order, the backward's sums order-free; OSQ_STRICT=1 / 0 force both; OSQ_FAST=0 / 1 the
    one-launch LayerNorm site."""
    import os
    width = int(os.environ.get("OSQ_STRICT_SIMD", "8"))
    strict = os.environ.get("OSQ_STRICT", "")
    if strict == "":
        set_strict(True, width, backward=False, _lib=_lib)
    else:
        set_strict(strict != "0", width, _lib=_lib)
    # unset: the default (one-launch LayerNorm site ON) -- a set_fast(False) of an earlier caller or test does not leak
    set_fast(os.environ.get("OSQ_FAST", "") != "0")


_apply_environment = reset_tier
