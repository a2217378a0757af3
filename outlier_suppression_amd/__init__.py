"""MI355X-native fake-quant / observer hot path of Outlier Suppression.

``outlier_suppression_amd.quantization`` mirrors the reference's
``quant_transformer.quantization`` import surface (same class and function names,
constructor arguments, buffers and state-dict keys); underneath, every reduction and
every scale-clip-round-dequant pass is a hand-written gfx950 HIP kernel reached
through the C ABI in ``include/osq_hip.h``.  There is no CPU path.
"""
__version__ = "0.1.0"


def set_strict(on=True, simd_width=8):
    """Sums in the REFERENCE's own order -- the DEFAULT of this package (round 4); ``set_strict(False)`` opts out.

    Two numbers of the path are sums over a whole activation: the loss of a per-tensor MSEFast search
    (`.pow(2).mean()`, quantization/observer.py:420-432) and the LSQ / LSQ+ parameter gradients (autograd's `sum_to_size`,
    quantization/util_quant.py:29-67).  The reference adds them with torch.sum on the CPU, whose order (ATen's
    cascade_sum) depends on the host's SIMD width and -- beyond 32768 elements -- on its thread count.

    By default (strict ON) both sums follow torch's order on a ONE-thread host with ``simd_width`` fp32 lanes per vector
    (8: x86 torch, AVX2 and AVX-512 builds alike; float64 sums use half as many), at any length: min_val / max_val /
    scale / zero_point of every MSEFast observer and scale.grad / zero_point.grad of every learnable quantizer --
    per-tensor (LSQ / LSQ+ activations: what every shipped configuration learns; any size) and per-channel weights
    (ch_axis = 0, rows of up to 3072 columns) -- equal that reference run bit for bit (tests/test_gpu_strict_order.py,
    fixtures made by running the reference at BERT-base site sizes; tests/test_gpu_parity.py for the per-channel case).
    Other per-channel layouts (an inner channel axis, longer rows) keep their float64 sums (2e-5 from autograd's fp32
    ones).  Per-channel (row) MSEFast searches follow the reference's order in either mode.

    ``set_strict(False)`` returns the correctly rounded sum instead (float64 / exact accumulation, rounded once:
    order-free, same bits whatever the reference host would have been), which can differ from ONE particular reference
    run in the last bits of the sum -- and, for MSEFast, in which of two tied candidates of its staircase loss the search
    keeps.  What it buys: a lone per-tensor search runs as one persistent launch (5.7-8.5 us per
    loss evaluation instead of 16-28), the LSQ+ backward runs 1.5x faster.  The searches of an observer PASS (deferred,
    quantization/deferred.py) are faster strict: one launch per round of evaluations of all sites (BASELINE configs[3]:
    1.09 s against 2.28 s).  From the environment: OSQ_STRICT=0 (and OSQ_STRICT_SIMD=16 for a 16-lane reference host)."""
    from . import ops
    if simd_width not in (8, 16):
        raise ValueError("simd_width must be 8 or 16")
    ops.set_tuning("mse_sum_order", simd_width if on else 0)
    ops.set_tuning("bwd_sum_order", simd_width if on else 0)


def set_fast(on=True):
    """Opt into the fusions that are NOT bit-comparable with the eager sequence (today: the one-launch LayerNorm site,
    util_layernorm.FUSE_LAYERNORM -- see there for the bound)."""
    from . import util_layernorm
    util_layernorm.FUSE_LAYERNORM = bool(on)


def _apply_environment():
    """Applied when the library is first loaded: the reference's summation order unless OSQ_STRICT=0; OSQ_FAST=1 opts
    into the fusions of set_fast."""
    import os
    set_strict(os.environ.get("OSQ_STRICT", "1") not in ("", "0"), int(os.environ.get("OSQ_STRICT_SIMD", "8")))
    if os.environ.get("OSQ_FAST", "0") not in ("", "0"):
        set_fast(True)
