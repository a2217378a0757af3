"""MI355X-native fake-quant / observer hot path of Outlier Suppression.

``outlier_suppression_amd.quantization`` mirrors the reference's
``quant_transformer.quantization`` import surface (same class and function names,
constructor arguments, buffers and state-dict keys); underneath, every reduction and
every scale-clip-round-dequant pass is a hand-written gfx950 HIP kernel reached
through the C ABI in ``include/osq_hip.h``.  There is no CPU path.
"""
__version__ = "0.1.0"
