"""Same import surface as the reference's ``quant_transformer.quantization`` (its __init__.py:1-4)."""
from .quantized_module import Quantizer, QuantizedModule  # noqa: F401
from .state import (enable_calibration_quantization, enable_calibration_woquantization,  # noqa: F401
                    enable_quantization, disable_all, set_observer_name)
