"""MI355X-native low-bit forward engine (gfx950).

Drop-in for the forward path of the reference's BinaryConv2D / BinaryDense
(layers/binary_layers.py) and QuantizedConv2D / QuantizedDense
(layers/quantized_layers.py) plus the activation clips of layers/*_ops.py.
Python keeps the Keras-compatible surface; all arithmetic runs in hand-written
HIP kernels behind the C ABI declared in include/qnn_abi.h (csrc/libqnn_hip.so).
There is no CPU fallback: without the built library and a GPU every op raises.

The directory name is not a Python identifier; import it with
``importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")`` or
through the ``qnn_amd`` alias module at the repository root.
"""
from . import _abi  # noqa: F401
from . import nets, engine, shard  # noqa: F401
from .layers.binary_ops import binary_tanh, binarize, binary_sigmoid  # noqa: F401
from .layers.quantized_ops import quantize, quantized_tanh  # noqa: F401
from .layers.ternary_ops import ternary_tanh, ternarize  # noqa: F401
from .layers.binary_layers import BinaryConv2D, BinaryDense, BinaryConvolution2D  # noqa: F401
from .layers.quantized_layers import (QuantizedConv2D, QuantizedDense,  # noqa: F401
                                      QuantizedConvolution2D)
from .layers.ternary_layers import TernaryConv2D, TernaryDense, TernaryConvolution2D  # noqa: F401

__all__ = [
    "binary_tanh", "binarize", "binary_sigmoid", "quantize", "quantized_tanh",
    "ternary_tanh", "ternarize", "BinaryConv2D", "BinaryDense", "BinaryConvolution2D",
    "QuantizedConv2D", "QuantizedDense", "QuantizedConvolution2D",
    "TernaryConv2D", "TernaryDense", "TernaryConvolution2D",
]
