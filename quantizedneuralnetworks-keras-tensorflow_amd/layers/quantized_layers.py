"""QuantizedDense / QuantizedConv2D with the reference's constructor surface
(layers/quantized_layers.py:32-206); forward through the gfx950 kernels."""
from .. import _abi
from ._base import LowBitConv2D, LowBitDense
from .binary_layers import Clip  # noqa: F401  (same class in the reference, quantized_layers.py:13-29)
from .quantized_ops import quantize  # noqa: F401


class QuantizedDense(LowBitDense):
    """quantized_layers.py:32-96.  call: x . quantize(W, nb) + b  (79-88)."""

    _wkind = _abi.W_QUANT

    def __init__(self, units, H=1., nb=16, kernel_lr_multiplier='Glorot', bias_lr_multiplier=None,
                 **kwargs):
        self.nb = int(nb)
        self._dense_init(units, H, kernel_lr_multiplier, bias_lr_multiplier, kwargs, "quantized_dense")


class QuantizedConv2D(LowBitConv2D):
    """quantized_layers.py:99-201.  call: conv2d(x, quantize(W, nb)) + b  (164-194)."""

    _wkind = _abi.W_QUANT

    def __init__(self, filters, kernel_regularizer=None, activity_regularizer=None,
                 kernel_lr_multiplier='Glorot', bias_lr_multiplier=None, H=1., nb=16, **kwargs):
        self.nb = int(nb)
        self._conv_init(filters, kernel_regularizer, activity_regularizer, H, kernel_lr_multiplier,
                        bias_lr_multiplier, kwargs, "quantized_conv2d")


# Aliases (quantized_layers.py:206)
QuantizedConvolution2D = QuantizedConv2D
