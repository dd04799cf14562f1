"""TernaryDense / TernaryConv2D with the reference's constructor surface
(layers/ternary_layers.py:30-186); forward through the gfx950 kernels.  Weights are
ternarised ONCE at prepack time (the reference re-runs ternarize every Session.run)."""
from .. import _abi
from ._base import LowBitConv2D, LowBitDense
from .binary_layers import Clip  # noqa: F401  (same class in the reference, ternary_layers.py:13-27)
from .ternary_ops import ternarize  # noqa: F401


class TernaryDense(LowBitDense):
    """ternary_layers.py:30-91.  call: x . ternarize(W) + b  (77-84)."""

    _wkind = _abi.W_TERNARY

    def __init__(self, units, H=1., kernel_lr_multiplier='Glorot', bias_lr_multiplier=None, **kwargs):
        self._dense_init(units, H, kernel_lr_multiplier, bias_lr_multiplier, kwargs, "ternary_dense")


class TernaryConv2D(LowBitConv2D):
    """ternary_layers.py:94-182.  call: conv2d(x, ternarize(W)) + b  (156-174)."""

    _wkind = _abi.W_TERNARY

    def __init__(self, filters, kernel_regularizer=None, activity_regularizer=None,
                 kernel_lr_multiplier='Glorot', bias_lr_multiplier=None, H=1., **kwargs):
        self._conv_init(filters, kernel_regularizer, activity_regularizer, H, kernel_lr_multiplier,
                        bias_lr_multiplier, kwargs, "ternary_conv2d")


# Aliases (ternary_layers.py:186)
TernaryConvolution2D = TernaryConv2D
