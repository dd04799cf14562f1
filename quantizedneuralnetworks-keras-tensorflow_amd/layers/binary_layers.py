"""BinaryDense / BinaryConv2D with the reference's constructor surface
(layers/binary_layers.py:31-199); forward through the gfx950 kernels."""
from .. import _abi
from ._base import LowBitConv2D, LowBitDense
from .binary_ops import binarize  # noqa: F401  (re-exported like the reference module)


class Clip:
    """binary_layers.py:13-28.  Kept for config compatibility; training-only."""

    def __init__(self, min_value, max_value=None):
        self.min_value = min_value
        self.max_value = max_value
        if not self.max_value:
            self.max_value = -self.min_value
        if self.min_value > self.max_value:
            self.min_value, self.max_value = self.max_value, self.min_value

    def __call__(self, p):
        return p.clamp(self.min_value, self.max_value)

    def get_config(self):
        return {"name": "__call__", "min_value": self.min_value, "max_value": self.max_value}


class BinaryDense(LowBitDense):
    """binary_layers.py:31-92.  call: x . binarize(W) + b  (78-85)."""

    _wkind = _abi.W_BINARY

    def __init__(self, units, H=1., kernel_lr_multiplier='Glorot', bias_lr_multiplier=None, **kwargs):
        self._dense_init(units, H, kernel_lr_multiplier, bias_lr_multiplier, kwargs, "binary_dense")


class BinaryConv2D(LowBitConv2D):
    """binary_layers.py:95-194.  call: conv2d(x, binarize(W)) + b  (160-187); the
    lr-multiplier trick around the conv (163-165,175-176) is the identity in the
    forward pass and is evaluated as such ("exact" mode)."""

    _wkind = _abi.W_BINARY

    def __init__(self, filters, kernel_regularizer=None, activity_regularizer=None,
                 kernel_lr_multiplier='Glorot', bias_lr_multiplier=None, H=1., **kwargs):
        self._conv_init(filters, kernel_regularizer, activity_regularizer, H, kernel_lr_multiplier,
                        bias_lr_multiplier, kwargs, "binary_conv2d")


# Aliases (binary_layers.py:199)
BinaryConvolution2D = BinaryConv2D
