"""Counterparts of the reference's layers/quantized_ops.py (forward only)."""
import torch

from .. import _abi


def quantized_tanh(W, nb=16):
    """quantized_ops.py:87-100: clip(round(W*m), -m, m-1)/m, m = 2**(nb-1).
    This is the activation the models use (model_factory.py:9,19-20)."""
    W = _abi.require_cuda(W, "quantized_tanh")
    y = torch.empty_like(W)
    _abi.check(_abi.load().qnn_quantized_tanh_f32(_abi.ptr(W), _abi.ptr(y), W.numel(), int(nb),
                                                  _abi.stream_ptr()), "quantized_tanh")
    return y


def quantize(W, nb=16, clip_through=False):
    """quantized_ops.py:49-66.  `clip_through` only changes the gradient."""
    return quantized_tanh(W, nb)
