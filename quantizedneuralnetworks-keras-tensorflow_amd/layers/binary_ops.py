"""Counterparts of the reference's layers/binary_ops.py on torch-ROCm tensors.

Forward semantics only (the straight-through-estimator terms are identities in
the forward pass).  Every function dispatches to a HIP kernel through the C ABI.
"""
import torch

from .. import _abi


def _unary(x, call, what):
    x = _abi.require_cuda(x, what)
    y = torch.empty_like(x)
    _abi.check(call(x, y), what)
    return y


def binary_tanh(x):
    """binary_ops.py:37-51: 2*round(clip(0.5x+0.5,0,1))-1 -> {-1,+1}; +1 iff x > 2**-24."""
    lib = _abi.load()
    return _unary(x, lambda a, b: lib.qnn_binary_tanh_f32(_abi.ptr(a), _abi.ptr(b), a.numel(),
                                                          _abi.stream_ptr()), "binary_tanh")


def binary_sigmoid(x):
    """binary_ops.py:27-34: round(clip(0.5x+0.5,0,1)) = (binary_tanh(x)+1)/2 (exact)."""
    return (binary_tanh(x) + 1.0) * 0.5


def binarize(W, H=1.0):
    """binary_ops.py:54-64: H * binary_tanh(W / H)."""
    if H == 1 or H == 1.0:
        return binary_tanh(W)
    H = float(H)
    return binary_tanh(W / H) * H
