"""Counterparts of the reference's layers/ternary_ops.py (forward only).

ternary_tanh thresholds at 0.7*mean|x| over the WHOLE tensor
(ternary_ops.py:23): the result depends on batch composition, so a batch that is
sharded over GPUs needs an all-reduce of (sum|x|, count) first; see shard.py.
"""
import torch

from .. import _abi


def ternary_tanh(x):
    """ternary_ops.py:52-54: ternarize(clip(x,-1,1))."""
    x = _abi.require_cuda(x, "ternary_tanh")
    y = torch.empty_like(x)
    ws = torch.empty(2, dtype=torch.float64, device=x.device)
    _abi.check(_abi.load().qnn_ternary_tanh_f32(_abi.ptr(x), _abi.ptr(y), x.numel(), _abi.ptr(ws),
                                                _abi.stream_ptr()), "ternary_tanh")
    return y


def ternarize(W, H=1.0):
    """ternary_ops.py:33-41 for weights in [-H, H] (the Clip constraint keeps them
    there, ternary_layers.py): same thresholding without the clip."""
    W = _abi.require_cuda(W, "ternarize")
    if H != 1 and H != 1.0:
        return ternarize(W / float(H)) * float(H)
    if float(W.abs().max()) > 1.0:
        raise _abi.QnnError("ternarize: weights outside [-H, H] are not supported")
    return ternary_tanh(W)
