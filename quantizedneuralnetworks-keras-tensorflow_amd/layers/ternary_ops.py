"""Counterparts of the reference's layers/ternary_ops.py (forward only).

ternary_tanh thresholds at 0.7*mean|x| over the WHOLE tensor
(ternary_ops.py:23): the result depends on batch composition, so a batch that is
sharded over GPUs needs an all-reduce of (sum|x|, count) first; see shard.py.
"""
import torch

from .. import _abi


def ternary_tanh(x, group=None):
    """ternary_ops.py:52-54: ternarize(clip(x,-1,1)).

    The cutoff is 0.7 * mean|clip(x)| over the WHOLE batch tensor (ternary_ops.py:23).  When the batch is
    sharded over processes (`shard.sharded(...)` is active, or `group` is given) the two partial sums
    {sum|clip(x)|, count} are all-reduced between the reduction kernel and the threshold kernel, so every
    shard thresholds at the global mean: the result equals the single-process one.  A shard that was padded
    to the common size (`shard.sharded(group, valid_rows=k)`) contributes only its first k batch rows to the
    two sums; the threshold is applied to every row."""
    from .. import shard
    x = _abi.require_cuda(x, "ternary_tanh")
    y = torch.empty_like(x)
    ws = torch.empty(2, dtype=torch.float64, device=x.device)
    lib = _abi.load()
    n_sum = x.numel()
    valid = shard.active_valid_rows()
    if valid is not None and x.dim() >= 1 and x.shape[0] > 0:
        n_sum = min(int(valid), x.shape[0]) * (x.numel() // x.shape[0])
    _abi.check(lib.qnn_ternary_abs_sum_f32(_abi.ptr(x), n_sum, _abi.ptr(ws), _abi.stream_ptr()),
               "ternary_tanh")
    shard.allreduce_sum_count(ws, group)
    _abi.check(lib.qnn_ternary_apply_f32(_abi.ptr(x), _abi.ptr(y), x.numel(), _abi.ptr(ws), _abi.stream_ptr()),
               "ternary_tanh")
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
