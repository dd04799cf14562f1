"""Data-parallel inference over the GPUs of one node (one process per GPU).

The forward pass is independent per image (BN uses moving statistics, the clips
are elementwise, weights are replicated), so the batch is cut into contiguous
equal shards with NO data-path collective; the only exchange is one all-gather
of the (B/G, classes) float32 logits (RCCL over xGMI on the GPU box, gloo in the
CPU tests).  40 bytes per image: latency-bound, no bucketing needed.

Not shardable this way: ternary_tanh (global mean|x| over the batch tensor,
ternary_ops.py:23) -- it needs all-reduced (sum, count) first; see
`allreduce_mean_abs`.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous equal shards; the last ones are padded by the caller if total %
    world != 0 (ranks must contribute equal-sized tensors to the all-gather)."""
    per = -(-total // world)
    lo = min(rank * per, total)
    hi = min(lo + per, total)
    return lo, hi, per


def shard_batch(x, rank, world):
    """Return this rank's slice of the global batch, zero-padded to the common size."""
    lo, hi, per = shard_bounds(x.shape[0], rank, world)
    part = x[lo:hi]
    if part.shape[0] < per:
        pad = torch.zeros((per - part.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        part = torch.cat([part, pad], dim=0)
    return part.contiguous()


def gather_logits(local, total=None, group=None):
    """All-gather the per-rank logits into (world*per, classes) and drop the padding."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local if total is None else local[:total]
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out if total is None else out[:total]


_ACTIVE = []          # stack of (process group, valid rows) inside `with sharded(group):`


class sharded:
    """Context in which batch tensors are SHARDS of a global batch: ops that reduce over the whole batch
    tensor (ternary_tanh, ternary_ops.py:23) all-reduce their partial sums over `group`.

    valid_rows: number of REAL images at the front of this rank's shard when it was zero-padded to the
    common size (shard_batch).  The padded images still produce non-zero pre-activations (bias, BN shift),
    so batch-wide statistics must not see them: reductions take the first `valid_rows` batch rows only."""

    def __init__(self, group=None, valid_rows=None):
        self.group = group
        self.valid_rows = valid_rows

    def __enter__(self):
        g = self.group if self.group is not None else (dist.group.WORLD if dist.is_initialized() else None)
        _ACTIVE.append((g, self.valid_rows))
        return self

    def __exit__(self, *exc):
        _ACTIVE.pop()
        return False


def active_valid_rows():
    """Valid batch rows of the innermost `sharded` context (None: every row is a real image)."""
    return _ACTIVE[-1][1] if _ACTIVE else None


def allreduce_sum_count(ws, group=None):
    """Sum the {sum|clip(x)|, count} pair of ternary_tanh over the shards (in place).  No-op outside a sharded
    context / for a single process."""
    if group is None and _ACTIVE:
        group = _ACTIVE[-1][0]
    elif group is None:
        return ws
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(ws, group=group)
    return ws


def allreduce_mean_abs(x, group=None):
    """Global mean(|clip(x,-1,1)|) of a batch-sharded tensor: the statistic ternary_tanh thresholds at,
    computed with torch ops (used by the CPU tests; the GPU path is ternary_ops.ternary_tanh)."""
    s = torch.stack([x.clamp(-1, 1).abs().double().sum(),
                     torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, group=group)
    return (s[0] / s[1]).float()


def sharded_forward(model, x_global, rank, world, group=None):
    """forward(shard) on every rank + all-gather of the logits.  Runs inside `sharded(group)`, so a
    full-tnn network thresholds its ternary activations at the global batch mean."""
    total = x_global.shape[0]
    lo, hi, per = shard_bounds(total, rank, world)
    with sharded(group, valid_rows=hi - lo):      # a ragged last shard is zero-padded: keep the padding out of the statistics
        local = model(shard_batch(x_global, rank, world))
    return gather_logits(local, total, group)
