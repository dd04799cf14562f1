"""N>1 path on CPU: two gloo ranks shard a batch, run a (CPU stand-in) per-image
forward and all-gather the logits; the result must equal the single-rank result.
The sharding / gather code is the same one bench.py uses with RCCL on the GPU box."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from qnn_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(x):
    # any per-image function stands in for the GPU forward (which has no CPU path)
    w = torch.arange(x[0].numel() * 10, dtype=torch.float32).reshape(-1, 10) / 1000.0
    return torch.tanh(x.reshape(x.shape[0], -1) @ w)


def _tnn_forward(x):
    """CPU stand-in for a ternary network layer with the structure of the GPU op (layers/ternary_ops.ternary_tanh):
    non-zero pre-activations for zero images (bias), partial {sum|clip|, count} over this shard's VALID rows only,
    all-reduce through the active `shard.sharded` context, threshold applied to every row."""
    pre = x.reshape(x.shape[0], -1) * 3.0 - 1.2 + 0.4            # "conv + bias": padded (zero) images give -0.8
    valid = shard.active_valid_rows()
    rows = pre if valid is None else pre[:valid]
    ws = torch.stack([rows.clamp(-1, 1).abs().double().sum(), torch.tensor(float(rows.numel()), dtype=torch.float64)])
    shard.allreduce_sum_count(ws)
    cut = 0.7 * (ws[0] / ws[1]).float()
    w = pre.clamp(-1, 1)
    t = torch.where(w > cut, torch.ones_like(w), torch.where(w <= -cut, -torch.ones_like(w), torch.zeros_like(w)))
    return t[:, :10].contiguous()


def _worker_tnn(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.rand((total, 4, 4, 3), generator=torch.Generator().manual_seed(1))
        y = shard.sharded_forward(_tnn_forward, x, rank, world)
        if rank == 0:
            torch.save({"y": y}, out)
    finally:
        dist.destroy_process_group()


def test_ragged_shards_keep_padding_out_of_the_ternary_statistics(tmp_path):
    """total = 13 on two ranks: rank 1 holds 6 images + 1 zero image.  The zero image's pre-activations (-0.8) must not
    enter the all-reduced mean: the gathered result equals the single-process one (shard.py:93)."""
    for total in (13, 16):
        x = torch.rand((total, 4, 4, 3), generator=torch.Generator().manual_seed(1))
        want = _tnn_forward(x)                                   # no sharded context: one process, every row valid
        out = str(tmp_path / ("tnn_%d.pt" % total))
        mp.spawn(_worker_tnn, args=(2, _free_port(), total, out), nprocs=2, join=True)
        got = torch.load(out, weights_only=True)["y"]
        assert torch.equal(got, want), total
    # the statistic really is sensitive to the padding: with every row of the padded shard counted it moves
    x = torch.rand((13, 4, 4, 3), generator=torch.Generator().manual_seed(1))
    padded = torch.cat([x, torch.zeros((1, 4, 4, 3))])
    assert not torch.equal(_tnn_forward(padded)[:13], _tnn_forward(x))


def _worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        x = torch.rand((total, 4, 4, 3), generator=g)
        y = shard.sharded_forward(_fake_forward, x, rank, world)
        lo, hi, _ = shard.shard_bounds(total, rank, world)
        xx = (x * 4 - 2)[lo:hi]                                # unpadded rows of this rank
        m = shard.allreduce_mean_abs(xx)
        # the pair ternary_ops.ternary_tanh all-reduces between its two kernels when the batch is sharded
        ws = torch.stack([xx.clamp(-1, 1).abs().double().sum(), torch.tensor(float(xx.numel()), dtype=torch.float64)])
        untouched = shard.allreduce_sum_count(ws.clone())      # outside a sharded context: no exchange
        with shard.sharded():
            shard.allreduce_sum_count(ws)
        if rank == 0:
            torch.save({"y": y, "m": m, "m2": (ws[0] / ws[1]).float(), "count": ws[1],
                        "local_count": untouched[1]}, out)
    finally:
        dist.destroy_process_group()


def _run(total, world, tmp_path):
    out = str(tmp_path / ("out_%d_%d.pt" % (total, world)))
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    return torch.load(out, weights_only=True)


def test_shard_bounds():
    assert shard.shard_bounds(4096, 3, 8) == (1536, 2048, 512)
    assert shard.shard_bounds(10, 3, 4) == (9, 10, 3)
    assert shard.shard_bounds(10, 0, 1) == (0, 10, 10)
    x = torch.arange(10.0).reshape(10, 1)
    parts = [shard.shard_batch(x, r, 4) for r in range(4)]
    assert all(p.shape == (3, 1) for p in parts)
    assert torch.equal(torch.cat(parts)[:10], x) and float(parts[3][1:].abs().sum()) == 0


def test_two_ranks_equal_single_rank(tmp_path):
    g = torch.Generator().manual_seed(0)
    for total in (16, 13):                      # even split and ragged (padded) split
        x = torch.rand((total, 4, 4, 3), generator=g.manual_seed(0))
        want = _fake_forward(x)
        got = _run(total, 2, tmp_path)
        assert got["y"].shape == want.shape
        assert torch.equal(got["y"], want)
        ref = (x * 4 - 2).clamp(-1, 1).abs().double().mean().float()
        assert abs(float(got["m"]) - float(ref)) < 1e-6
        assert abs(float(got["m2"]) - float(ref)) < 1e-6
        assert float(got["count"]) == total * 48 and float(got["local_count"]) < total * 48
