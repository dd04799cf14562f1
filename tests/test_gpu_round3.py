"""Round-3 regression tests on the GPU: hipGraph capture of a ternary network (the {sum, count} seed of
ternary_tanh must be a memset node, not a copy from a stack buffer), padded shards kept out of the ternary statistics,
and the pipelined product path (engine.Pipelined / nets.Model.predict) against the eager one."""
import numpy as np
import pytest
import torch

import qnn_amd
from qnn_amd import _abi, engine, nets, shard
from oracle import qnn_oracle as O
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu
F32 = np.float32


def _tnn():
    cf = nets.Config(network_type="full-tnn", architecture="VGG", nla=1, nlb=1, nlc=1, nfa=32, nfb=32, nfc=32)
    return cf, nets.build_spec(cf, 11)


def test_full_tnn_forward_replays_from_a_hipgraph():
    """ternary_tanh = memset + reduction kernel + threshold kernel: all three must be capturable, and a replay on NEW
    input data must give the new data's result (a dangling host pointer in the graph would replay the old seed)."""
    cf, spec = _tnn()
    m = engine.ResidualFusedModel(spec)
    xs = [nets.synthetic_images(cf, 16, s) for s in (1, 2, 3)]
    static = dev(xs[0]).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            m(static)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(static)
    for x in xs[::-1]:
        static.copy_(dev(x))
        g.replay()
        got = host(y)
        np.testing.assert_array_equal(got, host(m(dev(x))))            # eager
        np.testing.assert_array_equal(got, O.run_spec(spec, x, float_conv="device"))


def test_padded_shard_rows_stay_out_of_the_ternary_statistics():
    """shard.sharded(valid_rows=k): the reduction of ternary_tanh sees the first k batch rows only (ADVICE r2,
    shard.py:93).  One process: a 5-image batch zero-padded to 8 must give, on its first 5 rows, what the 5 images
    give alone; without the hint the padded images' pre-activations move the threshold."""
    rng = np.random.default_rng(5)
    x5 = rng.standard_normal((5, 6, 6, 8)).astype(F32)
    x8 = np.concatenate([x5, np.full((3, 6, 6, 8), 0.9, F32)])        # padded rows with non-zero "pre-activations"
    alone = host(qnn_amd.ternary_tanh(dev(x5)))
    with shard.sharded(None, valid_rows=5):
        padded = host(qnn_amd.ternary_tanh(dev(x8)))
    np.testing.assert_array_equal(padded[:5], alone)
    np.testing.assert_array_equal(alone, O.ternary_tanh(x5))
    unhinted = host(qnn_amd.ternary_tanh(dev(x8)))
    assert not np.array_equal(unhinted[:5], alone)
    # whole network through sharded_forward with world = 1 and a ragged total: identical to the plain forward
    cf, spec = _tnn()
    m = engine.ResidualFusedModel(spec)
    x = nets.synthetic_images(cf, 7, 4)
    np.testing.assert_array_equal(host(shard.sharded_forward(m, dev(x), 0, 1)), host(m(dev(x))))
