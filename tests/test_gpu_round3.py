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
    m = engine.ResidualFusedModel(spec, first_layer="exact")
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
    m = engine.ResidualFusedModel(spec, first_layer="exact")
    x = nets.synthetic_images(cf, 7, 4)
    np.testing.assert_array_equal(host(shard.sharded_forward(m, dev(x), 0, 1)), host(m(dev(x))))


@pytest.mark.parametrize("first", ["exact", "image", "fixed", "u8"])
def test_pipelined_predict_equals_the_eager_path(first):
    """engine.Pipelined (hipGraph per lane, round-robin streams) is what nets.Model.predict runs on: same bits as the
    eager engine, for full batches, a ragged tail, repeated calls and every input form."""
    cf = nets.baseline_config(2)
    spec = nets.build_spec(cf, nets.SEED_BASE + 2)
    model = nets.Model(cf, spec, first_layer="exact" if first == "u8" else first, lanes=2)
    xu8 = nets.synthetic_images_u8(cf, 5 * 64 + 17, 21)
    x = xu8 if first == "u8" else (xu8.astype(F32) / F32(255)).astype(F32)
    eager = host(model.engine(dev(x)))
    if first == "u8":
        np.testing.assert_array_equal(eager, O.run_spec_u8(spec, xu8))
    elif first == "exact":
        np.testing.assert_array_equal(eager[:64], O.run_spec(spec, x[:64], float_conv="device"))
    elif first == "image":
        np.testing.assert_array_equal(eager, O.run_spec_u8(spec, xu8))       # recognised as bytes: the uint8 entry's result
    for rep in range(2):
        got = model.predict(x, batch_size=64)                                # numpy in, numpy out
        assert isinstance(got, np.ndarray)
        np.testing.assert_array_equal(got, eager)
    xd = dev(x)
    got = model.predict(xd, batch_size=64)                                   # resident in, resident out
    assert isinstance(got, torch.Tensor) and got.is_cuda
    np.testing.assert_array_equal(host(got), eager)
    np.testing.assert_array_equal(host(model.predict(xd[:40], batch_size=64)), eager[:40])      # smaller than a batch
    pipe = model.pipeline(64)
    assert len(pipe.lanes_for(xd[:64])) == 2
    # three lanes, one lane
    for lanes in (1, 3):
        np.testing.assert_array_equal(host(engine.Pipelined(model.engine, lanes=lanes, batch_size=128)(xd)), eager)


def test_pipelined_first_layer_domain_is_checked_by_predict():
    cf = nets.baseline_config(2)
    spec = nets.build_spec(cf, nets.SEED_BASE + 2)
    x = nets.synthetic_images(cf, 128, 3)
    for first in ("image", "fixed"):
        model = nets.Model(cf, spec, first_layer=first)
        model.predict(x, batch_size=64)
        bad = x.copy()
        bad[70, 3, 3, 0] = 0.5001 if first == "image" else 1.25          # off the byte grid / outside [0, 1]
        with pytest.raises(_abi.QnnError, match="outside its domain"):
            model.predict(bad, batch_size=64)
        model.predict(x, batch_size=64)


def test_pipelined_residual_and_ternary_networks():
    cf = nets.Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET", nres=1, dim=32)
    spec = nets.build_spec(cf, 3)
    model = nets.Model(cf, spec, first_layer="exact")
    assert type(model.engine).__name__ == "ResidualFusedModel"
    x = nets.synthetic_images(cf, 40, 9)
    eager = np.concatenate([host(model.engine(dev(x[i:i + 16]))) for i in range(0, 40, 16)])
    np.testing.assert_array_equal(model.predict(x, batch_size=16), eager)
    cf, spec = _tnn()
    model = nets.Model(cf, spec, first_layer="exact")
    x = nets.synthetic_images(cf, 48, 2)
    want = np.concatenate([O.run_spec(spec, x[i:i + 16], float_conv="device") for i in range(0, 48, 16)])
    np.testing.assert_array_equal(model.predict(x, batch_size=16), want)   # batch statistics per 16-image batch


def test_bench_default_line_carries_the_contract_and_the_round3_blocks():
    """bench.py end to end at a small batch (a subprocess, as the driver runs it): one JSON line with the contract's
    keys, `roofline`, `cpu_baseline`, the first-layer alternatives, the product-call figure and the targets block."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "512",
                        "--repeats", "2"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["steps"] == 4 and out["n_gpus"] == 1 and out["config"]["first_layer"] == "auto"
    assert out["roofline"]["frac"] > 0 and out["roofline"]["bound"] in ("hbm", "mfma")
    assert set(out["first_layer_alternatives"]) == {"exact", "fixed", "u8"}
    assert all("value" in v for v in out["first_layer_alternatives"].values()), out["first_layer_alternatives"]
    assert out["first_layer_alternatives"]["u8"]["kernel"] == "mfma_i8_first_u8"
    assert out["model_predict_resident"]["value"] > 0
    t = out["targets"]
    assert t["xnor_m0_hbm_frac"] > 0 and t["int8_mfma_frac"]["layers"] == 8
    assert set(t["images_per_s"]) == {"vgg64_full_bnn", "vgg_large_full_qnn_w8a8", "imagenet224_resnet10_w4a4"}


# ---------------------------------------------------------------------------------------------------------------
# qnn_conv2d_dense_forward: last conv group + classifier in one launch
# ---------------------------------------------------------------------------------------------------------------
def _head_case(seed, N, cin, wkind, abits_in, act, dense_kind, units, bn_d=True):
    from test_gpu_parity import BIN_ACT, Q, _rand_bn
    rng = np.random.default_rng(seed)
    in_act = Q(abits_in) if abits_in > 1 else BIN_ACT
    x = O.run_spec([in_act], rng.standard_normal((N, 8, 8, cin)).astype(F32))
    conv = {"op": "conv", "kind": wkind, "kernel": rng.uniform(-1, 1, (3, 3, cin, 64)).astype(F32),
            "bias": (rng.standard_normal(64) * 0.05).astype(F32), "strides": (1, 1), "padding": "same"}
    if wkind == "quantized":
        conv["nb"] = 4
    dense = {"op": "dense", "kind": dense_kind, "kernel": rng.uniform(-1, 1, (1024, units)).astype(F32),
             "bias": (rng.standard_normal(units) * 0.1).astype(F32)}
    if dense_kind == "quantized":
        dense["nb"] = 4
    spec = [conv, _rand_bn(rng, 64, 9 * cin * 0.12), act, {"op": "maxpool", "size": 2}, {"op": "flatten"}, dense]
    if bn_d:
        spec.append(_rand_bn(rng, units, 1024 * 0.1))
    return x, in_act, spec


@pytest.mark.parametrize("N", [1, 3, 257])
@pytest.mark.parametrize("cin,wkind,abits_in,actname,dense_kind,units", [
    (64, "quantized", 4, "q4", "quantized", 10), (64, "quantized", 4, "q2", "quantized", 16),
    (64, "binary", 1, "bin", "binary", 10), (128, "quantized", 4, "q4", "quantized", 7),
    (64, "quantized", 2, "q3", "binary", 1)])
def test_fused_conv_and_classifier_equals_the_two_launches(N, cin, wkind, abits_in, actname, dense_kind, units):
    from test_gpu_parity import BIN_ACT, Q
    act = BIN_ACT if actname == "bin" else Q(int(actname[1]))
    x, in_act, spec = _head_case(N * 31 + cin + units, N, cin, wkind, abits_in, act, dense_kind, units, bn_d=units != 7)
    want = O.run_spec(spec, x)
    # drive the ABI directly: packed int4 input -> logits
    conv, bn_c, dense = spec[0], spec[1], spec[5]
    bn_d = spec[6] if len(spec) > 6 else None
    wc = engine._prepack(conv, _abi.STORE_I4, torch.device("cuda"))
    wd = engine._prepack(dense, _abi.STORE_I4, torch.device("cuda"))
    xp = _abi.pack(dev(x), cin, _abi.FN_GRID, abits_in, _abi.STORE_I4)
    ci, cs = (dev(a) for a in engine.bn_constants(bn_c))
    di, ds = (dev(a) for a in engine.bn_constants(bn_d)) if bn_d is not None else (None, None)
    fn, ab = engine._act_code(act)
    y = _abi.conv2d_dense(wc, wd, xp, _abi.STORE_I4, abits_in, N, 8, 8, ci, cs, fn, ab if fn == _abi.FN_QUANTIZED_TANH else 0,
                          di, ds)
    assert y is not None and _abi.last_kernel() in ("mfma_i4_halo64x64+dense", "mfma_i4_areg64x64+dense")
    np.testing.assert_array_equal(host(y), want)
    # ... and equals the two separate launches bit for bit
    h, _, _ = _abi.conv2d(wc, xp, _abi.STORE_I4, abits_in, N, 8, 8, ci, cs, fn, ab if fn == _abi.FN_QUANTIZED_TANH else 0,
                          2, _abi.STORE_I4)
    y2 = _abi.dense(wd, h, _abi.STORE_I4, ab, N, di, ds)
    assert torch.equal(y, y2)


def test_fused_classifier_only_where_the_kernel_exists():
    from test_gpu_parity import Q
    x, in_act, spec = _head_case(5, 2, 64, "quantized", 4, Q(4), "quantized", 10)
    wc = engine._prepack(spec[0], _abi.STORE_I4, torch.device("cuda"))
    wd = engine._prepack(spec[5], _abi.STORE_I4, torch.device("cuda"))
    ci, cs = (dev(a) for a in engine.bn_constants(spec[1]))
    x16 = _abi.pack(torch.zeros((2, 16, 16, 64), device="cuda"), 64, _abi.FN_GRID, 4, _abi.STORE_I4)
    assert _abi.conv2d_dense(wc, wd, x16, _abi.STORE_I4, 4, 2, 16, 16, ci, cs, _abi.FN_QUANTIZED_TANH, 4, None, None) is None
    # whole networks: the headline configs take the fused head, with the same logits as without it
    for idx in (1, 2):
        cf = nets.baseline_config(idx)
        sp = nets.build_spec(cf, nets.SEED_BASE + idx)
        xi = nets.synthetic_images(cf, 9, 3)
        m = engine.FusedModel(sp, first_layer="exact")
        m.kernel_log = []
        got = host(m(dev(xi)))
        assert m.kernel_log[-2:] == ["mfma_i4_halo64x64+dense", "(fused into the conv)"], m.kernel_log
        np.testing.assert_array_equal(got, O.run_spec(sp, xi, float_conv="device"))
        m2 = engine.FusedModel(sp, first_layer="exact")
        m2.fuse_head = False
        m2.kernel_log = []
        np.testing.assert_array_equal(host(m2(dev(xi))), got)
        assert m2.kernel_log[-1].startswith("dense_"), m2.kernel_log
    # VALU-only family: no matrix-pipe head
    _abi.set_conv_impl(_abi.IMPL_VALU)
    try:
        m3 = engine.FusedModel(nets.build_spec(nets.baseline_config(2), 3), first_layer="exact")
        m3.kernel_log = []
        m3(dev(nets.synthetic_images(nets.baseline_config(2), 2, 1)))
        assert not any("+dense" in k for k in m3.kernel_log)
    finally:
        _abi.set_conv_impl(_abi.IMPL_AUTO)


@pytest.mark.parametrize("rows,cols", [(64, 1000), (5, 10), (1, 1), (300, 7), (0, 10)])
def test_softmax_entry_matches_the_float64_definition(rows, cols):
    """qnn_softmax_f32 (the classifier's activation='softmax', models/resnet.py:137): float64 inside, one rounding -- equal to
    the oracle's definition to the last float32 bit except where exp differs by an ulp (atol 1e-7 on probabilities)."""
    rng = np.random.default_rng(rows * 31 + cols)
    x = (rng.standard_normal((rows, cols)) * 6).astype(np.float32)
    if rows:
        x[0, 0] = 80.0                                  # float32 exp would overflow without the max shift
    got = host(_abi.softmax(dev(x)))
    assert got.shape == x.shape
    if rows:
        np.testing.assert_allclose(got, O.softmax(x), rtol=0, atol=1e-7)
        np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-6)
        want_t = torch.softmax(dev(x).double(), dim=-1).float()
        np.testing.assert_allclose(got, host(want_t), rtol=0, atol=1e-7)
