"""QNN_STORE_T2: ternary activations and weights as sign / mask bit planes, contracted with two popcounts
(layers/ternary_layers.py:77-84,156-174, layers/ternary_ops.py).  Bit-exact against the oracle's
TernaryConv2D / TernaryDense arithmetic and against the int4-code path it replaces."""
import zlib

import numpy as np
import pytest
import torch

import qnn_amd
from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O
from test_gpu_parity import Q, _rand_bn, dev, host

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.mark.parametrize("C", [3, 16, 32, 37, 64, 96])
def test_t2_pack_layout_and_roundtrip(C):
    rng = np.random.default_rng(C)
    x = rng.integers(-1, 2, (5, 7, C)).astype(F32)
    pixels = 35
    p = _abi.pack(dev(x), C, _abi.FN_GRID, 1, _abi.STORE_T2)
    pairs = (C + 31) // 32
    assert p.shape == (pixels, 2 * pairs)
    words = host(p).view(np.uint32)
    flat = x.reshape(pixels, C)
    want = np.zeros_like(words)
    for c in range(C):
        want[:, 2 * (c // 32)] |= (flat[:, c] != 0).astype(np.uint32) << np.uint32(c % 32)        # mask plane
        want[:, 2 * (c // 32) + 1] |= (flat[:, c] > 0).astype(np.uint32) << np.uint32(c % 32)     # sign plane
    np.testing.assert_array_equal(words, want)
    np.testing.assert_array_equal(host(_abi.unpack(p, pixels, C, _abi.STORE_T2, 1)).reshape(x.shape), x)
    with pytest.raises(_abi.QnnError, match="already"):
        _abi.pack(dev(x), C, _abi.FN_QUANTIZED_TANH, 2, _abi.STORE_T2)


def _tern_case(name, shape, cout, k=3, stride=1, kind="ternary", bias=True):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = rng.integers(-1, 2, shape).astype(F32)
    op = {"op": "conv", "kind": kind, "kernel": rng.uniform(-1, 1, (k, k, shape[3], cout)).astype(F32),
          "strides": (stride, stride), "padding": "same"}
    if bias:
        op["bias"] = (rng.standard_normal(cout) * 0.05).astype(F32)
    return rng, x, op


def _oracle(x, op, bn, act, pool):
    spec = [dict(op)] + ([bn] if bn is not None else []) + ([act] if act is not None else [])
    if pool == 2:
        spec.append({"op": "maxpool", "size": 2})
    return O.run_spec(spec, x)


CASES = [("c32", (3, 12, 12, 32), 32, 3, 1, "ps_t2_cw2_k3"), ("c16_pad", (2, 9, 11, 16), 64, 3, 1, "ps_t2_cw2_k3"),
         ("c64", (2, 10, 10, 64), 64, 3, 1, "ps_t2_cw4_k3"), ("c128", (1, 6, 6, 128), 32, 3, 1, "ps_t2_cw8_k3"),
         ("c32_s2", (2, 15, 17, 32), 64, 3, 2, "ps_t2_cw2_k3"), ("c64_1x1_s2", (2, 16, 16, 64), 32, 1, 2, "ps_t2_cw4_k1"),
         ("c96_generic", (2, 8, 8, 96), 40, 3, 1, "generic"), ("c3", (2, 12, 12, 3), 16, 3, 1, "ps_t2_cw2_k3")]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_t2_conv_bit_exact_vs_oracle_and_vs_int4_codes(case):
    name, shape, cout, k, stride, kern = case
    rng, x, op = _tern_case(name, shape, cout, k, stride, bias="c16" not in name)
    N, H, W, C = shape
    w2 = engine._prepack(op, _abi.STORE_T2, torch.device("cuda"), stride=stride)
    w4 = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=stride)
    x2 = _abi.pack(dev(x), C, _abi.FN_GRID, 1, _abi.STORE_T2)
    x4 = _abi.pack(dev(x), C, _abi.FN_GRID, 1, _abi.STORE_I4)
    bn = _rand_bn(rng, cout, k * k * C * 0.3)
    inv, shift = (dev(a) for a in engine.bn_constants(bn))
    Ho, Wo = -(-H // stride), -(-W // stride)
    for use_bn, act, pool, store in ((False, None, 1, _abi.STORE_F32), (True, None, 1, _abi.STORE_F32),
                                     (True, Q(4), 2 if min(Ho, Wo) >= 2 else 1, _abi.STORE_I4),
                                     (True, Q(2), 1, _abi.STORE_F32)):
        fn, ab = (_abi.FN_NONE, 0) if act is None else (_abi.FN_QUANTIZED_TANH, act["nb"])
        y2, ho, wo = _abi.conv2d(w2, x2, _abi.STORE_T2, 1, N, H, W, inv if use_bn else None, shift if use_bn else None,
                                 fn, ab, pool, store)
        assert _abi.last_kernel() == kern
        y4, _, _ = _abi.conv2d(w4, x4, _abi.STORE_I4, 1, N, H, W, inv if use_bn else None, shift if use_bn else None,
                               fn, ab, pool, store)
        assert torch.equal(y2, y4)                                          # same bits as the int4-code path
        got = host(y2) if store == _abi.STORE_F32 else host(_abi.unpack(y2, N * ho * wo, cout, store, ab)).reshape(N, ho, wo, cout)
        np.testing.assert_array_equal(got, _oracle(x, op, bn if use_bn else None, act, pool))


def test_t2_binary_weights_against_ternary_activations():
    rng, x, op = _tern_case("binw", (2, 10, 10, 64), 32, kind="binary")
    w2 = engine._prepack(op, _abi.STORE_T2, torch.device("cuda"))
    x2 = _abi.pack(dev(x), 64, _abi.FN_GRID, 1, _abi.STORE_T2)
    y, _, _ = _abi.conv2d(w2, x2, _abi.STORE_T2, 1, 2, 10, 10)
    np.testing.assert_array_equal(host(y), _oracle(x, op, None, None, 1))


@pytest.mark.parametrize("K,units", [(64, 10), (1024, 10), (512, 40), (96, 7)])
def test_t2_dense(K, units):
    rng = np.random.default_rng(K + units)
    x = rng.integers(-1, 2, (33, K)).astype(F32)
    op = {"op": "dense", "kind": "ternary", "kernel": rng.uniform(-1, 1, (K, units)).astype(F32),
          "bias": (rng.standard_normal(units) * 0.1).astype(F32)}
    w = engine._prepack(op, _abi.STORE_T2, torch.device("cuda"))
    xp = _abi.pack(dev(x), K, _abi.FN_GRID, 1, _abi.STORE_T2)
    y = _abi.dense(w, xp, _abi.STORE_T2, 1, 33)
    assert _abi.last_kernel() in ("dense_t2", "generic")
    np.testing.assert_array_equal(host(y), O.run_spec([op], x))


def test_t2_rejects_what_it_cannot_hold():
    rng, x, op = _tern_case("rej", (1, 8, 8, 32), 32, kind="quantized")
    op["nb"] = 4
    with pytest.raises(_abi.QnnError, match="ternary"):
        engine._prepack(op, _abi.STORE_T2, torch.device("cuda"))
    op = dict(op, kind="ternary")
    w = engine._prepack(op, _abi.STORE_T2, torch.device("cuda"))
    x4 = _abi.pack(dev(x), 32, _abi.FN_GRID, 1, _abi.STORE_I4)
    with pytest.raises(_abi.QnnError, match="prepacked for store"):
        _abi.conv2d(w, x4, _abi.STORE_I4, 1, 1, 8, 8)
    with pytest.raises(_abi.QnnError, match="out_store"):
        _abi.conv2d(w, _abi.pack(dev(x), 32, _abi.FN_GRID, 1, _abi.STORE_T2), _abi.STORE_T2, 1, 1, 8, 8,
                    out_store=_abi.STORE_T2)


@pytest.mark.parametrize("arch,nf", [("VGG", 32), ("VGG", 64), ("RESNET", None)])
def test_full_tnn_networks_on_bit_planes(arch, nf):
    """Whole full-tnn networks: every engine, sign / mask planes on and off -- same bits, equal to the oracle; the
    ternary x ternary layers really run on the two-popcount kernels."""
    if arch == "VGG":
        cf = nets.Config(network_type="full-tnn", architecture="VGG", nla=1, nlb=1, nlc=1, nfa=nf, nfb=nf, nfc=nf)
    else:
        cf = nets.Config(network_type="full-tnn", architecture="RESNET", nres=1, dim=32)
    spec = nets.build_spec(cf, 17)
    x = nets.synthetic_images(cf, 6, 23)
    want = O.run_spec(spec, x, float_conv="device")
    results = {}
    for t2 in (True, False):
        engine.TERNARY_T2 = t2
        try:
            for cls in (engine.ResidualFusedModel, engine.GraphModel, engine.LayerModel):
                m = cls(spec)
                if cls is engine.ResidualFusedModel:
                    m.kernel_log = []
                got = host(m(dev(x)))
                if arch == "VGG":
                    np.testing.assert_array_equal(got, want, err_msg="%s t2=%s" % (cls.__name__, t2))
                else:
                    np.testing.assert_allclose(got, want, atol=1e-6, err_msg="%s t2=%s" % (cls.__name__, t2))   # softmax exp ulp
                results[(cls.__name__, t2)] = got
                if cls is engine.ResidualFusedModel:
                    used = any(k.startswith("ps_t2") or k == "dense_t2" for k in m.kernel_log)
                    assert used == t2, (t2, m.kernel_log)
        finally:
            engine.TERNARY_T2 = True
    for name in ("ResidualFusedModel", "GraphModel", "LayerModel"):
        np.testing.assert_array_equal(results[(name, True)], results[(name, False)])
