"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bar: bit-exact for every integer / grid-valued path (XNOR+popcount, packed int4,
int8, all epilogues); float-input layers are bit-exact against the oracle's
device-order FMA chain and within 1e-5*max(1,|y|) of the oracle's ideal
(float64-accumulated) convolution.
"""
import json
import os
import zlib

import numpy as np
import pytest
import torch

import qnn_amd
from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def within(got, want, mult=1.0):
    """|got - want| <= mult * 1e-5 * max(1, |want|); the failure message carries the measured ratio."""
    r = float((np.abs(got.astype(np.float64) - want) / (1e-5 * np.maximum(1.0, np.abs(want)))).max())
    assert r <= mult, "max |d| / (1e-5*max(1,|y|)) = %.3g > %g" % (r, mult)
    return r


def edge_values():
    e = [0.0, -0.0, 2.0 ** -24, 2.0 ** -23, -2.0 ** -24, 1e-9, -1e-9, 1e-45, 0.5, -0.5, 1.0, -1.0,
         0.0625, 0.1875, 0.3125, 0.4375, 0.9375, 0.96875, -0.0625, -0.1875, -0.3125, -0.9375, -1.2, 1.2,
         3.5 / 128, 2.5 / 128, 1.5 / 128, 0.5 / 128, -0.5 / 128, 127.5 / 128, 126.5 / 128, 7.5 / 8,
         6.5 / 8, 1e30, -1e30]
    return np.array(e, dtype=F32)


# ---------------------------------------------------------------------------
def test_native_library_is_what_runs():
    assert os.path.exists(_abi.lib_path())
    assert _abi.load().qnn_version() == 4


def test_binary_tanh_matches_oracle():
    rng = np.random.default_rng(0)
    x = np.concatenate([edge_values(), rng.standard_normal(100003).astype(F32),
                        (rng.standard_normal(5000) * 1e-7).astype(F32)])
    y = host(qnn_amd.binary_tanh(dev(x)))
    np.testing.assert_array_equal(y, O.binary_tanh(x))
    np.testing.assert_array_equal(host(qnn_amd.binarize(dev(x))), O.binarize(x))


@pytest.mark.parametrize("nb", [2, 3, 4, 8, 16])
def test_quantized_tanh_matches_oracle(nb):
    rng = np.random.default_rng(nb)
    x = np.concatenate([edge_values(), rng.uniform(-1.3, 1.3, 100001).astype(F32),
                        (np.arange(-300, 300) / 256.0).astype(F32)])
    y = host(qnn_amd.quantized_tanh(dev(x), nb))
    np.testing.assert_array_equal(y, O.quantized_tanh(x, nb))


def test_ternary_tanh_matches_oracle():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((64, 1000)).astype(F32)
    np.testing.assert_array_equal(host(qnn_amd.ternary_tanh(dev(x))), O.ternary_tanh(x))


@pytest.mark.parametrize("C", [3, 16, 32, 37, 64, 256])
def test_pack_unpack_roundtrip(C):
    rng = np.random.default_rng(C)
    x = rng.standard_normal((5, 7, C)).astype(F32)
    pixels = 35
    # binary
    p = _abi.pack(dev(x), C, _abi.FN_BINARY_TANH, 1, _abi.STORE_BIN)
    assert p.shape == (pixels, (C + 31) // 32)
    back = host(_abi.unpack(p, pixels, C, _abi.STORE_BIN, 1)).reshape(x.shape)
    np.testing.assert_array_equal(back, O.binary_tanh(x))
    # explicit bit layout: bit j of word w is channel 32w+j
    words = host(p).view(np.uint32)
    want = np.zeros_like(words)
    bits = (O.binary_tanh(x).reshape(pixels, C) > 0)
    for c in range(C):
        want[:, c // 32] |= bits[:, c].astype(np.uint32) << np.uint32(c % 32)
    np.testing.assert_array_equal(words, want)
    for nb, store in ((2, _abi.STORE_I4), (4, _abi.STORE_I4), (8, _abi.STORE_I8), (5, _abi.STORE_I8)):
        p = _abi.pack(dev(x), C, _abi.FN_QUANTIZED_TANH, nb, store)
        back = host(_abi.unpack(p, pixels, C, store, nb)).reshape(x.shape)
        np.testing.assert_array_equal(back, O.quantized_tanh(x, nb))
        q = O.quantized_tanh(x, nb)
        p2 = _abi.pack(dev(q), C, _abi.FN_GRID, nb, store)
        assert torch.equal(p, p2)


def _fixture(code):
    d = np.load(os.path.join(GOLD, "resnet3_%s.npz" % code))
    meta = json.loads(bytes(d["meta_json"]).decode())
    return d, meta


@pytest.mark.parametrize("code,nb", [("bb", None), ("44", 4), ("42", 2), ("48", 8)])
def test_weight_quantizers_on_trained_kernels(code, nb):
    d, _ = _fixture(code)
    for key in [k for k in d.files if k.endswith("_kernel")]:
        k = d[key]
        if nb is None:
            w = _abi.Weights(_abi.W_BINARY, 1, 1.0, dev(k), None, 1, True, _abi.STORE_F32)
            want = O.binarize(k)
        else:
            w = _abi.Weights(_abi.W_QUANT, nb, 1.0, dev(k), None, 1, True, _abi.STORE_F32)
            want = O.quantize(k, nb)
        got = host(w.dequant()).reshape(want.shape)
        np.testing.assert_array_equal(got, want)


# ---------------------------------------------------------------------------
# single-layer parity
# ---------------------------------------------------------------------------
def _oracle_group(x, op, bn, act, pool, float_conv="ideal"):
    spec = [dict(op)]
    if bn is not None:
        spec.append(bn)
    if act is not None:
        spec.append(act)
    if pool == 2:
        spec.append({"op": "maxpool", "size": 2})
    return O.run_spec(spec, x, float_conv=float_conv)


def _run_group(x_np, in_act, op, bn, act, pool, out_store):
    """Drive qnn_conv2d_forward directly.  in_act: None (float32 input) or an act dict
    describing the grid the input values are on."""
    N, H, W, C = x_np.shape
    st = tuple(op.get("strides", (1, 1)))
    wstore = engine._wstore(op)
    if in_act is None:
        x_store, x_bits, xin = _abi.STORE_F32, 0, dev(x_np)
    else:
        fn, bits = engine._act_code(in_act)
        x_store = engine._join_store(bits, wstore)
        x_bits = bits
        xin = _abi.pack(dev(x_np), C, _abi.FN_GRID, bits, x_store)
    w = engine._prepack(op, x_store, torch.device("cuda"), stride=st[0],
                        same_pad=op.get("padding", "same") == "same")
    inv = shift = None
    if bn is not None:
        i, s = engine.bn_constants(bn)
        inv, shift = dev(i), dev(s)
    fn, abits = _abi.FN_NONE, 0
    if act is not None:
        fn, abits = engine._act_code(act)
    y, Ho, Wo = _abi.conv2d(w, xin, x_store, x_bits, N, H, W, inv, shift, fn,
                            abits if fn == _abi.FN_QUANTIZED_TANH else 0, pool, out_store)
    kern = _abi.last_kernel()
    cout = op["kernel"].shape[3]
    if out_store == _abi.STORE_F32:
        return host(y), kern
    out = _abi.unpack(y, N * Ho * Wo, cout, out_store, abits if abits else 1)
    return host(out).reshape(N, Ho, Wo, cout), kern


def _rand_bn(rng, n, var):
    return dict(op="bn", eps=1e-4, gamma=rng.uniform(-1.5, 1.5, n).astype(F32),
                beta=(rng.standard_normal(n) * 0.5).astype(F32),
                mean=(rng.standard_normal(n) * 0.1 * np.sqrt(var)).astype(F32),
                var=(var * rng.uniform(0.8, 1.25, n)).astype(F32))


BIN_ACT = {"op": "act", "fn": "binary_tanh"}


def Q(nb):
    return {"op": "act", "fn": "quantized_tanh", "nb": nb}


LAYER_CASES = [
    # name, (N,H,W,Cin), Cout, k, stride, wkind, wnb, in_act, expected kernel prefix
    ("bin_64_64", (3, 16, 16, 64), 64, 3, 1, "binary", None, BIN_ACT, "ps_bin_cw2_k3"),
    ("bin_16_16_pad", (2, 9, 11, 16), 32, 3, 1, "binary", None, BIN_ACT, "ps_bin_cw1_k3"),
    ("bin_128_64", (2, 8, 8, 128), 64, 3, 1, "binary", None, BIN_ACT, "ps_bin_cw4_k3"),
    ("bin_256_64", (1, 6, 6, 256), 64, 3, 1, "binary", None, BIN_ACT, "ps_bin_cw8_k3"),
    ("bin_s2", (2, 16, 16, 32), 64, 3, 2, "binary", None, BIN_ACT, "ps_bin_cw1_k3"),
    ("bin_1x1_s2", (2, 16, 16, 32), 64, 1, 2, "binary", None, BIN_ACT, "ps_bin_cw1_k1"),
    ("bin_generic", (2, 8, 8, 96), 40, 3, 1, "binary", None, BIN_ACT, "generic"),
    ("i4_64_64", (3, 16, 16, 64), 64, 3, 1, "quantized", 4, Q(4), "ps_i4_cw8_k3"),
    ("i4_16_16", (2, 12, 12, 16), 16, 3, 1, "quantized", 4, Q(4), "ps_i4_cw2_k3"),
    ("i4_32_64_s2", (2, 16, 16, 32), 64, 3, 2, "quantized", 4, Q(4), "ps_i4_cw4_k3"),
    ("i4_128_32", (1, 8, 8, 128), 32, 3, 1, "quantized", 4, Q(4), "ps_i4_cw16_k3"),
    ("i4_1x1_s2", (2, 16, 16, 16), 32, 1, 2, "quantized", 4, Q(4), "ps_i4_cw2_k1"),
    ("i4_1x1_s2_32_64", (3, 15, 9, 32), 64, 1, 2, "quantized", 4, Q(4), "ps_i4_cw4_k1"),
    ("i4_1x1_s1_16_64", (2, 5, 7, 16), 64, 1, 1, "quantized", 4, Q(4), "ps_i4_cw2_k1"),
    ("i4_w2_a4", (2, 8, 8, 64), 64, 3, 1, "quantized", 2, Q(4), "ps_i4_cw8_k3"),
    ("i4_w4_a2", (2, 8, 8, 64), 64, 3, 1, "quantized", 4, Q(2), "ps_i4_cw8_k3"),
    ("i4_generic_c24", (2, 8, 8, 24), 24, 3, 1, "quantized", 4, Q(4), "generic"),
    ("i8_32_32", (2, 10, 10, 32), 32, 3, 1, "quantized", 8, Q(8), "ps_i8_cw8_k3"),
    ("i8_64_64", (1, 8, 8, 64), 64, 3, 1, "quantized", 8, Q(8), "ps_i8_cw16_k3"),
    ("i8_w4_a8", (1, 8, 8, 16), 16, 3, 1, "quantized", 4, Q(8), "ps_i8_cw4_k3"),
    ("binw_a4", (2, 8, 8, 64), 64, 3, 1, "binary", None, Q(4), "ps_i4_cw8_k3"),
    ("w4_abin", (2, 8, 8, 64), 64, 3, 1, "quantized", 4, BIN_ACT, "ps_i4_cw8_k3"),
    # shapes aimed at the int8-MFMA implicit GEMM (cin, cout multiples of 64)
    ("i8_256_256", (2, 8, 8, 256), 256, 3, 1, "quantized", 8, Q(8), "generic"),
    ("i8_64_128", (3, 10, 6, 64), 128, 3, 1, "quantized", 8, Q(8), "ps_i8_cw16_k3"),
    ("i4_128_128", (2, 12, 12, 128), 128, 3, 1, "quantized", 4, Q(4), "ps_i4_cw16_k3"),
    ("i4_64_64_s2", (2, 14, 14, 64), 64, 3, 2, "quantized", 4, Q(4), "ps_i4_cw8_k3"),
    ("i8_128_64_1x1_s2", (2, 12, 12, 128), 64, 1, 2, "quantized", 8, Q(8), "generic"),
    ("i4_64_64_big", (37, 16, 16, 64), 64, 3, 1, "quantized", 4, Q(4), "ps_i4_cw8_k3"),
    # two 64-channel groups per tap with one 64-filter slice (register-operand MFMA kernel, KC = 2)
    ("i4_128_64", (2, 12, 12, 128), 64, 3, 1, "quantized", 4, Q(4), "ps_i4_cw16_k3"),
    ("i8_128_64", (3, 9, 7, 128), 64, 3, 1, "quantized", 8, Q(8), ""),
    ("i4_128_64_s2", (5, 13, 13, 128), 64, 3, 2, "quantized", 4, Q(4), "ps_i4_cw16_k3"),
    # 256-filter multiples with int8 activations: the LDS-DMA staged 256x256 GEMM (one K-step, three 64-channel chunks per
    # tap, stride 2 with ragged tiles, two filter slices)
    ("i8_64_256_1x1", (2, 9, 7, 64), 256, 1, 1, "quantized", 8, Q(8), ""),
    ("i8_192_256", (1, 10, 6, 192), 256, 3, 1, "quantized", 8, Q(8), ""),
    ("i8_256_512_s2", (2, 12, 12, 256), 512, 3, 2, "quantized", 8, Q(8), ""),
    ("i8_w4_a8_64_256", (3, 16, 16, 64), 256, 3, 1, "quantized", 4, Q(8), ""),
]


def _mfma_eligible(case):
    name, xs, cout, k, stride, wkind, wnb, in_act, _ = case
    if in_act is BIN_ACT and wkind == "binary":
        return False
    return xs[3] % 64 == 0 and cout % 64 == 0


@pytest.fixture(params=[_abi.IMPL_VALU, _abi.IMPL_MFMA], ids=["valu", "mfma"])
def impl(request):
    _abi.set_conv_impl(request.param)
    yield request.param
    _abi.set_conv_impl(_abi.IMPL_AUTO)


@pytest.mark.parametrize("case", LAYER_CASES, ids=[c[0] for c in LAYER_CASES])
@pytest.mark.parametrize("pool", [1, 2])
def test_lowbit_conv_layer_bit_exact(case, pool, impl):
    name, xs, cout, k, stride, wkind, wnb, in_act, kernel_name = case
    if impl == _abi.IMPL_MFMA:
        if not _mfma_eligible(case):
            pytest.skip("shape not eligible for the MFMA kernel")
        kernel_name = "mfma_"
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    N, H, W, C = xs
    pre = rng.standard_normal(xs).astype(F32)
    x = O.run_spec([in_act], pre)          # values on the input grid
    op = {"op": "conv", "kind": wkind, "kernel": rng.uniform(-1, 1, (k, k, C, cout)).astype(F32),
          "bias": (rng.standard_normal(cout) * 0.05).astype(F32), "strides": (stride, stride),
          "padding": "same"}
    if wnb:
        op["nb"] = wnb
    var = k * k * C * (1.0 if wkind == "binary" else 0.3) * (1.0 if in_act is BIN_ACT else 0.4)
    bn = _rand_bn(rng, cout, var)
    Ho = -(-H // stride)
    Wo = -(-W // stride)
    if pool == 2 and (Ho < 2 or Wo < 2):
        pytest.skip("no pool window")
    # 1) plain float32 output, bias only (the Keras call() surface)
    got, kern = _run_group(x, in_act, op, None, None, pool, _abi.STORE_F32)
    if k == 1 and wnb == 4 and in_act == Q(4) and pool == 1 and impl != _abi.IMPL_MFMA:
        assert kern == "pw_i4_f32", kern          # 1x1 int4 -> float32: the projection-shortcut kernel
    else:
        assert kern.startswith(kernel_name), kern
    want = _oracle_group(x, op, None, None, pool)
    np.testing.assert_array_equal(got, want)
    # 2) fused BN + activation (+pool), packed output, every storage that can hold it
    for act in (BIN_ACT, Q(2), Q(4), Q(8)):
        fn, bits = engine._act_code(act)
        stores = [_abi.STORE_F32]
        if bits == 1:
            stores += [_abi.STORE_BIN, _abi.STORE_I4, _abi.STORE_I8]
        elif bits <= 4:
            stores += [_abi.STORE_I4, _abi.STORE_I8]
        else:
            stores += [_abi.STORE_I8]
        want = _oracle_group(x, op, bn, act, pool)
        for out_store in stores:
            got, _ = _run_group(x, in_act, op, bn, act, pool, out_store)
            np.testing.assert_array_equal(got, want, err_msg="act=%r store=%d" % (act, out_store))
    # 3) BN without activation -> float32
    got, _ = _run_group(x, in_act, op, bn, None, 1, _abi.STORE_F32)
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, None, 1))


FLOAT_CASES = [
    ("img3_64", (3, 32, 32, 3), 64, 3, 1, "quantized", 4, "ps_f32_cw3_k3"),
    ("img3_64_bin", (2, 32, 32, 3), 64, 3, 1, "binary", None, "ps_f32_cw3_k3"),
    ("img1_64", (2, 28, 28, 1), 64, 3, 1, "binary", None, "ps_f32_cw1_k3"),
    ("img3_16", (2, 17, 13, 3), 16, 3, 1, "quantized", 8, "ps_f32_cw3_k3"),
    ("float_generic", (2, 9, 9, 5), 6, 3, 2, "quantized", 4, "generic"),
    ("img3_256", (2, 12, 10, 3), 256, 3, 1, "quantized", 8, "generic"),
    ("img3_128", (2, 8, 8, 3), 128, 3, 1, "quantized", 4, "generic"),
]


@pytest.mark.parametrize("case", FLOAT_CASES, ids=[c[0] for c in FLOAT_CASES])
@pytest.mark.parametrize("pool", [1, 2])
def test_float_input_layer(case, pool, impl):
    name, xs, cout, k, stride, wkind, wnb, kernel_name = case
    if impl == _abi.IMPL_MFMA:
        if not (cout in (64, 128, 256) and xs[3] in (1, 3) and k == 3):
            pytest.skip("shape not eligible for the f32-MFMA first-layer kernel")
        kernel_name = "mfma_f32_first_cin%d" % xs[3]
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    N, H, W, C = xs
    x = (rng.integers(0, 256, xs).astype(F32) / F32(255)).astype(F32)
    op = {"op": "conv", "kind": wkind, "kernel": rng.uniform(-1, 1, (k, k, C, cout)).astype(F32),
          "bias": (rng.standard_normal(cout) * 0.05).astype(F32), "strides": (stride, stride),
          "padding": "same"}
    if wnb:
        op["nb"] = wnb
    got, kern = _run_group(x, None, op, None, None, pool, _abi.STORE_F32)
    assert kern == kernel_name or (impl == _abi.IMPL_VALU and kernel_name == "generic" and kern.startswith("ps_f32"))
    exact = _oracle_group(x, op, None, None, pool, float_conv="device")
    np.testing.assert_array_equal(got, exact)           # same FMA chain -> bit-exact
    ideal = _oracle_group(x, op, None, None, pool)
    tol = 1e-5 * np.maximum(1.0, np.abs(ideal))          # north_star: 1e-5 on the float path
    assert np.all(np.abs(got.astype(np.float64) - ideal) <= tol)
    bn = _rand_bn(rng, cout, k * k * C * 0.3)
    for act, store in ((BIN_ACT, _abi.STORE_BIN), (Q(4), _abi.STORE_I4), (Q(8), _abi.STORE_I8),
                       (Q(4), _abi.STORE_F32)):
        got, _ = _run_group(x, None, op, bn, act, pool, store)
        want = _oracle_group(x, op, bn, act, pool, float_conv="device")
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 64), (2, 8, 8, 64, 64), (2, 5, 7, 128, 64),
                                   (1, 32, 32, 256, 128), (2, 40, 24, 64, 192), (1, 3, 3, 64, 64)])
@pytest.mark.parametrize("with_bn", [False, True])
def test_layer_surface_xnor_fused(shape, with_bn):
    """float32 in -> binary_tanh on load -> XNOR conv -> float32 out in ONE kernel (M0 path)."""
    N, H, W, C, Cout = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    pre = rng.standard_normal((N, H, W, C)).astype(F32)
    pre.reshape(-1)[:7] = [0.0, 2.0 ** -24, 2.0 ** -23, -0.0, 1e-30, -1e-30, 1.0]   # threshold edge cases
    op = {"op": "conv", "kind": "binary", "kernel": rng.uniform(-1, 1, (3, 3, C, Cout)).astype(F32),
          "bias": (rng.standard_normal(Cout) * 0.05).astype(F32), "strides": (1, 1), "padding": "same"}
    bn = _rand_bn(rng, Cout, 9.0 * C) if with_bn else None
    w = engine._prepack(op, _abi.STORE_BIN, torch.device("cuda"))
    inv = shift = None
    if bn is not None:
        i, s_ = engine.bn_constants(bn)
        inv, shift = dev(i), dev(s_)
    _abi.set_conv_impl(_abi.IMPL_AUTO)
    y, _, _ = _abi.conv2d_f32in(w, dev(pre), _abi.FN_BINARY_TANH, 1, inv, shift)
    assert _abi.last_kernel().startswith("xnor_f32_cw")
    spec = [BIN_ACT, dict(op)] + ([bn] if bn is not None else [])
    np.testing.assert_array_equal(host(y), O.run_spec(spec, pre))
    # grid input (values already +-1) and the two-pass fallback agree with it
    xb = O.binary_tanh(pre)
    y2, _, _ = _abi.conv2d_f32in(w, dev(xb), _abi.FN_GRID, 1, inv, shift)
    assert torch.equal(y, y2)
    _abi.set_conv_impl(_abi.IMPL_MFMA)       # disables the fused kernel -> pack + conv
    y3, _, _ = _abi.conv2d_f32in(w, dev(pre), _abi.FN_BINARY_TANH, 1, inv, shift)
    _abi.set_conv_impl(_abi.IMPL_AUTO)
    assert torch.equal(y, y3)


@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 64), (2, 8, 8, 64, 64), (2, 5, 7, 128, 64), (1, 28, 28, 64, 128),
                                   (1, 32, 32, 256, 128), (2, 40, 24, 64, 192), (1, 2, 2, 64, 64), (2, 7, 7, 64, 64)])
@pytest.mark.parametrize("pool", [1, 2])
def test_packed_xnor_kernel(shape, pool):
    """packed bits in -> XNOR conv -> BN -> binary_tanh (-> 2x2 max-pool) -> packed bits out."""
    N, H, W, C, Cout = shape
    if pool == 2 and (H < 2 or W < 2):
        pytest.skip("no pool window")
    rng = np.random.default_rng(zlib.crc32(repr((shape, pool)).encode()))
    x = O.binary_tanh(rng.standard_normal((N, H, W, C)).astype(F32))
    op = {"op": "conv", "kind": "binary", "kernel": rng.uniform(-1, 1, (3, 3, C, Cout)).astype(F32),
          "bias": (rng.standard_normal(Cout) * 0.05).astype(F32), "strides": (1, 1), "padding": "same"}
    bn = _rand_bn(rng, Cout, 9.0 * C)
    _abi.set_conv_impl(_abi.IMPL_AUTO)
    got, kern = _run_group(x, BIN_ACT, op, bn, BIN_ACT, pool, _abi.STORE_BIN)
    assert kern.startswith("xnor_pk_cw"), kern
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, BIN_ACT, pool))
    got2, _ = _run_group(x, BIN_ACT, op, None, BIN_ACT, pool, _abi.STORE_BIN)     # no BN
    np.testing.assert_array_equal(got2, _oracle_group(x, op, None, BIN_ACT, pool))


@pytest.mark.parametrize("kind,nb,in_act", [("binary", None, BIN_ACT), ("quantized", 4, Q(4)),
                                            ("quantized", 8, Q(8)), ("quantized", 4, None)])
def test_dense_layer(kind, nb, in_act):
    rng = np.random.default_rng(11)
    N, K, U = 37, 1024, 10
    pre = rng.standard_normal((N, K)).astype(F32)
    x = O.run_spec([in_act], pre) if in_act is not None else pre
    op = {"op": "dense", "kind": kind, "kernel": rng.uniform(-1, 1, (K, U)).astype(F32),
          "bias": (rng.standard_normal(U) * 0.05).astype(F32)}
    if nb:
        op["nb"] = nb
    cls = qnn_amd.BinaryDense if kind == "binary" else qnn_amd.QuantizedDense
    layer = cls(U, **({"nb": nb} if nb else {}))
    layer.build((None, K))
    layer.set_weights([op["kernel"], op["bias"]])
    if in_act is not None:
        layer.input_domain = "binary" if in_act is BIN_ACT else ("quantized", in_act["nb"])
        got = host(layer(dev(x)))
        np.testing.assert_array_equal(got, O.run_spec([op], x))
        # generic float path must agree exactly on grid inputs
        layer.input_domain = None
        np.testing.assert_array_equal(host(layer(dev(x))), got)
    else:
        got = host(layer(dev(x)))
        want = O.run_spec([op], x)
        within(got, want)


def test_layer_classes_on_trained_weights():
    """Keras-surface call() on the reference's trained kernels (older topology:
    use_bias=True), float domain vs packed domain vs oracle."""
    rng = np.random.default_rng(5)
    for code, nb, act in (("bb", None, BIN_ACT), ("44", 4, Q(4))):
        d, meta = _fixture(code)
        for ci in (2, 8, 10, 16, 17, 21):
            k, b = d["conv%d_kernel" % ci], d["conv%d_bias" % ci]
            lm = meta["layers"]["conv%d" % ci]
            kh, kw, cin, cout = k.shape
            x = O.run_spec([act], rng.standard_normal((2, 16, 16, cin)).astype(F32))
            kwargs = dict(kernel_size=(kh, kw), strides=tuple(lm["strides"]), padding=lm["padding"])
            layer = (qnn_amd.BinaryConv2D(cout, **kwargs) if nb is None
                     else qnn_amd.QuantizedConv2D(cout, nb=nb, **kwargs))
            layer.build((None, 16, 16, cin))
            assert abs(float(layer.kernel_lr_multiplier) - lm["klm"]) < 1e-5
            layer.set_weights([k, b])
            op = {"op": "conv", "kind": "binary" if nb is None else "quantized", "kernel": k, "bias": b,
                  "strides": tuple(lm["strides"]), "padding": lm["padding"]}
            if nb:
                op["nb"] = nb
            want = O.run_spec([op], x)
            got_float = host(layer(dev(x)))
            layer.input_domain = "binary" if nb is None else ("quantized", nb)
            got_packed = host(layer(dev(x)))
            np.testing.assert_array_equal(got_packed, want)
            np.testing.assert_array_equal(got_float, want)
            # faithful replay of the lr-multiplier trick stays within the documented band
            fa = O.run_spec([dict(op, klm=np.float32(lm["klm"]))], x, mode="faithful")
            within(fa, want)


# ---------------------------------------------------------------------------
# whole networks
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("idx", [0, 1, 2])
def test_vgg_configs_end_to_end(idx, impl):
    cf = nets.baseline_config(idx)
    spec = nets.build_spec(cf, nets.SEED_BASE + idx)
    x = nets.synthetic_images(cf, 6, nets.SEED_BASE + idx)
    want = O.run_spec(spec, x, float_conv="device")
    fused = engine.FusedModel(spec, first_layer="exact")
    got = host(fused(dev(x)))
    np.testing.assert_array_equal(got, want)
    graph = host(engine.GraphModel(spec)(dev(x)))
    np.testing.assert_array_equal(graph, want)
    layer = host(engine.LayerModel(spec)(dev(x)))
    np.testing.assert_array_equal(layer, want)
    ideal = O.run_spec(spec, x)
    # vs the ideal (float64-accumulated) first-layer conv: identical unless a first-layer value sits within
    # an ulp of a quantisation threshold, where one activation code flips and the logits of that image move.
    # Measured on these seeds: no image of any of the three configurations differs.
    rows = int(np.any(got != ideal, axis=1).sum())
    assert rows == 0, rows


@pytest.mark.parametrize("nt,wb,ab,nfc", [("full-bnn", 1, 1, 16), ("full-bnn", 1, 1, 48), ("full-qnn", 4, 4, 20),
                                          ("full-qnn", 4, 4, 12), ("full-qnn", 8, 8, 6), ("full-qnn", 4, 4, 7)])
def test_flatten_of_channels_that_do_not_fill_packed_words(nt, wb, ab, nfc):
    """The conv in front of Flatten pads every pixel to a word boundary; the dense weights are packed as one
    contiguous K = H*W*C vector.  FusedModel widens the storage until the channels fill whole words, or
    declares the chain not fusable (nets.Model then takes the residual engine); never silently wrong."""
    cf = nets.Config(network_type=nt, wbits=wb, abits=ab, architecture="VGG", nfa=32, nfb=32, nfc=nfc)
    spec = nets.build_spec(cf, 5)
    x = nets.synthetic_images(cf, 3, 5)
    want = O.run_spec(spec, x, float_conv="device")
    try:
        fused = engine.FusedModel(spec, first_layer="exact")
    except _abi.NotFusable:
        assert nfc % 4 != 0                       # no packed storage whose words the channels fill
        fused = None
    if fused is not None:
        last_conv = [st for st in fused.steps if st["kind"] == "conv"][-1]
        assert nfc % _abi.per_word(last_conv["out_store"]) == 0
        np.testing.assert_array_equal(host(fused(dev(x))), want)
    model = nets.Model(cf, spec, first_layer="exact")
    assert type(model.engine).__name__ == ("FusedModel" if fused is not None else "ResidualFusedModel")
    np.testing.assert_array_equal(model.predict(x), want)


def test_model_does_not_mask_real_errors_as_not_fusable(monkeypatch):
    cf = nets.baseline_config(2)
    spec = nets.build_spec(cf, 1)

    def boom(*a, **k):
        raise _abi.QnnError("prepack failed")
    monkeypatch.setattr(engine, "_prepack", boom)
    with pytest.raises(_abi.QnnError, match="prepack failed"):
        nets.Model(cf, spec, first_layer="exact")


def test_one_bit_layers_on_the_matrix_pipe():
    """full-bnn VGG: the fused engine stores +-1 codes as int4 in front of 3x3 64->64 layers so
    that they run on the int8 MFMA kernel; VALU-only mode keeps the bit-packed XNOR path; both
    are bit-exact against the oracle."""
    cf = nets.baseline_config(1)
    spec = nets.build_spec(cf, nets.SEED_BASE + 1)
    x = nets.synthetic_images(cf, 5, nets.SEED_BASE + 1)
    want = O.run_spec(spec, x, float_conv="device")
    try:
        _abi.set_conv_impl(_abi.IMPL_AUTO)
        fused = engine.FusedModel(spec, first_layer="exact")
        stores = [st["x_store"] for st in fused.steps]
        assert _abi.STORE_I4 in stores and stores[0] == _abi.STORE_F32, stores
        got = host(fused(dev(x)))
        np.testing.assert_array_equal(got, want)
        _abi.set_conv_impl(_abi.IMPL_VALU)
        fused_v = engine.FusedModel(spec, first_layer="exact")
        assert _abi.STORE_I4 not in [st["x_store"] for st in fused_v.steps]
        np.testing.assert_array_equal(host(fused_v(dev(x))), want)
    finally:
        _abi.set_conv_impl(_abi.IMPL_AUTO)


@pytest.mark.parametrize("cin,cout,hw,n", [(16, 16, (20, 32), 3), (16, 32, (7, 16), 2), (32, 32, (9, 16), 5),
                                           (32, 64, (5, 48), 2), (16, 16, (33, 16), 9), (16, 16, (224, 224), 2),
                                           (32, 32, (112, 112), 3), (16, 16, (5, 16), 1), (32, 32, (1, 32), 2),
                                           (16, 16, (2, 48), 70), (16, 16, (9, 20), 3), (32, 32, (7, 56), 2),
                                           (64, 64, (56, 56), 3), (64, 64, (8, 8), 5), (64, 128, (5, 23), 2),
                                           (16, 16, (3, 7), 2), (64, 64, (1, 1), 3)])
def test_small_channel_layers_on_the_matrix_pipe(cin, cout, hw, n):
    """3x3 int4 layers with 16 / 32 / 64 input channels.  Row-walking strip kernel (any width: ragged last strip,
    one-pixel images, chunked rows) and, with the switch off, the tile kernels; every border class, 2- and 4-bit
    and binary output codes."""
    rng = np.random.default_rng(cin * 1000 + cout + hw[0])
    H, W = hw
    pre = rng.standard_normal((n, H, W, cin)).astype(F32)
    x = O.run_spec([Q(4)], pre)
    op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (3, 3, cin, cout)).astype(F32),
          "bias": (rng.standard_normal(cout) * 0.05).astype(F32), "strides": (1, 1), "padding": "same"}
    bn = _rand_bn(rng, cout, 9 * cin * 0.12)
    for strip in (1, 0):
        _abi.set_option("strip", strip)
        try:
            for act in (Q(4), Q(2), BIN_ACT):
                want = _oracle_group(x, op, bn, act, 1)
                got, kern = _run_group(x, Q(4), op, bn, act, 1, _abi.STORE_I4)
                if strip:                      # (round 3: the 64-channel layers take it too, with or without a merge)
                    assert kern == "strip_i4_c%d" % cin, kern
                elif cin < 64 and W % 16 == 0:
                    assert kern == "mfma_i4_small_c%d" % cin, kern
                else:
                    assert not kern.startswith("strip"), kern
                np.testing.assert_array_equal(got, want)
        finally:
            _abi.set_option("strip", 1)
    if cin == 64:      # without a bias (the ResNet layers have none) and as "never" / "always" for Cin 64
        op2 = dict(op, bias=None)
        for s64 in (0, 1, -1):
            _abi.set_option("strip64", s64)
            try:
                got, kern = _run_group(x, Q(4), op2, bn, Q(4), 1, _abi.STORE_I4)
                assert kern.startswith("strip") == (s64 != 0), (s64, kern)     # auto = always for un-pooled layers (round 3)
                np.testing.assert_array_equal(got, _oracle_group(x, op2, bn, Q(4), 1))
            finally:
                _abi.set_option("strip64", -1)


@pytest.mark.parametrize("cin,cout,hw,n", [(16, 32, (224, 224), 1), (32, 64, (112, 112), 2), (16, 32, (8, 8), 3),
                                           (32, 64, (7, 9), 2), (16, 64, (5, 33), 2), (32, 32, (1, 1), 2), (16, 32, (2, 40), 5)])
def test_stride2_layers_on_the_strip_kernel(cin, cout, hw, n):
    """The stride-2 3x3 convs that open a ResNet stage (resnet.py:108-112): SAME padding is asymmetric for even sizes
    (0 before / 1 after), symmetric for odd ones; ragged last strips; with and without a bias."""
    rng = np.random.default_rng(cin + cout * 7 + hw[1])
    H, W = hw
    x = O.run_spec([Q(4)], rng.standard_normal((n, H, W, cin)).astype(F32))
    for bias in (True, False):
        op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (3, 3, cin, cout)).astype(F32),
              "bias": (rng.standard_normal(cout) * 0.05).astype(F32) if bias else None, "strides": (2, 2), "padding": "same"}
        bn = _rand_bn(rng, cout, 9 * cin * 0.12)
        for act in (Q(4), Q(3), BIN_ACT):
            got, kern = _run_group(x, Q(4), op, bn, act, 1, _abi.STORE_I4)
            assert kern == "strip_i4_c%d_s2" % cin, kern
            np.testing.assert_array_equal(got, _oracle_group(x, op, bn, act, 1))


def test_vgg_large_8bit_small_batch(impl):
    cf = nets.baseline_config(3)
    spec = nets.build_spec(cf, nets.SEED_BASE + 3)
    x = nets.synthetic_images(cf, 2, nets.SEED_BASE + 3)
    want = O.run_spec(spec, x, float_conv="device")
    got = host(engine.FusedModel(spec, first_layer="exact")(dev(x)))
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("nt,wb,ab", [("full-qnn", 4, 4), ("full-bnn", 1, 1), ("qbnn", 1, 4),
                                      ("qnn", 4, 4), ("full-qnn", 2, 8)])
def test_resnet_small(nt, wb, ab):
    cf = nets.Config(network_type=nt, wbits=wb, abits=ab, architecture="RESNET", nres=1, dim=32)
    spec = nets.build_spec(cf, 42)
    x = nets.synthetic_images(cf, 3, 42)
    want = O.run_spec(spec, x, float_conv="device")
    got = host(engine.GraphModel(spec)(dev(x)))
    if nt == "qnn":   # LeakyReLU activations: float convs everywhere -> tolerance
        np.testing.assert_allclose(got, want, atol=2e-5)
    else:
        np.testing.assert_allclose(got, want, atol=1e-6)     # softmax exp differs in the last ulp
        lm = host(engine.LayerModel(spec)(dev(x)))
        np.testing.assert_allclose(lm, want, atol=1e-6)


@pytest.mark.parametrize("code,wb,ab", [("44", 4, 4), ("bb", None, None)])
def test_trained_reference_checkpoint_end_to_end(code, wb, ab):
    """The reference's own trained ResNet-20 (results/RESNET3/weights_*.hdf5, imported with
    tools/import_keras_hdf5.py; older topology: conv biases, no 0.5 scaling) through the
    engine and the oracle."""
    spec = nets.spec_from_keras_npz(os.path.join(GOLD, "resnet3_full_%s.npz" % code), wb, ab)
    assert sum(op["op"] == "conv" for op in spec) == 21
    x = nets.synthetic_images(nets.Config(dim=32), 5, 11)
    env = O.run_spec(spec, x, float_conv="device", return_all=True)
    want = list(env.values())[-1]
    got = host(engine.GraphModel(spec)(dev(x)))
    np.testing.assert_allclose(got, want, atol=1e-6)          # softmax: exp differs in the last ulp
    assert np.array_equal(got.argmax(-1), want.argmax(-1))
    np.testing.assert_allclose(host(engine.LayerModel(spec)(dev(x))), want, atol=1e-6)
    # the pre-softmax logits are bit-exact
    logits_name = [op["dst"] for op in spec if op["op"] == "dense"][0]
    g = engine.GraphModel(spec[:-1])
    np.testing.assert_array_equal(host(g(dev(x))), env[logits_name])


@pytest.mark.parametrize("nt,wb,ab", [("full-qnn", 4, 4), ("full-bnn", 1, 1), ("qbnn", 1, 4),
                                      ("full-qnn", 2, 8), ("full-qnn", 8, 4), ("qnn", 4, 4)])
@pytest.mark.parametrize("nres", [1, 2])
def test_residual_fused_model(nt, wb, ab, nres):
    """conv->BN->add(shortcut)->x0.5->act fused into one launch, activations packed end to end."""
    cf = nets.Config(network_type=nt, wbits=wb, abits=ab, architecture="RESNET", nres=nres, dim=32)
    spec = nets.build_spec(cf, 77)
    x = nets.synthetic_images(cf, 3, 77)
    want = O.run_spec(spec, x, float_conv="device")
    m = engine.ResidualFusedModel(spec, first_layer="exact")
    m.kernel_log = []
    got = host(m(dev(x)))
    np.testing.assert_allclose(got, want, atol=2e-5 if nt == "qnn" else 1e-6)
    if nt != "qnn":
        # logits before the softmax are bit-exact, and nothing fell back to the generic kernel
        logits_name = [op["dst"] for op in spec if op["op"] == "dense"][0]
        env = O.run_spec(spec, x, float_conv="device", return_all=True)
        np.testing.assert_array_equal(host(engine.ResidualFusedModel(spec[:-1], first_layer="exact")(dev(x))), env[logits_name])
        assert "generic" not in m.kernel_log, m.kernel_log
        # one launch per convolution; a projection shortcut (1x1 strides-2 conv) computed inside the launch of the block's
        # second convolution has none of its own (qnn_projection_t, tests/test_gpu_proj.py)
        assert len(m.kernel_log) == sum(op["op"] == "conv" for op in spec) - sum(k.endswith("_proj") for k in m.kernel_log)
        if (nt, wb, ab) == ("full-qnn", 4, 4):
            # the 16- and 32-channel stages (incl. their residual merges) run on the register-operand
            # MFMA kernel, the 64-channel stage on the LDS-weights one
            # (round 4: with a usable fold the 16-channel stage runs on the LDS-staged form, "strip_i4_c16_lds")
            assert m.kernel_log.count("strip_i4_c16") + m.kernel_log.count("strip_i4_c16_lds") >= 2 * nres, m.kernel_log
            # (the first 32-channel block starts with a stride-2 conv; its second conv merges the
            # float32 projection shortcut, which the kernel reads directly)
            assert m.kernel_log.count("strip_i4_c32") + m.kernel_log.count("strip_i4_c32_proj") >= 2 * nres - 1, m.kernel_log
            # 64-channel stage: the strip kernel too (round 3), with and without a residual merge
            assert m.kernel_log.count("strip_i4_c64") + m.kernel_log.count("strip_i4_c64_proj") >= 2 * nres - 1, m.kernel_log


def test_residual_fused_model_at_imagenet_geometry():
    """224 / 112 / 56 wide stages (14 / 7 / 3.5 sixteen-pixel segments per row, tiles straddling rows and
    images): every matrix-pipe kernel of the residual engine against the oracle, bit for bit."""
    base = nets.baseline_config(4)
    cf = nets.Config(network_type=base.network_type, wbits=base.wbits, abits=base.abits, architecture="RESNET",
                     nres=2, dim=base.dim, channels=base.channels, classes=base.classes)
    spec = nets.build_spec(cf, 11)[:-1]                     # logits
    x = nets.synthetic_images(cf, 2, 12)
    want = O.run_spec(spec, x, float_conv="device")
    m = engine.ResidualFusedModel(spec, first_layer="exact")
    m.kernel_log = []
    got = host(m(dev(x)))
    np.testing.assert_array_equal(got, want)
    for k in ("strip_i4_c16", "strip_i4_c32", "strip_i4_c64"):
        assert k in m.kernel_log or k + "_lds" in m.kernel_log, m.kernel_log
    _abi.set_option("strip64", 0)           # the LDS-weight kernel for the plain 64-channel layers: same logits
    try:
        m3 = engine.ResidualFusedModel(spec, first_layer="exact")
        m3.kernel_log = []
        np.testing.assert_array_equal(host(m3(dev(x))), want)
        assert "mfma_i4_areg64x64" in m3.kernel_log, m3.kernel_log
    finally:
        _abi.set_option("strip64", -1)
    # the tile kernel (strip switch off) gives the same logits
    _abi.set_option("strip", 0)
    try:
        m2 = engine.ResidualFusedModel(spec, first_layer="exact")
        m2.kernel_log = []
        np.testing.assert_array_equal(host(m2(dev(x))), want)
        assert "mfma_i4_small_c16" in m2.kernel_log and not any(k.startswith("strip_i4_c16") for k in m2.kernel_log)
    finally:
        _abi.set_option("strip", 1)


def test_config5_imagenet224_resnet_nres10_at_spec():
    """BASELINE config 5 as stated: ImageNet-224 ResNet, nres = 10 (63 convolutions, 62-layer deep chain of
    fused residual launches), full-qnn 4/4 -- logits of one image bit for bit against the oracle."""
    cf = nets.baseline_config(4)
    assert (cf.nres, cf.dim, cf.wbits, cf.abits) == (10, 224, 4, 4)
    spec = nets.build_spec(cf, nets.SEED_BASE + 4)
    assert sum(op["op"] == "conv" for op in spec) == 63
    x = nets.synthetic_images(cf, 1, 5)
    want = O.run_spec(spec[:-1], x, float_conv="device")
    m = engine.ResidualFusedModel(spec[:-1], first_layer="exact")
    m.kernel_log = []
    got = host(m(dev(x)))
    np.testing.assert_array_equal(got, want)
    # 61 launches: the two projection shortcuts are computed inside the launches of their blocks' second convolutions
    assert len(m.kernel_log) == 61 and m.kernel_log.count("strip_i4_c32_proj") == 1 and \
        m.kernel_log.count("strip_i4_c64_proj") == 1 and "generic" not in m.kernel_log, m.kernel_log
    # the softmax output through the general interpreter agrees too
    probs = host(engine.GraphModel(spec)(dev(x)))
    np.testing.assert_allclose(probs, O.softmax(want), atol=1e-6)


@pytest.mark.parametrize("store,bits,shape,size", [(_abi.STORE_I4, 4, (3, 56, 56, 64), 8), (_abi.STORE_I4, 2, (2, 9, 10, 24), 3),
                                                   (_abi.STORE_I8, 8, (2, 16, 16, 12), 8), (_abi.STORE_BIN, 1, (2, 8, 8, 64), 8),
                                                   (_abi.STORE_I4, 4, (1, 8, 8, 16), 8), (_abi.STORE_I4, 3, (5, 12, 20, 128), 4),
                                                   (_abi.STORE_I4, 4, (2, 9, 9, 64), 3)])
def test_average_pool_on_packed_codes(store, bits, shape, size):
    """AveragePooling2D behind the last activation (resnet.py:134) reads the packed codes: exact window sums."""
    rng = np.random.default_rng(bits + shape[1])
    N, H, W, C = shape
    act = BIN_ACT if bits == 1 else Q(bits)
    x = O.run_spec([act], rng.standard_normal(shape).astype(F32))
    fn, b = engine._act_code(act)
    p = _abi.pack(dev(x), C, _abi.FN_GRID, b, store)
    got = host(_abi.avgpool_packed(p, store, b, N, H, W, C, size))
    np.testing.assert_array_equal(got, O.avgpool2d(x, size))


@pytest.mark.parametrize("code,wb,ab", [("44", 4, 4), ("bb", None, None)])
def test_residual_fused_model_on_trained_checkpoint(code, wb, ab):
    spec = nets.spec_from_keras_npz(os.path.join(GOLD, "resnet3_full_%s.npz" % code), wb, ab)
    x = nets.synthetic_images(nets.Config(dim=32), 4, 12)
    env = O.run_spec(spec, x, float_conv="device", return_all=True)
    logits_name = [op["dst"] for op in spec if op["op"] == "dense"][0]
    np.testing.assert_array_equal(host(engine.ResidualFusedModel(spec[:-1], first_layer="exact")(dev(x))), env[logits_name])


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_residual_fused_model_runs_vgg_too(idx):
    cf = nets.baseline_config(idx)
    spec = nets.build_spec(cf, nets.SEED_BASE + idx)
    x = nets.synthetic_images(cf, 4, 3)
    np.testing.assert_array_equal(host(engine.ResidualFusedModel(spec, first_layer="exact")(dev(x))),
                                  O.run_spec(spec, x, float_conv="device"))


@pytest.mark.parametrize("in_act", [BIN_ACT, Q(4), Q(8), None])
def test_ternary_layers(in_act):
    """TernaryConv2D / TernaryDense (ternary_layers.py:77-84,156-174): weights ternarised with the
    global 0.7*mean|W| cutoff at prepack time."""
    rng = np.random.default_rng(21)
    pre = rng.standard_normal((2, 9, 9, 64)).astype(F32)
    x = O.run_spec([in_act], pre) if in_act is not None else pre
    k = rng.uniform(-1, 1, (3, 3, 64, 32)).astype(F32)
    b = (rng.standard_normal(32) * 0.05).astype(F32)
    layer = qnn_amd.TernaryConv2D(32, kernel_size=(3, 3), padding="same")
    layer.build((None, 9, 9, 64))
    layer.set_weights([k, b])
    np.testing.assert_array_equal(host(layer.quantized_kernel()), O._ternarize(k))
    np.testing.assert_array_equal(O._ternarize(k), O.ternarize(k))       # straight-through form is exact here
    want = O.run_spec([{"op": "conv", "kind": "ternary", "kernel": k, "bias": b}], x)
    if in_act is not None:
        layer.input_domain = "binary" if in_act is BIN_ACT else ("quantized", in_act["nb"])
        np.testing.assert_array_equal(host(layer(dev(x))), want)
    else:
        # unit-variance float inputs, K = 576: the float32 FMA chain (what any float32 convolution does, TF's
        # included) is 1.13e-5 * max(1, |y|) away from the float64-accumulated ideal here (measured): 2x band
        got = host(layer(dev(x)))
        within(got, want, 2.0)
    dk = rng.uniform(-1, 1, (64, 10)).astype(F32)
    d = qnn_amd.TernaryDense(10)
    d.build((None, 64))
    d.set_weights([dk, np.zeros(10, F32)])
    if in_act is not None:
        d.input_domain = layer.input_domain
        xv = x[:, 0, 0, :]
        np.testing.assert_array_equal(host(d(dev(xv))), O.run_spec([{"op": "dense", "kind": "ternary", "kernel": dk,
                                                                   "bias": np.zeros(10, F32)}], xv))


@pytest.mark.parametrize("nt", ["tnn", "qtnn", "full-tnn"])
@pytest.mark.parametrize("arch", ["VGG", "RESNET"])
def test_ternary_networks(nt, arch):
    """model_factory.py:49-58: ternary weights with LeakyReLU / quantized / ternary activations."""
    cf = nets.Config(network_type=nt, wbits=4, abits=4, architecture=arch, nres=1, dim=32)
    spec = nets.build_spec(cf, 31)
    x = nets.synthetic_images(cf, 3, 31)
    want = O.run_spec(spec, x, float_conv="device")
    tol = 2e-5 if nt == "tnn" else 1e-6
    for cls in (engine.GraphModel, engine.ResidualFusedModel, engine.LayerModel):
        got = host(cls(spec)(dev(x)))
        np.testing.assert_allclose(got, want, atol=tol, err_msg=cls.__name__)
    if nt == "qtnn" and arch == "VGG":
        np.testing.assert_array_equal(host(engine.FusedModel(spec, first_layer="exact")(dev(x))), want)


def test_build_model_predict_evaluate():
    cf = nets.baseline_config(2)
    model = nets.build_model(cf, nets.SEED_BASE + 2, first_layer="exact")
    assert type(model.engine).__name__ == "FusedModel" and model.count_params() > 80000
    x = nets.synthetic_images(cf, 10, 4)
    p = model.predict(x, batch_size=4)
    np.testing.assert_array_equal(p, O.run_spec(model.spec, x, float_conv="device"))
    acc = model.evaluate(x, np.eye(10, dtype=F32)[p.argmax(-1)])
    assert acc == 1.0
    lines = []
    model.summary(lines.append)
    assert "Total params" in lines[-1]
    rmodel = nets.build_model(nets.Config(architecture="RESNET", nres=1), 3, first_layer="exact")
    assert type(rmodel.engine).__name__ == "ResidualFusedModel"


def test_mnist_resnet_zero_padding():
    cf = nets.Config(network_type="full-bnn", architecture="RESNET", dataset="MNIST", dim=28,
                     channels=1, nres=1)
    spec = nets.build_spec(cf, 7)
    x = nets.synthetic_images(cf, 2, 7)
    want = O.run_spec(spec, x, float_conv="device")
    got = host(engine.GraphModel(spec)(dev(x)))
    np.testing.assert_allclose(got, want, atol=1e-6)


# ---------------------------------------------------------------------------
# full-size (batch 4096) size-independent properties
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("idx", [1, 2, 3])
def test_full_batch_properties(idx, impl):
    if idx == 3 and impl == _abi.IMPL_VALU:
        pytest.skip("VGG-large at batch 4096 on the VALU kernels alone takes minutes; covered at N=2")
    cf = nets.baseline_config(idx)
    spec = nets.build_spec(cf, nets.SEED_BASE + idx)
    fused = engine.FusedModel(spec, first_layer="exact")
    N = 4096
    x = dev(nets.synthetic_images(cf, N, 99))
    y = fused(x)
    # batch independence: any sub-batch gives the same rows
    perm = torch.randperm(N, device="cuda")
    y_perm = fused(x[perm].contiguous())
    assert torch.equal(y_perm, y[perm])
    assert torch.equal(fused(x[100:137].contiguous()), y[100:137])
    # the three engines agree bit for bit at full size
    assert torch.equal(engine.LayerModel(spec)(x[:512 if idx < 3 else 64].contiguous()), y[:512 if idx < 3 else 64])
    # and the head of the batch equals the oracle
    want = O.run_spec(spec, host(x[:4]), float_conv="device")
    np.testing.assert_array_equal(host(y[:4]), want)
    assert bool(torch.isfinite(y).all())


def test_full_batch_properties_config5_residual_engine():
    """Config 5 at its per-GPU batch (64 images of 224x224, nres = 10): batch independence of the fused residual
    engine (any sub-batch and any permutation give the same rows), finite outputs, head of the batch == oracle."""
    cf = nets.baseline_config(4)
    spec = nets.build_spec(cf, nets.SEED_BASE + 4)[:-1]           # logits
    m = engine.ResidualFusedModel(spec, first_layer="exact")
    N = 64
    x = dev(nets.synthetic_images(cf, N, 98))
    y = m(x)
    perm = torch.randperm(N, device="cuda")
    assert torch.equal(m(x[perm].contiguous()), y[perm])
    assert torch.equal(m(x[10:13].contiguous()), y[10:13])
    assert bool(torch.isfinite(y).all())
    want = O.run_spec(spec, host(x[:1]), float_conv="device")
    np.testing.assert_array_equal(host(y[:1]), want)
