"""k_conv_mfma_halo (csrc/qnn_mfma_areg.hip): the pooled int4 64-channel layers with the pixel operand staged ONCE per tile
through a wave-private LDS region (QuantizedConv2D / BinaryConv2D.call + BN + clip + MaxPooling2D, models/vgg.py:21-36).

Bit-exact against the oracle and against the kernel it replaces (qnn_set_option("halo", 0) -> k_conv_mfma_areg), over the
tilings it accepts (8 x 2 and 4 x 4 pooled rectangles, several tiles per image row / column), edge cases
of the zero padding, negative BN scales (min-pooling), bias, binary_tanh, odd batch sizes; shapes it does not tile keep the
old kernel.  The fused conv + classifier entry (qnn_conv2d_dense_forward) runs on it as well.
"""
import zlib

import numpy as np
import pytest
import torch

import qnn_amd  # noqa: F401
from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O
from test_gpu_parity import BIN_ACT, Q, _oracle_group, _rand_bn, _run_group, dev, host

pytestmark = pytest.mark.gpu
F32 = np.float32


def _case(name, shape, cout, kind="quantized", nb=4, abits=4, bias=False):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    m = 2 ** (abits - 1)
    x = (rng.integers(-m, m, shape) / m).astype(F32)
    op = {"op": "conv", "kind": kind, "kernel": rng.uniform(-1, 1, (3, 3, shape[3], cout)).astype(F32),
          "strides": (1, 1), "padding": "same"}
    if kind == "quantized":
        op["nb"] = nb
    if bias:
        op["bias"] = (rng.standard_normal(cout) * 0.5).astype(F32)
    return rng, x, op


#        name          input shape        cout  expected kernel
CASES = [("b0",        (5, 16, 16, 64),   64,  "mfma_i4_halo64x64"),     # 8 x 2 rectangles, four per image
         ("c0",        (7, 8, 8, 64),     64,  "mfma_i4_halo64x64"),     # 4 x 4: tile == image
         ("wide",      (2, 8, 32, 64),    64,  "mfma_i4_halo64x64"),     # two tiles per row: interior left / right halos
         ("big",       (1, 32, 32, 64),   64,  "mfma_i4_halo64x64"),     # 2 x 8 tiles per image
         ("cout128",   (2, 16, 16, 64),   128, "mfma_i4_256x128"),       # more than one filter slice: the tile kernel keeps it
         ("w24",       (3, 8, 24, 64),    64,  "mfma_i4_halo64x64"),     # pooled width 12 -> 4 x 4 rectangles, three per row
         ("tall",      (2, 24, 8, 64),    64,  "mfma_i4_halo64x64"),     # 4 x 4, three per column
         ("one",       (1, 4, 16, 64),    64,  "mfma_i4_halo64x64"),     # a single 8 x 2 tile
         ("many",      (300, 8, 8, 64),   64,  "mfma_i4_halo64x64"),     # more tiles than one wave per tile
         ("w12",       (2, 12, 12, 64),   64,  "mfma_i4_areg64x64"),     # pooled 6 x 6: no tiling -> old kernel
         ("h6",        (2, 6, 16, 64),    64,  "mfma_i4_areg64x64")]     # pooled height 3


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_halo_kernel_bit_exact_vs_oracle_and_vs_the_kernel_it_replaces(case):
    name, shape, cout, kern_want = case
    rng, x, op = _case(name, shape, cout, bias=name in ("wide", "c0"))
    bn = _rand_bn(rng, cout, 9 * 64 * 0.12)                       # gamma in [-1.5, 1.5]: min- and max-pooled channels
    for act in (Q(4), Q(2), BIN_ACT):
        got, kern = _run_group(x, Q(4), op, bn, act, 2, _abi.STORE_I4)
        assert kern == kern_want, (kern, kern_want)
        np.testing.assert_array_equal(got, _oracle_group(x, op, bn, act, 2))
        _abi.set_option("halo", 0)
        try:
            old, kern_old = _run_group(x, Q(4), op, bn, act, 2, _abi.STORE_I4)
        finally:
            _abi.set_option("halo", 1)
        assert "halo" not in kern_old
        np.testing.assert_array_equal(got, old)


def test_halo_kernel_all_positive_scales_and_extreme_codes():
    """gamma > 0 everywhere (the max-only pooling path) and inputs / weights pinned at the ends of their grids: the
    largest sums the int32 accumulators see (9 * 64 * 8 * 8 * 256 with both operands carrying * 16)."""
    rng, x, op = _case("pos", (3, 16, 16, 64), 64)
    x[0] = -1.0
    x[1] = 0.875
    op["kernel"][..., :8] = -1.0
    op["kernel"][..., 8:16] = 1.0
    bn = _rand_bn(rng, 64, 9 * 64 * 0.12)
    bn["gamma"] = np.abs(bn["gamma"]) + F32(0.05)
    got, kern = _run_group(x, Q(4), op, bn, Q(4), 2, _abi.STORE_I4)
    assert kern == "mfma_i4_halo64x64"
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, Q(4), 2))
    got, kern = _run_group(x, Q(4), op, None, Q(4), 2, _abi.STORE_I4)          # no BN at all
    assert kern == "mfma_i4_halo64x64"
    np.testing.assert_array_equal(got, _oracle_group(x, op, None, Q(4), 2))


def test_halo_kernel_padding_is_zero_not_neighbouring_rows():
    """An image whose border pixels are the only non-zero ones, next to images of all ones: a halo lane that read the
    neighbouring row / image instead of zero padding would change the border sums."""
    rng, x, op = _case("edge", (4, 16, 16, 64), 64)
    x[:] = 0.0
    x[1, 0, :, :] = 0.875
    x[1, -1, :, :] = -1.0
    x[1, :, 0, :] = 0.5
    x[1, :, -1, :] = -0.5
    x[0] = 0.875
    x[2] = -1.0
    got, kern = _run_group(x, Q(4), op, None, Q(4), 2, _abi.STORE_I4)
    assert kern == "mfma_i4_halo64x64"
    np.testing.assert_array_equal(got, _oracle_group(x, op, None, Q(4), 2))


def test_binary_network_layers_on_int4_codes_use_the_halo_kernel_too():
    rng, x, op = _case("bin", (4, 16, 16, 64), 64, kind="binary", abits=1)
    x = np.where(rng.random(x.shape) < 0.5, -1.0, 1.0).astype(F32)
    bn = _rand_bn(rng, 64, 9 * 64)
    got, kern = _run_group(x, BIN_ACT, op, bn, BIN_ACT, 2, _abi.STORE_I4)
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, BIN_ACT, 2))


@pytest.mark.parametrize("seed", range(10))
def test_halo_kernel_random_tilings(seed):
    """Random batch sizes, image sizes among those the kernel tiles, activation widths, bias on / off, weight kinds."""
    rng = np.random.default_rng(1000 + seed)
    H, W = [(4, 16), (8, 8), (8, 16), (16, 16), (12, 16), (16, 32), (8, 24), (16, 8), (20, 16), (24, 24)][seed]
    N = int(rng.integers(1, 40))
    kind = ["quantized", "binary"][seed % 2]
    abits = 4 if kind == "quantized" else 1
    _, x, op = _case("rnd%d" % seed, (N, H, W, 64), 64, kind=kind, nb=int(rng.integers(2, 5)), abits=abits, bias=bool(seed & 2))
    if kind == "binary":
        x = np.where(rng.random(x.shape) < 0.5, -1.0, 1.0).astype(F32)
    bn = _rand_bn(rng, 64, 9 * 64 * (0.12 if kind == "quantized" else 1.0))
    act = [Q(4), Q(3), Q(2), BIN_ACT][seed % 4]
    in_act = Q(4) if kind == "quantized" else BIN_ACT
    got, kern = _run_group(x, in_act, op, bn, act, 2, _abi.STORE_I4)
    if kind == "quantized":
        assert kern == "mfma_i4_halo64x64", kern
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, act, 2))


@pytest.mark.parametrize("units", [10, 16, 3])
def test_fused_conv_classifier_on_the_halo_kernel(units):
    rng, x, op = _case("head%d" % units, (9, 8, 8, 64), 64)
    bn = _rand_bn(rng, 64, 9 * 64 * 0.12)
    dop = {"op": "dense", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (1024, units)).astype(F32),
           "bias": (rng.standard_normal(units) * 0.1).astype(F32)}
    dbn = _rand_bn(rng, units, 1024 * 0.1)
    want = O.run_spec([dict(op), bn, Q(4), {"op": "maxpool", "size": 2}, {"op": "flatten"}, dict(dop), dbn], x)
    wc = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"))
    wd = engine._prepack(dop, _abi.STORE_I4, torch.device("cuda"))
    xp = _abi.pack(dev(x), 64, _abi.FN_GRID, 4, _abi.STORE_I4)
    inv, shift = (dev(a) for a in engine.bn_constants(bn))
    dinv, dshift = (dev(a) for a in engine.bn_constants(dbn))
    outs = {}
    for halo in (1, 0):
        _abi.set_option("halo", halo)
        try:
            y = _abi.conv2d_dense(wc, wd, xp, _abi.STORE_I4, 4, 9, 8, 8, inv, shift, _abi.FN_QUANTIZED_TANH, 4, dinv, dshift)
        finally:
            _abi.set_option("halo", 1)
        outs[halo] = host(y)
    np.testing.assert_array_equal(outs[1], outs[0])
    np.testing.assert_array_equal(outs[1], want)


def test_headline_network_same_logits_with_and_without_the_halo_kernel():
    cf = nets.baseline_config(2)
    spec = nets.build_spec(cf, nets.SEED_BASE + 2)
    x = nets.synthetic_images(cf, 96, 5)
    outs = {}
    for halo in (1, 0):
        _abi.set_option("halo", halo)
        try:
            m = engine.FusedModel(spec, first_layer="exact")
            m.kernel_log = []
            outs[halo] = host(m(dev(x)))
            used = any("halo" in k for k in m.kernel_log)
            assert used == bool(halo), m.kernel_log
        finally:
            _abi.set_option("halo", 1)
    np.testing.assert_array_equal(outs[1], outs[0])
    np.testing.assert_array_equal(outs[1], O.run_spec(spec, x, float_conv="device"))
