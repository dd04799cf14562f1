"""Known-answer tables that pin the oracle's elementwise semantics.

The reference holds no golden vectors, so these tables are
derived by hand from the formulas in layers/binary_ops.py:16-64,
layers/quantized_ops.py:49-66,87-100 and layers/ternary_ops.py:15-54 together
with tf.round = round-half-to-even (SURVEY.md section 7 hard part 3, appendix A.2).
"""
import numpy as np
import pytest

from oracle import qnn_oracle as O

F32 = np.float32


def test_binary_tanh_threshold_table():
    x = np.array([-2, -1, -.5, -1e-9, 0, 2.0 ** -24, 2.0 ** -23, 1e-7, .5, 1, 2], dtype=F32)
    want = np.array([-1, -1, -1, -1, -1, -1, +1, +1, +1, +1, +1], dtype=F32)
    np.testing.assert_array_equal(O.binary_tanh(x), want)
    np.testing.assert_array_equal(O.binarize(x), want)


def test_binary_tanh_is_gt_2pow_minus24():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(20000).astype(F32),
        (rng.standard_normal(20000) * 1e-7).astype(F32),
        np.array([np.nextafter(F32(2.0 ** -24), F32(1)), np.nextafter(F32(2.0 ** -24), F32(0)),
                  F32(np.inf), F32(-np.inf), F32(1e-45), F32(-0.0)], dtype=F32)])
    want = np.where(x > F32(2.0 ** -24), F32(1), F32(-1))
    np.testing.assert_array_equal(O.binary_tanh(x), want)


def test_quantized_tanh_nb4_table():
    x = np.array([-1.2, -1, -.9375, -.3125, -.1875, -.0625, 0, .0625, .1875, .3125, .9375, 1], dtype=F32)
    want8 = np.array([-8, -8, -8, -2, -2, 0, 0, 0, 2, 2, 7, 7], dtype=F32)
    np.testing.assert_array_equal(O.quantized_tanh(x, 4) * F32(8), want8)
    np.testing.assert_array_equal(O.quantize(x, 4) * F32(8), want8)


def test_quantize_nb2_table():
    x = np.array([-1.2, -1, -.9375, -.3125, -.1875, -.0625, 0, .0625, .1875, .3125, .9375, 1], dtype=F32)
    want2 = np.array([-2, -2, -2, -1, 0, 0, 0, 0, 0, 1, 1, 1], dtype=F32)
    np.testing.assert_array_equal(O.quantize(x, 2) * F32(2), want2)


@pytest.mark.parametrize("nb", [2, 3, 4, 8, 16])
def test_quantize_grid_and_range(nb):
    rng = np.random.default_rng(nb)
    x = rng.uniform(-1.5, 1.5, 50000).astype(F32)
    m = 2 ** (nb - 1)
    q = O.quantize(x, nb)
    k = q.astype(np.float64) * m
    assert np.array_equal(k, np.rint(k))
    assert k.min() == -m and k.max() == m - 1
    # idempotent on the grid
    np.testing.assert_array_equal(O.quantize(q, nb), q)
    # equals the closed form clip(rint(x*m))
    want = np.clip(np.rint(x * F32(m)), -m, m - 1) / F32(m)
    np.testing.assert_array_equal(q, want.astype(F32))


def test_round_through_is_rint():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-300, 300, 100000), np.arange(-20, 20) + 0.5]).astype(F32)
    np.testing.assert_array_equal(O.round_through(x), np.rint(x))


def test_ternary_tanh_table():
    x = np.array([-1.2, -1, -.9375, -.3125, -.1875, -.0625, 0, .0625, .1875, .3125, .9375, 1], dtype=F32)
    want = np.array([-1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 1, 1], dtype=F32)
    np.testing.assert_array_equal(O.ternary_tanh(x), want)


def test_same_padding_cases():
    # (in, k, s) -> (out, before, after); SURVEY.md 8c known answers
    assert O.same_padding(32, 3, 1) == (32, 1, 1)
    assert O.same_padding(32, 3, 2) == (16, 0, 1)
    assert O.same_padding(32, 1, 2) == (16, 0, 0)
    assert O.same_padding(28, 3, 1) == (28, 1, 1)
    assert O.same_padding(224, 3, 2) == (112, 0, 1)
    assert O.same_padding(7, 3, 2) == (4, 1, 1)


def test_maxpool_valid_7_to_3():
    x = np.arange(7 * 7, dtype=F32).reshape(1, 7, 7, 1)
    y = O.maxpool2d(x, 2)
    assert y.shape == (1, 3, 3, 1)
    assert y[0, 0, 0, 0] == 8 and y[0, 2, 2, 0] == 40


def test_glorot_klm_values():
    # binary_layers.py:129-132; values listed in SURVEY.md 8(a6)/appendix A.1
    assert O.glorot_klm(3, 3, 3, 64) == F32(20.0499382)
    assert O.glorot_klm(3, 3, 64, 64) == F32(27.7128124)
    assert O.glorot_klm(3, 3, 256, 256) == F32(55.4256248)
    assert O.glorot_klm(3, 3, 16, 16) == F32(13.8564062)
    assert O.glorot_klm(1, 1, 32, 64) == F32(8.0)


def test_conv2d_matches_direct_loops():
    rng = np.random.default_rng(3)
    x = rng.integers(-8, 8, (2, 7, 6, 5)).astype(F32)
    w = rng.integers(-8, 8, (3, 3, 5, 4)).astype(F32)
    for s in (1, 2):
        y = O.conv2d(x, w, (s, s), "same")
        Ho, pt, _ = O.same_padding(7, 3, s)
        Wo, pl, _ = O.same_padding(6, 3, s)
        want = np.zeros((2, Ho, Wo, 4), dtype=np.float64)
        for n in range(2):
            for oy in range(Ho):
                for ox in range(Wo):
                    for dy in range(3):
                        for dx in range(3):
                            iy, ix = oy * s + dy - pt, ox * s + dx - pl
                            if 0 <= iy < 7 and 0 <= ix < 6:
                                want[n, oy, ox] += x[n, iy, ix].astype(np.float64) @ w[dy, dx].astype(np.float64)
        np.testing.assert_array_equal(y, want.astype(F32))
        np.testing.assert_array_equal(O.int_conv2d(x, w, (s, s)), want.astype(np.int64))


def test_trick_constants_and_noise():
    # appendix A.1: the conv sees 0.99999994 for (64,64) under legacy promotion
    klm = O.glorot_klm(3, 3, 64, 64)
    c_in, s_in, c_out, s_out = O.trick_constants(klm, "legacy")
    one = O._trick(np.ones(1, F32), c_in, s_in)[0]
    assert one == F32(0.99999994) or one == F32(1.0)
    o = np.arange(-600, 601, dtype=F32)
    err = np.abs(O._trick(o, c_out, s_out).astype(np.float64) - o).max()
    assert err < 5e-3  # reference's own rounding noise, ~1e-3 at |o|~600
    # power-of-two multiplier is exact
    c_in, s_in, c_out, s_out = O.trick_constants(O.glorot_klm(1, 1, 32, 64), "legacy")
    np.testing.assert_array_equal(O._trick(o, c_out, s_out), o)


def test_binary_layer_outputs_are_integers_and_faithful_is_close():
    rng = np.random.default_rng(5)
    x = O.binary_tanh(rng.standard_normal((2, 8, 8, 64)).astype(F32))
    k = rng.uniform(-1, 1, (3, 3, 64, 64)).astype(F32)
    b = (rng.standard_normal(64) * 0.05).astype(F32)
    y = O.binary_conv2d_call(x, k, None, mode="exact")
    assert np.array_equal(y, np.rint(y)) and np.abs(y).max() <= 576
    yi = O.int_conv2d(O.signs_of(x), O.signs_of(O.binarize(k)))
    np.testing.assert_array_equal(y.astype(np.int64), yi)
    for promo in ("legacy", "nep50"):
        yf = O.binary_conv2d_call(x, k, b, mode="faithful", promotion=promo)
        ye = O.binary_conv2d_call(x, k, b, mode="exact")
        assert np.abs(yf.astype(np.float64) - ye).max() <= 1e-5 * max(1.0, np.abs(ye).max())


def test_quantized_layer_exact_on_grid():
    rng = np.random.default_rng(6)
    x = O.quantized_tanh(rng.standard_normal((2, 8, 8, 32)).astype(F32), 4)
    k = rng.uniform(-1, 1, (3, 3, 32, 16)).astype(F32)
    y = O.quantized_conv2d_call(x, k, None, nb=4, mode="exact")
    yi = O.int_conv2d(O.codes_of(x, 4), O.codes_of(O.quantize(k, 4), 4))
    np.testing.assert_array_equal(y.astype(np.float64) * 64.0, yi.astype(np.float64))


def test_bn_formula_two_roundings():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((4, 3, 3, 8)).astype(F32) * 30
    g, b, m = (rng.standard_normal(8).astype(F32) for _ in range(3))
    v = rng.uniform(0.5, 900, 8).astype(F32)
    inv, shift = O.bn_constants(g, b, m, v, 1e-4)
    y = O.batchnorm_inference(x, g, b, m, v, 1e-4)
    want = (x * inv).astype(F32) + shift
    np.testing.assert_array_equal(y, want)
    ref = (x.astype(np.float64) - m) / np.sqrt(v.astype(np.float64) + 1e-4) * g + b
    assert np.abs(y - ref).max() < 1e-4
