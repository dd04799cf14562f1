"""The OPT-IN fixed-point first layer (qnn_set_option("first_fixed", 1); csrc/qnn_first_fixed.hip).

It is deliberately NOT the oracle's float32 FMA chain, so its checks are different from every other kernel's:

 * against its own specification -- inputs rounded to 2^-23, exact integer sums, one rounding -- restated here in int64
   numpy: bit-exact, raw float32 output and packed codes alike;
 * against the IDEAL (float64) convolution: within the north star's 1e-5 (the bound is 27 * 2^-24 + half an ulp);
 * against the oracle's codes: the measured fraction of activation codes that differ (values whose pre-activation sits
   within ~1e-6 of a rounding threshold), each by one code step, is printed and bounded.
The default (exact) kernel is untouched: the last test checks the switch restores it.
"""
import zlib

import numpy as np
import pytest
import torch

import qnn_amd  # noqa: F401
from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O
from test_gpu_parity import BIN_ACT, Q, _oracle_group, _rand_bn, _run_group, host

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture
def fixed_first_layer():
    _abi.set_option("first_fixed", 1)
    try:
        yield
    finally:
        _abi.set_option("first_fixed", 0)


def _fixed_conv(x, op):
    """The kernel's specification: X = clip(rint(x * 2^23), 0, 2^23); integer weight codes; exact sum; one rounding."""
    if op["kind"] == "binary":
        wq, wshift = O.binarize(op["kernel"]), 0
    else:
        wq, wshift = O.quantize(op["kernel"], op["nb"]), op["nb"] - 1
    codes = np.rint(wq.astype(np.float64) * 2.0 ** wshift).astype(np.int64)
    assert np.array_equal(codes.astype(np.float64) * 2.0 ** -wshift, wq.astype(np.float64))
    X = np.clip(np.rint((x * F32(8388608.0)).astype(F32)).astype(np.int64), 0, 8388608)
    T = O.int_conv2d(X, codes)
    assert np.abs(T).max() < 2 ** 31
    v = T.astype(F32) * F32(2.0 ** -(23 + wshift))            # int32 -> float32: round to nearest even; the scale is exact
    if op.get("bias") is not None:
        v = O.bias_add(v, op["bias"])
    return v


def _tail(v, bn, act, pool):
    spec = []
    if bn is not None:
        spec.append(bn)
    if act is not None:
        spec.append(act)
    if pool == 2:
        spec.append({"op": "maxpool", "size": 2})
    return O.run_spec(spec, v) if spec else v


def _case(name, shape, kind, nb, bias=True):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = (rng.integers(0, 256, shape).astype(F32) / F32(255)).astype(F32)
    op = {"op": "conv", "kind": kind, "kernel": rng.uniform(-1, 1, (3, 3, 3, 64)).astype(F32),
          "strides": (1, 1), "padding": "same"}
    if bias:
        op["bias"] = (rng.standard_normal(64) * 0.05).astype(F32)
    if nb:
        op["nb"] = nb
    return rng, x, op


CASES = [("q4_32", (3, 32, 32, 3), "quantized", 4), ("q2_16x48", (2, 16, 48, 3), "quantized", 2),
         ("q3_nobias", (1, 8, 16, 3), "quantized", 3), ("bin_32", (2, 32, 32, 3), "binary", None),
         ("q4_tall", (5, 66, 16, 3), "quantized", 4), ("q4_many", (300, 4, 16, 3), "quantized", 4)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_raw_output_is_the_specified_fixed_point_sum_and_within_1e5_of_ideal(case, fixed_first_layer):
    name, shape, kind, nb = case
    rng, x, op = _case(name, shape, kind, nb, bias="nobias" not in name)
    got, kern = _run_group(x, None, op, None, None, 1, _abi.STORE_F32)
    assert kern == "mfma_i8x3_first_fixed"
    np.testing.assert_array_equal(got, _fixed_conv(x, op))
    ideal = _oracle_group(x, op, None, None, 1)                 # float64 accumulation
    err = np.abs(got.astype(np.float64) - ideal)
    assert np.all(err <= 1e-5 * np.maximum(1.0, np.abs(ideal)))     # north_star tolerance
    # the analytic bound: 27 taps * |w| <= 1 * 2^-24 input rounding + half an ulp of the sum (+ one for the bias)
    assert err.max() <= 27 * 2.0 ** -24 + 2 * np.spacing(F32(np.abs(ideal).max()))
    print("\n[first_fixed] %s: max |y - ideal| = %.3g" % (name, err.max()))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("act", [Q(4), Q(2), BIN_ACT], ids=["a4", "a2", "abin"])
def test_fused_codes_match_the_specification_and_rarely_differ_from_the_oracle(case, act, fixed_first_layer):
    name, shape, kind, nb = case
    rng, x, op = _case(name, shape, kind, nb, bias="nobias" not in name)
    bn = _rand_bn(rng, 64, 27 * 0.3)
    got, kern = _run_group(x, None, op, bn, act, 2, _abi.STORE_I4)
    assert kern == "mfma_i8x3_first_fixed"
    np.testing.assert_array_equal(got, _tail(_fixed_conv(x, op), bn, act, 2))       # its own specification: bit-exact
    want = _oracle_group(x, op, bn, act, 2, float_conv="device")                     # the default kernel's result
    diff = got != want
    step = 2.0 if act is BIN_ACT else 2.0 ** -(act["nb"] - 1)
    assert np.all(np.abs(got - want)[diff] == step)                                   # never more than one code step
    rate = diff.mean()
    print("\n[first_fixed] %s/%s: %d of %d codes differ from the float32-chain result (%.2e)"
          % (name, act.get("nb", "bin"), diff.sum(), diff.size, rate))
    assert rate <= 2e-3


def test_inputs_outside_the_unit_interval_saturate(fixed_first_layer):
    """The kernel's domain is images / 255; values outside [0, 1] are clamped (the specification's np.clip), so the
    digits never wrap -- AND the layer's domain flag is raised (tests/test_gpu_u8.py,
    test_fixed_point_first_layer_reports_inputs_outside_its_domain): the clamped result is not handed out silently."""
    rng, x, op = _case("q4_32", (2, 16, 32, 3), "quantized", 4)
    x = x.copy()
    x[0, :4] = 1.5
    x[0, 4:8] = -0.25
    x[1, 0, 0] = [0.0, 1.0, 1.0 - 2.0 ** -24]
    got, kern = _run_group(x, None, op, None, None, 1, _abi.STORE_F32)
    assert kern == "mfma_i8x3_first_fixed"
    np.testing.assert_array_equal(got, _fixed_conv(x, op))
    np.testing.assert_array_equal(got[0, 1:3], _fixed_conv(np.clip(x, 0, 1), op)[0, 1:3])


@pytest.mark.parametrize("shape", [(2, 30, 24, 3), (2, 15, 16, 3), (1, 16, 16, 1)], ids=["w24", "h15", "cin1"])
def test_shapes_outside_the_domain_keep_the_exact_kernel(shape, fixed_first_layer):
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 256, shape).astype(F32) / F32(255)).astype(F32)
    op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (3, 3, shape[3], 64)).astype(F32),
          "strides": (1, 1), "padding": "same"}
    bn = _rand_bn(rng, 64, 9 * shape[3] * 0.3)
    got, kern = _run_group(x, None, op, bn, Q(4), 2, _abi.STORE_I4)
    assert "fixed" not in kern
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, Q(4), 2, float_conv="device"))


def test_headline_network_with_the_fixed_point_first_layer(fixed_first_layer):
    """BASELINE configs[2] (VGG-64 4/4, the headline workload) at batch 256: logits with the fixed-point first layer against the oracle's."""
    cf = nets.baseline_config(2)
    spec = nets.build_spec(cf, nets.SEED_BASE + 2)
    rng = np.random.default_rng(77)
    x = (rng.integers(0, 256, (256, 32, 32, 3)).astype(F32) / F32(255)).astype(F32)
    model = engine.FusedModel(spec, first_layer="exact")
    model.kernel_log = []
    got = host(model(torch.from_numpy(x).cuda()))
    assert model.kernel_log[0] == "mfma_i8x3_first_fixed"
    want = O.run_spec(spec, x, float_conv="device")
    same_rows = np.all(got == want, axis=1).mean()
    agree = (got.argmax(1) == want.argmax(1)).mean()
    print("\n[first_fixed] VGG-64 4/4, 256 images: %.1f %% of the logit rows bit-identical to the oracle, "
          "argmax agreement %.2f %%, max |dlogit| %.3g" % (100 * same_rows, 100 * agree, np.abs(got - want).max()))
    # measured: every one of the 256 logit rows bit-identical.  One activation code in ~1.4 M sits close enough to a
    # rounding threshold to differ (test above), so a single differing row is tolerated; the decision never changes
    assert agree == 1.0 and same_rows >= 255.0 / 256.0, (agree, same_rows)


def test_the_switch_is_off_by_default_and_restores_the_exact_kernel():
    rng, x, op = _case("q4_32", (2, 32, 32, 3), "quantized", 4)
    bn = _rand_bn(rng, 64, 27 * 0.3)
    got, kern = _run_group(x, None, op, bn, Q(4), 2, _abi.STORE_I4)
    assert kern == "mfma_f32_first_cin3"
    np.testing.assert_array_equal(got, _oracle_group(x, op, bn, Q(4), 2, float_conv="device"))
    # shapes outside the fixed-point kernel's domain fall back to the exact kernel even when the switch is on
    _abi.set_option("first_fixed", 1)
    try:
        _, x8, op8 = _case("q8", (2, 32, 32, 3), "quantized", 8)
        got, kern = _run_group(x8, None, op8, bn, Q(4), 2, _abi.STORE_I4)
        assert kern == "mfma_f32_first_cin3"
        np.testing.assert_array_equal(got, _oracle_group(x8, op8, bn, Q(4), 2, float_conv="device"))
    finally:
        _abi.set_option("first_fixed", 0)
