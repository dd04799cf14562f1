"""CPU checks of the oracle's restatement of the typed image entry (include/qnn_abi.h, QNN_STORE_U8 / _F32_IMAGE):
the float32 FMA emulation is exact, the integer path agrees with the float path on the reference's own first-layer
vectors (tests/golden/ref_first.npz: BinaryConv2D / QuantizedConv2D .call() executed in place), and whole specs run."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import qnn_oracle as O
from qnn_amd import nets

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


def test_fma32_is_the_correctly_rounded_fused_multiply_add():
    """Against exact rational arithmetic, including the double-rounding traps (float64 sums that land on a float32 tie)."""
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.integers(-900000, 900000, 4000).astype(F32), F32([0, 1, -1, 16777215, -16777215, 3, 5])])
    b = np.concatenate([rng.standard_normal(4000).astype(F32) * F32(1e-3), F32([1e-3, 2.0 ** -24, 1 + 2.0 ** -23, 0.1, -0.1, 1 / 3, 1e-8])])
    c = np.concatenate([rng.standard_normal(4000).astype(F32) * F32(4), F32([0.5, 1.0, -1.0, 2.0 ** -25, 1e8, -1 / 3, 1.0])])
    # hand-made ties: a*b + c exactly half way between two float32 values
    a = np.concatenate([a, F32([1, 1, 3, 3])])
    b = np.concatenate([b, F32([2.0 ** -24, -2.0 ** -24, 2.0 ** -25, 2.0 ** -25])])
    c = np.concatenate([c, F32([1.0, 1.0, 1.0, 1 + 2.0 ** -23])])
    got = O.fma32(a, b, c)
    for x, y, z, g in zip(a, b, c, got):
        exact = Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z))
        lo = F32(float(exact))                      # float(Fraction) rounds to nearest float64; refine to float32 neighbours
        cands = sorted({float(np.nextafter(lo, F32(-np.inf))), float(lo), float(np.nextafter(lo, F32(np.inf)))})
        best = min(cands, key=lambda v: (abs(Fraction(v) - exact), int(np.float32(v).view(np.uint32)) & 1))
        assert float(g) == best, (x, y, z, g, best)


def _first_cases():
    d = np.load(os.path.join(GOLD, "ref_first.npz"))
    return d, json.loads(bytes(d["index_json"]).decode())["first"]


@pytest.mark.parametrize("tag", [c["tag"] for c in _first_cases()[1]])
def test_u8_specification_vs_the_references_first_layer_call(tag):
    d, cases = _first_cases()
    c = [k for k in cases if k["tag"] == tag][0]
    xu8 = d[tag + "_xu8"]
    op = {"op": "conv", "kind": c["kind"], "kernel": (d[tag + "_kernel"].astype(F32) / F32(32768)).astype(F32),
          "strides": (1, 1), "padding": "same"}
    if c["nb"]:
        op["nb"] = c["nb"]
    if c["use_bias"]:
        op["bias"] = d[tag + "_bias"]
    got = O.u8_conv_group(xu8, op)
    x = (xu8.astype(F32) / F32(255)).astype(F32)
    exact = O.run_spec([op], x)                       # the float32-input oracle (float64 accumulation)
    for prom in ("nep50", "legacy"):
        ref = d["%s_y_%s" % (tag, prom)]
        band = 1e-5 * np.maximum(1.0, np.abs(ref.astype(np.float64)))
        r_u8 = float((np.abs(got.astype(np.float64) - ref) / band).max())
        r_ex = float((np.abs(exact.astype(np.float64) - ref) / band).max())
        assert r_u8 <= 1.0 and r_u8 <= r_ex + 0.02, (tag, prom, r_u8, r_ex)
    # and against the real-number convolution of the exact quotients code / 255: within 1.5 ulp
    k, ws = O.weight_codes(op)
    real = O.int_conv2d(xu8.astype(np.int64), k).astype(np.float64) / (255.0 * 2.0 ** ws)
    if op.get("bias") is not None:
        real = real + op["bias"].astype(np.float64)
    assert np.abs(got - real).max() <= 2.0 * np.spacing(F32(np.abs(real).max()))


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_run_spec_u8_whole_networks(idx):
    """The uint8 entry changes the first conv group only; on these networks no activation code moves against the
    float32-input oracle (the quantity the GPU tests then compare bit for bit)."""
    cf = nets.baseline_config(idx)
    spec = nets.build_spec(cf, nets.SEED_BASE + idx)
    xu8 = nets.synthetic_images_u8(cf, 3, 11)
    y8 = O.run_spec_u8(spec, xu8, return_all=True)
    yf = O.run_spec(spec, (xu8.astype(F32) / F32(255)).astype(F32), return_all=True)
    bn_i, act_i, nxt = O._u8_group(spec)
    assert (bn_i, act_i) == (1, 2)                     # conv -> bn -> act fused behind the bytes
    flips = sum(int((y8[k] != yf[k]).sum()) for k in y8 if k in yf and y8[k] is not None)
    assert flips == 0
