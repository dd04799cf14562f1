"""The projection shortcut computed inside the second convolution's launch (include/qnn_abi.h, qnn_projection_t).

models/resnet.py:117-129: the first block of a stage adds the block input through a 1x1 strides-2 QuantizedConv2D (no BN, no
activation) to the BN output of the block's second 3x3 convolution, halves the sum and clips it with quantized_tanh.  These
tests run that block end as ONE launch (k_conv_strip<.., RES = 3>) against

  * the two-launch form (k_conv_pw_f32 -> float32 shortcut tensor -> k_conv_strip<.., RES = 2>): identical bits;
  * the oracle's float32 restatement of the reference's operations;
and whole CIFAR ResNets through engine.ResidualFusedModel with and without the fused form.
"""
import numpy as np
import pytest
import torch

from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O
from test_gpu_parity import dev, host
from test_gpu_fold import _layer, _packed_codes

pytestmark = pytest.mark.gpu
F32 = np.float32


def _proj_layer(rng, cin0, cout, bias):
    return {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (1, 1, cin0, cout)).astype(F32),
            "bias": (rng.standard_normal(cout) * 0.05).astype(F32) if bias else None, "strides": (2, 2), "padding": "same"}


@pytest.mark.parametrize("cin,hw0,n", [(32, (224, 224), 1), (64, (112, 112), 2), (32, (13, 17), 3), (64, (26, 63), 5),
                                        (32, (2, 2), 4), (64, (1, 1), 2), (32, (31, 66), 70), (64, (9, 40), 260)])
@pytest.mark.parametrize("bias", [(False, False), (True, True), (True, False)])
def test_projection_inside_the_launch_equals_the_two_launch_form(cin, hw0, n, bias):
    """Block input H0 x W0 x cin/2 (odd sizes: 'same' padding of a strides-2 1x1 window reads pixel (2y, 2x)), block
    body output ceil(H0/2) x ceil(W0/2) x cin: one launch vs two, and the oracle."""
    rng = np.random.default_rng(cin + hw0[0] * 7 + hw0[1] + n)
    H0, W0 = hw0
    H, W = -(-H0 // 2), -(-W0 // 2)
    x0, x0p = _packed_codes(rng, n, H0, W0, cin // 2)
    x, xp = _packed_codes(rng, n, H, W, cin)
    op, bn = _layer(rng, cin, cin, 3, bias[0])
    pop = _proj_layer(rng, cin // 2, cin, bias[1])
    w = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=1, same_pad=True)
    pw = engine._prepack(pop, _abi.STORE_I4, torch.device("cuda"), stride=2, same_pad=True)
    i, s = engine.bn_constants(bn)
    inv, shift = dev(i), dev(s)
    # two launches: the float32 shortcut tensor, then the conv that reads it
    sc, Hs, Ws = _abi.conv2d(pw, x0p, _abi.STORE_I4, 4, n, H0, W0, None, None, _abi.FN_NONE, 0, 1, _abi.STORE_F32)
    assert (Hs, Ws) == (H, W) and _abi.last_kernel() == "pw_i4_f32"
    y2, _, _ = _abi.conv2d(w, xp, _abi.STORE_I4, 4, n, H, W, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                           res=sc, res_store=_abi.STORE_F32, post_scale=0.5)
    assert _abi.last_kernel() == "strip_i4_c%d" % cin
    # one launch
    y1, _, _ = _abi.conv2d(w, xp, _abi.STORE_I4, 4, n, H, W, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                           post_scale=0.5, proj=(pw, x0p, H0, W0, 4))
    assert _abi.last_kernel() == "strip_i4_c%d_proj" % cin
    assert torch.equal(y1, y2)
    got = host(_abi.unpack(y1, n * H * W, cin, _abi.STORE_I4, 4)).reshape(n, H, W, cin)
    if n * H * W <= 20000:
        v = O.quantized_conv2d_call(x, op["kernel"], op["bias"], nb=4, strides=(1, 1))
        v = O.batchnorm_inference(v, bn["gamma"], bn["beta"], bn["mean"], bn["var"], bn["eps"])
        r = O.quantized_conv2d_call(x0, pop["kernel"], pop["bias"], nb=4, strides=(2, 2))
        np.testing.assert_array_equal(host(sc), r)
        v = ((r + v).astype(F32) * F32(0.5)).astype(F32)
        np.testing.assert_array_equal(got, O.quantized_tanh(v, 4))


def test_projection_is_refused_where_no_kernel_computes_it():
    """16 -> 16 channel layers, mismatched sizes, a shortcut tensor beside it: QNN_EUNSUPPORTED / QNN_EINVAL, never a
    silently different result."""
    rng = np.random.default_rng(1)
    x, xp = _packed_codes(rng, 1, 8, 16, 16)
    op, bn = _layer(rng, 16, 16, 3, False)
    w = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=1, same_pad=True)
    i, s = engine.bn_constants(bn)
    inv, shift = dev(i), dev(s)
    x0, x0p = _packed_codes(rng, 1, 16, 32, 8)
    pw = engine._prepack(_proj_layer(rng, 8, 16, False), _abi.STORE_I4, torch.device("cuda"), stride=2, same_pad=True)
    with pytest.raises(_abi.QnnUnsupported):
        _abi.conv2d(w, xp, _abi.STORE_I4, 4, 1, 8, 16, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                    post_scale=0.5, proj=(pw, x0p, 16, 32, 4))
    # a 32-channel layer whose block input has the wrong size
    x, xp = _packed_codes(rng, 1, 8, 16, 32)
    op, bn = _layer(rng, 32, 32, 3, False)
    w = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=1, same_pad=True)
    i, s = engine.bn_constants(bn)
    inv, shift = dev(i), dev(s)
    x0, x0p = _packed_codes(rng, 1, 16, 32, 16)
    pw = engine._prepack(_proj_layer(rng, 16, 32, False), _abi.STORE_I4, torch.device("cuda"), stride=2, same_pad=True)
    with pytest.raises(_abi.QnnUnsupported):
        _abi.conv2d(w, xp, _abi.STORE_I4, 4, 1, 8, 16, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                    post_scale=0.5, proj=(pw, x0p, 12, 32, 4))
    with pytest.raises(_abi.QnnError, match="proj excludes res"):
        _abi.conv2d(w, xp, _abi.STORE_I4, 4, 1, 8, 16, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                    res=xp, res_store=_abi.STORE_I4, res_bits=4, post_scale=0.5, proj=(pw, x0p, 16, 32, 4))
    y, _, _ = _abi.conv2d(w, xp, _abi.STORE_I4, 4, 1, 8, 16, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                          post_scale=0.5, proj=(pw, x0p, 16, 32, 4))
    assert _abi.last_kernel() == "strip_i4_c32_proj"


@pytest.mark.parametrize("nres", [1, 2])
def test_residual_engine_with_projection_blocks_in_one_launch(nres):
    """Whole CIFAR ResNets (models/resnet.py; stages of 16 / 32 / 64 filters, projection blocks at 32 and 64): the fused
    form is used, and the logits are the oracle's and the two-launch engine's bit for bit."""
    cf = nets.Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET", nres=nres, dim=32)
    spec = nets.build_spec(cf, 7)[:-1]
    x = nets.synthetic_images(cf, 3, 7)
    want = O.run_spec(spec, x, float_conv="device")
    m = engine.ResidualFusedModel(spec, first_layer="exact")
    m.kernel_log = []
    got = host(m(dev(x)))
    np.testing.assert_array_equal(got, want)
    assert m.kernel_log.count("strip_i4_c32_proj") == 1 and m.kernel_log.count("strip_i4_c64_proj") == 1, m.kernel_log
    assert "pw_i4_f32" not in m.kernel_log
    m0 = engine.ResidualFusedModel(spec, first_layer="exact", fuse_projection=False)
    m0.kernel_log = []
    np.testing.assert_array_equal(host(m0(dev(x))), want)
    assert m0.kernel_log.count("pw_i4_f32") == 2 and not any(k.endswith("_proj") for k in m0.kernel_log)
