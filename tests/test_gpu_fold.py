"""The epilogue as integer thresholds (include/qnn_abi.h, qnn_fold_prepare; csrc/qnn_fold.h).

What the reference computes behind a low-bit convolution -- K.bias_add, BatchNormalization (models/vgg.py:16,
models/resnet.py:61), the residual merge keras.layers.add + Lambda(x * 0.5) (models/resnet.py:127-128) and quantized_tanh
(quantized_ops.py:87-100) -- is a step function of the integer accumulator.  These tests

  * sweep the WHOLE accumulator domain of a layer (x all 16 shortcut codes), channel by channel, through the folded
    epilogue (qnn_fold_eval: the device function the kernels inline) against the oracle's float32 restatement of the
    chain (bias_add, batchnorm_inference, add, scale, quantized_tanh);
  * run the strip kernels with and without the fold on the same tensors: identical bits, and the oracle's;
  * run the residual engine both ways on whole networks.
"""
import numpy as np
import pytest
import torch

from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O
from test_gpu_parity import dev, host, Q, _rand_bn

pytestmark = pytest.mark.gpu
F32 = np.float32


def _layer(rng, cin, cout, k=3, bias=True, stride=1, gamma_sign=None):
    op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (k, k, cin, cout)).astype(F32),
          "bias": (rng.standard_normal(cout) * 0.05).astype(F32) if bias else None, "strides": (stride, stride),
          "padding": "same"}
    bn = _rand_bn(rng, cout, k * k * cin * 0.12)
    if gamma_sign is not None:
        bn["gamma"] = (np.abs(bn["gamma"]) * gamma_sign).astype(F32)
    return op, bn


def _chain_codes(acc, sc, c, op, bn, x_bits=4):
    """The reference chain on integer accumulators of channel c (float32, one rounding per operation): conv value =
    acc * 2^-(wshift + xshift) (exact), K.bias_add, BatchNormalization, [(shortcut + y) * 0.5], quantized_tanh(nb=4)."""
    v = (acc.astype(F32) * F32(2.0 ** -(3 + x_bits - 1))).astype(F32)
    if op["bias"] is not None:
        v = O.bias_add(v, op["bias"][c])
    v = O.batchnorm_inference(v, bn["gamma"][c], bn["beta"][c], bn["mean"][c], bn["var"][c], bn["eps"])
    if sc is not None:
        r = (sc.astype(F32) * F32(0.125)).astype(F32)
        v = ((r + v).astype(F32) * F32(0.5)).astype(F32)
    return np.rint(O.quantized_tanh(v, 4) * F32(8)).astype(np.int32)


def _prep(op, bn, res, x_bits=4):
    w = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=op["strides"][0], same_pad=True)
    i, s = engine.bn_constants(bn)
    inv, shift = dev(i), dev(s)
    kw = {}
    if res:
        kw = dict(res=torch.zeros(1, dtype=torch.int32, device="cuda"), res_store=_abi.STORE_I4, res_bits=4, post_scale=0.5)
    f = _abi.Fold.try_prepare(w, _abi.STORE_I4, x_bits, inv, shift, _abi.FN_QUANTIZED_TANH, 4, _abi.STORE_I4, **kw)
    return w, inv, shift, f


@pytest.mark.parametrize("cin,cout,k,bias,res", [(16, 16, 3, False, False), (16, 16, 3, False, True), (16, 32, 3, True, False),
                                                 (32, 32, 3, True, True), (64, 64, 3, False, True), (64, 64, 3, False, False),
                                                 (64, 64, 1, False, False)])
def test_whole_accumulator_domain_per_channel(cin, cout, k, bias, res):
    """Every accumulator value the layer can produce (from its quantized weights and the 4-bit input codes), for every
    channel and -- with a shortcut -- every shortcut code: folded epilogue == oracle chain.  Both signs of the BN scale."""
    rng = np.random.default_rng(cin * 100 + cout + k + 7 * bias + 13 * res)
    op, bn = _layer(rng, cin, cout, k, bias)
    w, inv, shift, f = _prep(op, bn, res)
    assert f is not None
    wc, wshift = O.weight_codes(op)                     # HWIO integer codes
    assert wshift == 3
    wc = wc.reshape(-1, cout)
    hi = np.maximum(wc * -8, wc * 7).sum(axis=0)
    lo = np.minimum(wc * -8, wc * 7).sum(axis=0)
    assert f.acc_lo == lo.min() and f.acc_hi == hi.max(), (f.acc_lo, f.acc_hi, lo.min(), hi.max())
    assert f.shortcut_codes == (16 if res else 1)
    A, beta = (host(t) for t in f.constants("cuda"))
    folded = A != 0
    assert folded.sum() == f.folded
    assert f.points == int(((hi - lo + 1)[folded] * (16 if res else 1)).sum())
    # the search must succeed almost everywhere: a fold that fails is a performance bug, never a correctness one
    assert f.folded >= cout - max(1, cout // 8), (f.folded, cout)
    checked = 0
    for c in np.nonzero(folded)[0]:
        acc = np.arange(lo[c], hi[c] + 1, dtype=np.int32)
        for sc in (range(-8, 8) if res else (None,)):
            scv = None if sc is None else np.full_like(acc, sc)
            got = host(f.eval(int(c), dev(acc), None if scv is None else dev(scv)))
            want = _chain_codes(acc, scv, c, op, bn)
            np.testing.assert_array_equal(got, want, err_msg="channel %d shortcut %r" % (c, sc))
            checked += acc.size
    assert checked == f.points
    print("[fold] %dx%d k%d res=%d: %d / %d channels folded, %d points swept against the oracle"
          % (cin, cout, k, res, f.folded, cout, checked))


def _packed_codes(rng, n, h, w_, c):
    pre = rng.standard_normal((n, h, w_, c)).astype(F32)
    x = O.run_spec([Q(4)], pre)
    return x, _abi.pack(dev(x), c, _abi.FN_GRID, 4, _abi.STORE_I4)


@pytest.mark.parametrize("cin,cout,hw,n,stride", [(16, 16, (20, 32), 3, 1), (16, 16, (33, 23), 2, 1), (32, 32, (9, 16), 5, 1),
                                                  (64, 64, (14, 14), 3, 1), (64, 128, (5, 23), 2, 1), (16, 32, (16, 16), 2, 2),
                                                  (32, 64, (7, 9), 2, 2), (16, 16, (224, 224), 1, 1), (16, 16, (1, 2), 2, 1),
                                                  (16, 16, (3, 18), 3, 1), (16, 16, (13, 50), 70, 1), (16, 16, (40, 16), 2, 1),
                                                  (16, 16, (25, 34), 400, 1), (32, 32, (112, 112), 2, 1), (64, 64, (56, 56), 2, 1),
                                                  (32, 32, (1, 1), 3, 1), (64, 64, (3, 17), 5, 1), (32, 32, (26, 33), 40, 1),
                                                  (64, 64, (13, 16), 300, 1)])
@pytest.mark.parametrize("gamma_sign", [None, 1.0, -1.0])
def test_strip_kernels_with_and_without_the_fold(cin, cout, hw, n, stride, gamma_sign):
    """k_conv_strip / k_conv_strip_s2 with the folded epilogue: the same bits as the float32 chain and as the oracle, with
    and without the residual merge, ragged strips, every border class."""
    rng = np.random.default_rng(cin + 3 * cout + hw[0] + stride)
    H, W = hw
    x, xp = _packed_codes(rng, n, H, W, cin)
    for bias in (False, True):
        op, bn = _layer(rng, cin, cout, 3, bias, stride, gamma_sign)
        for res in ((False, True) if stride == 1 else (False,)):
            w, inv, shift, f = _prep(op, bn, res)
            assert f is not None
            Ho, Wo = -(-H // stride), -(-W // stride)
            kw = {}
            sc = None
            if res:
                sc, scp = _packed_codes(rng, n, Ho, Wo, cout)
                kw = dict(res=scp, res_store=_abi.STORE_I4, res_bits=4, post_scale=0.5)
            outs = []
            for fold in (None, f):
                y, _, _ = _abi.conv2d(w, xp, _abi.STORE_I4, 4, n, H, W, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1,
                                      _abi.STORE_I4, fold=fold, **kw)
                want_kernel = "strip_i4_c%d_s2" % cin if stride == 2 else "strip_i4_c%d" % cin
                if fold is not None and fold.usable and (cin, cout, stride) == (16, 16, 1) and W % 2 == 0:
                    want_kernel = "strip_i4_c16_lds"         # the LDS-staged form of the 16-channel stage (qnn_mfma_strip16.hip)
                assert _abi.last_kernel() == want_kernel, (_abi.last_kernel(), want_kernel)
                outs.append(host(_abi.unpack(y, n * Ho * Wo, cout, _abi.STORE_I4, 4)).reshape(n, Ho, Wo, cout))
            np.testing.assert_array_equal(outs[1], outs[0])
            v = O.quantized_conv2d_call(x, op["kernel"], op["bias"], nb=4, strides=op["strides"])
            v = O.batchnorm_inference(v, bn["gamma"], bn["beta"], bn["mean"], bn["var"], bn["eps"])
            if res:
                v = ((sc + v).astype(F32) * F32(0.5)).astype(F32)
            np.testing.assert_array_equal(outs[1], O.quantized_tanh(v, 4))
            assert f.usable or f.folded < cout


def test_fold_handle_is_bound_to_its_layer_and_epilogue():
    """A fold prepared for one layer / BN / shortcut form is rejected (QNN_EINVAL) anywhere else; unsupported forms
    (other bit widths, float32 shortcut) fold nothing (None) and the call without a fold still works."""
    rng = np.random.default_rng(3)
    op, bn = _layer(rng, 16, 16)
    op2, bn2 = _layer(rng, 16, 16)
    w, inv, shift, f = _prep(op, bn, False)
    w2, inv2, shift2, f2 = _prep(op2, bn2, True)
    x, xp = _packed_codes(rng, 1, 8, 16, 16)
    args = (_abi.STORE_I4, 4, 1, 8, 16)
    with pytest.raises(_abi.QnnError, match="fold handle was prepared for another"):
        _abi.conv2d(w2, xp, *args, inv2, shift2, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4, fold=f)
    with pytest.raises(_abi.QnnError, match="fold handle was prepared for another"):
        _abi.conv2d(w, xp, *args, inv2, shift2, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4, fold=f)
    with pytest.raises(_abi.QnnError, match="fold handle was prepared for another"):      # prepared WITH a shortcut
        _abi.conv2d(w2, xp, *args, inv2, shift2, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4, fold=f2)
    assert _abi.Fold.try_prepare(w, _abi.STORE_I4, 4, inv, shift, _abi.FN_QUANTIZED_TANH, 3, _abi.STORE_I4) is None
    assert _abi.Fold.try_prepare(w, _abi.STORE_I4, 4, inv, shift, _abi.FN_BINARY_TANH, 0, _abi.STORE_I4) is None
    assert _abi.Fold.try_prepare(w, _abi.STORE_I4, 4, inv, shift, _abi.FN_QUANTIZED_TANH, 4, _abi.STORE_I4,
                                 res=torch.zeros(4, device="cuda"), res_store=_abi.STORE_F32, post_scale=0.5) is None


@pytest.mark.parametrize("nres", [1, 2])
def test_residual_engine_with_and_without_folds(nres):
    """Whole CIFAR ResNets (models/resnet.py) through engine.ResidualFusedModel: folds on (default) and off give the
    oracle's logits bit for bit, and the folds were actually used."""
    cf = nets.Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET", nres=nres, dim=32)
    spec = nets.build_spec(cf, 5)[:-1]
    x = nets.synthetic_images(cf, 3, 5)
    want = O.run_spec(spec, x, float_conv="device")
    m = engine.ResidualFusedModel(spec, first_layer="exact")
    got = host(m(dev(x)))
    np.testing.assert_array_equal(got, want)
    folds = [f for f in m._folds.values() if f is not None]
    assert len(folds) >= 6 * nres - 2 and sum(f.usable for f in folds) >= len(folds) - 2, \
        [(f.folded, f.channels) for f in folds]
    m0 = engine.ResidualFusedModel(spec, fold=False, first_layer="exact")
    np.testing.assert_array_equal(host(m0(dev(x))), want)
    assert not m0._folds


@pytest.mark.parametrize("gamma_sign", [None, -1.0])
def test_image_entry_first_layer_fold(gamma_sign):
    """Mode 3: the first layer on image bytes (QNN_STORE_U8 / QNN_STORE_F32_IMAGE).  The folded epilogue reproduces the
    entry's specification  clip(rint(fma(float(S), A, B)))  on every integer sum S the filter can produce (swept here
    against the oracle's u8_affine + fma32), and the kernel gives the same bits with and without it."""
    rng = np.random.default_rng(17)
    cf = nets.baseline_config(2)
    op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (3, 3, 3, 64)).astype(F32),
          "bias": (rng.standard_normal(64) * 0.05).astype(F32), "strides": (1, 1), "padding": "same"}
    bn = _rand_bn(rng, 64, 27 * 0.33 * 0.33)
    if gamma_sign is not None:
        bn["gamma"] = (np.abs(bn["gamma"]) * gamma_sign).astype(F32)
    w = engine._prepack(op, _abi.STORE_F32, torch.device("cuda"), stride=1, same_pad=True)
    i, s_ = engine.bn_constants(bn)
    inv, shift = dev(i), dev(s_)
    f = _abi.Fold.try_prepare(w, _abi.STORE_U8, 0, inv, shift, _abi.FN_QUANTIZED_TANH, 4, _abi.STORE_I4)
    assert f is not None and f.mode == 3 and f.usable, (f and (f.mode, f.folded))
    codes, ws = O.weight_codes(op)
    codes = codes.reshape(-1, 64)
    lo, hi = 255 * np.minimum(codes, 0).sum(axis=0), 255 * np.maximum(codes, 0).sum(axis=0)
    assert f.acc_lo == lo.min() and f.acc_hi == hi.max()
    A, B = O.u8_affine(op["bias"], bn, 8.0, 255.0 * 2.0 ** ws, 64)
    total = 0
    for c in range(64):
        S = np.arange(lo[c], hi[c] + 1, dtype=np.int32)
        want = np.clip(np.rint(O.fma32(S.astype(F32), A[c], B[c])), -8, 7).astype(np.int32)
        np.testing.assert_array_equal(host(f.eval(c, dev(S))), want, err_msg="channel %d" % c)
        total += S.size
    assert total == f.points
    xu8 = nets.synthetic_images_u8(cf, 6, 3)
    for x in (dev(xu8), dev((xu8.astype(F32) / F32(255)).astype(F32))):
        store = _abi.STORE_U8 if x.dtype == torch.uint8 else _abi.STORE_F32_IMAGE
        outs = []
        for fold in (None, f):
            y, hp, wp = _abi.conv2d(w, x, store, 0, 6, 32, 32, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 2, _abi.STORE_I4, fold=fold)
            outs.append(host(_abi.unpack(y, 6 * hp * wp, 64, _abi.STORE_I4, 4)))
        np.testing.assert_array_equal(outs[0], outs[1])
        want = O.maxpool2d(O.u8_conv_group(xu8, op, bn, Q(4)))
        np.testing.assert_array_equal(outs[1].reshape(want.shape), want)
    # a fold of the image entry is refused on the exact float32 call of the same layer
    with pytest.raises(_abi.QnnError, match="fold handle was prepared for another"):
        _abi.conv2d(w, dev((xu8.astype(F32) / F32(255)).astype(F32)), _abi.STORE_F32, 0, 6, 32, 32, inv, shift,
                    _abi.FN_QUANTIZED_TANH, 4, 2, _abi.STORE_I4, fold=f)


@pytest.mark.parametrize("gamma_sign", [None, 1.0, -1.0])
@pytest.mark.parametrize("hw,n", [((16, 16), 5), ((8, 8), 9), ((32, 32), 2)])
def test_pooled_64_channel_layers_with_the_fold(hw, n, gamma_sign):
    """k_conv_mfma_halo (CIFAR B0 / C0: 64 -> 64 channels, 2x2 max pool) with the folded epilogue in its "bits" form: pooling
    on the raw accumulators that carry the offset, same bits as the float32 chain and as the oracle; also through the fused
    conv + classifier launch (qnn_conv2d_dense_forward)."""
    rng = np.random.default_rng(hw[0] + n)
    H, W = hw
    x, xp = _packed_codes(rng, n, H, W, 64)
    op, bn = _layer(rng, 64, 64, 3, True, 1, gamma_sign)
    if gamma_sign is not None:       # BN scales away from zero, as trained ones are: the thresholds then lie inside the "bits"
        bn["gamma"] = (np.sign(bn["gamma"]) * np.maximum(np.abs(bn["gamma"]), 0.5)).astype(F32)      # form's exact range
    w, inv, shift, f = _prep(op, bn, False)
    assert f is not None and f.usable and (f.mode == 2 or gamma_sign is None), (f.folded, f.mode)
    # (a channel whose BN scale is nearly zero spreads its thresholds beyond +-2^22 accumulator units: the handle is then
    # in the conversion form, which this kernel does not take -- it keeps the float32 chain; same bits either way)
    outs = []
    for fold in (None, f):
        y, hp, wp = _abi.conv2d(w, xp, _abi.STORE_I4, 4, n, H, W, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 2, _abi.STORE_I4, fold=fold)
        assert _abi.last_kernel() == "mfma_i4_halo64x64", _abi.last_kernel()
        outs.append(host(_abi.unpack(y, n * hp * wp, 64, _abi.STORE_I4, 4)).reshape(n, hp, wp, 64))
    np.testing.assert_array_equal(outs[0], outs[1])
    v = O.quantized_conv2d_call(x, op["kernel"], op["bias"], nb=4)
    v = O.batchnorm_inference(v, bn["gamma"], bn["beta"], bn["mean"], bn["var"], bn["eps"])
    np.testing.assert_array_equal(outs[1], O.maxpool2d(O.quantized_tanh(v, 4)))
    if hw == (8, 8):                                     # 4 x 4 pooled map: the classifier rides in the same launch
        dense = {"op": "dense", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (1024, 10)).astype(F32),
                 "bias": (rng.standard_normal(10) * 0.1).astype(F32)}
        wd = engine._prepack(dense, _abi.STORE_I4, torch.device("cuda"))
        got = [host(_abi.conv2d_dense(w, wd, xp, _abi.STORE_I4, 4, n, H, W, inv, shift, _abi.FN_QUANTIZED_TANH, 4, None, None,
                                      fold=fold)) for fold in (None, f)]
        assert _abi.last_kernel() == "mfma_i4_halo64x64+dense", _abi.last_kernel()
        np.testing.assert_array_equal(got[0], got[1])
        want = O.quantized_dense_call(O.maxpool2d(O.quantized_tanh(v, 4)).reshape(n, -1), dense["kernel"], dense["bias"], nb=4)
        np.testing.assert_array_equal(got[1], want)
