"""Randomised sweep of qnn_conv2d_forward against the CPU oracle: random shapes (including
channel counts that are not multiples of the packing word), strides, paddings, weight and
activation widths, epilogues (bias / BN / clip / pool / residual merge) and output storages.
Every comparison is bit-exact: all tensors here are grid-valued."""
import numpy as np
import pytest
import torch

import qnn_amd
from qnn_amd import _abi, engine
from oracle import qnn_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def _act(bits):
    return {"op": "act", "fn": "binary_tanh"} if bits == 1 else {"op": "act", "fn": "quantized_tanh", "nb": bits}


def _one_case(rng):
    N = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 14)), int(rng.integers(1, 14))
    C = int(rng.choice([1, 3, 8, 16, 24, 32, 33, 64, 80, 128]))
    Cout = int(rng.choice([1, 4, 10, 16, 32, 40, 64, 128]))
    k = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2]))
    padding = str(rng.choice(["same", "same", "valid"]))
    wbits = int(rng.choice([1, 2, 3, 4, 5, 8]))
    abits = int(rng.choice([1, 2, 4, 8]))
    if padding == "valid" and (H < k or W < k):
        padding = "same"
    op = {"op": "conv", "kind": "binary" if wbits == 1 else "quantized",
          "kernel": rng.uniform(-1, 1, (k, k, C, Cout)).astype(F32),
          "bias": (rng.standard_normal(Cout) * 0.05).astype(F32) if rng.random() < 0.7 else None,
          "strides": (stride, stride), "padding": padding}
    if wbits > 1:
        op["nb"] = wbits
    x = O.run_spec([_act(abits)], rng.standard_normal((N, H, W, C)).astype(F32))
    return x, op, abits


@pytest.mark.parametrize("seed", range(48))
def test_random_conv_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    x, op, abits = _one_case(rng)
    N, H, W, C = x.shape
    Cout = op["kernel"].shape[3]
    k, stride = op["kernel"].shape[0], op["strides"][0]
    conv = O.run_spec([dict(op)], x)
    Ho, Wo = conv.shape[1], conv.shape[2]
    # random epilogue
    use_bn = rng.random() < 0.7
    bn = None
    if use_bn:
        var = max(1.0, float(k * k * C) * 0.3)
        bn = dict(op="bn", eps=1e-3, gamma=rng.uniform(-1.5, 1.5, Cout).astype(F32),
                  beta=(rng.standard_normal(Cout) * 0.5).astype(F32),
                  mean=(rng.standard_normal(Cout) * 0.1 * np.sqrt(var)).astype(F32),
                  var=(var * rng.uniform(0.8, 1.25, Cout)).astype(F32))
    out_bits = int(rng.choice([0, 1, 2, 4, 8]))          # 0 = no clip, float32 out
    pool = 2 if (rng.random() < 0.4 and Ho >= 2 and Wo >= 2) else 1
    use_res = pool == 1 and rng.random() < 0.4
    res_bits = int(rng.choice([0, 1, 4, 8])) if use_res else None     # 0 = float32 residual
    post = float(rng.choice([1.0, 0.5])) if use_res else 1.0
    # ---- oracle ----
    v = conv
    if bn is not None:
        v = O.batchnorm_inference(v, bn["gamma"], bn["beta"], bn["mean"], bn["var"], bn["eps"])
    res_val = None
    if use_res:
        pre = rng.standard_normal((N, Ho, Wo, Cout)).astype(F32)
        res_val = pre if res_bits == 0 else O.run_spec([_act(res_bits)], pre)
        v = ((res_val + v).astype(F32) * F32(post)).astype(F32)
    if out_bits:
        v = O.run_spec([_act(out_bits)], v)
    if pool == 2:
        v = O.maxpool2d(v, 2)
    want = v
    # ---- device ----
    wstore = engine._wstore(op)
    x_store = engine._join_store(abits, wstore)
    xp = _abi.pack(dev(x), C, _abi.FN_GRID, abits, x_store)
    w = engine._prepack(op, x_store, torch.device("cuda"), stride=stride, same_pad=op["padding"] == "same")
    inv = shift = None
    if bn is not None:
        i_, s_ = engine.bn_constants(bn)
        inv, shift = dev(i_), dev(s_)
    fn, ab = _abi.FN_NONE, 0
    stores = [_abi.STORE_F32]
    if out_bits == 1:
        fn = _abi.FN_BINARY_TANH
        stores += [_abi.STORE_BIN, _abi.STORE_I4, _abi.STORE_I8]
    elif out_bits:
        fn, ab = _abi.FN_QUANTIZED_TANH, out_bits
        stores += [_abi.STORE_I8] + ([_abi.STORE_I4] if out_bits <= 4 else [])
    rkw = {}
    if use_res:
        if res_bits == 0:
            rkw = dict(res=dev(res_val), res_store=_abi.STORE_F32, res_bits=0, post_scale=post)
        else:
            rstore = _abi.STORE_BIN if res_bits == 1 else _abi.store_for_bits(res_bits)
            rp = _abi.pack(dev(res_val), Cout, _abi.FN_GRID, res_bits, rstore)
            rkw = dict(res=rp, res_store=rstore, res_bits=res_bits, post_scale=post)
    for impl in (_abi.IMPL_VALU, _abi.IMPL_AUTO):
        _abi.set_conv_impl(impl)
        for out_store in stores:
            y, Hp, Wp = _abi.conv2d(w, xp, x_store, abits, N, H, W, inv, shift, fn, ab, pool, out_store, **rkw)
            if out_store == _abi.STORE_F32:
                got = host(y)
            else:
                got = host(_abi.unpack(y, N * Hp * Wp, Cout, out_store, out_bits)).reshape(N, Hp, Wp, Cout)
            np.testing.assert_array_equal(got, want, err_msg="seed=%d impl=%d out_store=%d kernel=%s"
                                          % (seed, impl, out_store, _abi.last_kernel()))
    _abi.set_conv_impl(_abi.IMPL_AUTO)
