"""Randomised sweep aimed at the matrix-pipe kernels (row-walking strips, register-operand and LDS-resident-filter
kernels, the tiled implicit GEMM and its LDS-DMA variant): channel counts that select them, image sizes that leave ragged
strips / tiles, strides 1 and 2, 4- and 8-bit operands, pooling, every output storage.  Bit-exact against the oracle (all
tensors are grid-valued).  The kernel family is left to the dispatcher; the test only requires that nothing falls back to
the generic any-shape kernel for these shapes when the matrix pipe is preferred."""
import numpy as np
import pytest

import qnn_amd  # noqa: F401
from qnn_amd import _abi, engine
from oracle import qnn_oracle as O
from test_gpu_parity import BIN_ACT, Q, _oracle_group, _rand_bn, _run_group

pytestmark = pytest.mark.gpu
F32 = np.float32


def _case(rng):
    bits = int(rng.choice([4, 4, 8]))
    cin = int(rng.choice([16, 32, 64, 64, 128, 192, 256]))
    cout = int(rng.choice([16, 32, 64, 64, 128, 256, 256, 512]))
    if bits == 8 and cin < 64:
        cin = 64
    if cin < 64 and cout > 128:
        cout = 64
    k = 3 if rng.random() < 0.85 else 1
    stride = 1 if rng.random() < 0.75 else 2
    big = cin * cout >= 128 * 256
    H = int(rng.integers(1, 14 if big else 41))
    W = int(rng.integers(1, 14 if big else 41))
    N = int(rng.integers(1, 4))
    wbits = int(rng.choice([2, 4] if bits == 4 else [4, 8]))
    return bits, N, H, W, cin, cout, k, stride, wbits


@pytest.mark.parametrize("seed", range(40))
def test_random_fast_path_layer(seed):
    rng = np.random.default_rng(77000 + seed)
    bits, N, H, W, cin, cout, k, stride, wbits = _case(rng)
    in_act = Q(bits)
    x = O.run_spec([in_act], rng.standard_normal((N, H, W, cin)).astype(F32))
    op = {"op": "conv", "kind": "quantized", "nb": wbits, "kernel": rng.uniform(-1, 1, (k, k, cin, cout)).astype(F32),
          "bias": (rng.standard_normal(cout) * 0.05).astype(F32) if rng.random() < 0.5 else None,
          "strides": (stride, stride), "padding": "same"}
    bn = _rand_bn(rng, cout, k * k * cin * 0.12)
    Ho, Wo = -(-H // stride), -(-W // stride)
    pools = [1] + ([2] if Ho >= 2 and Wo >= 2 and stride == 1 else [])
    _abi.set_conv_impl(_abi.IMPL_MFMA if rng.random() < 0.5 else _abi.IMPL_AUTO)
    try:
        for pool in pools:
            for act in (Q(bits), BIN_ACT) if bits == 4 else (Q(8), Q(4)):
                fn, abits = engine._act_code(act)
                stores = [_abi.STORE_F32, _abi.STORE_I8] + ([_abi.STORE_I4] if abits <= 4 else [])
                want = _oracle_group(x, op, bn, act, pool)
                for out_store in stores:
                    got, kern = _run_group(x, in_act, op, bn, act, pool, out_store)
                    np.testing.assert_array_equal(got, want, err_msg="%s pool=%d act=%r store=%d shape=%r" % (
                        kern, pool, act, out_store, (bits, N, H, W, cin, cout, k, stride, wbits)))
        got, kern = _run_group(x, in_act, op, None, None, 1, _abi.STORE_F32)      # the plain call() surface
        np.testing.assert_array_equal(got, _oracle_group(x, op, None, None, 1), err_msg=kern)
    finally:
        _abi.set_conv_impl(_abi.IMPL_AUTO)
