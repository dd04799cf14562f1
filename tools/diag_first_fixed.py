"""Debug helper: where does the fixed-point first layer differ from its specification?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import qnn_amd
from qnn_amd import _abi
import test_gpu_first_fixed as T
from test_gpu_parity import _run_group

_abi.set_option("first_fixed", 1)
for name, shape, kind, nb in T.CASES[:2]:
    rng, x, op = T._case(name, shape, kind, nb)
    got, kern = _run_group(x, None, op, None, None, 1, _abi.STORE_F32)
    want = T._fixed_conv(x, op)
    bad = got != want
    print(name, kern, "mismatch", bad.mean())
    print(" by image", bad.mean(axis=(1, 2, 3)))
    print(" by row", np.round(bad.mean(axis=(0, 2, 3)), 2))
    print(" by col", np.round(bad.mean(axis=(0, 1, 3)), 2))
    print(" by ch ", np.round(bad.mean(axis=(0, 1, 2)), 2))
    d = (got.astype(np.float64) - want)
    print(" diff sample", d[0, 5, 5, :8], "\n got", got[0, 5, 5, :8], "\n want", want[0, 5, 5, :8])
    # hypothesis: zero input
    x0 = np.zeros_like(x)
    g0, _ = _run_group(x0, None, op, None, None, 1, _abi.STORE_F32)
    print(" zero input: max|got - bias|", np.abs(g0 - op["bias"]).max())
    x1 = np.ones_like(x)
    g1, _ = _run_group(x1, None, op, None, None, 1, _abi.STORE_F32)
    w1 = T._fixed_conv(x1, op)
    print(" ones input mismatch", (g1 != w1).mean(), g1[0, 5, 5, :4], w1[0, 5, 5, :4])
    xs = np.zeros_like(x); xs[:, 5, 5, 0] = 1.0
    gs, _ = _run_group(xs, None, op, None, None, 1, _abi.STORE_F32)
    ws = T._fixed_conv(xs, op)
    print(" delta input mismatch", (gs != ws).mean())
    print(np.round((gs - op["bias"])[0, 3:8, 3:8, 0], 3)); print(np.round((ws - op["bias"])[0, 3:8, 3:8, 0], 3))
