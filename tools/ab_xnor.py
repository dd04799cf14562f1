#!/usr/bin/env python3
"""A/B of the 1-bit XNOR layer op (float32 NHWC in and out, CIFAR B0 16x16 and C0 8x8 at batch 4096) between library
builds on ONE box:  QNN_LIB=<variant .so> python tools/ab_xnor.py  -- alternate the builds several times in one gpurun call
(boxes differ by up to 10 % on this VALU-bound kernel).  Prints the fraction of the 8 TB/s HBM roof on the M0 bytes."""
import importlib, os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
abi = pkg._abi
import bench
rng = np.random.default_rng(0)
abi.set_conv_impl(abi.IMPL_VALU)
out = {}
for name, H in (("B0", 16), ("C0", 8)):
    n = 4096
    x = torch.randn((n, H, H, 64), device="cuda")
    k = torch.as_tensor(rng.uniform(-1, 1, (3, 3, 64, 64)).astype(np.float32)).cuda()
    w = abi.Weights(abi.W_BINARY, 1, 1.0, k, torch.zeros(64, device="cuda"), 1, True, abi.STORE_BIN)
    ms, _, _ = bench.time_launch(torch, lambda: abi.conv2d_f32in(w, x, abi.FN_BINARY_TANH, 1)[0], reps=40, rounds=5)
    m0 = 2.0 * n * H * H * 64 * 4
    out[name] = round(m0 / (ms * 1e-3) / 8e12, 4)
print(os.environ.get("QNN_LIB", "current")[-20:], out)
