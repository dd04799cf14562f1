#!/usr/bin/env python3
"""The 16 -> 16 channel stage alone (64 x 224 x 224, folded epilogue, with and without the shortcut merge): HIP-event time per
launch from a graph of back-to-back launches.  Usage: tools/bench_strip16.py [reps]   (A/B two builds with tools/ab_lib.sh)"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
_abi, engine = pkg._abi, pkg.engine
F32 = np.float32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
n, hw, c = 64, 224, 16
op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (3, 3, c, c)).astype(F32), "bias": None,
      "strides": (1, 1), "padding": "same"}
var = 9 * c * 0.12
bn = dict(op="bn", eps=1e-3, gamma=rng.uniform(0.5, 1.5, c).astype(F32), beta=(rng.standard_normal(c) * 0.5).astype(F32),
          mean=(rng.standard_normal(c) * 0.1 * np.sqrt(var)).astype(F32), var=(var * rng.uniform(0.8, 1.25, c)).astype(F32))
w = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=1, same_pad=True)
i, s = engine.bn_constants(bn)
inv, shift = torch.as_tensor(i).cuda(), torch.as_tensor(s).cuda()
x = torch.randint(-2**31, 2**31 - 1, (n * hw * hw, c // 8), dtype=torch.int32, device="cuda")
sc = torch.randint(-2**31, 2**31 - 1, (n * hw * hw, c // 8), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
row = {"lib": os.path.basename(_abi.lib_path())}
for res in (False, True):
    kw = dict(res=sc, res_store=_abi.STORE_I4, res_bits=4, post_scale=0.5) if res else {}
    f = _abi.Fold.try_prepare(w, _abi.STORE_I4, 4, inv, shift, _abi.FN_QUANTIZED_TANH, 4, _abi.STORE_I4, **kw)
    assert f is not None and f.usable

    def launch():
        _abi.conv2d(w, x, _abi.STORE_I4, 4, n, hw, hw, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4, out=y, fold=f, **kw)
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            launch()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    row["res" if res else "nores"] = round(min(ts), 2)
    row["kernel"] = _abi.last_kernel()
    row["digest_" + ("res" if res else "nores")] = int(y.to(torch.int64).sum().item())
print(json.dumps(row))
