#!/usr/bin/env python3
"""A/B of the strip kernels with the float32 chain and with the folded epilogue (qnn_fold_prepare) on the ResNet-224
layer shapes: HIP-event time per launch, back to back.  Usage: python tools/bench_fold.py [reps]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
_abi, engine = pkg._abi, pkg.engine
F32 = np.float32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
out = []
SHAPES = [(64, 224, 16, 16, 1), (64, 112, 32, 32, 1), (64, 56, 64, 64, 1), (64, 224, 16, 32, 2), (64, 112, 32, 64, 2)]
if os.environ.get("NS"):        # batch-size sweep of the 16-channel stage: fixed cost per launch vs cost per image
    SHAPES = [(int(v), 224, 16, 16, 1) for v in os.environ["NS"].split(",")]
for (n, hw, cin, cout, stride) in SHAPES:
    op = {"op": "conv", "kind": "quantized", "nb": 4, "kernel": rng.uniform(-1, 1, (3, 3, cin, cout)).astype(F32),
          "bias": None, "strides": (stride, stride), "padding": "same"}
    var = 9 * cin * 0.12
    bn = dict(op="bn", eps=1e-3, gamma=rng.uniform(0.5, 1.5, cout).astype(F32), beta=(rng.standard_normal(cout) * 0.5).astype(F32),
              mean=(rng.standard_normal(cout) * 0.1 * np.sqrt(var)).astype(F32), var=(var * rng.uniform(0.8, 1.25, cout)).astype(F32))
    w = engine._prepack(op, _abi.STORE_I4, torch.device("cuda"), stride=stride, same_pad=True)
    i, s = engine.bn_constants(bn)
    inv, shift = torch.as_tensor(i).cuda(), torch.as_tensor(s).cuda()
    ho = hw // stride
    x = torch.randint(-2**31, 2**31 - 1, (n * hw * hw, cin // 8), dtype=torch.int32, device="cuda")
    sc = torch.randint(-2**31, 2**31 - 1, (n * ho * ho, cout // 8), dtype=torch.int32, device="cuda")
    y = torch.empty((n * ho * ho, cout // 8), dtype=torch.int32, device="cuda")
    for res in ((False, True) if stride == 1 else (False,)):
        kw = dict(res=sc, res_store=_abi.STORE_I4, res_bits=4, post_scale=0.5) if res else {}
        f = _abi.Fold.try_prepare(w, _abi.STORE_I4, 4, inv, shift, _abi.FN_QUANTIZED_TANH, 4, _abi.STORE_I4, **kw)
        row = dict(shape=[n, hw, cin, cout, stride], res=res, folded=[f.folded, f.channels] if f else None)
        ref = None
        for name, fold, lds in (("chain", None, 1), ("fold", f, 1), ("fold_nolds", f, 0)):
            _abi.set_option("lds16", lds)            # 0: the register-staged strip kernel with the same folded epilogue
            def launch():
                _abi.conv2d(w, x, _abi.STORE_I4, 4, n, hw, hw, inv, shift, _abi.FN_QUANTIZED_TANH, 4, 1, _abi.STORE_I4,
                            out=y, fold=fold, **kw)
            for _ in range(10):
                launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    launch()
            g.replay(); torch.cuda.synchronize()
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            row[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / reps, 2)
            row[name + "_kernel"] = _abi.last_kernel()
            got = y.clone()
            if ref is None:
                ref = got
            else:
                row["same_bits"] = row.get("same_bits", True) and bool(torch.equal(ref, got))
        _abi.set_option("lds16", 1)
        out.append(row)
        print(json.dumps(row), flush=True)
