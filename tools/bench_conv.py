#!/usr/bin/env python3
"""Time ONE packed 3x3 (or 1x1) int4/int8 conv layer with a fused BN + quantized_tanh epilogue, as the fused engines
launch it.  Usage:  tools/bench_conv.py N H W CIN COUT [k=3] [stride=1] [bits=4] [res=0|1] [opt=key:val,...] [out=i4|f32]
[store=4|8 (packed storage of the codes; 8 = `bits`-bit codes kept in bytes)] [pool=1|2] [fold=0|1 (qnn_fold_prepare)]
Prints one JSON line per call: kernel tag, us, pixels/us, fraction of the 8 TB/s HBM roof on in + out (+ shortcut)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
abi, engine = pkg._abi, pkg.engine


def main():
    N, H, W, cin, cout = (int(v) for v in sys.argv[1:6])
    kw = dict(a.split("=", 1) for a in sys.argv[6:])
    k, stride, bits = int(kw.get("k", 3)), int(kw.get("stride", 1)), int(kw.get("bits", 4))
    res, out = int(kw.get("res", 0)), kw.get("out", "i4")
    for item in filter(None, kw.get("opt", "").split(",")):
        key, val = item.split(":")
        abi.set_option(key, int(val))
    rng = np.random.default_rng(0)
    store = int(kw["store"]) if "store" in kw else abi.store_for_bits(bits)
    pool = int(kw.get("pool", 1))
    op = {"op": "conv", "kind": "quantized", "nb": bits, "kernel": rng.uniform(-1, 1, (k, k, cin, cout)).astype(np.float32),
          "bias": None, "strides": (stride, stride), "padding": "same"}
    w = engine._prepack(op, store, torch.device("cuda"), stride=stride, same_pad=True)
    x = torch.randn((N, H, W, cin), device="cuda")
    xp = abi.pack(x, cin, abi.FN_QUANTIZED_TANH, bits, store)
    Ho, Wo = -(-H // stride), -(-W // stride)
    inv = torch.full((cout,), 0.05, device="cuda")
    shift = torch.zeros(cout, device="cuda")
    out_store = store if out == "i4" else abi.STORE_F32
    fn = abi.FN_QUANTIZED_TANH if out == "i4" else abi.FN_NONE
    rkw = {}
    if res:
        r = abi.pack(torch.randn((N, Ho, Wo, cout), device="cuda"), cout, abi.FN_QUANTIZED_TANH, bits, store)
        rkw = dict(res=r, res_store=store, res_bits=bits, post_scale=0.5)

    fold = None
    if int(kw.get("fold", 0)):
        inv = torch.as_tensor(rng.uniform(0.02, 0.08, cout).astype(np.float32)).cuda()
        shift = torch.as_tensor((rng.standard_normal(cout) * 0.5).astype(np.float32)).cuda()
        fold = abi.Fold.try_prepare(w, store, bits, inv, shift, fn, bits, out_store, **rkw)
        assert fold is not None and fold.usable, (fold and (fold.folded, fold.channels))

    def launch():
        return abi.conv2d(w, xp, store, bits, N, H, W, inv if out == "i4" else None, shift if out == "i4" else None,
                          fn, bits if out == "i4" else 0, pool, out_store, fold=fold, **rkw)[0]

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    sb = store if store in (4, 8) else bits
    in_b = N * H * W * cin * sb / 8
    out_b = N * (Ho // pool) * (Wo // pool) * cout * (sb / 8 if out == "i4" else 4)
    tot = in_b + out_b * (2 if res else 1)
    print(json.dumps({"kernel": abi.last_kernel(), "shape": [N, H, W, cin, cout, k, stride], "res": res, "out": out, "store": store, "pool": pool,
                      "opt": kw.get("opt", ""), "fold": int(kw.get("fold", 0)), "us": round(best * 1e3, 2),
                      "Mpix_per_s": round(N * Ho * Wo / best / 1e3, 1),
                      "TMACps": round(N * Ho * Wo * k * k * cin * cout / best / 1e9, 1),
                      "hbm_frac": round(tot / best / 1e6 / 8000.0, 3)}))


if __name__ == "__main__":
    main()
