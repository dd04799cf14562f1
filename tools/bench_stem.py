#!/usr/bin/env python3
"""Time a 3-channel first layer with packed un-pooled output on its own: tools/bench_stem.py N H W COUT [bits=4] [abits=4]
[mode=u8|image|exact].  ResNet stem: 64 224 224 16; VGG-large first layer: 4096 32 32 256 bits=8 abits=8."""
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
N, H, W, cout = (int(v) for v in sys.argv[1:5])
kw = dict(a.split("=", 1) for a in sys.argv[5:])
bits, abits, mode = int(kw.get("bits", 4)), int(kw.get("abits", 4)), kw.get("mode", "u8")
rng = np.random.default_rng(0)
op = {"op": "conv", "kind": "quantized", "nb": bits, "kernel": rng.uniform(-1, 1, (3, 3, 3, cout)).astype(np.float32),
      "bias": None, "strides": (1, 1), "padding": "same"}
w = engine._prepack(op, abi.STORE_F32, torch.device("cuda"))
xu8 = torch.as_tensor(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
x = xu8 if mode == "u8" else (xu8.float() / 255.0)
inv = torch.full((cout,), 0.3, device="cuda")
shift = torch.zeros(cout, device="cuda")
store = abi.store_for_bits(abits)
if mode == "image":
    abi.set_option("first_image", 1)
out = None


def launch():
    global out
    out = abi.conv2d(w, x, abi.STORE_U8 if mode == "u8" else abi.STORE_F32, 0, N, H, W, inv, shift, abi.FN_QUANTIZED_TANH, abits,
                     1, store, out=out)[0]


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
print(json.dumps({"kernel": abi.last_kernel(), "shape": [N, H, W, cout], "mode": mode, "us": round(best * 1e3, 2),
                  "GBps": round((x.numel() * x.element_size() + N * H * W * cout * abits / 8) / best / 1e6, 1)}))
