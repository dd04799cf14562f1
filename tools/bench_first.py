#!/usr/bin/env python3
"""Time the float-input first layer of a BASELINE config on its own (HIP events behind queued launches).
Env: IDX (baseline config index, default 2), N (batch, default 4096), FIXED=1 (fixed-point variant), U8=1 (typed uint8
image entry, QNN_STORE_U8), QNN_FIRST_ABL (kernel ablations, experiment builds only).
Prints one JSON line: kernel tag, us per launch, fraction of the 157.3 TFLOP/s f32 matrix peak."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
nets, engine, abi = pkg.nets, pkg.engine, pkg._abi

if os.environ.get("FIXED", "0") == "1":          # the opt-in fixed-point variant (csrc/qnn_first_fixed.hip)
    abi.set_option("first_fixed", 1)
idx = int(os.environ.get("IDX", "2"))
N = int(os.environ.get("N", "4096"))
cf = nets.baseline_config(idx)
spec = nets.build_spec(cf, nets.SEED_BASE + idx)
model = engine.FusedModel(spec)
st = model.steps[0]
U8 = os.environ.get("U8", "0") == "1"
if os.environ.get("IMAGE", "0") == "1":          # float32 bytes / 255 recognised as bytes (csrc/qnn_first_u8.hip, F32IN)
    abi.set_option("first_image", 1)
# ROTATE=K: K distinct input batches used in turn, so the kernel reads HBM and not a batch that stays in the 256 MB
# Infinity Cache between launches (one float32 CIFAR batch is 50 MB)
ROT = int(os.environ.get("ROTATE", "1"))
xs = [torch.as_tensor(nets.synthetic_images_u8(cf, N, 1 + r) if U8 else nets.synthetic_images(cf, N, 1 + r)).cuda()
      for r in range(ROT)]
x = xs[0]
outb = None
cnt = [0]


def launch():
    global outb
    xi = xs[cnt[0] % ROT]
    cnt[0] += 1
    o, _, _ = abi.conv2d(st["w"], xi, abi.STORE_U8 if U8 else st["x_store"], st["x_bits"], N, cf.dim, cf.dim, st["inv"],
                         st["shift"], st["fn"], st["act_bits"], st["pool"], st["out_store"], out=outb)
    outb = o
    return o


for _ in range(5):
    launch()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        launch()
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
kh, kw, cin, cout = st["w"].shape
flops = 2.0 * N * cf.dim * cf.dim * kh * kw * cin * cout
print(json.dumps({"kernel": abi.last_kernel(), "idx": idx, "N": N, "rotate": ROT, "abl": os.environ.get("QNN_FIRST_ABL", "0"),
                  "us": round(best * 1e3, 2), "TFLOPs": round(flops / best / 1e9, 1),
                  "GBps": round((x.numel() * x.element_size() + N * (cf.dim // st["pool"]) ** 2 * cout // 2) / best / 1e6, 1),
                  "frac_f32_mfma": round(flops / best / 1e9 / 157.3, 3)}))
