#!/usr/bin/env python3
"""BASELINE config 5: synthetic ImageNet-224 ResNet (nres=10, pfilt=1) full-qnn 4/4 through GraphModel."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
nets, engine = pkg.nets, pkg.engine
from oracle import qnn_oracle as O

cf = nets.baseline_config(4)
spec = nets.build_spec(cf, nets.SEED_BASE + 4)
model = engine.GraphModel(spec)
B = int(os.environ.get("B", "64"))
if os.environ.get("CHECK", "1") == "1":
    x1 = nets.synthetic_images(cf, 1, 5)
    t0 = time.time(); want = O.run_spec(spec, x1, float_conv="device"); t_or = time.time() - t0
    got = model(torch.as_tensor(x1).cuda()).cpu().numpy()
    print(json.dumps({"check": "resnet224 N=1 vs oracle", "max_abs_diff": float(np.abs(got - want).max()),
                      "oracle_s": round(t_or, 1)}))
x = torch.as_tensor(nets.synthetic_images(cf, B, 6)).cuda()
for _ in range(2):
    y = model(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    y = model(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": "imagenet224_resnet10_w4a4", "batch": B, "ms": round(dt * 1e3, 2),
                  "images_per_s": round(B / dt, 1), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}))
