#!/usr/bin/env python3
"""Parity of the fused residual engine at ImageNet geometry (224 / 112 / 56 wide stages) against the oracle:
ResNet with NRES blocks per stage (default 2), full-qnn 4/4, N images (default 2: tiles straddle images)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
nets, engine = pkg.nets, pkg.engine
from oracle import qnn_oracle as O

base = nets.baseline_config(4)
cf = nets.Config(network_type=base.network_type, wbits=base.wbits, abits=base.abits, architecture="RESNET",
                 nres=int(os.environ.get("NRES", "2")), dim=base.dim, channels=base.channels, classes=base.classes)
spec = nets.build_spec(cf, 11)
N = int(os.environ.get("N", "2"))
x = nets.synthetic_images(cf, N, 12)
t0 = time.time()
want = O.run_spec(spec[:-1], x, float_conv="device")          # logits (before the softmax)
t_or = time.time() - t0
m = engine.ResidualFusedModel(spec[:-1])
m.kernel_log = []
got = m(torch.as_tensor(x).cuda()).cpu().numpy()
kinds = sorted(set(m.kernel_log))
print(json.dumps({"check": "resnet dim=%d nres=%d N=%d logits vs oracle" % (cf.dim, cf.nres, N),
                  "bit_exact": bool(np.array_equal(got, want)), "max_abs_diff": float(np.abs(got - want).max()),
                  "oracle_s": round(t_or, 1), "kernels": kinds}))
sys.exit(0 if np.array_equal(got, want) else 1)
