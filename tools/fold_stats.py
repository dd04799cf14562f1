#!/usr/bin/env python3
"""Fold statistics of a workload's layers (qnn_fold_prepare): channels folded, mode, points swept.  Usage: tools/fold_stats.py [config index]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
nets, engine = pkg.nets, pkg.engine
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cf = nets.baseline_config(idx)
spec = nets.build_spec(cf, nets.SEED_BASE + idx)
m = engine.ResidualFusedModel(spec, first_layer="image")
x = torch.as_tensor(nets.synthetic_images(cf, 2, 1)).cuda()
m(x)
torch.cuda.synchronize()
tot = dict(layers=0, usable=0, points=0, mode2=0)
for key, f in sorted(m._folds.items(), key=lambda kv: kv[0][0]):
    if f is None:
        print(key, None); continue
    tot["layers"] += 1; tot["usable"] += f.usable; tot["points"] += f.points; tot["mode2"] += f.mode == 2
    if not f.usable or f.mode != 2:
        print(key, dict(folded=f.folded, channels=f.channels, mode=f.mode, res=f.shortcut_codes == 16, acc=[f.acc_lo, f.acc_hi]))
print(json.dumps(tot))
