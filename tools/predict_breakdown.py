#!/usr/bin/env python3
"""Where nets.Model.predict's time goes on resident data (headline workload, 64 batches of 4096 float32 images): host time of the
launch loop, the flag read-back, the whole call; against the hipGraph replay loop bench.py times.  Usage: tools/predict_breakdown.py [lanes]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
nets, engine = pkg.nets, pkg.engine
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cf = nets.baseline_config(2)                 # BASELINE.json configs[2]: CIFAR-10 VGG-64 full-qnn 4/4
spec = nets.build_spec(cf, nets.SEED_BASE + 2)
m = nets.Model(cf, spec, lanes=lanes)
N, nb = 4096, 64
rng = np.random.default_rng(0)
xb = torch.as_tensor((rng.integers(0, 256, (8 * N, cf.dim, cf.dim, cf.channels), dtype=np.uint8).astype(np.float32) / np.float32(255))).cuda().repeat(nb // 8, 1, 1, 1)
pipe = m.pipeline(N)
for _ in range(2):
    m.predict(xb, batch_size=N)
torch.cuda.synchronize()
rows = {}
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.predict(xb, batch_size=N); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
rows["predict_ms"] = round(1e3 * float(np.median(ts)), 3)
# the launch loop alone (host time until the last launch is queued, then until the GPU is done)
bound = pipe._bound_lanes(xb[:N])
outs = torch.empty((xb.shape[0],) + bound[0]["yshape"][1:], dtype=bound[0]["ydtype"], device="cuda")
flags = torch.zeros(nb, dtype=torch.int32, device="cuda")
xp, xs = xb.data_ptr(), N * xb.stride(0) * 4
yp, ys = outs.data_ptr(), N * outs.stride(0) * outs.element_size()
fp = flags.data_ptr()
hs, gs = [], []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(nb):
        ln = bound[i % len(bound)]
        ln["plan"](ln["stream"].cuda_stream, xp + i * xs, yp + i * ys, fp + 4 * i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); gs.append(t2 - t0)
rows["loop_host_ms"] = round(1e3 * float(np.median(hs)), 3)
rows["loop_total_ms"] = round(1e3 * float(np.median(gs)), 3)
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.nonzero(flags).flatten().tolist(); ts.append(time.perf_counter() - t0)
rows["flag_readback_ms"] = round(1e3 * float(np.median(ts)), 3)
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = torch.empty_like(outs); f = torch.zeros(nb, dtype=torch.int32, device="cuda")
    ts.append(time.perf_counter() - t0)
rows["alloc_ms"] = round(1e3 * float(np.median(ts)), 3)
# hipGraph replay loop over the same batches copied into 8 rotating static inputs per lane (what bench.py's value times)
gl = pipe.lanes_for(xb[:N], 1, 8)
for li, ln in enumerate(gl):
    for j, xi in enumerate(ln["xs"]):
        xi.copy_(xb[((li * 8 + j) % 8) * N:((li * 8 + j) % 8 + 1) * N])
torch.cuda.synchronize()
gs = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(nb):
        ln = gl[i % len(gl)]
        with torch.cuda.stream(ln["stream"]):
            ln["graphs"][(i // len(gl)) % 8].replay()
    torch.cuda.synchronize(); gs.append(time.perf_counter() - t0)
rows["graph_loop_ms"] = round(1e3 * float(np.median(gs)), 3)
# fixed cost vs cost per batch: the same loops over 32 / 64 / 128 batches (wrapping around the resident tensor)
for name in ("bound", "graph"):
    for n_ in (32, 64, 128):
        gs = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(n_):
                if name == "bound":
                    ln = bound[i % len(bound)]
                    ln["plan"](ln["stream"].cuda_stream, xp + (i % nb) * xs, yp + (i % nb) * ys, fp + 4 * (i % nb))
                else:
                    ln = gl[i % len(gl)]
                    with torch.cuda.stream(ln["stream"]):
                        ln["graphs"][(i // len(gl)) % 8].replay()
            torch.cuda.synchronize(); gs.append(time.perf_counter() - t0)
        rows["%s_%d_ms" % (name, n_)] = round(1e3 * float(np.median(gs)), 3)
    rows[name + "_us_per_batch"] = round((rows[name + "_128_ms"] - rows[name + "_32_ms"]) / 96 * 1e3, 2)
    rows[name + "_fixed_us"] = round(rows[name + "_32_ms"] * 1e3 - 32 * rows[name + "_us_per_batch"], 1)
rows["lanes"] = lanes
print(json.dumps(rows))
