"""CPU baseline for bench.py -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT.

"Restated reference float path (not TensorFlow)": TensorFlow/Keras are absent from
this image and from the GPU box, so the reference's own CPU path cannot be timed.
This module replays, on PyTorch-CPU with every host core, exactly the op sequence
the reference's graph executes per Session.run (TF-op table):
weights fake-quantized every forward (binary_layers.py:161 / quantized_layers.py:165),
the lr-multiplier identity trick as three elementwise passes on the input and three
on the output (binary_layers.py:163-165,175-176), a float32 NHWC conv / matmul,
bias add, inference BN as mul+add, the activation clip as its chain of elementwise
ops, and max-pooling.  kind = "port".
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import qnn_oracle as O


def _round_through(x):
    return x + (torch.round(x) - x)          # torch.round is half-to-even


def _binary_tanh(x):
    h = torch.clamp(0.5 * x + 0.5, 0.0, 1.0)
    return 2.0 * _round_through(h) - 1.0


def _quantize(x, nb):
    m = float(2 ** (nb - 1))
    return torch.clamp(_round_through(x * m), -m, m - 1) / m


def _trick(x, c, s):
    return (x - c * x) * s


def forward(spec, x):
    """x: float32 NHWC torch CPU tensor.  Sequential specs (models/vgg.py) and residual ones (models/resnet.py:
    ops name their inputs with `src` / `a`, `b` and their output with `dst`)."""
    env = {"input": x}
    cur = x
    for i, op in enumerate(spec):
        k = op["op"]
        src = env[op["src"]] if "src" in op else cur
        if k == "conv":
            w = op["_w"]
            wq = _binary_tanh(w) if op["kind"] == "binary" else _quantize(w, op["nb"])
            c_in, s_in, c_out, s_out = op["_trick"]
            xin = _trick(src, c_in, s_in)
            st = tuple(op.get("strides", (1, 1)))
            xin = xin.permute(0, 3, 1, 2)
            if w.shape[0] == 3:                      # TF 'SAME': 1/1 at stride 1, 0/1 at stride 2 on even sizes
                pt, pb = O.same_padding(xin.shape[2], 3, st[0])[1:]
                pl, pr = O.same_padding(xin.shape[3], 3, st[1])[1:]
                xin = F.pad(xin, (pl, pr, pt, pb))
            y = F.conv2d(xin, wq.permute(3, 2, 0, 1), None, stride=st)
            y = _trick(y.permute(0, 2, 3, 1), c_out, s_out)
            cur = y + op["_b"] if op.get("bias") is not None else y
        elif k == "dense":
            w = op["_w"]
            wq = _binary_tanh(w) if op["kind"] == "binary" else _quantize(w, op["nb"])
            cur = src @ wq
            if op.get("bias") is not None:
                cur = cur + op["_b"]
        elif k == "bn":
            cur = src * op["_inv"] + op["_shift"]
        elif k == "act":
            cur = _binary_tanh(src) if op["fn"] == "binary_tanh" else _quantize(src, op["nb"])
        elif k == "maxpool":
            cur = F.max_pool2d(src.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        elif k == "avgpool":
            cur = F.avg_pool2d(src.permute(0, 3, 1, 2), op.get("size", 8)).permute(0, 2, 3, 1)
        elif k == "flatten":
            cur = src.reshape(src.shape[0], -1)
        elif k == "add":
            cur = env[op["a"]] + env[op["b"]]
        elif k == "scale":
            cur = src * float(op["value"])
        elif k == "softmax":
            cur = torch.softmax(src, dim=-1)
        elif k == "zeropad":
            p = op["pad"]
            cur = F.pad(src, (0, 0, p, p, p, p))
        else:
            raise ValueError(k)
        env[op.get("dst", "t%d" % i)] = cur
    return cur


def prepare(spec):
    out = []
    for op in spec:
        op = dict(op)
        if op["op"] in ("conv", "dense"):
            op["_w"] = torch.as_tensor(op["kernel"])
            if op.get("bias") is not None:
                op["_b"] = torch.as_tensor(op["bias"])
            if op["op"] == "conv":
                kh, kw, ci, co = op["kernel"].shape
                op["_trick"] = tuple(float(v) for v in O.trick_constants(O.glorot_klm(kh, kw, ci, co)))
        elif op["op"] == "bn":
            inv, shift = O.bn_constants(op["gamma"], op["beta"], op["mean"], op["var"], op["eps"])
            op["_inv"], op["_shift"] = torch.as_tensor(inv), torch.as_tensor(shift)
        out.append(op)
    return out


def _usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU
    quota (a GPU box hands each job a share of a much larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, int(os.environ.get("QNN_CPU_BASELINE_THREADS", "32"))))


def run(cf, spec, seconds=12.0, batch=None):
    cores = _usable_cores()
    torch.set_num_threads(cores)
    prepared = prepare(spec)
    if batch is None:                              # ~a second of work per call on a 16-core host
        batch = 256 if cf.dim <= 64 else 4
    rng = np.random.default_rng(1)
    x = torch.as_tensor((rng.integers(0, 256, (batch, cf.dim, cf.dim, cf.channels)).astype(np.float32)
                         / np.float32(255)))
    with torch.no_grad():
        forward(prepared, x)                      # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            forward(prepared, x)
            n += batch
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d images (batches of %d) of the same workload in %.1f s; restated reference "
                      "float path (not TensorFlow) on PyTorch-CPU, %s" % (n, batch, dt, cpu)}
