#!/usr/bin/env python3
"""Headline benchmark: images/sec @ batch 4096, CIFAR-10 VGG full-qnn 4/4.

    python bench.py --gpus N --steps K --warmup W

One "step" = one forward pass of the fused low-bit pipeline over one synthetic
batch of 4096 CIFAR-shaped images per GPU (inputs resident in HBM before the
timed region), followed -- for N > 1 -- by the RCCL all-gather of the logits.
Rank 0 prints ONE JSON line (contract in the task statement) carrying, besides
the throughput, `roofline` (dominant kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the restated reference float path on the host cores).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PKG = "quantizedneuralnetworks-keras-tensorflow_amd"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "i8": 5000.0}   # dense matrix peaks (f32: spec; i8: ~5 POP/s dense)
BATCH = 4096
WORKLOADS = {
    "vgg64_full_qnn_w4a4": 2,   # BASELINE.json configs[2]: the config the metric is quoted on
    "vgg64_full_bnn": 1,
    "vgg_large_full_qnn_w8a8": 3,
    "imagenet224_resnet10_w4a4": 4,   # BASELINE.json configs[4]; ResidualFusedModel, 64 images / GPU
}


def step_bytes(st, N, H, W):
    """Algorithmic bytes of one fused step under traffic model M1:
    input as stored + output as stored (weights amortise to ~0 at N=4096)."""
    abi = importlib.import_module(PKG + "._abi")
    kh, kw, cin, cout = st["w"].shape
    if st["kind"] == "conv":
        Ho = abi.out_hw(H, kh, st["w"].stride, st["w"].same_pad) // st["pool"]
        Wo = abi.out_hw(W, kw, st["w"].stride, st["w"].same_pad) // st["pool"]
        pix_in, pix_out = N * H * W, N * Ho * Wo
    else:
        Ho = Wo = 1
        pix_in = pix_out = N
    def nbytes(store, pixels, ch):
        return pixels * ch * 4 if store == abi.STORE_F32 else pixels * abi.words(store, ch) * 4
    return nbytes(st["x_store"], pix_in, cin) + nbytes(st["out_store"], pix_out, cout), Ho, Wo


def cpu_baseline(cf, spec, seconds=12.0):
    """Restated reference float path (not TensorFlow) on the host cores."""
    base = importlib.import_module("oracle.cpu_baseline")
    return base.run(cf, spec, seconds=seconds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vgg64_full_qnn_w4a4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="replay the forward from a hipGraph")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches kept in flight: graphs replayed round-robin on as many streams (measured: 1 -> 20.8 M, 2 -> 22.0 M, 3 -> 21.5 M img/s)")
    ap.add_argument("--impl", default="auto", choices=["auto", "valu", "mfma"],
                    help="conv kernel family (results are bit-identical)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("QNN_DIST_BACKEND", "nccl")   # "gloo" only to rehearse N>1 on a 1-GPU box
    # QNN_BENCH_FORCE_DIST=1: run the N>1 code path (process group, logits all-gather) with a
    # single rank -- the only way to exercise the RCCL calls on a 1-GPU box
    use_dist = world > 1 or os.environ.get("QNN_BENCH_FORCE_DIST") == "1"
    json_fd = None
    if use_dist:
        # stdout carries exactly one JSON line, but RCCL writes its banner and warnings to file
        # descriptor 1: point fd 1 at stderr for the whole run and keep the real stdout for the result
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    pkg = importlib.import_module(PKG)
    nets, engine, shard, abi = pkg.nets, pkg.engine, pkg.shard, pkg._abi
    abi.set_conv_impl({"auto": 0, "valu": 1, "mfma": 2}[args.impl])
    idx = WORKLOADS[args.workload]
    cf = nets.baseline_config(idx)
    spec = nets.build_spec(cf, nets.SEED_BASE + idx)
    fused = idx != 4
    model = engine.FusedModel(spec) if fused else engine.ResidualFusedModel(spec)
    N = args.batch if fused or args.batch != BATCH else 64
    # every rank owns a full batch (weak scaling: per-GPU work fixed)
    x = torch.as_tensor(nets.synthetic_images(cf, N, nets.SEED_BASE + idx + 1000 * rank)).cuda()

    def step():
        y = model(x)
        if not use_dist:
            return y
        if backend == "gloo":                      # rehearsal path: gloo gathers host tensors
            return shard.gather_logits(y.cpu())
        return shard.gather_logits(y)

    # ---- per-kernel HIP-event timing (same stream the kernels are launched on) ----
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    H, W = cf.dim, cf.dim
    per_kernel = []
    cur = x
    hh, ww = H, W
    for st in (model.steps if fused else []):
        nbytes, ho, wo = step_bytes(st, N, hh, ww)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps, lead = 20, 4
        outs = None

        def launch():
            if st["kind"] == "conv":
                o, _, _ = abi.conv2d(st["w"], cur, st["x_store"], st["x_bits"], N, hh, ww, st["inv"],
                                     st["shift"], st["fn"], st["act_bits"], st["pool"], st["out_store"])
                return o
            return abi.dense(st["w"], cur, st["x_store"], st["x_bits"], N, st["inv"], st["shift"],
                             st["fn"], st["act_bits"], st["out_store"])

        torch.cuda.synchronize()
        # the first event is recorded BEHIND a few queued launches, so the host's launch latency
        # is not inside the measured interval: (ev1 - ev0) / reps is the kernel's own duration,
        # the figure rocprofv3's kernel trace reports
        for _ in range(lead):
            outs = launch()
        ev0.record()
        for _ in range(reps):
            outs = launch()
        ev1.record()
        torch.cuda.synchronize()
        per_kernel.append(dict(kernel=abi.last_kernel(), ms=ev0.elapsed_time(ev1) / reps, bytes=nbytes,
                               macs=N * (ho * wo * st["pool"] ** 2 if st["kind"] == "conv" else 1)
                               * st["w"].shape[0] * st["w"].shape[1] * st["w"].shape[2] * st["w"].shape[3]))
        cur, hh, ww = outs, ho, wo
    if not fused:
        # residual topology: ~200 launches per forward; report the whole forward against the
        # float32-surface (M0) bytes of SURVEY.md 8d instead of a single kernel
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            model(x)
        ev1.record()
        torch.cuda.synchronize()
        per_kernel.append(dict(kernel="residual_forward(all launches)", ms=ev0.elapsed_time(ev1) / 3,
                               bytes=30307912 * N, macs=6855277184 * N))   # M1 bytes/img, SURVEY.md 8d
    dom = max(range(len(per_kernel)), key=lambda i: per_kernel[i]["ms"])

    # ---- hipGraph of the model forward (launch-bound inner loop); the logits all-gather stays
    # outside the graph and runs on RCCL's own stream, overlapped with the next batch.
    # --inflight L keeps L batches in flight: L graphs (each with its own intermediate and
    # output tensors) replayed round-robin on L streams, so one batch's kernel tails, launch
    # boundaries and the graph-launch gap are filled by the other batch's kernels ----
    lanes = []                                 # per lane: dict(stream, graph, y)
    if args.graph:
        try:
            for _ in range(max(1, args.inflight)):
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(2):
                        model(x)
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    y_l = model(x)
                lanes.append(dict(stream=torch.cuda.Stream(), graph=g, y=y_l))
        except Exception as exc:  # pragma: no cover
            print("hipGraph capture failed (%s); running eagerly" % exc, file=sys.stderr)
            lanes = []
    graph = lanes[0]["graph"] if lanes else None
    y_static = lanes[0]["y"] if lanes else None
    torch.cuda.synchronize()

    # logits exchange (N > 1): the (B, classes) float32 block of this rank is copied to a staging
    # buffer of its lane and all-gathered asynchronously; the next forward does not wait for it, the
    # buffer is only reused after its gather has been waited for (stream-level wait, no host block)
    pipelined = use_dist and backend != "gloo" and graph is not None
    for ln in lanes:
        ln["work"] = None
        if pipelined:
            ln["stage"] = [torch.empty_like(ln["y"]) for _ in range(2)]
            ln["gathered"] = [torch.empty((world * ln["y"].shape[0],) + tuple(ln["y"].shape[1:]),
                                          dtype=ln["y"].dtype, device=ln["y"].device) for _ in range(2)]
            ln["works"] = [None, None]
            ln["count"] = 0
    counter = [0]

    def run_step():
        if graph is None:
            step()
            return
        ln = lanes[counter[0] % len(lanes)]
        counter[0] += 1
        with torch.cuda.stream(ln["stream"]):
            ln["graph"].replay()
            if not use_dist:
                return
            if not pipelined:
                shard.gather_logits(ln["y"].cpu() if backend == "gloo" else ln["y"])
                return
            k = ln["count"] & 1
            ln["count"] += 1
            if ln["works"][k] is not None:
                ln["works"][k].wait()
            ln["stage"][k].copy_(ln["y"])
            ln["works"][k] = dist.all_gather_into_tensor(ln["gathered"][k], ln["stage"][k], async_op=True)

    def drain():
        for ln in lanes:
            with torch.cuda.stream(ln["stream"]):
                for k in range(2):
                    if pipelined and ln["works"][k] is not None:
                        ln["works"][k].wait()
                        ln["works"][k] = None

    for _ in range(args.warmup):
        run_step()
    drain()
    # dominant-kernel events inside the timed region only make sense eagerly; with a
    # graph the per-kernel figure above (same launches, same stream) is reported
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    drain()                                        # every gather of the timed steps has completed ...
    torch.cuda.synchronize()                       # ... before the clock stops
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if pipelined and rank == 0:
        # the gathered block must hold this rank's logits at its own offset
        ln = lanes[(counter[0] - 1) % len(lanes)]
        nloc = ln["y"].shape[0]
        torch.testing.assert_close(ln["gathered"][(ln["count"] - 1) & 1][rank * nloc:(rank + 1) * nloc],
                                   ln["y"], rtol=0, atol=0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * N * args.steps / dt
        d = per_kernel[dom]
        hbm_gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                prefix = ("k_conv_first" if d["kernel"].startswith("mfma_f32_first") else
                          "k_conv_mfma" if d["kernel"].startswith("mfma_i") else
                          "k_conv_ps" if d["kernel"].startswith("ps_") else
                          "k_dense_packed" if d["kernel"].startswith("dense_") else "k_conv_generic")
                cands = [v for k, v in tj.items() if k.startswith(prefix)]
                ent = cands[0] if len(cands) == 1 else {}
                if ent:
                    # rocprofv3 FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts 64 of every
                    # 128 fetched bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM): doubled
                    traffic = (2.0 * ent.get("fetch_kb", 0.0) + ent.get("write_kb", 0.0)) * 1024.0
            except Exception:
                traffic = None
        if d["kernel"].startswith("mfma_"):
            kind = "f32" if d["kernel"].startswith("mfma_f32") else "i8"
            achieved = 2.0 * d["macs"] / (d["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": d["kernel"], "layer_index": dom, "achieved": achieved,
                    "peak": MFMA_PEAK_TFLOPS[kind], "unit": "TFLOP/s",
                    "frac": achieved / MFMA_PEAK_TFLOPS[kind], "traffic": traffic,
                    "mfma_dtype": kind, "algorithmic_flops_per_launch": 2.0 * d["macs"],
                    "algorithmic_bytes_per_launch": d["bytes"], "avg_launch_ms": d["ms"],
                    "hbm_GBps": hbm_gbs, "hbm_frac": hbm_gbs / HBM_PEAK_GBS}
        else:
            roof = {"bound": "hbm", "kernel": d["kernel"], "layer_index": dom, "achieved": hbm_gbs,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": d["bytes"], "avg_launch_ms": d["ms"],
                    "note": "VALU-bound packed kernel; see kernels[].TMACps and DESIGN.md"}
        m0_bytes = {1: 442408, 2: 442408, 3: 7237672, 4: 238248232}[idx]     # SURVEY.md 8d, float32-surface traffic per image
        out = {
            "metric": "images/sec @ batch 4096, CIFAR-10 VGG full-qnn 4/4; % HBM roofline",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {1: "u1", 2: "int4", 3: "int8", 4: "int4"}[idx],
            "data": "synthetic",
            "config": {"workload": args.workload, "batch_per_gpu": N, "global_batch": N * world,
                       "traffic_model": "M1 (packed inter-layer tensors)", "engine": "FusedModel" if fused else "ResidualFusedModel",
                       "conv_impl": args.impl, "hipgraph": graph is not None, "batches_in_flight": len(lanes) if lanes else 1,
                       "parallelism": "dp%d" % world,
                       # the metric's "% HBM roofline" in BASELINE.md's sense: float32-surface (M0) bytes
                       # per image x images/s over 8 TB/s (the fused engine does not move those bytes)
                       "pct_of_m0_hbm_roofline": 100.0 * value / world * m0_bytes / (HBM_PEAK_GBS * 1e9)},
            "roofline": roof,
            "kernels": [{"kernel": k["kernel"], "ms": round(k["ms"], 5),
                         "GBps": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 2),
                         "TMACps": round(k["macs"] / (k["ms"] * 1e-3) / 1e12, 3)} for k in per_kernel],
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(cf, spec)
            except Exception as exc:  # pragma: no cover
                out["cpu_baseline"] = {"error": str(exc)}
        line = json.dumps(out) + "\n"
        if json_fd is None:
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            os.write(json_fd, line.encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
