#!/usr/bin/env python3
"""Headline benchmark: images/sec @ batch 4096, CIFAR-10 VGG full-qnn 4/4.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--workload ...]

One "step" = one forward pass of the fused low-bit pipeline over one synthetic batch
(4096 CIFAR-shaped images per GPU in weak scaling; the 4096-image global batch cut into
contiguous shards in strong scaling), inputs resident in HBM before the timed region,
followed -- for N > 1 -- by the RCCL all-gather of the logits.

Launching.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a
launcher: before anything touches the GPU it starts N fresh child processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), relays rank 0's
JSON line and exits non-zero if any rank fails.  Under torchrun (WORLD_SIZE set) the process
is a rank; --gpus must then equal WORLD_SIZE.  Asking for more ranks than the box has GPUs
is an error unless --rehearse is given (all ranks on the GPUs that exist, gloo exchange).

Timing.  W warm-up steps, then the K-step region -- barrier + torch.cuda.synchronize() on both
sides, MAX over ranks -- is timed R >= 5 times (more until ~0.3 s have been timed, the region
of the default workload is only a few ms) and the MEDIAN region gives `ms_per_step` / `value`.

Rank 0 prints ONE JSON line carrying, besides the throughput, `roofline` (dominant kernel,
HIP-event timed on the stream it is launched on) and `cpu_baseline` (the restated reference
float path on the host cores, N = 1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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
M1_BYTES = {1: 17704, 2: 33832, 3: 1560616, 4: 30307912}       # packed inter-layer traffic per image
MACS = {1: 13576192, 2: 13576192, 3: 1781309440, 4: 6855277184}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vgg64_full_qnn_w4a4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per GPU; strong: --batch images in all, cut into contiguous shards")
    ap.add_argument("--repeats", type=int, default=5, help="minimum number of timed K-step regions (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="replay the forward from a hipGraph")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches kept in flight: graphs replayed round-robin on as many streams "
                         "(0 = per workload: 3 for vgg_large_full_qnn_w8a8, measured +2.5 %, and for imagenet224_resnet10_w4a4 "
                         "-- round 4: 80.6 / 84.7 / 85.0 K img/s with 2 / 3 / 4 since its kernels run at 1-3 waves per SIMD -- else 2)")
    ap.add_argument("--impl", default="auto", choices=["auto", "valu", "mfma"],
                    help="conv kernel family (results are bit-identical)")
    ap.add_argument("--first-layer", default="auto", choices=["auto", "exact", "image", "fixed", "u8"],
                    help="how the images enter: float32 = bytes / 255 with the kernel named (auto: the PRODUCT DEFAULT -- "
                         "byte kernel with a per-batch domain flag, a batch that is not bytes / 255 is recomputed on the "
                         "exact kernel; image: byte kernel, QnnError on other inputs; exact: float32 FMA chain; fixed: fixed "
                         "point for [0, 1]) or u8 = the bytes themselves through the typed QNN_STORE_U8 entry.  The other "
                         "entries are reported beside the headline")
    ap.add_argument("--rotate", type=int, default=8,
                    help="distinct input batches per lane the timed region rotates through (8 x 50 MB > the 256 MB "
                         "Infinity Cache: the images stream from HBM; 1 = replay one cache-resident batch, reported "
                         "beside the headline as `cache_resident`)")
    ap.add_argument("--no-targets", action="store_true", help="skip the north-star target block (layer ops, other workloads)")
    ap.add_argument("--no-alternatives", action="store_true",
                    help="skip the other first-layer entries and the Model.predict figure (profiling runs)")
    ap.add_argument("--gather-every", type=int, default=16,
                    help="N > 1: batches of a lane whose logits go into one RCCL all-gather (measured on the single-rank "
                         "RCCL path, headline workload: 1 -> 43.4 M, 4 -> 54.0 M, 16 -> 56.0 M img/s; no exchange 59.6 M)")
    ap.add_argument("--rehearse", action="store_true",
                    help="allow more ranks than GPUs (ranks share devices, logits exchanged over gloo)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# launcher: runs before anything touches the GPU
# ---------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU on this image


def launch_ranks(args, argv):
    """Start one fresh process per rank; returns the exit status for the launcher."""
    ndev = visible_gpus()
    env_extra = {}
    if ndev < args.gpus:
        if not args.rehearse:
            sys.stderr.write("bench.py: --gpus %d but this box has %d GPU(s).  Refusing to report n_gpus=%d from "
                             "fewer devices (pass --rehearse to run the ranks on the GPUs that exist with a gloo "
                             "exchange; such a run is a rehearsal of the code path, not a scaling measurement).\n"
                             % (args.gpus, ndev, args.gpus))
            return 2
        if ndev == 0:
            sys.stderr.write("bench.py: no GPU visible; the product path has no CPU fallback.\n")
            return 2
        env_extra["QNN_DIST_BACKEND"] = "gloo"
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env_extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    return supervise(procs, deadline_s=float(os.environ.get("QNN_BENCH_DEADLINE_S", "3600")))


def supervise(procs, deadline_s=3600.0, poll_s=0.05):
    """Wait for the ranks WITHOUT ever blocking on one of them: rank 0's stdout is drained on a thread while every
    child is polled.  The first rank that exits non-zero (or the deadline) kills the others -- they would sit in the
    rendezvous / a collective until c10d's timeout -- and its status becomes the launcher's, within a poll interval.
    Rank 0's last stdout line is relayed only when every rank exited 0."""
    import threading
    chunks = []

    def drain(pipe):
        for chunk in iter(lambda: pipe.read(65536), b""):
            chunks.append(chunk)

    reader = None
    if procs and procs[0].stdout is not None:
        reader = threading.Thread(target=drain, args=(procs[0].stdout,), daemon=True)
        reader.start()
    rc = 0
    deadline = time.time() + deadline_s
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
        if rc == 0 and live and time.time() > deadline:
            sys.stderr.write("bench.py: ranks still running after %.0f s; killing them\n" % deadline_s)
            rc = 124
        if rc != 0:
            for q in live:                    # one rank failed: the others would wait in a collective for ever
                q.kill()
            for q in live:
                try:
                    q.wait(timeout=10)
                except subprocess.TimeoutExpired:   # pragma: no cover
                    pass
            live = []
        elif live:
            time.sleep(poll_s)
    if reader is not None:
        reader.join(timeout=10)
    lines = [l for l in b"".join(chunks).decode(errors="replace").splitlines() if l.strip()]
    if rc == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------
def step_bytes(abi, st, N, H, W, x_elem_bytes=4):
    """Algorithmic bytes of one fused step under traffic model M1:
    input as stored + output as stored (weights amortise to ~0 at N=4096)."""
    kh, kw, cin, cout = st["w"].shape
    if st["kind"] == "conv":
        Ho = abi.out_hw(H, kh, st["w"].stride, st["w"].same_pad) // st["pool"]
        Wo = abi.out_hw(W, kw, st["w"].stride, st["w"].same_pad) // st["pool"]
        pix_in, pix_out = N * H * W, N * Ho * Wo
    else:
        Ho = Wo = 1
        pix_in = pix_out = N

    def nbytes(store, pixels, ch):
        return pixels * ch * x_elem_bytes if store == abi.STORE_F32 else pixels * abi.words(store, ch) * 4
    return nbytes(st["x_store"], pix_in, cin) + nbytes(st["out_store"], pix_out, cout), Ho, Wo


def time_launch(torch, launch, reps=20, lead=4, rounds=3):
    """Duration of one launch, HIP events on the launching stream, two ways:
      back-to-back   (ev1 - ev0) / reps around `reps` queued launches; the first event is recorded BEHIND a few queued
                     launches so the host's launch latency is not inside the interval.  Includes the kernel boundary
                     (with 33 MB of dirty output the write-back between two launches is ~6 us);
      per launch     an event pair around every launch, median of the intervals: the boundary falls between two
                     intervals.  Only meaningful while the host stays ahead of the GPU (long kernels).
    The smaller of the two is the kernel's own duration, the figure rocprofv3's kernel trace reports; the median over a
    few rounds is taken because the first round can still see the clock ramp.  Returns (ms, back_to_back_ms, output)."""
    out, agg, pair = None, [], []
    for _ in range(rounds):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for _ in range(lead):
            out = launch()
        ev0.record()
        for _ in range(reps):
            out = launch()
        ev1.record()
        torch.cuda.synchronize()
        agg.append(ev0.elapsed_time(ev1) / reps)
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for _ in range(lead):
            out = launch()
        for i in range(reps):
            starts[i].record()
            out = launch()
            ends[i].record()
        torch.cuda.synchronize()
        pair.append(sorted(a.elapsed_time(b) for a, b in zip(starts, ends))[reps // 2])
    b2b = sorted(agg)[len(agg) // 2]
    return min(b2b, sorted(pair)[len(pair) // 2]), b2b, out


def layer_roof_ms(k):
    """Roofline time of one launch: the slower of its HBM transfer and its matrix-pipe work."""
    peak = MFMA_PEAK_TFLOPS["f32" if k["pipe"] == "f32" else "i8"] * 1e12
    return max(k["bytes"] / (HBM_PEAK_GBS * 1e9), 2.0 * k["macs"] / peak) * 1e3


def load_traffic(tag, workload):
    """HBM bytes per launch of kernel `tag` from the committed PMC passes of `workload` (profiles/latest_traffic.json:
    one `by_tag` table per profiled workload, written by tools/summarize_profile.py; exact tag match, a tag that
    several profiled kernels share is not listed).  rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE
    counts 64 of every 128 fetched bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM): doubled -- except for the
    strip kernels, whose loads are 8 and 2 bytes per lane ("other access widths are uncalibrated: calibrate on a known
    byte count"): a 64 x 224^2 x 16 layer with packed shortcut must read between 51.4 MB and 60.4 MB and reports
    FETCH_SIZE 52.3 MB, factor 1.  Calibration of the first-layer kernels (profiles/r03): the uint8 entry must read
    12.58 MB of image bytes and reports 6 223 KiB (x2 = 12.7 MB), the float32 "image" entry 50.3 MB and reports
    24 660 KiB (x2 = 50.5 MB): factor 2 for byte and dword loads of consecutive lanes alike."""
    path = os.path.join(ROOT, "profiles", "latest_traffic.json")
    try:
        ent = json.load(open(path)).get("by_workload", {}).get(workload, {}).get(tag)
    except (OSError, ValueError):
        return None, None
    if not ent or "fetch_kb" not in ent or "write_kb" not in ent:
        return None, None
    factor = 1.0 if tag.startswith("strip_") else 2.0
    return (factor * ent["fetch_kb"] + ent["write_kb"]) * 1024.0, ent.get("rocprof_kernel")


def replay_rate(torch, lanes, steps, regions, images_per_step):
    """images/s of `steps` round-robin graph replays, median over `regions` regions (the first is a warm-up)."""
    import numpy as np
    times = []
    for rep in range(regions + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ln = lanes[i % len(lanes)]
            with torch.cuda.stream(ln["stream"]):
                ln["graph"].replay()
        torch.cuda.synchronize()
        if rep:
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return images_per_step * steps / dt, dt / steps * 1e3, len(times)


def measure_targets(torch, pkg, args, budget_s=40.0):
    """The north-star targets next to the headline, measured in the same run (rank 0, N = 1):
      xnor_m0_hbm_frac   1-bit XNOR layer ops behind the float32 Keras surface (traffic model M0: float32 NHWC in and
                         out), CIFAR B0 (16^2 x 64 -> 64) and C0 (8^2 x 64 -> 64) at batch 4096: algorithmic bytes /
                         HIP-event time / 8 TB/s.  Target >= 0.60.
      int8_mfma_frac     every int8 x int8 conv of VGG-large 8/8 at batch 4096: MACs / HIP-event time / 2500 T MAC/s
                         (= 2 x MACs / 5 POP/s); min and max over the layers.  Target >= 0.40 on every layer.
      images_per_s       one timed region (hipGraph, batches in flight) of the three other GPU workloads."""
    import numpy as np
    nets, engine, abi = pkg.nets, pkg.engine, pkg._abi
    t_start = time.time()
    out = {}
    # ---- M0 layer ops (tools/bench_layers.py, same code path) ----
    rng = np.random.default_rng(0)
    abi.set_conv_impl(abi.IMPL_VALU)
    try:
        for name, H in (("B0", 16), ("C0", 8)):
            n = args.batch
            x = torch.randn((n, H, H, 64), device="cuda")
            k = torch.as_tensor(rng.uniform(-1, 1, (3, 3, 64, 64)).astype(np.float32)).cuda()
            w = abi.Weights(abi.W_BINARY, 1, 1.0, k, torch.zeros(64, device="cuda"), 1, True, abi.STORE_BIN)
            ms, _, _ = time_launch(torch, lambda: abi.conv2d_f32in(w, x, abi.FN_BINARY_TANH, 1)[0], reps=20, rounds=2)
            kern = abi.last_kernel()
            m0 = 2.0 * n * H * H * 64 * 4
            out["xnor_m0_" + name] = {"kernel": kern, "ms": round(ms, 5), "m0_bytes": m0,
                                      "hbm_frac": round(m0 / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
            del x, w
    finally:
        abi.set_conv_impl({"auto": 0, "valu": 1, "mfma": 2}[args.impl])
    out["xnor_m0_hbm_frac"] = min(out["xnor_m0_B0"]["hbm_frac"], out["xnor_m0_C0"]["hbm_frac"])
    # ---- the other workloads: one region each; VGG-large also per layer ----
    rates = {}
    for wl, steps in (("vgg64_full_bnn", 50), ("vgg_large_full_qnn_w8a8", 6), ("imagenet224_resnet10_w4a4", 20)):
        if wl == args.workload or time.time() - t_start > budget_s:
            continue
        idx = WORKLOADS[wl]
        cf = nets.baseline_config(idx)
        spec = nets.build_spec(cf, nets.SEED_BASE + idx)
        fused = idx != 4
        # every workload takes the images the way the headline does (VGG-large's 3 -> 256 first layer and the ResNet
        # stem run on the un-pooled form of the byte kernel)
        model = engine.FusedModel(spec, first_layer=args.first_layer if args.first_layer != "u8" else "exact") \
            if fused else engine.ResidualFusedModel(spec, first_layer=args.first_layer if args.first_layer in ("auto", "image") else "exact")
        n = BATCH if fused else 64
        xi = nets.synthetic_images_u8(cf, n, nets.SEED_BASE + idx) if args.first_layer == "u8" \
            else nets.synthetic_images(cf, n, nets.SEED_BASE + idx)
        x = torch.as_tensor(xi).cuda()
        nl = 3 if wl in ("vgg_large_full_qnn_w8a8", "imagenet224_resnet10_w4a4") else 2
        lanes = engine.Pipelined(model, lanes=nl, batch_size=n).lanes_for(x)
        v, ms, _ = replay_rate(torch, lanes, steps, 2, n)
        rates[wl] = {"images_per_s": round(v, 1), "ms_per_step": round(ms, 4), "batch": n, "batches_in_flight": nl}
        if wl == "vgg_large_full_qnn_w8a8":
            fr = []
            cur, hh, ww = x, cf.dim, cf.dim
            for si, st in enumerate(model.steps):
                if st["kind"] != "conv":
                    break
                kh, kw, cin, cout = st["w"].shape
                ms_l, _, outs = time_launch(torch, lambda si=si, cur=cur, hh=hh, ww=ww: model.run_step(si, cur, n, hh, ww)[0],
                                            reps=6, lead=2, rounds=1)
                ho = abi.out_hw(hh, kh, st["w"].stride, st["w"].same_pad)
                if st["x_store"] == abi.STORE_I8:
                    macs = n * ho * ho * kh * kw * cin * cout
                    fr.append(macs / (ms_l * 1e-3) / 2500e12)
                cur, hh, ww = outs, ho // st["pool"], ho // st["pool"]
            out["int8_mfma_frac"] = {"min": round(min(fr), 4), "max": round(max(fr), 4), "layers": len(fr),
                                     "per_layer": [round(f, 4) for f in fr]}
        del lanes, model, x
        torch.cuda.empty_cache()
    out["images_per_s"] = rates
    out["seconds"] = round(time.time() - t_start, 1)
    return out


class RingSchedule:
    """Bookkeeping of ONE lane's two logits rings (N > 1): slot j of 2 G consecutive steps, ring k = j // G.  A ring is
    handed to one collective when its last slot has been written, and is rewritten only after that collective has been
    waited for.  Pure index arithmetic over two callables, so that tests can drive it without a process group:
        gather(k)  -> starts the collective on ring k, returns a handle
        wait(h)    -> blocks until the collective behind the handle is done
    `fresh[k]` = slots of ring k written since its last gather (what drain() sends of a partly filled ring is the whole
    ring, of which only these are new); `log` records ('wait' | 'gather' | 'drain', k, fresh) for the tests."""

    def __init__(self, G, gather, wait):
        self.G, self.gather, self.wait = int(G), gather, wait
        self.count = 0
        self.works = [None, None]
        self.fresh = [0, 0]
        self.log = []

    def begin_step(self):
        """-> (slot j in 0 .. 2G-1) the next step writes; waits for the ring's gather if the step starts rewriting it."""
        j = self.count % (2 * self.G)
        k = j // self.G
        self.count += 1
        if j % self.G == 0:
            if self.works[k] is not None:
                self.wait(self.works[k])
                self.works[k] = None
                self.log.append(("wait", k, self.fresh[k]))
            self.fresh[k] = 0
        return j

    def end_step(self, j):
        """the step wrote slot j; a full ring goes into its collective"""
        k = j // self.G
        self.fresh[k] = j % self.G + 1
        if j % self.G == self.G - 1:
            self.works[k] = self.gather(k)
            self.log.append(("gather", k, self.fresh[k]))

    def drain(self):
        """gather the partly filled ring (if any) and wait for everything outstanding; the next step starts a ring"""
        j = self.count % (2 * self.G)
        if j % self.G != 0:
            k = j // self.G
            if self.works[k] is not None:          # (cannot happen: begin_step waited when it entered the ring)
                self.wait(self.works[k])
            self.works[k] = self.gather(k)
            self.log.append(("drain", k, self.fresh[k]))
            self.count += self.G - (j % self.G)    # the ring counts as used up
        for k in range(2):
            if self.works[k] is not None:
                self.wait(self.works[k])
                self.works[k] = None


def main_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
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
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())       # single-rank rehearsal only
        if backend == "nccl" and torch.cuda.device_count() > 0:
            # bind the communicator to this rank's GPU up front (no "using the device under current context" guess)
            dev_id = torch.device("cuda", local_rank % torch.cuda.device_count())
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev_id)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    pkg = importlib.import_module(PKG)
    nets, engine, shard, abi = pkg.nets, pkg.engine, pkg.shard, pkg._abi
    abi.set_conv_impl({"auto": 0, "valu": 1, "mfma": 2}[args.impl])
    idx = WORKLOADS[args.workload]
    if args.inflight <= 0:
        args.inflight = 3 if args.workload in ("vgg_large_full_qnn_w8a8", "imagenet224_resnet10_w4a4") else 2
    cf = nets.baseline_config(idx)
    spec = nets.build_spec(cf, nets.SEED_BASE + idx)
    fused = idx != 4
    u8 = args.first_layer == "u8"
    if not fused and args.first_layer == "fixed":
        args.first_layer = "exact"                      # the residual engine's stem has no fixed-point kernel

    def make_model(first):
        if not fused:
            return engine.ResidualFusedModel(spec, first_layer=first if first in ("auto", "image") else "exact")
        return engine.FusedModel(spec, first_layer="exact" if first == "u8" else first)

    def make_input(first, n, seed):
        a = nets.synthetic_images_u8(cf, n, seed)       # the images ARE bytes; float32 forms are bytes / 255
        return a if first == "u8" else (a.astype(np.float32) / np.float32(255)).astype(np.float32)

    model = make_model(args.first_layer)
    batch = args.batch if fused or args.batch != BATCH else 64
    if args.scaling == "weak":
        # every rank owns a full batch: per-GPU work fixed as N grows
        N = batch
        x = torch.as_tensor(make_input(args.first_layer, N, nets.SEED_BASE + idx + 1000 * rank)).cuda()
        global_batch = N * world
    else:
        # the global batch cut into contiguous equal shards (north_star: "inference batches shard embarrassingly")
        xg = torch.as_tensor(make_input(args.first_layer, batch, nets.SEED_BASE + idx))
        x = shard.shard_batch(xg, rank, world).cuda()
        N = x.shape[0]
        global_batch = batch
        del xg

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
    per_kernel = []
    if fused:
        cur, hh, ww = x, cf.dim, cf.dim
        for si, st in enumerate(model.steps):
            kh, kw, cin, cout = st["w"].shape
            if si == len(model.steps) - 2 and model.run_head(cur, N, hh, ww) is not None:
                # the last conv group and the classifier run as ONE launch (qnn_conv2d_dense_forward): bytes = the conv's
                # packed input + the float32 logits, MACs of both layers
                dn = model.steps[-1]["w"].shape
                nb_in = step_bytes(abi, dict(st, out_store=abi.STORE_F32, pool=1), N, hh, ww)[0] - N * hh * ww * cout * 4
                ms, b2b, outs = time_launch(torch, lambda cur=cur, hh=hh, ww=ww: model.run_head(cur, N, hh, ww))
                per_kernel.append(dict(kernel=abi.last_kernel(), ms=ms, b2b_ms=b2b, bytes=nb_in + N * dn[3] * 4, launches=1,
                                       pipe="i8", macs=N * hh * ww * kh * kw * cin * cout + N * dn[2] * dn[3]))
                break
            nbytes, ho, wo = step_bytes(abi, st, N, hh, ww, x_elem_bytes=1 if (u8 and si == 0) else 4)
            ms, b2b, outs = time_launch(torch, lambda si=si, cur=cur, hh=hh, ww=ww: model.run_step(si, cur, N, hh, ww)[0])
            kname = abi.last_kernel()
            # the integer first-layer kernels (uint8 entry, "image", "fixed") run the problem's MACs on the int8 pipe:
            # priced like every other integer layer, so HBM bounds them; the exact first layer is an f32-pipe kernel
            per_kernel.append(dict(kernel=kname, ms=ms, b2b_ms=b2b, bytes=nbytes, launches=1,
                                   pipe="f32" if kname.startswith("mfma_f32") or kname.startswith("dense_f32") else "i8",
                                   macs=N * (ho * wo * st["pool"] ** 2 if st["kind"] == "conv" else 1)
                                   * kh * kw * cin * cout))
            cur, hh, ww = outs, ho, wo
    else:
        # residual topology: ~130 launches per forward.  Every launch is captured with its operands and
        # re-issued back to back for timing; launches of the same kernel on the same shape form one group
        model.capture = []
        model(x)
        torch.cuda.synchronize()
        groups = {}
        for c in model.capture:
            ms, _, _ = time_launch(torch, c["launch"], reps=8, lead=2, rounds=1)
            g = groups.setdefault((c["kernel"], c["shape"]), dict(kernel=c["kernel"], shape=c["shape"], ms_total=0.0,
                                                                  launches=0, bytes=c["bytes"], macs=c["macs"],
                                                                  pipe=c["pipe"]))
            g["ms_total"] += ms
            g["launches"] += 1
        model.capture = None
        for g in groups.values():
            per_kernel.append(dict(kernel=g["kernel"], shape=g["shape"], ms=g["ms_total"] / g["launches"],
                                   launches=g["launches"], bytes=g["bytes"], macs=g["macs"], pipe=g["pipe"]))
    dom = max(range(len(per_kernel)), key=lambda i: per_kernel[i]["ms"] * per_kernel[i]["launches"])

    # ---- the product's pipeline (engine.Pipelined: one hipGraph of the forward per lane, lanes replayed round-robin
    # on their own streams; what nets.Model.predict runs on).  bench.py only adds what belongs to the measurement: the
    # logits all-gather, which stays outside the graphs on RCCL's own stream, overlapped with the next batch ----
    # N > 1: the logits of G consecutive batches of a lane are gathered with ONE RCCL all-gather (G * 160 KB instead of G
    # latency-bound 160 KB collectives, and one collective call of host time per G steps instead of per step: with one
    # gather per 68 us step the host, not the GPU, sets the pace -- 43 M instead of 60 M img/s in a single-rank run of
    # the RCCL path, 56 M with G = 16).  The last kernel of a forward writes straight into its ring slot (one hipGraph
    # per slot), so a step is one graph launch and nothing else.  Two rings per lane alternate: a ring is only rewritten
    # after its gather has been waited for.
    G = max(1, args.gather_every)
    R = max(1, args.rotate)
    pipelined = use_dist and backend != "gloo" and bool(args.graph)
    lanes = []
    pipe = None
    if args.graph:
        try:
            pipe = engine.Pipelined(model, lanes=max(1, args.inflight), batch_size=N)
            if pipelined:
                lanes = [dict(ln) for ln in pipe.lanes_for(x, slots=2 * G, inputs=R)]
            else:
                lanes = [dict(ln) for ln in pipe.lanes_for(x, inputs=R)]
        except Exception as exc:  # pragma: no cover
            print("hipGraph capture failed (%s); running eagerly" % exc, file=sys.stderr)
            lanes = []
    graph = lanes[0]["graph"] if lanes else None
    pipelined = pipelined and graph is not None
    # every static input buffer of every lane gets its own synthetic batch: R x 50 MB per lane is more than the 256 MB
    # Infinity Cache, so the timed region reads its images from HBM
    for li, ln in enumerate(lanes):
        for ri, xi in enumerate(ln.get("xs", [])):
            if li == 0 and ri == 0:
                continue
            xi.copy_(torch.as_tensor(make_input(args.first_layer, N, nets.SEED_BASE + idx + 1000 * rank + 17 * (li * R + ri))))
        ln["nin"] = len(ln.get("xs", [])) or 1
        ln["rot"] = 0
    torch.cuda.synchronize()
    for ln in lanes:
        if pipelined:
            B = ln["y"].shape[0]
            ln["B"] = B
            ln["rings"] = [ln["ring"][:G * B], ln["ring"][G * B:]]
            ln["gathered"] = [torch.empty((world * G * B,) + tuple(ln["y"].shape[1:]), dtype=ln["y"].dtype,
                                          device=ln["y"].device) for _ in range(2)]
            ln["sched"] = RingSchedule(
                G, lambda k, ln=ln: dist.all_gather_into_tensor(ln["gathered"][k], ln["rings"][k], async_op=True),
                lambda h: h.wait())
            # the `--gather-every 1` figure beside the headline: one collective per batch, two small buffers alternate
            ln["g1"] = [torch.empty((world * B,) + tuple(ln["y"].shape[1:]), dtype=ln["y"].dtype, device=ln["y"].device)
                        for _ in range(2)]
            ln["w1"] = [None, None]
            ln["c1"] = 0
    counter = [0]
    rotate_on = [True]
    every1 = [False]

    def run_step():
        if graph is None:
            step()
            return
        ln = lanes[counter[0] % len(lanes)]
        counter[0] += 1
        with torch.cuda.stream(ln["stream"]):
            if not pipelined:
                r = ln["rot"] % ln["nin"] if rotate_on[0] else 0
                ln["rot"] += 1
                (ln["graphs"][r] if "graphs" in ln else ln["graph"]).replay()
                if use_dist:
                    yr = ln["ys"][r] if "ys" in ln else ln["y"]
                    shard.gather_logits(yr.cpu() if backend == "gloo" else yr)
                return
            sch = ln["sched"]
            if every1[0]:
                # one all-gather per batch (what north_star describes literally); the ring slots are only a place to write
                j = sch.count % (2 * G)
                sch.count += 1
                if ln["direct"]:
                    ln["graphs"][j].replay()
                else:
                    ln["graph"].replay()
                    ln["ring"][j * ln["B"]:(j + 1) * ln["B"]].copy_(ln["y"], non_blocking=True)
                k1 = ln["c1"] % 2
                ln["c1"] += 1
                if ln["w1"][k1] is not None:
                    ln["w1"][k1].wait()
                ln["w1"][k1] = dist.all_gather_into_tensor(ln["g1"][k1], ln["ring"][j * ln["B"]:(j + 1) * ln["B"]], async_op=True)
                return
            j = sch.begin_step()                   # slot inside the lane's two rings (waits before a ring is rewritten)
            if ln["direct"]:
                ln["graphs"][j].replay()           # the last kernel writes straight into slot j
            else:
                ln["graph"].replay()
                ln["ring"][j * ln["B"]:(j + 1) * ln["B"]].copy_(ln["y"], non_blocking=True)
            sch.end_step(j)

    def drain():
        """Gather what the timed steps left in a partly filled ring and wait for every gather.  Every rank is at the same
        count; the whole ring goes into the collective (one fixed-size all-gather), of which only the slots the steps of
        this region wrote are new -- the rest still hold what an earlier, already gathered round wrote there (zeros before
        the first).  `ln["fresh"]` records how many slots of each ring are current, for the check behind the region."""
        for ln in lanes:
            if not pipelined:
                continue
            with torch.cuda.stream(ln["stream"]):
                if every1[0]:
                    for k1 in range(2):
                        if ln["w1"][k1] is not None:
                            ln["w1"][k1].wait()
                            ln["w1"][k1] = None
                    ln["sched"].count += (-ln["sched"].count) % G      # the next ring-mode step starts a ring
                else:
                    ln["sched"].drain()

    def timed_region():
        """EXACTLY --steps steps between two (barrier + synchronize) brackets; MAX over ranks."""
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
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        run_step()
    drain()
    regions = []
    # the number of regions must be the same on every rank: rank 0 decides, from max-over-ranks times
    while True:
        regions.append(timed_region())
        more = len(regions) < args.repeats or (sum(regions) < 0.3 and len(regions) < 50)
        if world > 1:
            flag = torch.tensor([1 if more else 0], dtype=torch.int32, device="cpu" if backend == "gloo" else "cuda")
            dist.broadcast(flag, src=0)
            more = bool(flag.item())
        if not more:
            break
    dt = float(np.median(regions))
    # "image" / "fixed" / "auto": every input of the run was inside the byte kernel's domain (raises otherwise: a replay
    # loop, unlike Pipelined.forward, recomputes nothing)
    (pipe if pipe is not None else model).check_domain()

    # the cache-resident figure beside the headline: the same replay loop on ONE input batch per lane
    cache_resident = None
    if lanes and R > 1 and not pipelined:
        rotate_on[0] = False
        for _ in range(args.warmup):
            run_step()
        rr = [timed_region() for _ in range(max(3, args.repeats))]
        rotate_on[0] = True
        cache_resident = {"value": global_batch * args.steps / float(np.median(rr)), "unit": "images/s",
                          "note": "every lane replays ONE 50 MB input batch, which stays in the 256 MB Infinity Cache"}

    ring_check = None
    if pipelined and rank == 0:
        # a gathered block must hold this rank's ring at its own offset, bit for bit and finite; and a ring slot the
        # region wrote must be what the eager forward gives for that slot's input batch
        torch.cuda.synchronize()
        ln = lanes[0]
        nloc = G * ln["B"]
        for k in range(2):
            mine = ln["gathered"][k][rank * nloc:(rank + 1) * nloc]
            assert bool(torch.isfinite(mine).all()), "non-finite logits in the gathered ring"
            torch.testing.assert_close(mine, ln["rings"][k], rtol=0, atol=0)
        j0 = 0
        eager = model(ln["xs"][j0 % len(ln["xs"])] if "xs" in ln else ln["x"])
        torch.testing.assert_close(ln["gathered"][0][rank * nloc:rank * nloc + ln["B"]], eager, rtol=0, atol=0)
        ring_check = "gathered == ring (finite, bit for bit); slot 0 == eager forward of its input batch"

    gather_every_1 = None
    if pipelined and G > 1:
        every1[0] = True
        for _ in range(args.warmup):
            run_step()
        drain()
        r1 = [timed_region() for _ in range(max(3, args.repeats))]
        every1[0] = False
        gather_every_1 = {"value": global_batch * args.steps / float(np.median(r1)), "unit": "images/s",
                          "note": "one RCCL all-gather of the logits per batch instead of one per %d batches of a lane" % G}

    # which physical devices took part: every rank reports its own (uuid, PCI bus id, name); gathered on rank 0
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    me = {"rank": rank, "uuid": str(getattr(props, "uuid", "")), "name": props.name,
          "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                            getattr(props, "pci_device_id", 0)),
          "local_rank": local_rank, "host": socket.gethostname()}
    devices = [me]
    if use_dist:
        got = [None] * world
        dist.all_gather_object(got, me)
        devices = got
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = global_batch * args.steps / dt
        d = per_kernel[dom]
        hbm_gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        tkey = d["kernel"] + (":%s" % d["shape"][-1] if d.get("shape") and str(d["shape"][-1]).startswith("res_") else "")
        traffic, rocprof_name = load_traffic(tkey, args.workload + ("" if (args.first_layer == "exact" or not fused)
                                                                    else "+first_" + args.first_layer))
        common = {"kernel": d["kernel"], "rocprof_kernel": rocprof_name, "layer_index": dom,
                  "launches_per_step": d["launches"], "avg_launch_ms": d["ms"],
                  "back_to_back_launch_ms": d.get("b2b_ms"),
                  "algorithmic_bytes_per_launch": d["bytes"], "traffic": traffic}
        if d["kernel"].startswith("mfma_") and 2.0 * d["macs"] / (MFMA_PEAK_TFLOPS["f32" if d["pipe"] == "f32" else "i8"] * 1e12) \
                >= d["bytes"] / (HBM_PEAK_GBS * 1e9):
            kind = "f32" if d["pipe"] == "f32" else "i8"
            achieved = 2.0 * d["macs"] / (d["ms"] * 1e-3) / 1e12
            roof = dict(common, bound="mfma", achieved=achieved, peak=MFMA_PEAK_TFLOPS[kind], unit="TFLOP/s",
                        frac=achieved / MFMA_PEAK_TFLOPS[kind], mfma_dtype=kind,
                        algorithmic_flops_per_launch=2.0 * d["macs"], hbm_GBps=hbm_gbs,
                        hbm_frac=hbm_gbs / HBM_PEAK_GBS)
        else:
            roof = dict(common, bound="hbm", achieved=hbm_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=hbm_gbs / HBM_PEAK_GBS)
        # one end-to-end fraction: the sum of every launch's own roofline time over the measured step
        roof_ms = sum(layer_roof_ms(k) * k["launches"] for k in per_kernel)
        per_gpu_step_ms = ms_per_step
        roof["pipeline_roof_ms"] = roof_ms
        roof["pipeline_frac"] = roof_ms / per_gpu_step_ms
        m1_bytes = M1_BYTES[idx] - (3 * cf.dim * cf.dim * cf.channels if u8 else 0)     # bytes instead of float32 images
        out = {
            "metric": "images/sec @ batch 4096, CIFAR-10 VGG full-qnn 4/4; % HBM roofline",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": {1: "u1", 2: "int4", 3: "int8", 4: "int4"}[idx], "data": "synthetic",
            "config": {"workload": args.workload, "batch_per_gpu": N, "global_batch": global_batch,
                       "traffic_model": "M1 (packed inter-layer tensors)",
                       "engine": ("FusedModel" if fused else "ResidualFusedModel") + " under engine.Pipelined",
                       "conv_impl": args.impl, "first_layer": args.first_layer,
                       "input": "uint8 NHWC image bytes (QNN_STORE_U8)" if u8 else "float32 NHWC (bytes / 255)",
                       "first_layer_note": FIRST_LAYER_NOTE[args.first_layer],
                       "hipgraph": graph is not None, "batches_in_flight": len(lanes) if lanes else 1,
                       "parallelism": "dp%d" % world,
                       "dist_backend": backend if use_dist else None,
                       "rccl_world_size": (dist.get_world_size() if (use_dist and backend == "nccl") else (1 if not use_dist else 0)),
                       "devices": devices, "distinct_devices": len(set(d["uuid"] for d in devices)),
                       "ring_check": ring_check,
                       "input_batches_per_lane": R if lanes else 1,
                       "input_bytes_rotated_per_lane": (R if lanes else 1) * int(x.numel()) * x.element_size(),
                       "cache_resident": cache_resident, "gather_every_1": gather_every_1,
                       "logits_gather": ("one all-gather per %d batches of a lane, asynchronous" % G) if pipelined
                       else ("per batch" if use_dist else None),
                       "timed_regions": len(regions),
                       "region_ms_min_median_max": [round(min(regions) * 1e3, 4), round(dt * 1e3, 4),
                                                    round(max(regions) * 1e3, 4)],
                       # the metric's "% HBM roofline" for what the fused engine really moves: M1 bytes per
                       # image x images/s per GPU over 8 TB/s (the path is compute-bound, see roofline)
                       "m1_hbm_frac": value / world * m1_bytes / (HBM_PEAK_GBS * 1e9)},
            "roofline": roof,
            "kernels": [{"kernel": k["kernel"], "launches": k["launches"], "ms": round(k["ms"], 5),
                         "GBps": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 2),
                         "TMACps": round(k["macs"] / (k["ms"] * 1e-3) / 1e12, 3),
                         "roof_ms": round(layer_roof_ms(k), 5),
                         **({"shape": list(k["shape"])} if "shape" in k else {})} for k in per_kernel],
        }
        if world == 1 and fused and lanes and not use_dist and not args.no_alternatives:
            # the same workload through the other first-layer entries, same graphs-in-flight scheme and the same number
            # of steps per region (NOT the headline; `config.first_layer` names the one `value` was measured with)
            alts = {}
            for first in ("exact", "image", "fixed", "u8"):
                if first == args.first_layer or (first == "image" and args.first_layer == "auto"):
                    continue                                 # ("auto" on dataset images IS the byte kernel of "image")
                try:
                    m2 = make_model(first)
                    x2 = torch.as_tensor(make_input(first, N, nets.SEED_BASE + idx)).cuda()
                    m2.kernel_log = []
                    m2(x2)
                    k0 = m2.kernel_log[0]
                    m2.kernel_log = None
                    l2 = engine.Pipelined(m2, lanes=len(lanes), batch_size=N).lanes_for(x2)
                    v2, ms2, nreg = replay_rate(torch, l2, args.steps, max(3, args.repeats), global_batch)
                    m2.check_domain()
                    alts[first] = {"kernel": k0, "value": v2, "unit": "images/s", "ms_per_step": ms2,
                                   "timed_regions": nreg, "note": FIRST_LAYER_NOTE[first]}
                    del l2, m2, x2
                except Exception as exc:  # pragma: no cover
                    alts[first] = {"error": str(exc)}
            out["first_layer_alternatives"] = alts
            # the PRODUCT call on resident data: nets.Model.predict on a CUDA tensor of 64 batches (3.2 GB of float32
            # images, far more than the 256 MB Infinity Cache): engine.Pipelined's bound launch plans read every batch
            # in place and write the logits in place; the one host synchronisation is predict's domain check at the end
            try:
                mp = nets.Model(cf, spec, lanes=len(lanes)) if args.first_layer == "auto" else \
                    nets.Model(cf, spec, first_layer="exact" if u8 else args.first_layer, lanes=len(lanes))
                nb = 64 if N * cf.dim * cf.dim * cf.channels * 4 * 64 < 8e9 else 8
                xb = torch.cat([torch.as_tensor(make_input(args.first_layer, N, nets.SEED_BASE + idx + 31 * b)) for b in range(8)]
                               ).cuda().repeat(nb // 8, 1, 1, 1)
                # warm-up: the GPU has idled while the 3.2 GB of images were built on the host; its clocks need a few
                # hundred ms of load to come back (tools/predict_breakdown.py: the first ~30 ms of calls run 7 % slower)
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < 0.5:
                    mp.predict(xb, batch_size=N)
                    torch.cuda.synchronize()
                ts = []
                for _ in range(9):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    yb = mp.predict(xb, batch_size=N)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                pv = nb * N / float(np.median(ts))
                # the replay loop of `value` again, right behind the predict calls (same clocks, same minute): what the
                # product call costs over the bare graph replays; `ratio_to_value` is against the headline itself
                adj = None
                if world == 1 and lanes:
                    adj = global_batch * args.steps / float(np.median([timed_region() for _ in range(max(3, args.repeats))]))
                out["model_predict_resident"] = {"value": pv, "unit": "images/s", "images": nb * N,
                                                 "call": "nets.Model(cf, spec).predict(x)  (no arguments: the defaults)"
                                                 if args.first_layer == "auto" else "nets.Model(cf, spec, first_layer=%r)" % args.first_layer,
                                                 "calls_timed": len(ts), "ratio_to_value": pv / value,
                                                 "replay_loop_adjacent": adj,
                                                 "ratio_to_replay_loop_adjacent": (pv / adj) if adj else None}
                del mp, xb, yb
            except Exception as exc:  # pragma: no cover
                out["model_predict_resident"] = {"error": str(exc)}
        if args.workload == "vgg64_full_qnn_w4a4":
            # the first layer's codes on THESE 4096 images against the reference's own first conv group, as generated by
            # tests/golden/make_fixtures_from_reference.py and asserted on the GPU by tests/test_gpu_u8.py
            try:
                import numpy as _np
                zf = _np.load(os.path.join(ROOT, "tests", "golden", "ref_bench_first.npz"))
                bf = json.loads(bytes(zf["index_json"]).decode())["bench_first"]
                own = "exact" if args.first_layer in ("exact", "fixed") else "u8"
                out["first_layer_vs_reference"] = {
                    "codes": bf["codes"], "entry": args.first_layer,
                    "flips_vs_reference_numpy1_promotion": bf["flips_%s_vs_legacy" % own],
                    "flips_vs_reference_numpy2_promotion": bf["flips_%s_vs_nep50" % own],
                    "max_code_step": max(bf["maxabs_%s_vs_legacy" % own], bf["maxabs_%s_vs_nep50" % own]),
                    "flips_exact_float32_chain_vs_reference": [bf["flips_exact_vs_legacy"], bf["flips_exact_vs_nep50"]],
                    "flips_between_the_reference_promotions": bf["flips_nep50_vs_legacy"],
                    "source": "tests/golden/ref_bench_first.npz (reference run on the benchmark's 4096 images, rank 0 batch); "
                              "tests/test_gpu_u8.py::test_benchmark_first_layer_codes_against_the_reference_run asserts "
                              "these counts on the GPU (fixed-point entry not covered)"}
            except Exception as exc:  # pragma: no cover
                out["first_layer_vs_reference"] = {"error": str(exc)}
        if world == 1 and not use_dist and not args.no_targets and args.workload == "vgg64_full_qnn_w4a4":
            try:
                out["targets"] = measure_targets(torch, pkg, args)
            except Exception as exc:  # pragma: no cover
                out["targets"] = {"error": str(exc)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = importlib.import_module("oracle.cpu_baseline").run(cf, spec, seconds=12.0)
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


FIRST_LAYER_NOTE = {
    "exact": "float32 images, any values: float32 FMA chain on the f32 matrix pipe, bit-exact vs the oracle",
    "auto": "the product default: float32 images run on the byte kernel of 'image' with one domain-flag word per batch; a "
            "batch that is not image bytes / 255 is recomputed in-process on the exact kernel (any float tensor is accepted, "
            "as by the reference's call()); first-layer codes against the reference's own first conv + BN + activation on "
            "the 4096 benchmark images: see `first_layer_vs_reference` (tests/test_gpu_u8.py asserts the same counts)",
    "image": "float32 images that are bytes / 255 (utils/load_data.py:40): recognised as bytes (|255 x - k| <= 2^-15, else "
             "the layer's domain flag -> QnnError), exact integer sum on the int8 matrix pipe + one FMA; identical to the "
             "uint8 entry; see `first_layer_vs_reference` for the measured code differences",
    "fixed": "float32 images in [0, 1] (else domain flag -> QnnError): fixed point at 2^-23, three int8 digit passes",
    "u8": "uint8 image bytes through the typed QNN_STORE_U8 entry of the C ABI: exact integer sum + one FMA",
}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    main_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
