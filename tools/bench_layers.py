#!/usr/bin/env python3
"""Layer-op micro-benchmark behind the Keras-compatible float32 surface (traffic model M0,
SURVEY.md 8d): float32 NHWC in -> [activation clip + pack] -> low-bit conv -> float32 NHWC out.
Reports each kernel's HIP-event time and its fraction of the HBM roofline (8 TB/s) on the
M0 algorithmic bytes 4*(N*H*W*Cin + N*Ho*Wo*Cout)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
abi = pkg._abi
HBM = 8000.0


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(os.environ.get("N", "4096"))
    rng = np.random.default_rng(0)
    rows = []
    layer_cases = [] if os.environ.get("CLIPS_ONLY") == "1" else None
    for name, (H, C, Cout), kind, nb, impl in layer_cases if layer_cases is not None else [
        ("B0 1-bit XNOR (16x16x64->64)", (16, 64, 64), "binary", 1, abi.IMPL_VALU),
        ("C0 1-bit XNOR (8x8x64->64)", (8, 64, 64), "binary", 1, abi.IMPL_VALU),
        ("B0 4-bit dot8 (16x16x64->64)", (16, 64, 64), "quantized", 4, abi.IMPL_VALU),
        ("B0 4-bit MFMA (16x16x64->64)", (16, 64, 64), "quantized", 4, abi.IMPL_MFMA),
        ("L 8-bit MFMA (32x32x256->256)", (32, 256, 256), "quantized", 8, abi.IMPL_MFMA),
    ]:
        n = N if C * H * H * N * 4 < 6e9 else 512
        abi.set_conv_impl(impl)
        x = torch.randn((n, H, H, C), device="cuda")
        k = torch.as_tensor(rng.uniform(-1, 1, (3, 3, C, Cout)).astype(np.float32)).cuda()
        b = torch.zeros(Cout, device="cuda")
        if kind == "binary":
            store, bits, fn, wk = abi.STORE_BIN, 1, abi.FN_BINARY_TANH, abi.W_BINARY
        else:
            store, bits, fn, wk = abi.store_for_bits(nb), nb, abi.FN_QUANTIZED_TANH, abi.W_QUANT
        w = abi.Weights(wk, nb, 1.0, k, b, 1, True, store)
        xp = abi.pack(x, C, fn, nb, store)
        t_pack = timeit(lambda: abi.pack(x, C, fn, nb, store))
        t_conv = timeit(lambda: abi.conv2d(w, xp, store, bits, n, H, H))
        kern = abi.last_kernel()
        t_fused = timeit(lambda: abi.conv2d_f32in(w, x, fn, nb))
        kern_fused = abi.last_kernel()
        in_b, out_b = n * H * H * C * 4, n * H * H * Cout * 4
        pk_b = xp.numel() * 4
        macs = n * H * H * 9 * C * Cout
        rows.append({
            "layer": name, "N": n, "conv_kernel": kern,
            "pack_ms": round(t_pack, 4), "pack_GBps": round((in_b + pk_b) / t_pack / 1e6, 1),
            "pack_hbm_frac": round((in_b + pk_b) / t_pack / 1e6 / HBM, 3),
            "conv_ms": round(t_conv, 4), "conv_GBps": round((pk_b + out_b) / t_conv / 1e6, 1),
            "conv_hbm_frac": round((pk_b + out_b) / t_conv / 1e6 / HBM, 3),
            "layer_ms": round(t_pack + t_conv, 4),
            "m0_GBps": round((in_b + out_b) / (t_pack + t_conv) / 1e6, 1),
            "m0_hbm_frac": round((in_b + out_b) / (t_pack + t_conv) / 1e6 / HBM, 3),
            "TMACps": round(macs / t_conv / 1e9, 1),
            "f32in_kernel": kern_fused, "f32in_ms": round(t_fused, 4),
            "f32in_m0_GBps": round((in_b + out_b) / t_fused / 1e6, 1),
            "f32in_m0_hbm_frac": round((in_b + out_b) / t_fused / 1e6 / HBM, 3)})
        print(json.dumps(rows[-1]))
    abi.set_conv_impl(abi.IMPL_AUTO)
    # elementwise clips (float32 -> float32)
    x = torch.randn((N, 32, 32, 64), device="cuda")
    for nm, f in (("binary_tanh", pkg.binary_tanh), ("quantized_tanh(4)", lambda t: pkg.quantized_tanh(t, 4))):
        t = timeit(lambda: f(x))
        print(json.dumps({"op": nm, "ms": round(t, 4), "GBps": round(2 * x.numel() * 4 / t / 1e6, 1),
                          "hbm_frac": round(2 * x.numel() * 4 / t / 1e6 / HBM, 3)}))


if __name__ == "__main__":
    main()
