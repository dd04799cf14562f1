#!/usr/bin/env python3
"""Host-side sanitizer build of the C-ABI library: every translation unit compiled with
-fsanitize=address,undefined on the HOST half only (-fno-gpu-sanitize: the device code is untouched; GPU ASan is not
available on this pool).  Output: csrc/variants/libqnn_hip_asan.so (+ a stamp with the source hash).
Used by tests/test_abi_sanitized.py, which drives the argument-validation and no-device error paths through it in the
build container (no GPU needed)."""
import importlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd._build")
VDIR = os.path.join(b.CSRC, "variants")
LIB = os.path.join(VDIR, "libqnn_hip_asan.so")
STAMP = LIB + ".srchash"
SAN = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-fno-omit-frame-pointer", "-g"]


def asan_runtime():
    out = subprocess.run([b._hipcc().replace("hipcc", "../lib/llvm/bin/clang") if False else "/opt/rocm/lib/llvm/bin/clang",
                          "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == b.source_hash():
        return LIB
    os.makedirs(VDIR, exist_ok=True)
    flags = [f for f in b.CFLAGS if f != "-O3"] + ["-O1"] + SAN
    objs = []

    def one(src):
        obj = os.path.join(VDIR, os.path.basename(src)[:-4] + "_asan.o")
        subprocess.run([b._hipcc()] + flags + ["-c", src, "-o", obj], check=True, stderr=subprocess.DEVNULL)
        return obj

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
        objs = list(ex.map(one, b._sources()))
    subprocess.run([b._hipcc()] + b.LDFLAGS + SAN + ["-o", LIB] + objs, check=True)
    with open(STAMP, "w") as f:
        f.write(b.source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(asan_runtime())
