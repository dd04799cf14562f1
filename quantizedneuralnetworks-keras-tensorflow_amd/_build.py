"""Build csrc/libqnn_hip.so with hipcc for gfx950 (in-tree, no JIT cache).

Every .hip file is compiled to its own object (in parallel) and the objects are linked into the
shared library.  Staleness is decided by CONTENT, not by mtime (a gpurun snapshot does not keep
mtimes): the library carries a stamp file holding the hash of all sources, headers and flags it
was built from; `needs_build()` / `_abi.load()` compare it with the sources present.
"""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libqnn_hip.so")
STAMP = os.path.join(CSRC, "libqnn_hip.srchash")
OBJDIR = os.path.join(CSRC, "build")
SOURCES = ["qnn_api.hip", "qnn_elementwise.hip", "qnn_conv.hip", "qnn_mfma.hip", "qnn_mfma_areg.hip",
           "qnn_mfma_small.hip", "qnn_mfma_strip.hip", "qnn_mfma_strip16.hip", "qnn_first.hip", "qnn_first_fixed.hip", "qnn_first_u8.hip", "qnn_stem.hip", "qnn_fold.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-Wno-cuda-compat",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build libqnn_hip.so")


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return hs + [os.path.join(ROOT, "include", "qnn_abi.h")]


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def source_hash():
    """Hash of everything the library is built from (sources, headers, flags)."""
    return _digest(_sources() + _headers(), CFLAGS[:5] + LDFLAGS)


def built_hash():
    try:
        return open(STAMP).read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or built_hash() != source_hash()


def build(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    headers = _headers()
    todo, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        tag = _digest([src] + headers, CFLAGS[:5])
        tagf = obj + ".hash"
        objs.append(obj)
        fresh = os.path.exists(obj) and os.path.exists(tagf) and open(tagf).read().strip() == tag
        if force or not fresh:
            todo.append((src, obj, tag, tagf))

    def compile_one(item):
        src, obj, tag, tagf = item
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(tagf, "w") as f:
            f.write(tag)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc] + LDFLAGS + ["-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
