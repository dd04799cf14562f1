"""Loader for tests/golden/ref_*.npz -- vectors produced by running the reference's own
layers/*.py and models/*.py in place (tests/golden/make_fixtures_from_reference.py).
Shared by the CPU tests (oracle == fixtures) and the GPU tests (HIP == fixtures)."""
import json
import os

import numpy as np

from qnn_amd import nets

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


def _load(name):
    return np.load(os.path.join(GOLD, name))


def index():
    return json.loads(bytes(_load("ref_ops.npz")["index_json"]).decode())


def ops():
    return _load("ref_ops.npz")


def layer_cases():
    d = _load("ref_layers.npz")
    return d, index()["layers"]


def trained(case):
    """Kernel / bias of a layer case from the reference's trained checkpoint fixtures."""
    d = _load(case["file"] + ".npz")
    name = case["layer"]
    bias = d[name + "_bias"] if case.get("use_bias", True) else None
    return d[name + "_kernel"], bias


def net_names():
    return [n["tag"] for n in index()["nets"]]


PARAM_SLOTS = {"conv": ("kernel", "bias"), "dense": ("kernel", "bias"), "bn": ("gamma", "beta", "mean", "var")}
REF_NAME = {"mean": "moving_mean", "var": "moving_variance"}


def net(tag):
    """-> (cf, spec, x, y_ref, trace).  The spec's TOPOLOGY comes from the product's nets.build_spec;
    its parameters are replaced, in creation order, by the ones the reference's layers asked for while
    models/vgg.py / models/resnet.py built the network -- a mismatch in count, order, name or shape of any
    parameter is a topology difference and fails here."""
    d = _load("ref_models.npz")
    meta = [n for n in index()["nets"] if n["tag"] == tag][0]
    c = meta["cf"]
    cf = nets.Config(network_type=c["network_type"], wbits=c["wbits"], abits=c["abits"],
                     architecture=c["architecture"], dataset=c["dataset"], dim=c["dim"],
                     channels=c["channels"], classes=c["classes"], nla=c["nla"], nfa=c["nfa"],
                     nlb=c["nlb"], nfb=c["nfb"], nlc=c["nlc"], nfc=c["nfc"], nres=c["nres"], pfilt=c["pfilt"])
    spec = nets.build_spec(cf, seed=1)
    params = meta["params"]
    i = 0
    for op in spec:
        for slot in PARAM_SLOTS.get(op["op"], ()):
            if op.get(slot) is None:
                continue
            cls, name, shape = params[i]
            assert name == REF_NAME.get(slot, slot), (tag, i, op["op"], slot, params[i])
            w = d["%s_p%03d" % (tag, i)]
            if w.dtype == np.int16:
                w = (w.astype(F32) / F32(32768)).astype(F32)
            assert tuple(w.shape) == tuple(np.asarray(op[slot]).shape) == tuple(shape), (tag, i, slot, shape)
            op[slot] = w
            i += 1
    assert i == len(params), (tag, i, len(params))
    trace = []
    for j, cls in enumerate(meta["trace"]):
        key = "%s_c%03d" % (tag, j)
        if key in d.files:
            trace.append((cls, "codes", (d[key].astype(F32) / F32(128)).astype(F32)))
        else:
            trace.append((cls, "head", (d["%s_h%03d" % (tag, j)], d["%s_s%03d" % (tag, j)])))
    return cf, spec, d[tag + "_x"], d[tag + "_y"], trace


def align_trace(spec, trace):
    """Pair spec op indices with the reference's per-layer trace entries.  The reference's Dense carries
    its softmax inside the layer (resnet.py:136-140); the spec has it as a separate op."""
    pairs = []
    j = 0
    n = len(spec)
    for i, op in enumerate(spec):
        if op["op"] == "dense" and i + 1 < n and spec[i + 1]["op"] == "softmax":
            continue                     # pre-softmax value is not visible in the reference
        pairs.append((i, j))
        j += 1
    assert j == len(trace), (j, len(trace))
    expect = {"conv": ("Conv2D",), "dense": ("Dense",), "softmax": ("Dense",), "bn": ("BatchNormalization",),
              "act": ("Activation", "LeakyReLU"), "maxpool": ("MaxPooling2D",), "avgpool": ("AveragePooling2D",),
              "flatten": ("Flatten",), "add": ("Add",), "scale": ("Lambda",), "zeropad": ("ZeroPadding2D",)}
    for i, j in pairs:
        cls = trace[j][0]
        assert any(cls.endswith(e) for e in expect[spec[i]["op"]]), (i, spec[i]["op"], cls)
    return pairs
