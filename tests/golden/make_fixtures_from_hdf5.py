"""Extract a few layers of the reference's trained Keras HDF5 checkpoints into
small .npz fixtures (inputs only: latent fp32 kernels, biases, BN statistics).

Run IN THE BUILD CONTAINER ONLY (needs h5py, which lives in the conda python):

    /opt/conda/bin/python3.9 tests/golden/make_fixtures_from_hdf5.py

Source files: /root/reference/results/RESNET3/weights_{bb,44,42,48}.hdf5
(Keras 2.1.3 full-model HDF5; read with h5py -- no pickle, nothing executed).
The fixtures carry DATA only; no reference source travels.  They were saved from
an older ResNet topology (use_bias=True on every conv, no Lambda(0.5)), which is
why biases are present.
"""
import json
import os
import sys

import h5py
import numpy as np

SRC = "/root/reference/results/RESNET3"
DST = os.path.dirname(os.path.abspath(__file__))

# (conv index, following BN index or None) -- covers 3->16 stem, 16->16,
# 16->32 stride 2, 1x1 projection, 32->64 stride 2, 64->64, dense 64->10.
PICKS = {
    "bb": [(1, 1), (2, 2), (8, 8), (10, None), (15, 14), (16, 15), (17, None), (21, 19)],
    "44": [(1, 1), (2, 2), (8, 8), (10, None), (15, 14), (16, 15), (17, None), (21, 19)],
    "42": [(1, 1), (2, 2), (10, None)],
    "48": [(1, 1), (2, 2), (10, None)],
}


def main():
    for code, picks in PICKS.items():
        path = os.path.join(SRC, "weights_%s.hdf5" % code)
        f = h5py.File(path, "r")
        mw = f["model_weights"]
        cfg = f.attrs["model_config"]
        if isinstance(cfg, bytes):
            cfg = cfg.decode()
        cfg = json.loads(cfg)
        lcfg = {l["config"]["name"]: l for l in cfg["config"]["layers"]}
        names = [n.decode() if isinstance(n, bytes) else n for n in mw.attrs["layer_names"]]
        conv_prefix = [n for n in names if n.endswith("conv2d_1")][0][: -len("_1")]
        dense_name = [n for n in names if "dense" in n][0]
        out = {}
        meta = {"source": "results/RESNET3/weights_%s.hdf5" % code,
                "keras_version": str(f.attrs["keras_version"]), "layers": {}}
        for ci, bi in picks:
            cname = "%s_%d" % (conv_prefix, ci)
            g = mw[cname]
            out["conv%d_kernel" % ci] = np.asarray(g[cname + "/kernel:0"], dtype=np.float32)
            out["conv%d_bias" % ci] = np.asarray(g[cname + "/bias:0"], dtype=np.float32)
            c = lcfg[cname]["config"]
            meta["layers"]["conv%d" % ci] = {
                "class": lcfg[cname]["class_name"], "strides": c["strides"],
                "padding": c["padding"], "klm": c.get("kernel_lr_multiplier"),
                "H": c.get("H"), "use_bias": c["use_bias"], "bn": bi}
            if bi is not None:
                bname = "batch_normalization_%d" % bi
                gb = mw[bname]
                for k in ("gamma", "beta", "moving_mean", "moving_variance"):
                    out["bn%d_%s" % (bi, k)] = np.asarray(gb["%s/%s:0" % (bname, k)], dtype=np.float32)
                meta["layers"]["bn%d" % bi] = {"epsilon": lcfg[bname]["config"]["epsilon"]}
        g = mw[dense_name]
        out["dense_kernel"] = np.asarray(g[dense_name + "/kernel:0"], dtype=np.float32)
        out["dense_bias"] = np.asarray(g[dense_name + "/bias:0"], dtype=np.float32)
        meta["layers"]["dense"] = {"class": lcfg[dense_name]["class_name"],
                                   "klm": lcfg[dense_name]["config"].get("kernel_lr_multiplier")}
        out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        dst = os.path.join(DST, "resnet3_%s.npz" % code)
        np.savez_compressed(dst, **out)
        print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    sys.exit(main())
