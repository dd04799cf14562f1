#!/opt/conda/bin/python3.9
"""Keras 2.1.3 full-model HDF5 -> .npz (weights + model_config JSON).

Counterpart of the reference's checkpoint path (train.py:50 ModelCheckpoint, test_resnet.py:66
model.load_weights).  Needs h5py, which only the conda interpreter of the build container has:

    /opt/conda/bin/python3.9 tools/import_keras_hdf5.py <weights.hdf5> <out.npz>

HDF5 is read as plain data (datasets + attributes); nothing in the file is executed.  The
.npz holds one array per Keras weight ("<layer>/<weight>") plus "model_config_json" (uint8).
`nets.spec_from_keras_npz()` rebuilds the network from it.
"""
import json
import sys

import h5py
import numpy as np


def convert(src, dst):
    f = h5py.File(src, "r")
    cfg = f.attrs["model_config"]
    cfg = cfg.decode() if isinstance(cfg, bytes) else cfg
    mw = f["model_weights"]
    out = {"model_config_json": np.frombuffer(cfg.encode(), dtype=np.uint8),
           "keras_version": np.frombuffer(str(f.attrs["keras_version"]).encode(), dtype=np.uint8)}
    for lname in mw.attrs["layer_names"]:
        lname = lname.decode() if isinstance(lname, bytes) else lname
        g = mw[lname]
        for wname in g.attrs["weight_names"]:
            wname = wname.decode() if isinstance(wname, bytes) else wname
            key = wname.split(":")[0]                       # "<layer>/<weight>"
            out[key] = np.asarray(g[wname], dtype=np.float32)
    np.savez_compressed(dst, **out)
    return len(out) - 2


if __name__ == "__main__":
    n = convert(sys.argv[1], sys.argv[2])
    print("wrote %s (%d weight tensors)" % (sys.argv[2], n))
