"""Topology builders for the benchmark configurations (callers of the hot path).

Counterpart of models/model_factory.py:18-72, models/vgg.py:5-44 and
models/resnet.py:15-147 plus the field names of utils/config_utils.py:10-46.
A network is described as a *net spec*: a plain list of op dicts holding numpy
arrays (Keras layouts).  The same spec is (a) interpreted by the CPU oracle,
(b) instantiated as Keras-compatible layer objects (LayerModel) and (c) compiled
into the fused packed pipeline (engine.FusedModel).
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

F32 = np.float32


@dataclass
class Config:
    """Field names follow utils/config_utils.py:10-46 / config/config*.py."""
    network_type: str = "full-qnn"      # float|qnn|full-qnn|bnn|qbnn|full-bnn
    wbits: int = 4
    abits: int = 4
    architecture: str = "VGG"           # VGG | RESNET
    dataset: str = "CIFAR-10"
    dim: int = 32
    channels: int = 3
    classes: int = 10
    nla: int = 1
    nfa: int = 64
    nlb: int = 1
    nfb: int = 64
    nlc: int = 1
    nfc: int = 64
    nres: int = 3
    pfilt: int = 1
    bits: Optional[int] = None

    def __post_init__(self):
        if self.bits is not None:       # config_utils.py:88-95
            self.wbits = self.abits = int(self.bits)


# The five BASELINE.json configurations (index = seed offset).
def baseline_config(index):
    if index == 0:   # MNIST full-bnn tiny
        return Config(network_type="full-bnn", architecture="VGG", dataset="MNIST", dim=28,
                      channels=1, nla=1, nlb=1, nlc=1, nfa=64, nfb=64, nfc=64)
    if index == 1:   # CIFAR-10 VGG full-bnn
        return Config(network_type="full-bnn", architecture="VGG")
    if index == 2:   # CIFAR-10 VGG full-qnn 4/4 (headline)
        return Config(network_type="full-qnn", wbits=4, abits=4, architecture="VGG")
    if index == 3:   # CIFAR-10 VGG-large full-qnn 8/8
        return Config(network_type="full-qnn", wbits=8, abits=8, architecture="VGG",
                      nla=3, nlb=3, nlc=3, nfa=256, nfb=256, nfc=256)
    if index == 4:   # ImageNet-224 ResNet nres=10 full-qnn 4/4
        return Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET",
                      dataset="IMAGENET-224", dim=224, nres=10, pfilt=1)
    raise ValueError(index)


def _layer_kinds(cf):
    """model_factory.py:24-61 -> (conv kind, conv nb, dense kind, dense nb, act op).

    Quirk kept from the reference: the Dense layer of a (full-)qnn gets
    nb=cf.abits, not cf.wbits (model_factory.py:31)."""
    nt = cf.network_type
    if nt == "float":
        conv, fc, act = ("float", None), ("float", None), {"op": "act", "fn": "leaky_relu", "alpha": 0.3}
    elif nt in ("qnn", "full-qnn"):
        conv, fc = ("quantized", cf.wbits), ("quantized", cf.abits)
        act = ({"op": "act", "fn": "leaky_relu", "alpha": 0.3} if nt == "qnn"
               else {"op": "act", "fn": "quantized_tanh", "nb": cf.abits})
    elif nt in ("bnn", "qbnn", "full-bnn"):
        conv, fc = ("binary", None), ("binary", None)
        if nt == "bnn":
            act = {"op": "act", "fn": "leaky_relu", "alpha": 0.3}
        elif nt == "qbnn":
            act = {"op": "act", "fn": "quantized_tanh", "nb": cf.abits}
        else:
            act = {"op": "act", "fn": "binary_tanh"}
    elif nt in ("tnn", "qtnn", "full-tnn"):      # model_factory.py:49-58
        conv, fc = ("ternary", None), ("ternary", None)
        if nt == "tnn":
            act = {"op": "act", "fn": "leaky_relu", "alpha": 0.3}
        elif nt == "qtnn":
            act = {"op": "act", "fn": "quantized_tanh", "nb": cf.abits}
        else:
            act = {"op": "act", "fn": "ternary_tanh"}
    else:
        raise ValueError("wrong network type, the supported network types in this repo are "
                         "float, qnn, full-qnn, bnn and full-bnn")
    return conv, fc, act


def _klm(kh, kw, cin, cout):
    return np.float32(1.0 / np.sqrt(1.5 / (int(cin * kh * kw) + int(cout * kh * kw))))


class _ParamGen:
    """Synthetic parameters, SURVEY.md 8d: latent kernels U(-1,1) (the reference's
    own initializer, binary_layers.py:136), biases N(0,0.05), BN gamma U(.5,1.5),
    beta N(0,.5), mean N(0,.1*sigma), variance ~ Var(conv output) estimated
    analytically so post-BN activations have about unit variance."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def kernel(self, shape):
        return self.rng.uniform(-1.0, 1.0, shape).astype(F32)

    def bias(self, n):
        return (self.rng.standard_normal(n) * 0.05).astype(F32)

    def bn(self, n, var_est):
        sigma = float(np.sqrt(var_est))
        return {"gamma": self.rng.uniform(0.5, 1.5, n).astype(F32),
                "beta": (self.rng.standard_normal(n) * 0.5).astype(F32),
                "mean": (self.rng.standard_normal(n) * 0.1 * sigma).astype(F32),
                "var": (var_est * self.rng.uniform(0.8, 1.25, n)).astype(F32)}


def _w_var(kind, nb):
    if kind == "binary":
        return 1.0
    if kind == "ternary":
        return 0.65                 # U(-1,1) weights, cutoff 0.35: P(|w|=1) ~ 0.65
    if kind == "quantized":
        m = 2.0 ** (nb - 1)
        return ((2 * m) ** 2 - 1) / 12.0 / (m * m)
    return 1.0 / 3.0


def _act_second_moment(act):
    if act is None:
        return 1.0 / 3.0            # raw image, U[0,1]: E[x^2] = 1/3
    if act["fn"] == "binary_tanh":
        return 1.0
    if act["fn"] == "quantized_tanh":
        return 0.45                 # ~N(0,1) clipped to [-1,1)
    if act["fn"] == "ternary_tanh":
        return 0.6
    return 0.6


def _conv_op(gen, kind, nb, kh, cin, cout, strides, use_bias, prev_act):
    op = {"op": "conv", "kind": kind, "kernel": gen.kernel((kh, kh, cin, cout)),
          "bias": gen.bias(cout) if use_bias else None, "strides": (strides, strides),
          "padding": "same", "klm": _klm(kh, kh, cin, cout)}
    if kind == "quantized":
        op["nb"] = int(nb)
    var = kh * kh * cin * _act_second_moment(prev_act) * _w_var(kind, nb)
    return op, var


def vgg_spec(cf, seed=0):
    """models/vgg.py:5-44."""
    (ck, cnb), (fk, fnb), act = _layer_kinds(cf)
    gen = _ParamGen(seed)
    spec = []
    cin, prev_act = cf.channels, None
    size = cf.dim

    def block(n_layers, filters):
        nonlocal cin, prev_act
        for _ in range(n_layers):
            op, var = _conv_op(gen, ck, cnb, 3, cin, filters, 1, True, prev_act)
            spec.append(op)
            spec.append(dict(op="bn", eps=1e-4, **gen.bn(filters, var)))
            spec.append(dict(act))
            cin, prev_act = filters, act

    block(1, cf.nfa)                    # vgg.py:15-17
    block(cf.nla - 1, cf.nfa)           # vgg.py:19-22
    spec.append({"op": "maxpool", "size": 2}); size //= 2
    block(cf.nlb, cf.nfb)
    spec.append({"op": "maxpool", "size": 2}); size //= 2
    block(cf.nlc, cf.nfc)
    spec.append({"op": "maxpool", "size": 2}); size //= 2
    spec.append({"op": "flatten"})
    k = size * size * cin
    dense = {"op": "dense", "kind": fk, "kernel": gen.kernel((k, cf.classes)),
             "bias": gen.bias(cf.classes)}
    if fk == "quantized":
        dense["nb"] = int(fnb)
    spec.append(dense)
    var = k * _act_second_moment(prev_act) * _w_var(fk, fnb)
    spec.append(dict(op="bn", eps=1e-4, **gen.bn(cf.classes, var)))
    return spec


def resnet_spec(cf, seed=0):
    """models/resnet.py:72-144 (ResNet v1, depth 6n+2, use_bias=False, 0.5*(x+y))."""
    (ck, cnb), (fk, fnb), act = _layer_kinds(cf)
    gen = _ParamGen(seed)
    spec = []
    uid = [0]

    def name(prefix):
        uid[0] += 1
        return "%s%d" % (prefix, uid[0])

    def resnet_layer(src, cin, filters, ksize=3, strides=1, bn=True, activation=True, prev_act=None):
        op, var = _conv_op(gen, ck, cnb, ksize, cin, filters * cf.pfilt, strides, False, prev_act)
        op["src"] = src
        op["dst"] = name("c")
        spec.append(op)
        last = op["dst"]
        if bn:
            b = dict(op="bn", eps=1e-3, src=last, dst=name("b"), **gen.bn(filters * cf.pfilt, var))
            spec.append(b)
            last = b["dst"]
        if activation:
            a = dict(act)
            a.update(src=last, dst=name("a"))
            spec.append(a)
            last = a["dst"]
        return last

    src = "input"
    size = cf.dim
    if cf.dataset in ("MNIST", "FASHION"):      # resnet.py:101-102
        spec.append({"op": "zeropad", "pad": 2, "src": "input", "dst": "padded"})
        src = "padded"
        size += 4
    num_filters = 16
    x = resnet_layer(src, cf.channels, num_filters, prev_act=None)
    cin = num_filters * cf.pfilt
    for stack in range(3):
        for res_block in range(cf.nres):
            strides = 2 if (stack > 0 and res_block == 0) else 1
            y = resnet_layer(x, cin, num_filters, strides=strides, prev_act=act)
            y = resnet_layer(y, num_filters * cf.pfilt, num_filters, activation=False, prev_act=act)
            if stack > 0 and res_block == 0:
                x = resnet_layer(x, cin, num_filters, ksize=1, strides=strides, bn=False,
                                 activation=False, prev_act=act)
                size //= 2
            s = name("s")
            spec.append({"op": "add", "a": x, "b": y, "dst": s})
            h = name("h")
            spec.append({"op": "scale", "value": 0.5, "src": s, "dst": h})
            a = dict(act)
            a.update(src=h, dst=name("a"))
            spec.append(a)
            x = a["dst"]
            cin = num_filters * cf.pfilt
        num_filters *= 2
    spec.append({"op": "avgpool", "size": 8, "src": x, "dst": "pooled"})
    size //= 8
    spec.append({"op": "flatten", "src": "pooled", "dst": "flat"})
    k = size * size * cin
    dense = {"op": "dense", "kind": fk, "kernel": gen.kernel((k, cf.classes)), "bias": None,
             "src": "flat", "dst": "logits"}
    if fk == "quantized":
        dense["nb"] = int(fnb)
    spec.append(dense)
    spec.append({"op": "softmax", "src": "logits", "dst": "probs"})
    return spec


def build_spec(cf, seed=0):
    """model_factory.py:63-68."""
    if cf.architecture == "VGG":
        return vgg_spec(cf, seed)
    if cf.architecture == "RESNET":
        return resnet_spec(cf, seed)
    raise ValueError("Error: type " + str(cf.architecture) + " is not supported")


class Model:
    """What models/model_factory.py:18-72 build_model(cf) hands to its callers (train.py:163-169,
    test_resnet.py:63-81), reduced to the inference surface: predict / evaluate / summary /
    set of weights, executing on the fused GPU engines."""

    def __init__(self, cf, spec, device="cuda", first_layer="auto", lanes=None):
        """first_layer: kernel for float32 images ("auto" | "exact" | "image" | "fixed", engine.FusedModel; "auto", the
        default, takes every float tensor like the reference's call(): the byte kernel where a batch is image bytes /
        255, the exact kernel where it is not); uint8 images always take the typed QNN_STORE_U8 entry.  lanes: batches
        kept in flight by predict() (engine.Pipelined; default 2 for the chain engine, 3 for the residual engine, whose
        kernels run at one to three waves per SIMD and leave room for a third batch: +5 % on the ImageNet-224 ResNet)."""
        from . import engine, _abi
        self.cf, self.spec = cf, spec
        try:
            self.engine = engine.FusedModel(spec, device, first_layer=first_layer)     # chains (VGG)
        except _abi.NotFusable:                                     # residual / non-fusable topologies
            self.engine = engine.ResidualFusedModel(spec, device,
                                                    first_layer=first_layer if first_layer in ("auto", "image") else "exact")
        self.layers = [op for op in spec if op["op"] in ("conv", "dense")]
        self.lanes = int(lanes) if lanes is not None else (2 if isinstance(self.engine, engine.FusedModel) else 3)
        self.upload_batches = 8              # predict() on a host array: batches uploaded (and resident) at a time
        self._pipes = {}

    def pipeline(self, batch_size=4096):
        """The hipGraph pipeline predict() runs on (engine.Pipelined), one per batch size."""
        from . import engine
        if batch_size not in self._pipes:
            self._pipes[batch_size] = engine.Pipelined(self.engine, lanes=self.lanes, batch_size=batch_size)
        return self._pipes[batch_size]

    def predict(self, x, batch_size=4096):
        """Keras `model.predict`: images (N, H, W, C) -> outputs (N, classes).

        x may be a numpy array -- uint8 (the dataset's bytes: uploaded as bytes, /255 inside the first layer) or
        anything else (converted to float32) -- and then a numpy array comes back; or a CUDA tensor (uint8 / float32),
        already resident in HBM, and then a CUDA tensor comes back without any host synchronisation.  Full batches are
        replayed from hipGraphs with `lanes` batches in flight; a ragged tail runs eagerly."""
        import torch
        resident = isinstance(x, torch.Tensor) and x.is_cuda
        restricted = hasattr(self.engine, "check_domain") and getattr(self.engine, "first_layer", "exact") in ("image", "fixed")
        if resident:
            y = self.pipeline(batch_size)(x)
            if restricted:
                self.engine.check_domain()   # restricted-domain first layer: never hand out results unchecked
            return y
        # host arrays are uploaded in chunks of a few batches (device memory bounded by the chunk, not by the dataset,
        # like Keras' predict(batch_size)); whole batches per chunk, so batch-dependent activations see the same batches
        a = np.asarray(x)
        a = np.ascontiguousarray(a if a.dtype == np.uint8 else a.astype(F32, copy=False))
        chunk = max(1, int(self.upload_batches)) * int(batch_size)
        outs = []
        for i in range(0, max(len(a), 1), chunk):
            xc = torch.from_numpy(a[i:i + chunk]).cuda()
            yc = self.pipeline(batch_size)(xc)
            if restricted:
                self.engine.check_domain()
            outs.append(yc.cpu())
        return torch.cat(outs).numpy()

    def evaluate(self, x, y, batch_size=4096):
        """Top-1 accuracy; y is one-hot (or +-1 hinge targets, utils/load_data.py:84-88) or class ids."""
        p = self.predict(x, batch_size)
        y = np.asarray(y)
        labels = y.argmax(-1) if y.ndim == 2 else y
        return float((p.argmax(-1) == labels).mean())

    def conv_output(self, x, number, batch_size=256):
        """Output of the `number`-th convolution (1-based, the order Keras names them conv2d_1, conv2d_2 ...), float32
        NHWC, computed layer by layer behind the Keras-compatible float32 surface: the intermediate model of
        test_resnet.py:70-81."""
        import torch
        from . import engine
        convs = [i for i, op in enumerate(self.spec) if op["op"] == "conv"]
        if not 1 <= number <= len(convs):
            raise ValueError("conv number %d outside 1..%d" % (number, len(convs)))
        if getattr(self, "_layer_model", None) is None:
            self._layer_model = engine.LayerModel(self.spec)
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=F32))
        outs = [self._layer_model.forward(x[i:i + batch_size].cuda(), upto=convs[number - 1]).cpu()
                for i in range(0, x.shape[0], batch_size)]
        return torch.cat(outs).numpy()

    def count_params(self):
        n = 0
        for op in self.spec:
            for k in ("kernel", "bias", "gamma", "beta", "mean", "var"):
                if op.get(k) is not None:
                    n += int(np.asarray(op[k]).size)
        return n

    def summary(self, print_fn=print):
        print_fn("%-4s %-10s %-28s %s" % ("#", "op", "detail", "params"))
        for i, op in enumerate(self.spec):
            d = ""
            if op["op"] in ("conv", "dense"):
                d = "%s %s%s" % (op["kind"], tuple(op["kernel"].shape), " nb=%d" % op["nb"] if "nb" in op else "")
            elif op["op"] == "act":
                d = op["fn"] + (" nb=%d" % op["nb"] if "nb" in op else "")
            p = sum(int(np.asarray(op[k]).size) for k in ("kernel", "bias", "gamma", "beta", "mean", "var")
                    if op.get(k) is not None)
            print_fn("%-4d %-10s %-28s %d" % (i, op["op"], d, p))
        print_fn("Total params: %d   engine: %s" % (self.count_params(), type(self.engine).__name__))


def build_model(cf, seed=0, device="cuda", first_layer="auto", lanes=None):
    """model_factory.py:18-72: config -> model (synthetic weights; use spec_from_keras_npz +
    Model(cf, spec) to run an imported checkpoint)."""
    return Model(cf, build_spec(cf, seed), device, first_layer=first_layer, lanes=lanes)


def activation_range_probe(model, x, number, limit=63.0, batch_size=256):
    """The check of test_resnet.py:70-89: does any value of the `number`-th convolution's output exceed `limit` in
    magnitude (63 = what a 7-bit accumulator holds)?  Returns {"count", "max_abs", "first"} with `first` the
    (element, value) pair the reference's loop would print first, or None."""
    y = model.conv_output(x, number, batch_size)
    a = np.abs(y)
    over = a > limit
    first = None
    if over.any():
        idx = np.argwhere(over)[0]                  # C order = the reference's elem / i / j / k loop nest
        first = (int(idx[0]), float(a[tuple(idx)]))
    return {"count": int(over.sum()), "max_abs": float(a.max()) if a.size else 0.0, "first": first}


class Dataset:
    """utils/load_data.py:38-42: X = uint8 / 255 as float32 (with a trailing channel axis for grey images), y as is."""

    def __init__(self, dset, add_dim=False):
        norm = np.asarray(dset[0]).astype("float32") / 255
        self.X = norm.reshape(norm.shape + (1,)) if add_dim else norm
        self.y = np.asarray(dset[1])


def load_dataset(dataset, path, architecture="VGG", classes=10):
    """utils/load_data.py:45-95 without the download: `path` is an .npz with x_train / y_train / x_test / y_test (the
    layout of Keras' own `mnist.npz`; CIFAR-10 converted to the same four arrays).  Returns (train, valid, test) with
    the reference's split (45 000 / 50 000 training images), its /255 normalisation, labels as one-hot rows
    (load_data.py:79-82), mapped to +-1 for the VGG nets' hinge loss (84-88)."""
    sizes = {"CIFAR-10": (45000, False), "MNIST": (50000, True), "FASHION": (50000, True)}
    if dataset not in sizes:
        raise ValueError(str(dataset) + " is not supported")
    size, add_dim = sizes[dataset]
    with np.load(path, allow_pickle=False) as z:
        tr = (z["x_train"], z["y_train"])
        te = (z["x_test"], z["y_test"])
    size = min(size, len(tr[0]))
    sets = [Dataset((tr[0][:size], tr[1][:size]), add_dim), Dataset((tr[0][size:], tr[1][size:]), add_dim),
            Dataset(te, add_dim)]
    for d in sets:
        lab = d.y.reshape(-1).astype(np.int64)
        onehot = np.zeros((lab.size, classes), dtype=F32)
        onehot[np.arange(lab.size), lab] = 1.0
        d.y = 2.0 * onehot - 1.0 if architecture == "VGG" else onehot
    return tuple(sets)


def synthetic_images_u8(cf, n, seed=0):
    """uint8 U{0..255} NHWC: the dataset's own bytes (what keras.datasets hands to utils/load_data.py:38)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, cf.dim, cf.dim, cf.channels), dtype=np.uint8)


def synthetic_images(cf, n, seed=0):
    """uint8 U{0..255} / 255 as float32 NHWC (utils/load_data.py:40)."""
    return (synthetic_images_u8(cf, n, seed).astype(F32) / F32(255)).astype(F32)


def spec_from_keras_npz(path, wbits=None, abits=None):
    """Rebuild a net spec from a Keras 2.1.3 checkpoint converted by tools/import_keras_hdf5.py
    (counterpart of test_resnet.py:44-66 build + load_weights).  The topology comes from the
    embedded model_config, so the shipped results/RESNET3/*.hdf5 files -- saved from an older
    ResNet variant with use_bias=True and without the 0.5 scaling -- load as they were trained.
    `nb` is not serialised by the reference (get_config omits it, quantized_layers.py:91-96):
    pass wbits / abits (the two-character code of the file name: '44' -> 4, 4)."""
    import json
    d = np.load(path)
    cfg = json.loads(bytes(d["model_config_json"]).decode())
    layers = cfg["config"]["layers"]
    spec = []
    first = True
    for l in layers:
        cls, c, name = l["class_name"], l["config"], l["name"]
        inbound = [ib[0] for ib in l["inbound_nodes"][0]] if l["inbound_nodes"] else []
        if cls == "InputLayer":
            spec_alias = name
            continue
        src = inbound[0] if inbound else None
        if src == spec_alias:
            src = "input"
        op = None
        if cls in ("QuantizedConv2D", "BinaryConv2D", "Conv2D"):
            kind = {"QuantizedConv2D": "quantized", "BinaryConv2D": "binary", "Conv2D": "float"}[cls]
            op = {"op": "conv", "kind": kind, "kernel": d[name + "/kernel"],
                  "bias": d[name + "/bias"] if c.get("use_bias", True) else None,
                  "strides": tuple(c["strides"]), "padding": c["padding"],
                  "klm": np.float32(c.get("kernel_lr_multiplier") or 1.0)}
            if kind == "quantized":
                op["nb"] = int(wbits)
        elif cls in ("QuantizedDense", "BinaryDense", "Dense"):
            kind = {"QuantizedDense": "quantized", "BinaryDense": "binary", "Dense": "float"}[cls]
            op = {"op": "dense", "kind": kind, "kernel": d[name + "/kernel"],
                  "bias": d[name + "/bias"] if c.get("use_bias", True) else None}
            if kind == "quantized":
                op["nb"] = int(abits)          # model_factory.py:31: Fc gets nb=cf.abits
        elif cls == "BatchNormalization":
            op = {"op": "bn", "eps": float(c["epsilon"]), "gamma": d[name + "/gamma"],
                  "beta": d[name + "/beta"], "mean": d[name + "/moving_mean"],
                  "var": d[name + "/moving_variance"]}
        elif cls == "Activation":
            fn = c["activation"]
            if fn == "quantized_relu":         # local name of quantize_op, model_factory.py:19-20
                op = {"op": "act", "fn": "quantized_tanh", "nb": int(abits)}
            elif fn in ("binary_tanh", "ternary_tanh"):
                op = {"op": "act", "fn": fn}
            else:
                raise ValueError("unsupported activation %r" % fn)
        elif cls == "LeakyReLU":
            op = {"op": "act", "fn": "leaky_relu", "alpha": float(c.get("alpha", 0.3))}
        elif cls == "Add":
            op = {"op": "add", "a": "input" if inbound[0] == spec_alias else inbound[0], "b": inbound[1]}
            src = None
        elif cls == "Lambda":
            op = {"op": "scale", "value": 0.5}
        elif cls == "AveragePooling2D":
            op = {"op": "avgpool", "size": int(c["pool_size"][0])}
        elif cls == "MaxPooling2D":
            op = {"op": "maxpool", "size": int(c["pool_size"][0])}
        elif cls == "ZeroPadding2D":
            op = {"op": "zeropad", "pad": int(c["padding"][0][0])}
        elif cls == "Flatten":
            op = {"op": "flatten"}
        else:
            raise ValueError("unsupported layer class %r" % cls)
        if src is not None:
            op["src"] = src
        op["dst"] = name
        spec.append(op)
        first = False
        if cls in ("QuantizedDense", "BinaryDense", "Dense") and c.get("activation") == "softmax":
            spec.append({"op": "softmax", "src": name, "dst": name + "_softmax"})
    return spec


SEED_BASE = 20240607
