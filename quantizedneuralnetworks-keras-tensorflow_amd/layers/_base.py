"""Shared machinery of the Keras-compatible low-bit layers.

The reference subclasses keras.layers.Dense / Conv2D; Keras is not a dependency
here, so this base restates the part of that surface the reference's callers
use (models/vgg.py:8-13, models/resnet.py:48-55,136-140): constructor kwargs,
build(input_shape), call(inputs), __call__, get_weights / set_weights with Keras
layouts ([kernel HWIO or (in,out), bias]), get_config, compute_output_shape.
"""
import math

import numpy as np
import torch

from .. import _abi


def _pair(v):
    if isinstance(v, (tuple, list)):
        assert len(v) == 2
        return (int(v[0]), int(v[1]))
    return (int(v), int(v))


def glorot_lr_multiplier(nb_input, nb_output):
    """binary_layers.py:129-132: np.float32(1. / np.sqrt(1.5 / (nb_input + nb_output)))."""
    return np.float32(1.0 / np.sqrt(1.5 / (nb_input + nb_output)))


def glorot_H(nb_input, nb_output):
    """binary_layers.py:123-126."""
    return np.float32(np.sqrt(1.5 / (nb_input + nb_output)))


class LowBitLayer:
    """Base of the four layer classes.  Subclasses set ``_wkind`` and may define ``nb``."""

    _wkind = _abi.W_FLOAT
    _counter = {}

    def _init_common(self, H, kernel_lr_multiplier, bias_lr_multiplier, kwargs, default_name):
        self.H = H
        self.kernel_lr_multiplier = kernel_lr_multiplier
        self.bias_lr_multiplier = bias_lr_multiplier
        self.use_bias = bool(kwargs.pop("use_bias", True))
        self.activation = kwargs.pop("activation", None)
        if isinstance(self.activation, str):
            if self.activation in ("linear",):
                self.activation = None
            elif self.activation == "softmax":
                self.activation = lambda t: torch.softmax(t, dim=-1)
            else:
                raise ValueError("unsupported activation %r" % self.activation)
        self.input_shape_arg = kwargs.pop("input_shape", None)
        self.kernel_initializer = kwargs.pop("kernel_initializer", None)  # overwritten in build
        self.bias_initializer = kwargs.pop("bias_initializer", "zeros")
        self.kernel_regularizer = kwargs.pop("kernel_regularizer", getattr(self, "kernel_regularizer", None))
        self.bias_regularizer = kwargs.pop("bias_regularizer", None)
        self.activity_regularizer = kwargs.pop("activity_regularizer",
                                               getattr(self, "activity_regularizer", None))
        self.kernel_constraint = kwargs.pop("kernel_constraint", None)
        self.bias_constraint = kwargs.pop("bias_constraint", None)
        n = LowBitLayer._counter.get(default_name, 0) + 1
        LowBitLayer._counter[default_name] = n
        self.name = kwargs.pop("name", "%s_%d" % (default_name, n))
        self.trainable = kwargs.pop("trainable", True)
        self.device = torch.device(kwargs.pop("device", "cuda"))
        self.seed = kwargs.pop("seed", None)
        self.built = False
        self.kernel = None
        self.bias = None
        self._packed = {}
        # Extension (not in the reference): what is known about the input tensor.
        #   None                      any float32 values (generic float kernel)
        #   "binary"                  values are exactly +-1        -> XNOR/popcount path
        #   ("quantized", nb)         values are k/2**(nb-1)        -> packed int path
        #   "ternary"                 values are exactly {-1,0,+1}  -> sign/mask planes (ternary layers) or int4
        #   ("binary_tanh",)          apply binary_tanh on load (fuses the Activation layer)
        #   ("quantized_tanh", nb)    apply quantized_tanh(nb) on load
        self.input_domain = kwargs.pop("input_domain", None)
        return kwargs

    # ---- weights ---------------------------------------------------------
    def _init_weights(self, kernel_shape, units):
        g = torch.Generator(device="cpu")
        if self.seed is not None:
            g.manual_seed(int(self.seed))
        else:
            g.seed()
        H = float(self.H)
        k = (torch.rand(kernel_shape, generator=g, dtype=torch.float32) * 2.0 - 1.0) * H
        self.kernel = k.to(self.device)
        self.bias = torch.zeros(units, dtype=torch.float32, device=self.device) if self.use_bias else None
        self.lr_multipliers = ([self.kernel_lr_multiplier, self.bias_lr_multiplier]
                               if self.use_bias else [self.kernel_lr_multiplier])
        self._packed = {}

    def get_weights(self):
        ws = [self.kernel.detach().cpu().numpy()]
        if self.use_bias:
            ws.append(self.bias.detach().cpu().numpy())
        return ws

    def set_weights(self, weights):
        expect = 2 if self.use_bias else 1
        if len(weights) != expect:
            raise ValueError("You called `set_weights(weights)` on layer \"%s\" with a weight list of "
                             "length %d, but the layer was expecting %d weights."
                             % (self.name, len(weights), expect))
        k = torch.as_tensor(np.asarray(weights[0], dtype=np.float32))
        if tuple(k.shape) != tuple(self.kernel.shape):
            raise ValueError("Layer weight shape %s not compatible with provided weight shape %s"
                             % (tuple(self.kernel.shape), tuple(k.shape)))
        self.kernel = k.to(self.device).contiguous()
        if self.use_bias:
            b = torch.as_tensor(np.asarray(weights[1], dtype=np.float32))
            if tuple(b.shape) != tuple(self.bias.shape):
                raise ValueError("Layer weight shape %s not compatible with provided weight shape %s"
                                 % (tuple(self.bias.shape), tuple(b.shape)))
            self.bias = b.to(self.device).contiguous()
        self._packed = {}   # re-quantize + re-pack lazily

    def count_params(self):
        n = self.kernel.numel()
        return n + (self.bias.numel() if self.use_bias else 0)

    def _wbits(self):
        return int(getattr(self, "nb", 1))

    def quantized_kernel(self):
        """The kernel the forward pass contracts with (binarize / quantize applied)."""
        return self._weights(_abi.STORE_F32).dequant()

    def _weights(self, store):
        w = self._packed.get(store)
        if w is None:
            stride = getattr(self, "strides", (1, 1))[0]
            same = getattr(self, "padding", "valid") == "same"
            w = _abi.Weights(self._wkind, self._wbits(), float(self.H), self.kernel, self.bias,
                             stride, same, store)
            self._packed[store] = w
        return w

    # ---- input domain -> (store, bits, fn) --------------------------------
    def _plan(self):
        dom = self.input_domain
        if dom is None or dom == "float":
            return None
        if self._wkind == _abi.W_BINARY:
            wbits_store = _abi.STORE_BIN
        elif self._wkind == _abi.W_TERNARY:
            wbits_store = _abi.STORE_I4          # codes {-1, 0, 1}
        else:
            wbits_store = _abi.store_for_bits(self._wbits())
        if self._wkind == _abi.W_TERNARY and (dom == "ternary" or dom == ("quantized", 1)):
            from .. import engine
            if engine.TERNARY_T2:          # values are {-1, 0, +1}: sign / mask planes against ternary weights
                return _abi.STORE_T2, 1, _abi.FN_GRID, 1
        if dom == "ternary":
            dom = ("quantized", 1)
        if dom == "binary" or dom == ("binary_tanh",) or dom == "binary_tanh":
            fn = _abi.FN_GRID if dom == "binary" else _abi.FN_BINARY_TANH
            store = _abi.STORE_BIN if wbits_store == _abi.STORE_BIN else wbits_store
            return store, 1, fn, 1
        kind, nb = dom
        nb = int(nb)
        fn = _abi.FN_GRID if kind == "quantized" else _abi.FN_QUANTIZED_TANH
        astore = _abi.store_for_bits(nb)
        store = max(astore, wbits_store) if wbits_store != _abi.STORE_BIN else astore
        return store, nb, fn, nb

    def __call__(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape))
        return self.call(inputs)

    def _base_config(self):
        return {"name": self.name, "trainable": self.trainable, "use_bias": self.use_bias,
                "kernel_regularizer": None, "bias_regularizer": None,
                "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}

    def _own_config(self):
        def py(v):
            return float(v) if isinstance(v, (np.floating, float)) else v
        # reference get_config (binary_layers.py:87-92,189-194) omits `nb`; it is
        # serialised here so a config round-trip rebuilds the same layer
        cfg = {"H": py(self.H), "kernel_lr_multiplier": py(self.kernel_lr_multiplier),
               "bias_lr_multiplier": py(self.bias_lr_multiplier)}
        if hasattr(self, "nb"):
            cfg["nb"] = self.nb
        return cfg


class LowBitDense(LowBitLayer):
    def _dense_init(self, units, H, kernel_lr_multiplier, bias_lr_multiplier, kwargs, name):
        self.units = int(units)
        kwargs = self._init_common(H, kernel_lr_multiplier, bias_lr_multiplier, dict(kwargs), name)
        if kwargs:
            raise TypeError("unexpected keyword arguments: %s" % sorted(kwargs))

    def build(self, input_shape):
        assert len(input_shape) >= 2
        input_dim = int(input_shape[1])
        if self.H == "Glorot":
            self.H = glorot_H(input_dim, self.units)
        if self.kernel_lr_multiplier == "Glorot":
            self.kernel_lr_multiplier = glorot_lr_multiplier(input_dim, self.units)
        self._init_weights((input_dim, self.units), self.units)
        self.input_dim = input_dim
        self.built = True

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.units)

    def call(self, inputs):
        x = _abi.require_cuda(inputs, self.name + ".call")
        if x.dim() != 2 or x.shape[1] != self.input_dim:
            raise ValueError("Input 0 is incompatible with layer %s: expected shape (None, %d), "
                             "found %s" % (self.name, self.input_dim, tuple(x.shape)))
        N = x.shape[0]
        plan = self._plan()
        if plan is None:
            y = _abi.dense(self._weights(_abi.STORE_F32), x, _abi.STORE_F32, 0, N)
        else:
            store, bits, fn, nb = plan
            xp = _abi.pack(x, self.input_dim, fn, nb, store)
            y = _abi.dense(self._weights(store), xp, store, bits, N)
        if self.activation is not None:
            y = self.activation(y)
        return y

    def get_config(self):
        cfg = self._base_config()
        cfg.update({"units": self.units})
        cfg.update(self._own_config())
        return cfg


class LowBitConv2D(LowBitLayer):
    def _conv_init(self, filters, kernel_regularizer, activity_regularizer, H,
                   kernel_lr_multiplier, bias_lr_multiplier, kwargs, name):
        self.filters = int(filters)
        kwargs = dict(kwargs)
        if "kernel_size" not in kwargs:
            raise TypeError("__init__() missing 1 required positional argument: 'kernel_size'")
        self.kernel_size = _pair(kwargs.pop("kernel_size"))
        self.strides = _pair(kwargs.pop("strides", (1, 1)))
        self.padding = str(kwargs.pop("padding", "valid")).lower()
        self.data_format = kwargs.pop("data_format", "channels_last") or "channels_last"
        self.dilation_rate = _pair(kwargs.pop("dilation_rate", (1, 1)))
        self.kernel_regularizer = kernel_regularizer
        self.activity_regularizer = activity_regularizer
        kwargs = self._init_common(H, kernel_lr_multiplier, bias_lr_multiplier, kwargs, name)
        if kwargs:
            raise TypeError("unexpected keyword arguments: %s" % sorted(kwargs))
        if self.padding not in ("same", "valid"):
            raise ValueError("The `padding` argument must be one of \"valid\", \"same\". Received: "
                             + self.padding)
        if self.data_format != "channels_last":
            raise _abi.QnnError("only data_format='channels_last' (NHWC) is supported")
        if self.dilation_rate != (1, 1):
            raise _abi.QnnError("dilation_rate != 1 is not supported")
        if self.strides[0] != self.strides[1]:
            raise _abi.QnnError("non-square strides are not supported")

    def build(self, input_shape):
        channel_axis = -1
        if input_shape[channel_axis] is None:
            raise ValueError("The channel dimension of the inputs should be defined. Found `None`.")
        input_dim = int(input_shape[channel_axis])
        kernel_shape = self.kernel_size + (input_dim, self.filters)
        base = self.kernel_size[0] * self.kernel_size[1]
        nb_input, nb_output = int(input_dim * base), int(self.filters * base)
        if self.H == "Glorot":
            self.H = glorot_H(nb_input, nb_output)
        if self.kernel_lr_multiplier == "Glorot":
            self.kernel_lr_multiplier = glorot_lr_multiplier(nb_input, nb_output)
        self._init_weights(kernel_shape, self.filters)
        self.input_dim = input_dim
        self.built = True

    def compute_output_shape(self, input_shape):
        n, h, w, _ = input_shape
        same = self.padding == "same"
        return (n, _abi.out_hw(h, self.kernel_size[0], self.strides[0], same),
                _abi.out_hw(w, self.kernel_size[1], self.strides[1], same), self.filters)

    def call(self, inputs):
        u8 = isinstance(inputs, torch.Tensor) and inputs.dtype == torch.uint8
        x = (_abi.require_cuda_u8(inputs, self.name + ".call") if u8
             else _abi.require_cuda(inputs, self.name + ".call"))
        if x.dim() != 4 or x.shape[-1] != self.input_dim:
            raise ValueError("Input 0 is incompatible with layer %s: expected axis -1 of input shape "
                             "to have value %d but got shape %s"
                             % (self.name, self.input_dim, tuple(x.shape)))
        N, H, W, C = x.shape
        plan = self._plan()
        if u8:
            # the dataset's own image bytes (extension; value = code / 255, utils/load_data.py:40): typed
            # QNN_STORE_U8 entry, exact integer sum and one rounding (include/qnn_abi.h)
            y, _, _ = _abi.conv2d(self._weights(_abi.STORE_F32), x, _abi.STORE_U8, 0, N, H, W)
        elif plan is None:
            y, _, _ = _abi.conv2d(self._weights(_abi.STORE_F32), x, _abi.STORE_F32, 0, N, H, W)
        else:
            store, bits, fn, nb = plan
            y, _, _ = _abi.conv2d_f32in(self._weights(store), x, fn, nb)
        if self.activation is not None:
            y = self.activation(y)
        return y

    def get_config(self):
        cfg = self._base_config()
        cfg.update({"filters": self.filters, "kernel_size": self.kernel_size, "strides": self.strides,
                    "padding": self.padding, "data_format": self.data_format,
                    "dilation_rate": self.dilation_rate})
        cfg.update(self._own_config())
        return cfg
