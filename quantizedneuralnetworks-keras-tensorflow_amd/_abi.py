"""ctypes binding of include/qnn_abi.h (csrc/libqnn_hip.so).

The library is loaded eagerly and loudly: a missing or unloadable .so raises at
import time of any op that needs it (there is no CPU implementation to fall back
to).  Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``)
plus the current HIP stream of torch.
"""
import ctypes
import os

import torch

from . import _build

QNN_OK = 0
QNN_EUNSUPPORTED = -2
STORE_F32, STORE_BIN, STORE_T2, STORE_I4, STORE_I8, STORE_U8 = 0, 1, 2, 4, 8, 16
STORE_F32_IMAGE, STORE_F32_UNIT = 17, 18      # float32 input with a declared domain (first layer; qnn_abi.h)
W_FLOAT, W_BINARY, W_QUANT, W_TERNARY = 0, 1, 2, 3
FN_NONE, FN_BINARY_TANH, FN_QUANTIZED_TANH, FN_TERNARY_TANH, FN_GRID = 0, 1, 2, 3, 4

EXPORTS = [
    "qnn_version", "qnn_last_error", "qnn_last_kernel", "qnn_set_conv_impl",
    "qnn_binary_tanh_f32", "qnn_quantized_tanh_f32", "qnn_ternary_tanh_f32",
    "qnn_ternary_abs_sum_f32", "qnn_ternary_apply_f32",
    "qnn_packed_bytes", "qnn_pack_f32", "qnn_unpack_f32", "qnn_avgpool_packed_f32", "qnn_softmax_f32",
    "qnn_prepack_weights", "qnn_free_weights", "qnn_weights_dequant", "qnn_weights_check",
    "qnn_conv2d_forward", "qnn_dense_forward", "qnn_conv2d_forward_f32in", "qnn_conv2d_workspace_bytes",
    "qnn_conv2d_dense_forward",
    "qnn_fold_prepare", "qnn_fold_free", "qnn_fold_info", "qnn_fold_constants", "qnn_fold_eval",
]


class Projection(ctypes.Structure):
    """qnn_projection_t: the shortcut as a 1x1 strides-2 convolution of the block input, computed inside the launch."""
    _fields_ = [("w", ctypes.c_void_p), ("x", ctypes.c_void_p), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("x_bits", ctypes.c_int32)]


class Epilogue(ctypes.Structure):
    _fields_ = [("bn_inv", ctypes.c_void_p), ("bn_shift", ctypes.c_void_p),
                ("fn", ctypes.c_int32), ("act_bits", ctypes.c_int32),
                ("pool", ctypes.c_int32), ("out_store", ctypes.c_int32),
                ("res", ctypes.c_void_p), ("res_store", ctypes.c_int32), ("res_bits", ctypes.c_int32),
                ("post_scale", ctypes.c_float), ("trick_c", ctypes.c_float), ("trick_s", ctypes.c_float),
                ("fold", ctypes.c_void_p), ("flags", ctypes.c_uint32), ("domain_flag", ctypes.c_void_p),
                ("proj", ctypes.c_void_p)]


class FoldInfo(ctypes.Structure):
    _fields_ = [("channels", ctypes.c_int32), ("folded", ctypes.c_int32), ("usable", ctypes.c_int32),
                ("shortcut_codes", ctypes.c_int32), ("points", ctypes.c_int64),
                ("acc_lo", ctypes.c_int32), ("acc_hi", ctypes.c_int32), ("mode", ctypes.c_int32)]


class QnnError(RuntimeError):
    pass


class QnnUnsupported(QnnError):
    """QNN_EUNSUPPORTED from an optional fused form (a projection shortcut computed inside the launch): the caller keeps
    the form it had."""


class NotFusable(QnnError):
    """A net spec that engine.FusedModel cannot express as one packed chain (the caller picks another
    engine).  Every other failure -- prepack, out of memory, a planner bug -- stays a plain QnnError."""


_lib = None


def lib_path():
    # QNN_LIB: an alternative build of the same ABI (tools/build_variant.py, A/B kernel experiments only)
    return os.environ.get("QNN_LIB") or _build.LIB


def load():
    """Return the loaded library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise QnnError(
            "libqnn_hip.so is not built (%s). Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (needs hipcc). There is no CPU fallback." % path)
    if path == _build.LIB and _build.built_hash() != _build.source_hash() and os.environ.get("QNN_ALLOW_STALE_LIB") != "1":
        raise QnnError(
            "libqnn_hip.so was built from other sources than the ones in csrc/ (stamp %s, sources %s). Rebuild: "
            "`python -c \"import __graft_entry__ as g; g.build()\"`." % (_build.built_hash(), _build.source_hash()))
    lib = ctypes.CDLL(path)
    vp, ci, sz, fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float
    lib.qnn_version.restype = ci
    lib.qnn_last_error.restype = ctypes.c_char_p
    lib.qnn_last_kernel.restype = ctypes.c_char_p
    lib.qnn_set_conv_impl.argtypes = [ci]
    lib.qnn_binary_tanh_f32.argtypes = [vp, vp, sz, vp]
    lib.qnn_quantized_tanh_f32.argtypes = [vp, vp, sz, ci, vp]
    lib.qnn_ternary_tanh_f32.argtypes = [vp, vp, sz, vp, vp]
    lib.qnn_ternary_abs_sum_f32.argtypes = [vp, sz, vp, vp]
    lib.qnn_ternary_apply_f32.argtypes = [vp, vp, sz, vp, vp]
    lib.qnn_packed_bytes.argtypes = [ci, sz, ci]
    lib.qnn_packed_bytes.restype = sz
    lib.qnn_pack_f32.argtypes = [vp, vp, sz, ci, ci, ci, ci, vp]
    lib.qnn_unpack_f32.argtypes = [vp, vp, sz, ci, ci, ci, vp]
    lib.qnn_avgpool_packed_f32.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
    lib.qnn_softmax_f32.argtypes = [vp, vp, ctypes.c_size_t, ci, vp]
    lib.qnn_prepack_weights.argtypes = [ci, ci, fl, vp, ci, ci, ci, ci, vp, ci, ci, ci, vp,
                                        ctypes.POINTER(vp)]
    lib.qnn_free_weights.argtypes = [vp]
    lib.qnn_weights_dequant.argtypes = [vp, vp, vp]
    lib.qnn_weights_check.argtypes = [vp, vp]
    lib.qnn_conv2d_forward.argtypes = [vp, vp, ci, ci, ci, ci, ci, ctypes.POINTER(Epilogue), vp, vp]
    lib.qnn_dense_forward.argtypes = [vp, vp, ci, ci, ci, ctypes.POINTER(Epilogue), vp, vp]
    lib.qnn_conv2d_dense_forward.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ctypes.POINTER(Epilogue),
                                             ctypes.POINTER(Epilogue), vp, vp]
    lib.qnn_fold_prepare.argtypes = [vp, ci, ci, ctypes.POINTER(Epilogue), vp, ctypes.POINTER(vp)]
    lib.qnn_fold_free.argtypes = [vp]
    lib.qnn_fold_info.argtypes = [vp, ctypes.POINTER(FoldInfo)]
    lib.qnn_fold_constants.argtypes = [vp, vp, vp, vp, vp]
    lib.qnn_fold_eval.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    lib.qnn_conv2d_workspace_bytes.argtypes = [vp, ci, ci, ci]
    lib.qnn_conv2d_workspace_bytes.restype = sz
    lib.qnn_conv2d_forward_f32in.argtypes = [vp, vp, ci, ci, ci, ci, ci, ctypes.POINTER(Epilogue), vp,
                                             vp, sz, vp]
    for name in EXPORTS:   # every symbol the header declares must be exported
        getattr(lib, name)
    _lib = lib
    return lib


def check(rc, what):
    if rc != QNN_OK:
        msg = load().qnn_last_error().decode(errors="replace")
        raise QnnError("%s failed (%d): %s" % (what, rc, msg))


IMPL_AUTO, IMPL_VALU, IMPL_MFMA = 0, 1, 2


_conv_impl = IMPL_AUTO


def set_conv_impl(impl):
    """0 auto, 1 VALU kernels only, 2 prefer int8 MFMA (bit-identical results)."""
    global _conv_impl
    check(load().qnn_set_conv_impl(int(impl)), "qnn_set_conv_impl")
    _conv_impl = int(impl)


EPI_NO_STRIP, EPI_NO_STRIP64, EPI_NO_HALO, EPI_NO_LDS16 = 1, 2, 4, 8      # qnn_epilogue_t.flags (qnn_abi.h)
_default_flags = 0
_default_first = None


def set_option(key, value):
    """TEST / TOOL HELPER of this binding -- the C library keeps no such state (round 4: qnn_set_option is gone from the
    ABI).  Sets what calls made THROUGH THIS MODULE pass per call when the caller says nothing:
      "strip" 0 / 1, "strip64" -1 / 0 / 1, "halo" 0 / 1, "lds16" 0 / 1   -> qnn_epilogue_t.flags (QNN_EPI_NO_*): kernel
                  selection only, results bit-identical;
      "first_fixed" / "first_image" 0 / 1   -> a plain float32 input store is declared QNN_STORE_F32_UNIT /
                  QNN_STORE_F32_IMAGE (the typed stores the engines pass explicitly)."""
    global _default_flags, _default_first
    bits = {"strip": EPI_NO_STRIP, "strip64": EPI_NO_STRIP64, "halo": EPI_NO_HALO, "lds16": EPI_NO_LDS16}
    if key in bits:
        _default_flags = (_default_flags | bits[key]) if int(value) == 0 else (_default_flags & ~bits[key])
    elif key in ("first_fixed", "first_image"):
        store = STORE_F32_UNIT if key == "first_fixed" else STORE_F32_IMAGE
        if int(value):
            _default_first = store
        elif _default_first == store:
            _default_first = None
    else:
        raise QnnError("set_option: unknown key %r" % (key,))


def _first_store(x_store):
    return _default_first if (x_store == STORE_F32 and _default_first is not None) else x_store


def conv_impl():
    """The kernel-family preference last set through set_conv_impl()."""
    return _conv_impl


def last_kernel():
    return load().qnn_last_kernel().decode()


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, what):
    """The product path runs on the GPU only; fail loudly otherwise."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor, got %r" % (what, type(t)))
    if not t.is_cuda:
        raise QnnError("%s: tensor is on %s; the low-bit engine has no CPU path" % (what, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s: expected float32, got %s" % (what, t.dtype))
    return t.contiguous()


def require_cuda_u8(t, what):
    """Typed image input (QNN_STORE_U8): a uint8 NHWC CUDA tensor, value = code / 255 (utils/load_data.py:40)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor, got %r" % (what, type(t)))
    if not t.is_cuda:
        raise QnnError("%s: tensor is on %s; the low-bit engine has no CPU path" % (what, t.device))
    if t.dtype != torch.uint8:
        raise TypeError("%s: expected uint8 image bytes, got %s" % (what, t.dtype))
    return t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


def per_word(store):
    """Channels a whole packed word (T2: word pair) covers."""
    return {STORE_BIN: 32, STORE_T2: 32, STORE_I4: 8, STORE_I8: 4}[store]


def words(store, channels):
    if store == STORE_T2:                      # (mask, sign) word pairs per 32 channels
        return 2 * ((channels + 31) // 32)
    pw = per_word(store)
    return (channels + pw - 1) // pw


def store_for_bits(bits):
    """Packed storage able to hold signed `bits`-bit codes (2..8)."""
    if bits <= 4:
        return STORE_I4
    if bits <= 8:
        return STORE_I8
    raise QnnError("no packed storage for %d-bit codes (max 8)" % bits)


# ---------------------------------------------------------------------------
# thin typed wrappers
# ---------------------------------------------------------------------------
def pack(x, channels, fn, nb, store):
    """x: float32 (..., channels) CUDA -> int32 tensor (pixels, words)."""
    x = require_cuda(x, "pack")
    pixels = x.numel() // channels
    out = torch.empty((pixels, words(store, channels)), dtype=torch.int32, device=x.device)
    check(load().qnn_pack_f32(ptr(x), ptr(out), pixels, channels, fn, nb, store, stream_ptr()),
          "qnn_pack_f32")
    return out


def unpack(p, pixels, channels, store, nb):
    out = torch.empty((pixels, channels), dtype=torch.float32, device=p.device)
    check(load().qnn_unpack_f32(ptr(p), ptr(out), pixels, channels, store, nb, stream_ptr()),
          "qnn_unpack_f32")
    return out


def avgpool_packed(p, store, bits, N, H, W, C, size):
    """AveragePooling2D(size) 'valid' of a packed NHWC tensor -> float32 (N, H//size, W//size, C)."""
    out = torch.empty((N, H // size, W // size, C), dtype=torch.float32, device=p.device)
    check(load().qnn_avgpool_packed_f32(ptr(p), store, bits, N, H, W, C, size, ptr(out), stream_ptr()),
          "qnn_avgpool_packed_f32")
    return out


def softmax(x, out=None):
    """qnn_softmax_f32 over the last axis (float64 inside, one rounding): Dense(..., activation='softmax')."""
    x = require_cuda(x, "softmax input").contiguous()
    assert x.dtype == torch.float32
    y = out if out is not None else torch.empty_like(x)
    cols = x.shape[-1]
    check(load().qnn_softmax_f32(ptr(x), ptr(y), x.numel() // cols, cols, stream_ptr()), "qnn_softmax_f32")
    return y


# Handles released while a HIP stream capture is in progress (Python's cycle collector can run a __del__ anywhere, e.g.
# inside `with torch.cuda.graph(g)`): hipFree is not permitted during a capture and would invalidate it, so the handle is
# parked here and freed at the next safe point (the next handle creation, or release_deferred()).
_deferred = []


def _release(fn_name, handle):
    if not (handle and handle.value) or _lib is None:
        return
    try:
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:
        capturing = False
    if capturing:
        _deferred.append((fn_name, handle.value))
    else:
        getattr(_lib, fn_name)(handle)


def release_deferred():
    """Free the handles whose owners died during a stream capture (no-op while a capture is in progress)."""
    if not _deferred or _lib is None:
        return
    try:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
    except Exception:
        return
    while _deferred:
        fn_name, value = _deferred.pop()
        getattr(_lib, fn_name)(ctypes.c_void_p(value))


class Weights:
    """Owner of an opaque qnn_weights_t handle."""

    def __init__(self, wkind, wbits, H, kernel, bias, stride, same_pad, store):
        kernel = require_cuda(kernel, "prepack kernel")
        if kernel.dim() == 2:
            kh = kw = 1
            cin, cout = kernel.shape
        else:
            kh, kw, cin, cout = kernel.shape
        bias = require_cuda(bias, "prepack bias") if bias is not None else None
        self.shape = (kh, kw, cin, cout)
        self.wkind = wkind
        self.store = store
        self.stride = stride
        self.same_pad = same_pad
        self.handle = ctypes.c_void_p(None)
        release_deferred()
        check(load().qnn_prepack_weights(wkind, wbits, float(H), ptr(kernel), kh, kw, cin, cout,
                                         ptr(bias), stride, 1 if same_pad else 0, store,
                                         stream_ptr(), ctypes.byref(self.handle)),
              "qnn_prepack_weights")
        self.device = kernel.device

    def check(self):
        """qnn_weights_check: synchronise the current stream and raise QnnError if a restricted-domain kernel of this
        layer (the opt-in fixed-point first layer: inputs in [0, 1]) met a value outside its domain since the last check."""
        check(load().qnn_weights_check(self.handle, stream_ptr()), "qnn_weights_check")

    def dequant(self):
        kh, kw, cin, cout = self.shape
        out = torch.empty((kh, kw, cin, cout), dtype=torch.float32, device=self.device)
        check(load().qnn_weights_dequant(self.handle, ptr(out), stream_ptr()), "qnn_weights_dequant")
        return out

    def __del__(self):
        try:
            _release("qnn_free_weights", self.handle)
            self.handle = ctypes.c_void_p(None)
        except Exception:
            pass


def out_hw(size, k, stride, same_pad):
    if same_pad:
        return -(-size // stride)
    return (size - k) // stride + 1


def make_epilogue(bn_inv, bn_shift, fn, act_bits, pool, out_store, res=None, res_store=STORE_F32,
                  res_bits=0, post_scale=1.0, trick=None, fold=None, domain_flag=None, flags=None, proj=None):
    """trick: None (the reference's lr-multiplier identity trick is the identity) or the (c, s) float32 pair of its
    OUTPUT side, `faithful_trick(klm, promotion)`.  fold: a Fold prepared for exactly this layer and epilogue.
    proj: (weights of the 1x1 strides-2 projection, packed block input, H, W, x_bits) -- the shortcut computed inside the
    launch (qnn_projection_t); the returned epilogue keeps the descriptor alive as `_proj`."""
    tc, ts = (float(trick[0]), float(trick[1])) if trick is not None else (0.0, 0.0)
    pj = None
    if proj is not None:
        pw, px, pH, pW, pbits = proj
        pj = Projection(pw.handle.value, ptr(px).value, int(pH), int(pW), int(pbits))
    e = Epilogue(ptr(bn_inv).value, ptr(bn_shift).value, fn, act_bits, pool, out_store,
                 ptr(res).value, res_store, res_bits, float(post_scale), tc, ts,
                 fold.handle.value if fold is not None else None,
                 _default_flags if flags is None else int(flags),
                 ptr(domain_flag).value if domain_flag is not None else None,
                 ctypes.addressof(pj) if pj is not None else None)
    e._proj = pj
    return e


class Fold:
    """Owner of a qnn_fold_t: the epilogue of one layer (bias, BN, [shortcut merge], quantized_tanh) folded to a slope and
    an accumulator offset per channel and PROVEN equal to the float32 chain on every point of the layer's accumulator
    domain (qnn_fold_prepare, include/qnn_abi.h).  `usable` False = some channel has no exact fold: the kernels keep the
    chain (bit-identical results either way).  Returns None from `try_prepare` where the library folds nothing
    (other bit widths, float32 shortcut, ...)."""

    def __init__(self, w, x_store, x_bits, bn_inv, bn_shift, fn, act_bits, out_store, res=None, res_store=STORE_F32,
                 res_bits=0, post_scale=1.0):
        self._keep = (w, bn_inv, bn_shift)
        self.handle = ctypes.c_void_p(None)
        release_deferred()
        epi = make_epilogue(bn_inv, bn_shift, fn, act_bits, 1, out_store, res, res_store, res_bits, post_scale)
        rc = load().qnn_fold_prepare(w.handle, x_store, x_bits, ctypes.byref(epi), stream_ptr(), ctypes.byref(self.handle))
        self.rc = rc
        if rc != QNN_OK:
            self.handle = ctypes.c_void_p(None)
            if rc != QNN_EUNSUPPORTED:
                check(rc, "qnn_fold_prepare")
            return
        info = FoldInfo()
        check(load().qnn_fold_info(self.handle, ctypes.byref(info)), "qnn_fold_info")
        self.channels, self.folded, self.usable = info.channels, info.folded, bool(info.usable)
        self.points, self.acc_lo, self.acc_hi = info.points, info.acc_lo, info.acc_hi
        self.shortcut_codes, self.mode = info.shortcut_codes, info.mode

    @classmethod
    def try_prepare(cls, *a, **kw):
        f = cls(*a, **kw)
        return f if f.handle.value else None

    def constants(self, device):
        A = torch.empty(self.channels, dtype=torch.float32, device=device)
        b = torch.empty(self.channels, dtype=torch.int32, device=device)
        check(load().qnn_fold_constants(self.handle, ptr(A), ptr(b), None, stream_ptr()), "qnn_fold_constants")
        return A, b

    def eval(self, c, acc, sc=None):
        """The folded epilogue of channel c on int32 CUDA tensors acc (true accumulator units) [and sc]: int32 codes."""
        out = torch.empty_like(acc)
        check(load().qnn_fold_eval(self.handle, int(c), ptr(acc), ptr(sc), ptr(out), acc.numel(), stream_ptr()),
              "qnn_fold_eval")
        return out

    def __del__(self):
        try:
            _release("qnn_fold_free", self.handle)
            self.handle = ctypes.c_void_p(None)
        except Exception:
            pass


def faithful_trick(klm, promotion="nep50"):
    """The two float32 constants of `(o - (1. - 1./klm) * o) * klm` (binary_layers.py:175-176) as the reference forms
    them: `promotion` = "nep50" (numpy >= 2: everything stays float32) or "legacy" (numpy < 2: `1. - 1./np.float32`
    is float64, cast to float32 when it meets the tensor)."""
    import numpy as np
    k = np.float32(klm)
    if promotion == "legacy":
        c = np.float32(1.0 - 1.0 / float(k))
    elif promotion == "nep50":
        c = np.float32(1.0) - np.float32(1.0) / k
    else:
        raise ValueError(promotion)
    return float(np.float32(c)), float(k)


def conv2d(w, x, x_store, x_bits, N, H, W, bn_inv=None, bn_shift=None, fn=FN_NONE, act_bits=0,
           pool=1, out_store=STORE_F32, res=None, res_store=STORE_F32, res_bits=0, post_scale=1.0, out=None, trick=None,
           fold=None, domain_flag=None, proj=None):
    """Run qnn_conv2d_forward; x is a float32 NHWC tensor, a uint8 NHWC tensor or an int32 packed tensor.
    Returns (y, Hp, Wp): y float32 (N,Hp,Wp,cout) or int32 (N*Hp*Wp, words); `out` = a tensor of that shape to write
    into instead of a fresh one.  proj: see make_epilogue (raises QnnUnsupported where no kernel computes it)."""
    kh, kw, cin, cout = w.shape
    Ho = out_hw(H, kh, w.stride, w.same_pad) // pool
    Wo = out_hw(W, kw, w.stride, w.same_pad) // pool
    if out_store == STORE_F32:
        shape, dt = (N, Ho, Wo, cout), torch.float32
    else:
        shape, dt = (N * Ho * Wo, words(out_store, cout)), torch.int32
    if out is None:
        y = torch.empty(shape, dtype=dt, device=x.device)
    else:
        if tuple(out.shape) != shape or out.dtype != dt or not out.is_contiguous() or out.device != x.device:
            raise QnnError("conv2d: `out` must be a contiguous %s tensor of shape %s on %s" % (dt, shape, x.device))
        y = out
    epi = make_epilogue(bn_inv, bn_shift, fn, act_bits, pool, out_store, res, res_store, res_bits, post_scale, trick, fold,
                        domain_flag, proj=proj)
    rc = load().qnn_conv2d_forward(w.handle, ptr(x), _first_store(x_store), x_bits, N, H, W, ctypes.byref(epi),
                                   ptr(y), stream_ptr())
    if proj is not None and rc == QNN_EUNSUPPORTED:
        raise QnnUnsupported("qnn_conv2d_forward: " + load().qnn_last_error().decode(errors="replace"))
    check(rc, "qnn_conv2d_forward")
    return y, Ho, Wo



def conv2d_dense(wc, wd, x, x_store, x_bits, N, H, W, c_inv, c_shift, c_fn, c_act_bits, d_inv, d_shift, out=None, fold=None):
    """qnn_conv2d_dense_forward: the last conv group (2x2 pool, packed int4 codes) and the dense head behind it in one
    launch.  Returns the (N, units) float32 logits, or None when no fused kernel covers the pair (QNN_EUNSUPPORTED)."""
    units = wd.shape[3]
    y = out if out is not None else torch.empty((N, units), dtype=torch.float32, device=x.device)
    ec = make_epilogue(c_inv, c_shift, c_fn, c_act_bits, 2, STORE_I4, fold=fold)
    ed = make_epilogue(d_inv, d_shift, FN_NONE, 0, 1, STORE_F32)
    rc = load().qnn_conv2d_dense_forward(wc.handle, wd.handle, ptr(x), x_store, x_bits, N, H, W, ctypes.byref(ec),
                                         ctypes.byref(ed), ptr(y), stream_ptr())
    if rc == QNN_EUNSUPPORTED:
        return None
    check(rc, "qnn_conv2d_dense_forward")
    return y


class BoundHead:
    """qnn_conv2d_dense_forward with everything bound once (see BoundStep)."""

    def __init__(self, wc, wd, x_store, x_bits, N, H, W, c_inv, c_shift, c_fn, c_act_bits, d_inv, d_shift, x, fold=None):
        self._keep = (wc, wd, c_inv, c_shift, d_inv, d_shift, x, fold)
        self._ec = make_epilogue(c_inv, c_shift, c_fn, c_act_bits, 2, STORE_I4, fold=fold)
        self._ed = make_epilogue(d_inv, d_shift, FN_NONE, 0, 1, STORE_F32)
        self._x = x.data_ptr()
        self._fn = load().qnn_conv2d_dense_forward
        self._a = (wc.handle, wd.handle)
        self._mid = (x_store, x_bits, N, H, W, ctypes.byref(self._ec), ctypes.byref(self._ed))

    def __call__(self, stream, x_ptr=None, y_ptr=None):
        rc = self._fn(self._a[0], self._a[1], ctypes.c_void_p(x_ptr or self._x), *self._mid, ctypes.c_void_p(y_ptr),
                      ctypes.c_void_p(stream))
        if rc != QNN_OK:
            check(rc, "qnn_conv2d_dense_forward")


class BoundStep:
    """qnn_conv2d_forward / qnn_dense_forward with everything bound once: a launch is then one foreign call, not a page
    of Python.  `x` / `y` are the step's static input / output tensors; either pointer can be overridden per call (a
    pipeline's first step reads the caller's batch in place, its last step writes into the caller's result)."""

    def __init__(self, kind, w, x_store, x_bits, N, H, W, bn_inv, bn_shift, fn, act_bits, pool, out_store, x, y,
                 trick=None, fold=None):
        self._keep = (w, bn_inv, bn_shift, x, y, fold)
        self._epi = make_epilogue(bn_inv, bn_shift, fn, act_bits, pool, out_store, trick=trick, fold=fold)
        self._x = x.data_ptr() if x is not None else 0
        self._y = y.data_ptr() if y is not None else 0
        lib = load()
        if kind == "conv":
            self._fn = lib.qnn_conv2d_forward
            self._mid = (_first_store(x_store), x_bits, N, H, W, ctypes.byref(self._epi))
        else:
            self._fn = lib.qnn_dense_forward
            self._mid = (x_store, x_bits, N, ctypes.byref(self._epi))
        self._h = w.handle
        self._what = "qnn_%s_forward" % ("conv2d" if kind == "conv" else "dense")

    def __call__(self, stream, x_ptr=None, y_ptr=None, flag_ptr=None):
        if flag_ptr is not None:
            self._epi.domain_flag = flag_ptr       # the caller's domain-flag word of THIS batch (engine "auto" mode)
        rc = self._fn(self._h, ctypes.c_void_p(x_ptr or self._x), *self._mid, ctypes.c_void_p(y_ptr or self._y),
                      ctypes.c_void_p(stream))
        if rc != QNN_OK:
            check(rc, self._what)


def conv2d_f32in(w, x, in_fn, in_bits, bn_inv=None, bn_shift=None, fn=FN_NONE, act_bits=0, pool=1,
                 out_store=STORE_F32):
    """qnn_conv2d_forward_f32in: float32 NHWC x, activation clip `in_fn` fused on load."""
    x = require_cuda(x, "conv2d_f32in")
    N, H, W, _ = x.shape
    kh, kw, cin, cout = w.shape
    Ho = out_hw(H, kh, w.stride, w.same_pad) // pool
    Wo = out_hw(W, kw, w.stride, w.same_pad) // pool
    if out_store == STORE_F32:
        y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    else:
        y = torch.empty((N * Ho * Wo, words(out_store, cout)), dtype=torch.int32, device=x.device)
    need = load().qnn_conv2d_workspace_bytes(w.handle, N, H, W)
    ws = torch.empty((max(need, 4) + 3) // 4, dtype=torch.int32, device=x.device)
    epi = make_epilogue(bn_inv, bn_shift, fn, act_bits, pool, out_store)
    check(load().qnn_conv2d_forward_f32in(w.handle, ptr(x), in_fn, in_bits, N, H, W, ctypes.byref(epi),
                                          ptr(y), ptr(ws), ws.numel() * 4, stream_ptr()),
          "qnn_conv2d_forward_f32in")
    return y, Ho, Wo


def dense(w, x, x_store, x_bits, N, bn_inv=None, bn_shift=None, fn=FN_NONE, act_bits=0,
          out_store=STORE_F32, out=None):
    cout = w.shape[3]
    if out_store == STORE_F32:
        shape, dt = (N, cout), torch.float32
    else:
        shape, dt = (N, words(out_store, cout)), torch.int32
    if out is None:
        y = torch.empty(shape, dtype=dt, device=x.device)
    else:
        if tuple(out.shape) != shape or out.dtype != dt or not out.is_contiguous() or out.device != x.device:
            raise QnnError("dense: `out` must be a contiguous %s tensor of shape %s on %s" % (dt, shape, x.device))
        y = out
    epi = make_epilogue(bn_inv, bn_shift, fn, act_bits, 1, out_store)
    check(load().qnn_dense_forward(w.handle, ptr(x), x_store, x_bits, N, ctypes.byref(epi), ptr(y),
                                   stream_ptr()), "qnn_dense_forward")
    return y
