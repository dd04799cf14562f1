"""CPU oracle for the low-bit forward path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file restates, in float32 numpy, exactly the op sequence the reference's
Python emits (file:line cited per function) on top of the *documented* TF
semantics it delegates to (tf.round = round-half-to-even, SAME padding,
NHWC/HWIO cross-correlation, tf.nn.batch_normalization op order).

PINNED by vectors produced by running the reference's OWN source in place
(tests/golden/make_fixtures_from_reference.py imports layers/*_ops.py,
layers/*_layers.py, models/vgg.py, models/resnet.py, models/model_factory.py from
/root/reference under an eager numpy stand-in for the few keras.backend /
tensorflow primitives they call, and writes tests/golden/ref_{ops,layers,models}.npz;
tests/test_reference_fixtures.py checks every function below against them), by the
hand-derived known-answer tables in tests/test_oracle_ops.py and by the real trained
weights in tests/golden/resnet3_*.npz (extracted from the reference's
results/RESNET3/*.hdf5 by tests/golden/make_fixtures_from_hdf5.py).
STILL UNPINNED: TensorFlow itself was never run (not installable offline), so the
summation order inside its Conv2D / MatMul kernels and its rsqrt are restated from
documentation, not measured.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path never calls it and has no CPU fallback.

Two conv modes (hard part 2):
  * ``exact``    - the lr-multiplier "identity trick" of the reference
                   (binary_layers.py:163-165,175-176) is treated as the identity,
                   so low-bit layers produce exactly-representable values.
  * ``faithful_out`` - only the OUTPUT side of the trick is replayed (the product's optional
                   trick_c / trick_s epilogue, include/qnn_abi.h); the contraction stays exact.
  * ``faithful`` - the trick is replayed in float32 with the constants the
                   reference forms (numpy-1.x or NEP-50 scalar promotion).
The contraction itself is accumulated in float64 and rounded once to float32:
for grid-valued operands (every low-bit x low-bit layer) that equals ANY float32
summation order as long as the sum stays below 2**24 grid units; for the
float-input first layer it is the correctly rounded result, which TF, this
oracle and the HIP kernels all approximate to ~1e-6 relative.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _f32(x):
    return np.asarray(x, dtype=np.float32)


# --------------------------------------------------------------------------
# layers/binary_ops.py
# --------------------------------------------------------------------------
def round_through(x):
    """binary_ops.py:8-13 / quantized_ops.py:8-14: x + stop_gradient(round(x) - x).

    K.round -> tf.round -> round-half-to-even (np.rint).  Replayed literally.
    """
    x = _f32(x)
    rounded = np.rint(x)
    return (x + (rounded - x)).astype(np.float32)


def hard_sigmoid(x):
    """binary_ops.py:16-24: clip(0.5*x + 0.5, 0, 1) (mul, then add, in float32)."""
    x = _f32(x)
    y = (F32(0.5) * x) + F32(0.5)
    return np.clip(y, F32(0), F32(1)).astype(np.float32)


def binary_sigmoid(x):
    """binary_ops.py:27-34."""
    return round_through(hard_sigmoid(x))


def binary_tanh(x):
    """binary_ops.py:37-51: 2*round_through(_hard_sigmoid(x)) - 1  -> {-1,+1}.

    Consequence (tests pin it): +1 iff x > 2**-24; 0 and 2**-24 map to -1.
    """
    return (F32(2) * round_through(hard_sigmoid(x)) - F32(1)).astype(np.float32)


def binarize(W, H=1.0):
    """binary_ops.py:54-64: H * binary_tanh(W / H)."""
    W = _f32(W)
    H = F32(H)
    return (H * binary_tanh(W / H)).astype(np.float32)


# --------------------------------------------------------------------------
# layers/quantized_ops.py
# --------------------------------------------------------------------------
def quantize(W, nb=16):
    """quantized_ops.py:49-66 (clip_through=False branch):
    clip(round_through(W*m), -m, m-1) / m with m = 2**(nb-1)."""
    W = _f32(W)
    m = F32(pow(2, nb - 1))
    Wq = np.clip(round_through(W * m), -m, m - F32(1)) / m
    return Wq.astype(np.float32)


def quantized_tanh(W, nb=16):
    """quantized_ops.py:87-100; same formula as quantize.  This is the function
    the models use as activation (model_factory.py:9,19-20: quantize_op)."""
    return quantize(W, nb)


def quantized_relu_unused(W, nb=16):
    """quantized_ops.py:69-84.  NOT used by any model (shadowed in
    model_factory.py:19-20); restated for completeness only."""
    W = _f32(W)
    p = F32(pow(2, nb))
    hs = np.clip((W + F32(1)) / F32(2), F32(0), F32(1))
    v = F32(2.0) * (round_through(hs * p) / p) - F32(1.0)
    hi = F32(1 - 1.0 / pow(2, nb - 1))
    return np.clip(v, F32(0), hi).astype(np.float32)


# --------------------------------------------------------------------------
# layers/ternary_ops.py
# --------------------------------------------------------------------------
def _ternarize(W, H=1.0):
    """ternary_ops.py:15-30: cutoff = 0.7*mean(|W/H|) over the WHOLE tensor;
    W > cutoff -> 1, W <= -cutoff -> -1, else 0; times H.

    K.mean is a float32 reduction whose order TF does not specify; the oracle
    reduces in float64 and rounds once (documented tolerance: the cutoff may
    differ by 1 ulp, which only matters for elements within 1 ulp of it).
    """
    W = _f32(W)
    H = F32(H)
    W = W / H
    mean_abs = F32(np.mean(np.abs(W), dtype=np.float64))
    cutoff = F32(0.7) * mean_abs
    Wt = np.where(W > cutoff, F32(1), np.where(W <= -cutoff, F32(-1), F32(0)))
    return (Wt.astype(np.float32) * H).astype(np.float32)


def ternarize(W, H=1.0):
    """ternary_ops.py:33-41: W + stop_gradient(Wt - W)."""
    W = _f32(W)
    Wt = _ternarize(W, H)
    return (W + (Wt - W)).astype(np.float32)


def ternary_tanh(x):
    """ternary_ops.py:52-54: ternarize(clip(x, -1, 1))."""
    x = np.clip(_f32(x), F32(-1), F32(1))
    return ternarize(x)


# --------------------------------------------------------------------------
# TF primitives the layers delegate to
# --------------------------------------------------------------------------
def same_padding(in_size, k, stride):
    """TF 'SAME': out = ceil(in/s); total = max((out-1)*s + k - in, 0);
    before = total // 2, after = total - before."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d(x, w, strides=(1, 1), padding="same"):
    """K.conv2d -> tf.nn.convolution, NHWC x HWIO, cross-correlation (no flip).
    float64 accumulation, one rounding to float32 (see module docstring)."""
    x = _f32(x)
    w = _f32(w)
    N, H, W_, C = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C, (ci, C)
    sh, sw = strides
    if padding == "same":
        Ho, pt, pb = same_padding(H, kh, sh)
        Wo, pl, pr = same_padding(W_, kw, sw)
    elif padding == "valid":
        Ho = (H - kh) // sh + 1
        Wo = (W_ - kw) // sw + 1
        pt = pb = pl = pr = 0
    else:
        raise ValueError(padding)
    xp = np.zeros((N, H + pt + pb, W_ + pl + pr, C), dtype=np.float64)
    xp[:, pt:pt + H, pl:pl + W_, :] = x
    w64 = w.astype(np.float64)
    out = np.zeros((N, Ho, Wo, co), dtype=np.float64)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (Ho - 1) * sh + 1:sh, dx:dx + (Wo - 1) * sw + 1:sw, :]
            out += patch.reshape(-1, C).dot(w64[dy, dx]).reshape(N, Ho, Wo, co)
    return out.astype(np.float32)


def fma32(a, b, c):
    """Correctly rounded float32 fused multiply-add a*b+c (what v_fma_f32 computes),
    emulated in float64: the product of two float32 is exact in float64, TwoSum gives
    the exact error of the float64 addition, and the final rounding to float32 is
    repaired when the float64 sum sits exactly on a float32 tie."""
    a64 = np.asarray(a, dtype=np.float32).astype(np.float64)
    b64 = np.asarray(b, dtype=np.float32).astype(np.float64)
    c64 = np.asarray(c, dtype=np.float32).astype(np.float64)
    p = a64 * b64
    s = p + c64
    bb = s - p
    e = (p - (s - bb)) + (c64 - bb)            # s + e == p + c exactly
    r = s.astype(np.float32)
    t = r.astype(np.float64)
    d = s - t                                  # exact
    toward = np.where(d > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32)
    nb = np.nextafter(r, toward)
    half = (nb.astype(np.float64) - t) * 0.5
    tie = (d != 0) & (d == half)
    fix = tie & (e != 0) & (np.sign(e) == np.sign(d))
    return np.where(fix, nb, r).astype(np.float32)


def conv2d_device_order(x, w, strides=(1, 1), padding="same"):
    """Float32 conv in the accumulation order of the HIP kernels: acc = 0, then one
    float32 FMA per (dy, dx, ci) in that order (out-of-image taps contribute
    fma(0, w, acc) = acc).  This is one valid instance of the unspecified summation
    order of tf.nn.convolution; it lets float-input layers be compared bit-for-bit."""
    x = _f32(x)
    w = _f32(w)
    N, H, W_, C = x.shape
    kh, kw, ci, co = w.shape
    sh, sw = strides
    if padding == "same":
        Ho, pt, pb = same_padding(H, kh, sh)
        Wo, pl, pr = same_padding(W_, kw, sw)
    else:
        Ho = (H - kh) // sh + 1
        Wo = (W_ - kw) // sw + 1
        pt = pb = pl = pr = 0
    xp = np.zeros((N, H + pt + pb, W_ + pl + pr, C), dtype=np.float32)
    xp[:, pt:pt + H, pl:pl + W_, :] = x
    acc = np.zeros((N, Ho, Wo, co), dtype=np.float32)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (Ho - 1) * sh + 1:sh, dx:dx + (Wo - 1) * sw + 1:sw, :]
            for c in range(C):
                acc = fma32(patch[..., c:c + 1], w[dy, dx, c][None, None, None, :], acc)
    return acc


def _on_grid(x):
    """True if every value is a multiple of 2**-7 in [-1, 1] (any <=8-bit activation
    grid, or +-1): sums of such values times <=8-bit weights are exact in any order."""
    x = np.asarray(x, dtype=np.float32)
    if x.size == 0:
        return True
    k = x.astype(np.float64) * 128.0
    return bool(np.abs(x).max() <= 1.0 and np.array_equal(k, np.rint(k)))


def dot(x, w):
    """K.dot -> tf.matmul, (N,K)x(K,units); float64 accumulate, one rounding."""
    return _f32(x).astype(np.float64).dot(_f32(w).astype(np.float64)).astype(np.float32)


def bias_add(x, b):
    """K.bias_add: float32 per-channel add on the last axis."""
    return (_f32(x) + _f32(b)).astype(np.float32)


def bn_constants(gamma, beta, mean, var, eps):
    """tf.nn.batch_normalization: inv = rsqrt(var+eps) * gamma;
    shift = beta - mean*inv.  All float32.  rsqrt restated as 1/sqrt (IEEE)."""
    gamma, beta, mean, var = map(_f32, (gamma, beta, mean, var))
    inv = (F32(1) / np.sqrt(var + F32(eps))).astype(np.float32) * gamma
    shift = beta - mean * inv
    return inv.astype(np.float32), shift.astype(np.float32)


def batchnorm_inference(x, gamma, beta, mean, var, eps):
    """Keras 2.1.3 BatchNormalization.call (inference) -> K.batch_normalization ->
    tf.nn.batch_normalization: x*inv + (beta - mean*inv), two roundings (no FMA).
    vgg.py:16 eps=1e-4; resnet.py:61 default eps=1e-3."""
    inv, shift = bn_constants(gamma, beta, mean, var, eps)
    return ((_f32(x) * inv).astype(np.float32) + shift).astype(np.float32)


def maxpool2d(x, size=2):
    """MaxPooling2D(pool_size=(2,2)) 'valid', stride = size (vgg.py:23,30,37)."""
    x = _f32(x)
    N, H, W_, C = x.shape
    Ho, Wo = H // size, W_ // size
    x = x[:, :Ho * size, :Wo * size, :].reshape(N, Ho, size, Wo, size, C)
    return x.max(axis=(2, 4))


def avgpool2d(x, size=8):
    """AveragePooling2D(pool_size=8) 'valid' (resnet.py:134).  tf.nn.avg_pool
    sums the window in float32 then divides; window sums of grid values are
    exact, so float64-sum/round == float32-sum here.  Then one division."""
    x = _f32(x)
    N, H, W_, C = x.shape
    Ho, Wo = H // size, W_ // size
    x = x[:, :Ho * size, :Wo * size, :].reshape(N, Ho, size, Wo, size, C)
    s = x.astype(np.float64).sum(axis=(2, 4)).astype(np.float32)
    return (s / F32(size * size)).astype(np.float32)


def leaky_relu(x, alpha=0.3):
    """keras LeakyReLU() default alpha=0.3 (model_factory.py:26,36,46,56)."""
    x = _f32(x)
    return np.where(x >= 0, x, F32(alpha) * x).astype(np.float32)


def softmax(x):
    x = _f32(x).astype(np.float64)
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


# --------------------------------------------------------------------------
# The lr-multiplier "identity trick"
# --------------------------------------------------------------------------
def glorot_klm(kh, kw, cin, cout):
    """binary_layers.py:129-132 / quantized_layers.py:133-136:
    np.float32(1. / np.sqrt(1.5 / (nb_input + nb_output)))."""
    nb_input = int(cin * kh * kw)
    nb_output = int(cout * kh * kw)
    return np.float32(1.0 / np.sqrt(1.5 / (nb_input + nb_output)))


def glorot_klm_dense(input_dim, units):
    """binary_layers.py:51-52 / quantized_layers.py:53-54."""
    return np.float32(1.0 / np.sqrt(1.5 / (input_dim + units)))


def trick_constants(klm, promotion="legacy"):
    """Constants of binary_layers.py:162-165,175-176 as float32 tensors-constants.

    ``legacy``: numpy<2 scalar promotion, python float (op) np.float32 -> float64;
    TF then casts the float64 scalar to the tensor dtype (float32).
    ``nep50``: numpy>=2, everything stays float32.
    Returns (c_in, s_in, c_out, s_out).
    """
    klm32 = np.float32(klm)
    if promotion == "legacy":
        inv = 1.0 / float(klm32)
        c_in = F32(1.0 - 1.0 / inv)
        s_in = F32(inv)
        c_out = F32(1.0 - 1.0 / float(klm32))
        s_out = klm32
    elif promotion == "nep50":
        inv = F32(1.0) / klm32
        c_in = F32(1.0) - F32(1.0) / inv
        s_in = inv
        c_out = F32(1.0) - F32(1.0) / klm32
        s_out = klm32
    else:
        raise ValueError(promotion)
    return F32(c_in), F32(s_in), F32(c_out), F32(s_out)


def _trick(x, c, s):
    x = _f32(x)
    return ((x - (c * x).astype(np.float32)).astype(np.float32) * s).astype(np.float32)


# --------------------------------------------------------------------------
# layers/binary_layers.py and layers/quantized_layers.py :: call()
# --------------------------------------------------------------------------
FLOAT_CONV = {"order": "ideal"}   # "ideal" (float64 accumulate) | "device" (FMA chain)


def _conv(x, qkernel, strides, padding):
    if FLOAT_CONV["order"] == "device" and not _on_grid(x):
        return conv2d_device_order(x, qkernel, strides, padding)
    return conv2d(x, qkernel, strides, padding)


def _conv_call(x, qkernel, bias, klm, strides, padding, mode, promotion):
    if mode == "exact":
        out = _conv(x, qkernel, strides, padding)
    elif mode == "faithful":
        c_in, s_in, c_out, s_out = trick_constants(klm, promotion)
        xin = _trick(x, c_in, s_in)
        o = conv2d(xin, qkernel, strides, padding)
        out = _trick(o, c_out, s_out)
    elif mode == "faithful_out":
        # the product's optional middle ground (include/qnn_abi.h, trick_c / trick_s): exact contraction, then the
        # reference's OUTPUT-side trick (binary_layers.py:175-176) in float32; the input side stays the identity
        c_in, s_in, c_out, s_out = trick_constants(klm, promotion)
        out = _trick(_conv(x, qkernel, strides, padding), c_out, s_out)
    else:
        raise ValueError(mode)
    if bias is not None:
        out = bias_add(out, bias)
    return out


def binary_conv2d_call(x, kernel, bias=None, H=1.0, klm=None, strides=(1, 1),
                       padding="same", mode="exact", promotion="legacy"):
    """BinaryConv2D.call, binary_layers.py:160-187 (activation=None in all models)."""
    kh, kw, ci, co = kernel.shape
    if klm is None:
        klm = glorot_klm(kh, kw, ci, co)
    return _conv_call(x, binarize(kernel, H), bias, klm, strides, padding, mode, promotion)


def quantized_conv2d_call(x, kernel, bias=None, nb=16, klm=None, strides=(1, 1),
                          padding="same", mode="exact", promotion="legacy"):
    """QuantizedConv2D.call, quantized_layers.py:164-194."""
    kh, kw, ci, co = kernel.shape
    if klm is None:
        klm = glorot_klm(kh, kw, ci, co)
    return _conv_call(x, quantize(kernel, nb), bias, klm, strides, padding, mode, promotion)


def binary_dense_call(x, kernel, bias=None, H=1.0):
    """BinaryDense.call, binary_layers.py:78-85 (no trick in the dense layers)."""
    out = dot(x, binarize(kernel, H))
    return bias_add(out, bias) if bias is not None else out


def quantized_dense_call(x, kernel, bias=None, nb=16):
    """QuantizedDense.call, quantized_layers.py:79-88."""
    out = dot(x, quantize(kernel, nb))
    return bias_add(out, bias) if bias is not None else out


def _ternary_kernel(kernel, H, mode):
    """ternarize(W, H) (ternary_ops.py:33-41).  'faithful' replays W + (Wt - W) in float32,
    which leaves some weights one ulp off {-H, 0, H}; 'exact' takes it as Wt."""
    return _ternarize(kernel, H) if mode in ("exact", "faithful_out") else ternarize(kernel, H)


def ternary_conv2d_call(x, kernel, bias=None, H=1.0, strides=(1, 1), padding="same", mode="exact"):
    """TernaryConv2D.call, ternary_layers.py:156-174 (no lr-multiplier trick in this layer)."""
    out = _conv(x, _ternary_kernel(kernel, H, mode), strides, padding)
    return bias_add(out, bias) if bias is not None else out


def ternary_dense_call(x, kernel, bias=None, H=1.0, mode="exact"):
    """TernaryDense.call, ternary_layers.py:77-84."""
    out = dot(x, _ternary_kernel(kernel, H, mode))
    return bias_add(out, bias) if bias is not None else out


def float_conv2d_call(x, kernel, bias=None, strides=(1, 1), padding="same"):
    """Stock keras Conv2D (network_type 'float', model_factory.py:24-26)."""
    out = _conv(x, kernel, strides, padding)
    return bias_add(out, bias) if bias is not None else out


# --------------------------------------------------------------------------
# Integer oracle for the XNOR/popcount and packed-int contractions
# --------------------------------------------------------------------------
def int_conv2d(xa, wa, strides=(1, 1), padding="same"):
    """Exact int64 NHWC x HWIO cross-correlation with zero padding."""
    xa = np.asarray(xa, dtype=np.int64)
    wa = np.asarray(wa, dtype=np.int64)
    N, H, W_, C = xa.shape
    kh, kw, ci, co = wa.shape
    sh, sw = strides
    if padding == "same":
        Ho, pt, pb = same_padding(H, kh, sh)
        Wo, pl, pr = same_padding(W_, kw, sw)
    else:
        Ho = (H - kh) // sh + 1
        Wo = (W_ - kw) // sw + 1
        pt = pb = pl = pr = 0
    xp = np.zeros((N, H + pt + pb, W_ + pl + pr, C), dtype=np.int64)
    xp[:, pt:pt + H, pl:pl + W_, :] = xa
    out = np.zeros((N, Ho, Wo, co), dtype=np.int64)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (Ho - 1) * sh + 1:sh, dx:dx + (Wo - 1) * sw + 1:sw, :]
            out += patch.reshape(-1, C).dot(wa[dy, dx]).reshape(N, Ho, Wo, co)
    return out


def codes_of(xq, nb):
    """Grid value k/2**(nb-1) -> integer code k (exact)."""
    m = float(pow(2, nb - 1))
    k = np.asarray(xq, dtype=np.float64) * m
    ki = np.rint(k).astype(np.int64)
    assert np.array_equal(ki.astype(np.float64), k), "not on the grid"
    return ki


def signs_of(xb):
    """{-1,+1} float -> int64 +-1 (asserts the domain)."""
    xi = np.asarray(xb).astype(np.int64)
    assert np.all(np.abs(xi) == 1)
    return xi


# --------------------------------------------------------------------------
# Typed image input (include/qnn_abi.h, QNN_STORE_U8): the dataset's bytes, value = code / 255
# (utils/load_data.py:40).  Restated specification of what the C ABI documents, NOT reference code: the
# reference only ever sees the float32 quotient.  S = sum code * k is an exact integer (k = weight * 2^wshift),
# and everything fused behind it is ONE float32 FMA t = fma(S, A, B).
# --------------------------------------------------------------------------
def weight_codes(op):
    """(integer codes HWIO, wshift) of a low-bit conv op of a net spec."""
    kind = op["kind"]
    if kind == "binary":
        return signs_of(binarize(op["kernel"], op.get("H", 1.0))), 0
    if kind == "ternary":
        return np.rint(ternarize(op["kernel"], op.get("H", 1.0))).astype(np.int64), 0
    if kind == "quantized":
        nb = int(op["nb"])
        return codes_of(quantize(op["kernel"], nb), nb), nb - 1
    raise ValueError("QNN_STORE_U8 input needs a low-bit layer, got %r" % kind)


def u8_affine(bias, bn, m, D, cout):
    """A = inv*m/D, B = (bias*inv + shift)*m: float64 arithmetic on the float32 constants, one rounding each."""
    if bn is not None:
        inv, shift = bn_constants(bn["gamma"], bn["beta"], bn["mean"], bn["var"], bn["eps"])
    else:
        inv, shift = np.ones(cout, F32), np.zeros(cout, F32)
    b = _f32(bias) if bias is not None else np.zeros(cout, F32)
    i64, s64, b64 = inv.astype(np.float64), shift.astype(np.float64), b.astype(np.float64)
    A = (i64 * float(m) / float(D)).astype(np.float32)
    B = ((b64 * i64 + s64) * float(m)).astype(np.float32)
    return A, B


def u8_conv_group(x_u8, op, bn=None, act=None):
    """conv (uint8 image in) [-> bn] [-> binary_tanh | quantized_tanh]: float32 NHWC values."""
    x_u8 = np.asarray(x_u8)
    assert x_u8.dtype == np.uint8
    k, ws = weight_codes(op)
    S = int_conv2d(x_u8.astype(np.int64), k, tuple(op.get("strides", (1, 1))), op.get("padding", "same"))
    assert np.abs(S).max(initial=0) < 2 ** 24
    m = float(2 ** (int(act["nb"]) - 1)) if act is not None and act["fn"] == "quantized_tanh" else 1.0
    A, B = u8_affine(op.get("bias"), bn, m, 255.0 * 2.0 ** ws, k.shape[3])
    t = fma32(S.astype(np.float32), A[None, None, None, :], B[None, None, None, :])
    if act is None:
        return t
    if act["fn"] == "binary_tanh":
        return np.where(t > F32(2.0 ** -24), F32(1), F32(-1)).astype(np.float32)
    if act["fn"] == "quantized_tanh":
        return (np.clip(np.rint(t), -m, m - 1) / F32(m)).astype(np.float32)
    raise ValueError(act["fn"])


def _u8_group(spec):
    """Ops the engines fuse behind a uint8-input first conv: the conv, a BN that is its only consumer and a low-bit
    activation that is the BN's (or conv's) only consumer.  Returns (bn index or None, act index or None, next index)."""
    def consumers(i):
        name = spec[i].get("dst", "t%d" % i)
        cons = []
        for j, o in enumerate(spec):
            if j == i:
                continue
            if o["op"] == "add":
                if name in (o["a"], o["b"]):
                    cons.append(j)
            elif "src" in o:
                if o["src"] == name:
                    cons.append(j)
            elif j == i + 1:
                cons.append(j)
        return cons
    bn_i = act_i = None
    nxt = 1
    if len(spec) > nxt and spec[nxt]["op"] == "bn" and consumers(0) == [nxt]:
        bn_i = nxt
        nxt += 1
    last = bn_i if bn_i is not None else 0
    if len(spec) > nxt and spec[nxt]["op"] == "act" and consumers(last) == [nxt] and \
            (spec[nxt]["fn"] == "binary_tanh" or (spec[nxt]["fn"] == "quantized_tanh" and spec[nxt]["nb"] <= 8)):
        act_i = nxt
        nxt += 1
    return bn_i, act_i, nxt


def run_spec_u8(spec, x_u8, return_all=False):
    """run_spec for uint8 images through the typed entry: the first conv group follows u8_conv_group, every later op
    is the exact-mode interpreter (all later layers are grid x grid, hence bit-exact integers)."""
    assert spec[0]["op"] == "conv"
    bn_i, act_i, nxt = _u8_group(spec)
    y = u8_conv_group(x_u8, spec[0], spec[bn_i] if bn_i is not None else None,
                      spec[act_i] if act_i is not None else None)
    last = nxt - 1
    env0 = {"input": None, spec[last].get("dst", "t%d" % last): y}
    return _run_spec(spec, y, "exact", "legacy", return_all, start=nxt, env0=env0)


# --------------------------------------------------------------------------
# Whole-network interpreter over a neutral "net spec" (list of op dicts).
# Topologies follow models/vgg.py:5-44 and models/resnet.py:26-144; the spec is
# produced by the product's nets.py builder (plain numpy data, no code shared).
# --------------------------------------------------------------------------
def run_spec(spec, x, mode="exact", promotion="legacy", return_all=False, float_conv="ideal"):
    """Interpret ``spec`` on NHWC float32 input ``x``.  Each op dict may name
    ``src`` (default: previous output) and ``dst``.  Returns the last tensor (or
    the dict of all named tensors).  ``float_conv="device"`` evaluates convolutions
    of non-grid inputs (the raw image) in the HIP kernels' FMA order."""
    FLOAT_CONV["order"] = float_conv
    try:
        return _run_spec(spec, x, mode, promotion, return_all)
    finally:
        FLOAT_CONV["order"] = "ideal"


def _run_spec(spec, x, mode, promotion, return_all, start=0, env0=None):
    env = {"input": _f32(x)} if env0 is None else dict(env0)
    cur = _f32(x)
    for i, op in enumerate(spec):
        if i < start:
            continue
        kind = op["op"]
        src = env[op["src"]] if "src" in op else cur
        if kind == "conv":
            lk = op["kind"]
            st = tuple(op.get("strides", (1, 1)))
            pad = op.get("padding", "same")
            if lk == "binary":
                y = binary_conv2d_call(src, op["kernel"], op.get("bias"), op.get("H", 1.0),
                                       op.get("klm"), st, pad, mode, promotion)
            elif lk == "quantized":
                y = quantized_conv2d_call(src, op["kernel"], op.get("bias"), op["nb"],
                                          op.get("klm"), st, pad, mode, promotion)
            elif lk == "ternary":
                y = ternary_conv2d_call(src, op["kernel"], op.get("bias"), op.get("H", 1.0), st, pad, mode)
            elif lk == "float":
                y = float_conv2d_call(src, op["kernel"], op.get("bias"), st, pad)
            else:
                raise ValueError(lk)
        elif kind == "dense":
            lk = op["kind"]
            if lk == "binary":
                y = binary_dense_call(src, op["kernel"], op.get("bias"), op.get("H", 1.0))
            elif lk == "quantized":
                y = quantized_dense_call(src, op["kernel"], op.get("bias"), op["nb"])
            elif lk == "ternary":
                y = ternary_dense_call(src, op["kernel"], op.get("bias"), op.get("H", 1.0), mode)
            elif lk == "float":
                y = dot(src, op["kernel"])
                if op.get("bias") is not None:
                    y = bias_add(y, op["bias"])
            else:
                raise ValueError(lk)
        elif kind == "bn":
            y = batchnorm_inference(src, op["gamma"], op["beta"], op["mean"], op["var"], op["eps"])
        elif kind == "act":
            fn = op["fn"]
            if fn == "binary_tanh":
                y = binary_tanh(src)
            elif fn == "quantized_tanh":
                y = quantized_tanh(src, op["nb"])
            elif fn == "ternary_tanh":
                y = ternary_tanh(src)
            elif fn == "leaky_relu":
                y = leaky_relu(src, op.get("alpha", 0.3))
            else:
                raise ValueError(fn)
        elif kind == "maxpool":
            y = maxpool2d(src, op.get("size", 2))
        elif kind == "avgpool":
            y = avgpool2d(src, op.get("size", 8))
        elif kind == "zeropad":
            p = op["pad"]
            y = np.pad(src, ((0, 0), (p, p), (p, p), (0, 0)))
        elif kind == "flatten":
            y = src.reshape(src.shape[0], -1)
        elif kind == "add":
            y = (env[op["a"]] + env[op["b"]]).astype(np.float32)
        elif kind == "scale":
            y = (src * F32(op["value"])).astype(np.float32)
        elif kind == "softmax":
            y = softmax(src)
        else:
            raise ValueError(kind)
        cur = y
        env[op.get("dst", "t%d" % i)] = y
    return env if return_all else cur
