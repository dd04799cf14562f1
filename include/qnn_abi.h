/*
 * qnn_abi.h -- C ABI of the MI355X (gfx950) low-bit forward engine.
 *
 * This is the drop-in boundary for the reference's low-bit forward path.  The
 * reference (victorjoos/QuantizedNeuralNetworks-Keras-Tensorflow) has no FFI of
 * its own: the path sits behind the Keras Layer API and runs as a TensorFlow
 * sub-graph.  Each entry point below replaces the TF ops one reference function
 * emits (file:line cited per function); the Python host classes in
 * quantizedneuralnetworks-keras-tensorflow_amd/layers/ keep the Keras surface
 * and bind these symbols through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *   - every function returns 0 (QNN_OK) or a negative QNN_E* code;
 *     qnn_last_error() returns a thread-local human-readable message.
 *   - the CALLER owns every device buffer passed in; the library allocates
 *     device memory only inside an opaque prepacked-weights handle.
 *   - all work is enqueued on the caller's stream (a hipStream_t passed as
 *     void*; NULL = default stream); nothing synchronises the device.
 *   - tensors are NHWC, kernels HWIO, dense kernels (in, units): Keras layouts.
 *   - functions are re-entrant for distinct streams / handles.
 *
 * Packed activation storage (NHWC, channel-fastest, little-endian words):
 *   QNN_STORE_BIN  1 bit / channel, 32 channels per uint32, bit j of word w is
 *                  channel 32*w+j, bit = 1 <=> +1, bit = 0 <=> -1; channels are
 *                  padded with 0 bits up to a multiple of 32.
 *   QNN_STORE_I4   signed 4-bit two's-complement codes, 8 per uint32, nibble j
 *                  of word w is channel 8*w+j; value = code / 2^(abits-1);
 *                  2- and 3-bit activations are stored sign-extended in 4 bits.
 *   QNN_STORE_I8   signed 8-bit codes, 4 per uint32.
 *   Words per pixel = ceil(C / (32 / bits)).
 *   QNN_STORE_T2   ternary codes {-1, 0, +1} as two bit planes, 2 bits / channel (layers/ternary_layers.py,
 *                  ternary_ops.py): per 32 channels one MASK word (bit j = 1 <=> channel 32w+j is non-zero) followed
 *                  by one SIGN word (bit j = 1 <=> +1; 0 where the mask is 0); 2 * ceil(C / 32) words per pixel.
 *                  A ternary x ternary dot product is two popcounts: with m = mask_a & mask_w and d = sign_a ^ sign_w,
 *                  sum = popc(m & ~d) - popc(m & d) = popc(m) - 2 popc(m & d).  A zero-padding tap is an all-zero
 *                  mask: no border corrections.  Input-side storage (activations and weights); outputs are float32
 *                  because ternary_tanh thresholds at a mean over the whole batch tensor (ternary_ops.py:23).
 *
 * Typed image input (first layer only):
 *   QNN_STORE_U8   the dataset's own bytes: NHWC, C unsigned bytes per pixel, no padding; value = code / 255,
 *                  which is how the reference forms its float32 images (utils/load_data.py:40:
 *                  `dset[0].astype('float32') / 255`).  The domain [0, 1] is guaranteed by the type, the /255
 *                  happens inside the layer, and the contraction is an exact integer sum (see qnn_conv2d_forward).
 */
#ifndef QNN_ABI_H
#define QNN_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QNN_ABI_VERSION 4

/* status codes */
#define QNN_OK            0
#define QNN_EINVAL       -1   /* bad argument (shape, kind, null pointer) */
#define QNN_EUNSUPPORTED -2   /* valid request this build has no kernel for */
#define QNN_EHIP         -3   /* a HIP runtime call failed */
#define QNN_ENOMEM       -4

/* storage kinds of activation / weight tensors */
#define QNN_STORE_F32 0
#define QNN_STORE_BIN 1
#define QNN_STORE_T2  2   /* ternary sign / mask bit planes */
#define QNN_STORE_I4  4
#define QNN_STORE_I8  8
#define QNN_STORE_U8  16  /* input only: unsigned image bytes, value = code / 255 */
/* float32 input WITH A DECLARED DOMAIN (first layer of a network, 3x3 on 3 channels; any other layer treats them as
 * QNN_STORE_F32).  The declaration selects the restricted-domain kernel for this call only -- no process-wide switch --
 * and a value outside the domain raises the layer's domain flag (qnn_weights_check):
 *   QNN_STORE_F32_IMAGE  the values are image bytes / 255 (what utils/load_data.py:40 produces): a value x is read as the
 *                        byte k = rint(255 x) when |255 x - k| <= 2^-15 (every float32 quotient k/255 passes) and raises
 *                        the domain flag otherwise.  Result = the typed uint8 entry's (see qnn_conv2d_forward): exact
 *                        integer sum, one float32 FMA.
 *   QNN_STORE_F32_UNIT   the values lie in [0, 1] (64 filters of <= 4 bits): fixed point -- inputs rounded to 2^-23, exact
 *                        int32 sums on the int8 matrix pipe, one rounding: within 27 * 2^-24 + half an ulp of the
 *                        REAL-number convolution, hence inside the 1e-5 contract, but not the float32 FMA chain of the
 *                        exact kernel -- activation codes whose pre-activation sits that close to a rounding threshold can
 *                        differ from the oracle's.  Inputs outside [0, 1] (NaN included) are NOT silently saturated: the
 *                        kernel raises the domain flag, see qnn_weights_check() / qnn_epilogue_t.domain_flag.
 * (Round 4: the process-wide qnn_set_option() of ABI 3 is gone; these typed stores are the only way in.) */
#define QNN_STORE_F32_IMAGE 17
#define QNN_STORE_F32_UNIT  18

/* weight quantizers (what the layer applies to its latent fp32 kernel) */
#define QNN_W_FLOAT    0   /* stock Conv2D / Dense: kernel used as is          */
#define QNN_W_BINARY   1   /* binarize(W, H)      layers/binary_ops.py:54-64   */
#define QNN_W_QUANT    2   /* quantize(W, nb)     layers/quantized_ops.py:49-66 */
#define QNN_W_TERNARY  3   /* ternarize(W, H)     layers/ternary_ops.py:33-41  */

/* activation functions (layers/ *_ops.py) */
#define QNN_FN_NONE            0
#define QNN_FN_BINARY_TANH     1   /* binary_ops.py:37-51                       */
#define QNN_FN_QUANTIZED_TANH  2   /* quantized_ops.py:87-100 (= quantize_op)    */
#define QNN_FN_TERNARY_TANH    3   /* ternary_ops.py:52-54                      */
#define QNN_FN_GRID            4   /* input is already on the grid: encode only  */

typedef struct qnn_weights qnn_weights_t;   /* opaque prepacked layer weights */
typedef struct qnn_fold qnn_fold_t;         /* opaque: one layer's epilogue folded over its accumulator domain (below) */

/*
 * Epilogue fused behind the contraction.  Steps, in the reference's op order:
 *   v = acc * 2^-(wshift+xshift)             conv / matmul result (exact)
 *   v = (v - trick_c*v) * trick_s             only if trick_s != 0 ("faithful" output side, see below)
 *   v = v + bias[c]                           K.bias_add   (if the layer has bias)
 *   v = v*bn_inv[c] + bn_shift[c]             inference BatchNormalization, two
 *                                             roundings (vgg.py:16, resnet.py:61)
 *   v = (res[pixel][c] + v) * post_scale      residual merge of models/resnet.py:127-128:
 *                                             keras.layers.add([x, y]) then Lambda(x*0.5);
 *                                             only if res != NULL (pool must be 1)
 *   v = fn(v)                                 binary_tanh / quantized_tanh(act_bits)
 *   v = max over a pool x pool window         MaxPooling2D(2,2) (vgg.py:23,30,37)
 *   store as out_store (F32 value, or packed code)
 * bn_inv / bn_shift are device pointers to per-channel float32 constants formed
 * by the host exactly as tf.nn.batch_normalization forms them; NULL = no BN.
 *
 * trick_c / trick_s -- the reference wraps its convolutions in an lr-multiplier "identity trick"
 * (layers/binary_layers.py:163-165,175-176, quantized_layers.py:167-169,179-180):
 *     outputs = (o - (1. - 1./klm) * stop_gradient(o)) * klm
 * which is the identity in real numbers and a float32 no-op up to rounding noise (~klm * ulp(o)).  By default
 * (trick_s = 0, "exact" mode) it is evaluated as the identity.  A caller who wants the reference's OUTPUT-side noise
 * reproduced bit for bit -- it moves 8-bit activation codes sitting on rounding ties -- passes the two float32 constants
 * the reference forms, trick_c = 1 - 1/klm and trick_s = klm; the three float32 operations then run on the exact conv
 * result before the bias.  Only the VALU kernel families (and the generic kernel) implement it: a launch with
 * trick_s != 0 bypasses the matrix-pipe kernels.  The INPUT-side trick (163-165) perturbs grid-valued inputs by at most
 * one ulp and would make the contraction a float32 problem; it stays the identity.
 */
/*
 * A projection shortcut computed INSIDE the launch (ABI 4).  models/resnet.py:117-124: the first block of a stage adds
 * the block input through a 1x1, strides-2 QuantizedConv2D (no BN, no activation) to the BN output of the block's second
 * 3x3 convolution.  As two launches that float32 shortcut is written and read once (8x the bytes of the packed block
 * input it is computed from: 103 MB against 12.8 MB per 64 images at 224^2 -> 112^2); as `proj` of the second
 * convolution's epilogue the launch reads the block input itself -- output pixel (y, x) takes input pixel (2y, 2x) --
 * and forms   float32(sum_k code_x code_w) * 2^-(wshift + x_bits - 1) [+ bias]   per output value: bit for bit what
 * qnn_conv2d_forward(w, x, out_store = QNN_STORE_F32, no BN) stores, then added like a QNN_STORE_F32 `res`
 * (post_scale applies).  Supported where the row-walking strip kernel runs the layer (3x3 stride-1 int4 layer with
 * cin = cout in {32, 64}, packed int4 output, w->cin == cin / 2): anything else returns QNN_EUNSUPPORTED -- the caller
 * keeps the two-launch form.  `res` and `fold` must be NULL.
 */
typedef struct qnn_projection {
    const qnn_weights_t* w;  /* prepacked 1x1 kernel, strides 2, QNN_STORE_I4; w->cout == the layer's cout            */
    const void* x;           /* DEVICE: the block input, packed int4 codes, N x H x W x w->cin                        */
    int32_t H, W;            /* its spatial size; ceil(H / 2) x ceil(W / 2) must be the layer's output size           */
    int32_t x_bits;          /* value = code / 2^(x_bits - 1)                                                         */
} qnn_projection_t;

typedef struct qnn_epilogue {
    const float* bn_inv;     /* [cout] or NULL                                  */
    const float* bn_shift;   /* [cout] or NULL                                  */
    int32_t fn;              /* QNN_FN_NONE | _BINARY_TANH | _QUANTIZED_TANH     */
    int32_t act_bits;        /* nb of quantized_tanh (ignored otherwise)         */
    int32_t pool;            /* 1 = none, 2 = 2x2 max pool stride 2 'valid'      */
    int32_t out_store;       /* QNN_STORE_F32 | _BIN | _I4 | _I8                 */
    /* residual input (shortcut branch), same pixels x cout as the output, or NULL */
    const void* res;         /* DEVICE: float32 NHWC, or packed codes              */
    int32_t res_store;       /* QNN_STORE_F32 | _BIN | _I4 | _I8                   */
    int32_t res_bits;        /* packed: value = code / 2^(res_bits-1) (BIN: +-1)    */
    float post_scale;        /* multiplier after the add (0.5 in resnet.py:128; 1 = none) */
    float trick_c;           /* output-side identity trick: 1 - 1/klm (ignored when trick_s == 0) */
    float trick_s;           /* klm, or 0 = the trick is the identity (default)                    */
    const qnn_fold_t* fold;  /* qnn_fold_prepare() of exactly this layer + epilogue, or NULL (ABI 4)  */
    uint32_t flags;          /* QNN_EPI_* kernel-selection bits for THIS call (0 = the defaults); see below (ABI 4) */
    uint32_t* domain_flag;   /* DEVICE-visible word of the CALLER: a restricted-domain first-layer kernel
                              * (QNN_STORE_F32_IMAGE / _UNIT) that meets an input outside its domain stores a
                              * non-zero value there instead of raising the layer's own flag -- one word per
                              * batch in flight lets the caller recompute exactly the affected batch on the
                              * exact kernel (engine "auto" mode).  NULL = the handle's flag (ABI 4)          */
    const qnn_projection_t* proj;   /* the shortcut as a 1x1 strides-2 convolution of the block input, computed inside
                                     * the launch (see qnn_projection_t), or NULL (ABI 4)                             */
} qnn_epilogue_t;

/*
 * qnn_epilogue_t.flags -- kernel selection for ONE call, for A/B measurements and tests.  Results are bit-identical
 * under every combination; there is no process-wide switch (the library keeps no global state besides
 * qnn_set_conv_impl's family preference).
 *   QNN_EPI_NO_STRIP     3x3 stride-1 / -2 int4 layers with 16 / 32 / 64 channels: not the row-walking strip kernels
 *                        (the tile kernels take them)
 *   QNN_EPI_NO_STRIP64   only the 64-channel layers leave the strip kernel (the LDS-weight kernel takes them)
 *   QNN_EPI_NO_HALO      pooled int4 layers with 64 input channels: per-tap operand fetch (k_conv_mfma_areg) instead of
 *                        the receptive field staged once through LDS (k_conv_mfma_halo)
 *   QNN_EPI_NO_LDS16     16 -> 16 channel layers with a fold: k_conv_strip instead of the LDS-staged k_conv_strip16_lds
 * The restricted-domain first-layer kernels are selected by the TYPED input stores QNN_STORE_F32_IMAGE /
 * QNN_STORE_F32_UNIT of the call (above), never by a switch.
 */
#define QNN_EPI_NO_STRIP    1u
#define QNN_EPI_NO_STRIP64  2u
#define QNN_EPI_NO_HALO     4u
#define QNN_EPI_NO_LDS16    8u

/* ---- library ------------------------------------------------------------ */
int         qnn_version(void);
const char* qnn_last_error(void);
/* Kernel family for eligible conv layers: 0 = auto (fastest), 1 = VALU kernels only
 * (XNOR+popcount / v_dot8 / v_dot4), 2 = prefer the int8 MFMA implicit GEMM.
 * Process-wide; results are bit-identical across families. */
int         qnn_set_conv_impl(int impl);

/* ---- elementwise activation clips on float32 tensors --------------------- */
/* binary_ops.binary_tanh, layers/binary_ops.py:37-51 */
int qnn_binary_tanh_f32(const float* x, float* y, size_t n, void* stream);
/* quantized_ops.quantized_tanh (and quantize), layers/quantized_ops.py:49-66,87-100 */
int qnn_quantized_tanh_f32(const float* x, float* y, size_t n, int nb, void* stream);
/* ternary_ops.ternary_tanh, layers/ternary_ops.py:52-54.  Needs the global
 * mean(|clip(x)|) first: `workspace16` is 16 bytes of caller-owned device memory
 * for the running sum.  Memset + two kernels on `stream`. */
int qnn_ternary_tanh_f32(const float* x, float* y, size_t n, void* workspace16, void* stream);
/* The same op in its two halves, for a batch tensor that is sharded over processes (the mean of
 * ternary_ops.py:23 is over the WHOLE tensor): abs_sum leaves {sum |clip(x,-1,1)|, n} as two doubles
 * in `workspace16`; the caller all-reduces (sums) those 16 bytes across the shards; apply thresholds
 * at 0.7 * workspace16[0] / workspace16[1]. */
int qnn_ternary_abs_sum_f32(const float* x, size_t n, void* workspace16, void* stream);
int qnn_ternary_apply_f32(const float* x, float* y, size_t n, const void* workspace16, void* stream);

/* ---- pack / unpack between float32 NHWC and packed storage --------------- */
/* bytes of a packed tensor of `pixels` pixels x `channels` channels */
size_t qnn_packed_bytes(int store, size_t pixels, int channels);
/* y = encode(fn(x)):  fn = QNN_FN_BINARY_TANH -> BIN; QNN_FN_QUANTIZED_TANH(nb)
 * -> I4/I8 codes; QNN_FN_GRID -> x is already +-1 (BIN: bit = x > 0) or k/2^(nb-1). */
int qnn_pack_f32(const float* x, void* y, size_t pixels, int channels,
                 int fn, int nb, int store, void* stream);
/* inverse of the encoding: float32 value of every stored code */
int qnn_unpack_f32(const void* x, float* y, size_t pixels, int channels,
                   int store, int nb, void* stream);

/* AveragePooling2D(pool_size=size) 'valid' of a packed NHWC tensor into float32 (N, H/size, W/size, C):
 * models/resnet.py:134 behind the last activation.  Exact window sums of the codes, one float32 division. */
int qnn_avgpool_packed_f32(const void* x, int store, int bits, int N, int H, int W, int C, int size,
                           float* y, void* stream);
/* softmax over the last axis of a (rows, cols) float32 matrix, evaluated in float64 and rounded once: the classifier's
 * activation='softmax' (models/resnet.py:137).  x == y allowed. */
int qnn_softmax_f32(const float* x, float* y, size_t rows, int cols, void* stream);

/* ---- weights -------------------------------------------------------------- */
/*
 * Quantize + pack a layer's latent float32 kernel once (the reference re-runs
 * binarize/quantize every Session.run: binary_layers.py:161, quantized_layers.py:165).
 *   wkind/wbits : QNN_W_* and nb of quantize()
 *   H           : the layers' H (binarize(W,H)); 1.0 in every model
 *   kernel      : DEVICE pointer, float32 HWIO (kh,kw,cin,cout); dense: kh=kw=1
 *   bias        : DEVICE pointer [cout] or NULL (use_bias=False)
 *   stride      : 1 or 2 (square); same_pad: 1 = 'same', 0 = 'valid'
 *   store       : packed kind to prepare for the integer path (QNN_STORE_BIN only
 *                 for QNN_W_BINARY; I4 needs wbits<=4; I8 needs wbits<=8; ternary
 *                 weights {-1,0,1} use T2 -- against ternary activations -- or I4 / I8;
 *                 binary weights may also be prepared as T2: +-1 with a full mask), or
 *                 QNN_STORE_F32 for "float32 inputs only" (first layer).
 */
int qnn_prepack_weights(int wkind, int wbits, float H, const float* kernel,
                        int kh, int kw, int cin, int cout, const float* bias,
                        int stride, int same_pad, int store, void* stream,
                        qnn_weights_t** out);
int qnn_free_weights(qnn_weights_t* w);
/* read back the quantized kernel as float32 HWIO into a DEVICE buffer (tests) */
int qnn_weights_dequant(const qnn_weights_t* w, float* kernel_hwio, void* stream);
/*
 * Domain flag of a layer.  A kernel with a restricted input domain (today: the "first_fixed" variant, inputs in
 * [0, 1]) raises a flag inside the handle when it meets a value outside it; the affected outputs are unspecified.
 * The flag is reported -- and cleared -- as QNN_EINVAL by
 *   - qnn_weights_check(w, stream): synchronises `stream` first, so every launch enqueued on it has been seen
 *     (the sync point a caller places before it trusts the results of a batch);
 *   - the next qnn_conv2d_forward on the handle, if the flag has already become visible to the host by then
 *     (no synchronisation; a convenience, not a guarantee).
 * Layers whose kernels accept every input never raise it: the call is then just the stream synchronisation.
 */
int qnn_weights_check(const qnn_weights_t* w, void* stream);

/* ---- the epilogue as integer thresholds ------------------------------------- */
/*
 * Everything the reference computes behind a low-bit convolution -- K.bias_add, the inference BatchNormalization,
 * the residual merge and quantized_tanh (models/vgg.py:16-17,23; models/resnet.py:59-63,127-129) -- is, per output
 * channel, a MONOTONE STEP FUNCTION of the integer accumulator (and of the shortcut code): at most 2^act_bits - 1
 * thresholds.  The accumulator domain of a layer is finite and known from its weights:
 *     acc in [ sum_k min(w_k a_min, w_k a_max),  sum_k max(w_k a_min, w_k a_max) ],   a = the input codes' range.
 * qnn_fold_prepare evaluates the reference-order float32 chain (exactly the arithmetic of the un-folded kernels) on
 * EVERY point of that domain (x every shortcut code), on the GPU, and looks per channel for two constants -- a float32
 * slope A[c] and an integer offset beta[c] in accumulator units / 256 -- such that
 *     code = sat16( round( (acc*256 + beta[c]) * A[c] * 32767 ) ) >> 12                    (no shortcut)
 *     code = sat16( 2 * sat16( round((acc*256 + beta[c]) * A[c] * 32767) + (sc + 8) * 1024 ) ) >> 12     (shortcut code sc)
 * (v_cvt_f32_i32, v_mul_f32, v_cvt_pknorm_i16_f32, v_pk_add_i16 clamp: 3 to 5 vector instructions per value instead
 * of 6 to 10) reproduces the chain's code on every point.  The offset costs nothing: it is the initial value of the
 * MFMA accumulator.  Where it also reproduces the chain on the whole domain, the "bits" form is used instead
 * (mode 2; exact while |acc*256 + beta| < 2^22, monotone and saturating beyond): the stored offset carries 0x4B400000, the accumulator is then the bit pattern of the float
 * 12582912 + acc*256 + beta, and round(...) is taken of fma(that float, A[c], C[c]) with C = float(-12582912 A): one
 * v_fma_f32 in place of the conversion and the multiply.  A fold is accepted only if the exhaustive comparison finds no differing point; otherwise the
 * handle is marked "not folded" and the kernels keep evaluating the float32 chain -- results are bit-identical either
 * way, which is what the sweep proves.
 *
 * Supported today: weights prepacked for QNN_STORE_I4 with an int8 matrix-pipe image (3x3 / 1x1 convolutions),
 * x_store = QNN_STORE_I4, fn = QNN_FN_QUANTIZED_TANH with act_bits = 4, out_store = QNN_STORE_I4, no output-side
 * trick; a shortcut must be QNN_STORE_I4 codes of 4 bits with post_scale = 0.5.  Anything else: QNN_EUNSUPPORTED.
 * The call synchronises `stream` (it is a set-up call, like qnn_prepack_weights).  epi->res / epi->fold are ignored;
 * whether epi->res is NULL decides between the two forms.  The handle records the layer and the epilogue constants it
 * was built for; qnn_conv2d_forward rejects it (QNN_EINVAL) for any other combination.  The BN constant arrays must
 * not change while the handle is in use.
 */
int qnn_fold_prepare(const qnn_weights_t* w, int x_store, int x_bits, const qnn_epilogue_t* epi, void* stream,
                     qnn_fold_t** out);
int qnn_fold_free(qnn_fold_t* f);
typedef struct qnn_fold_info {
    int32_t channels;        /* cout                                                                  */
    int32_t folded;          /* channels for which (A, beta) reproduce the chain on the whole domain  */
    int32_t usable;          /* 1 if folded == channels: the kernels use the fold                      */
    int32_t shortcut_codes;  /* 16 with a shortcut, else 1                                            */
    int64_t points;          /* (accumulator, shortcut) points compared, summed over the channels      */
    int32_t acc_lo, acc_hi;  /* union of the channels' accumulator domains (true integer units)        */
    int32_t mode;            /* 1: conversion + multiply; 2: "bits" (the accumulator carries the float's bit pattern) */
} qnn_fold_info_t;
int qnn_fold_info(const qnn_fold_t* f, qnn_fold_info_t* info);
/* DEVICE copies of A[cout] (float32), beta[cout] (int32; mode 2: + 0x4B400000) and, if C != NULL, C[cout] (float32) --
 * tests and diagnostics */
int qnn_fold_constants(const qnn_fold_t* f, float* A, int32_t* beta, float* C, void* stream);
/* Evaluate the FOLDED epilogue (the very device function the kernels inline) for channel `c` on n accumulator values
 * acc[i] (true integer units, DEVICE int32) and, with a shortcut, shortcut codes sc[i] (DEVICE int32, else NULL):
 * codes[i] (DEVICE int32) = the 4-bit output code.  Lets a test sweep the whole domain against an independent
 * restatement of the reference chain. */
int qnn_fold_eval(const qnn_fold_t* f, int c, const int32_t* acc, const int32_t* sc, int32_t* codes, size_t n,
                  void* stream);

/* ---- the contractions ------------------------------------------------------ */
/*
 * BinaryConv2D.call  (layers/binary_layers.py:160-187) and
 * QuantizedConv2D.call (layers/quantized_layers.py:164-194), with the
 * lr-multiplier identity trick treated as the identity ("exact" mode).
 *   x        : DEVICE, NHWC; x_store = QNN_STORE_F32 (any float32 values: the
 *              first layer, or the generic fallback), a packed kind, or QNN_STORE_U8
 *   x_bits   : abits of the packed codes (value = code/2^(x_bits-1); BIN: 1; U8: ignored)
 *
 * QNN_STORE_U8 (image bytes, value = code/255; low-bit weights of <= 8 bits, no residual): with the integer weight
 * codes k = w * 2^wshift the convolution is the EXACT integer S = sum code * k, and everything the launch fuses
 * behind it is one affine map of S, evaluated as ONE float32 fused multiply-add per value:
 *     t = fma((float)S, A[c], B[c]),   A = inv[c] * m / (255 * 2^wshift),  B = (bias[c] * inv[c] + shift[c]) * m
 * (A, B formed in float64 from the float32 constants and rounded once; inv = 1, shift = 0 without BN; m = 2^(act_bits-1)
 * for quantized_tanh, else 1), then  quantized_tanh: code = clip(rint(t), -m, m-1);  binary_tanh: +1 iff t > 2^-24;
 * none: t.  t is within 1.5 ulp of the real-number value of the reference's expression on the same image, i.e. closer
 * to it than any float32 evaluation order of the reference's own op sequence; it is NOT bit-identical to the
 * float32-input path on x = code/255 (whose products are rounded one by one).
 *   y        : DEVICE, NHWC (N, Ho/pool, Wo/pool, cout) float32 or packed
 * Output geometry: Ho = ceil(H/stride) for 'same'.
 */
int qnn_conv2d_forward(const qnn_weights_t* w, const void* x, int x_store, int x_bits,
                       int N, int H, int W, const qnn_epilogue_t* epi, void* y,
                       void* stream);
/*
 * The same layer behind the Keras float32 surface with the PRECEDING activation layer
 * fused on load: y = conv2d(in_fn(x), W) (+ epilogue).  x is float32 NHWC; in_fn is
 * QNN_FN_BINARY_TANH (models/model_factory.py:47: Activation(binary_tanh)),
 * QNN_FN_QUANTIZED_TANH with in_bits = abits (model_factory.py:36), or QNN_FN_GRID when x
 * already holds grid values.  The weights must have been prepacked for a packed store.
 * `workspace` is DEVICE scratch of qnn_conv2d_workspace_bytes() bytes; it is only used
 * when no single fused kernel covers the shape (then: clip+pack pass, conv pass).
 */
size_t qnn_conv2d_workspace_bytes(const qnn_weights_t* w, int N, int H, int W);
int qnn_conv2d_forward_f32in(const qnn_weights_t* w, const float* x, int in_fn, int in_bits,
                             int N, int H, int W, const qnn_epilogue_t* epi, void* y,
                             void* workspace, size_t workspace_bytes, void* stream);
/*
 * BinaryDense.call (layers/binary_layers.py:78-85) / QuantizedDense.call
 * (layers/quantized_layers.py:79-88): x (N, in) float32 or packed -> y (N, units).
 * pool must be 1.
 */
int qnn_dense_forward(const qnn_weights_t* w, const void* x, int x_store, int x_bits,
                      int N, const qnn_epilogue_t* epi, void* y, void* stream);

/*
 * The last convolution group of models/vgg.py (conv -> BN -> act -> MaxPooling2D, lines 32-37) and the classifier
 * behind it (Flatten -> Dense -> BN, lines 38-42) in ONE launch:  y = dense(flatten(conv_group(x))), float32 (N, units).
 * Bit-identical to qnn_conv2d_forward(wc, ..., epi_conv{out_store = QNN_STORE_I4}) followed by
 * qnn_dense_forward(wd, ...): the activation codes never leave the chip.  Covers: int4-stored input, 3x3 stride-1 'same'
 * conv of 64 / 128 channels into 64 filters, 2x2 pool down to a 4 x 4 map, dense 1024 -> at most 16 units prepacked for
 * QNN_STORE_I4, float32 output without activation.  Anything else returns QNN_EUNSUPPORTED: issue the two calls.
 */
int qnn_conv2d_dense_forward(const qnn_weights_t* wconv, const qnn_weights_t* wdense, const void* x, int x_store,
                             int x_bits, int N, int H, int W, const qnn_epilogue_t* epi_conv,
                             const qnn_epilogue_t* epi_dense, float* y, void* stream);

/* Name of the kernel variant the last qnn_conv2d_forward / qnn_dense_forward on
 * this thread dispatched to ("ps_bin_cw2_k3", "generic", "mfma_i8", ...). */
const char* qnn_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* QNN_ABI_H */
