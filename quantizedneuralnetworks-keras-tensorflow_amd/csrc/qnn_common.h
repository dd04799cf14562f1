// Shared host/device helpers of the gfx950 low-bit forward engine.
// Written for CDNA4 only (wave64); no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "qnn_abi.h"

#define QNN_WAVE 64

// ---- error plumbing --------------------------------------------------------
void qnn_set_error(const char* fmt, ...);
void qnn_set_kernel_name(const char* name);

#define QNN_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            qnn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                          __FILE__, __LINE__);                                 \
            return QNN_EHIP;                                                   \
        }                                                                      \
    } while (0)

#define QNN_REQUIRE(cond, code, ...)                                           \
    do {                                                                       \
        if (!(cond)) {                                                         \
            qnn_set_error(__VA_ARGS__);                                        \
            return (code);                                                     \
        }                                                                      \
    } while (0)

// ---- measurement scaffolding ------------------------------------------------------
// A/B switches read from the environment and the operand-load ablations of the GEMM exist ONLY in builds made with
// -DQNN_EXPERIMENTS (tools/build_variant.py <name> <file.hip> -DQNN_EXPERIMENTS ...).  The default library reads no
// environment variable on its dispatch paths and carries no ablation code: every switch is its compiled-in default.
#ifdef QNN_EXPERIMENTS
#define QNN_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define QNN_ENV_STR(name) getenv(name)
#define QNN_ABLATE(word, bit) (((word) & (bit)) != 0)
#else
#define QNN_ENV_INT(name, dflt) (dflt)
#define QNN_ENV_STR(name) ((const char*)nullptr)
#define QNN_ABLATE(word, bit) false
#endif

// ---- opaque weights handle --------------------------------------------------
struct qnn_weights {
    int wkind, wbits;
    float H;
    int kh, kw, cin, cout;
    int stride, same_pad;
    int store;        // QNN_STORE_F32 (float path only) | BIN | I4 | I8
    int cw;           // packed words per tap (ceil(cin / per_word))
    int kwords;       // kh*kw*cw
    int wshift;       // quantized value = code * 2^-wshift
    uint32_t* d_packed;   // [cout][kh*kw][cw]   packed codes (store != F32)
    float* d_wq;          // [cout][kh*kw][cin]  quantized values as float32
    float* d_bias;        // [cout] or nullptr
    int32_t* d_corr;      // BIN + same_pad: [64][cout] zero-padding corrections
    uint8_t* d_mfma;      // int8 image [cout][kh*kw][cin] for the MFMA kernel (may alias d_packed)
    void* d_mfma_own;     // owned allocation behind d_mfma (I4 store), or nullptr
    void* d_aux;          // ternary: the 0.7*mean|W| cutoff (1 float), else nullptr
    uint32_t* d_head;     // dense 1024 -> <= 16 int4 heads: the per-lane table of the fused conv + classifier kernel
    uint32_t* h_flag;     // domain flag (pinned, device-visible host word; qnn_weights_check) or nullptr
    uint32_t* d_flag;     // the same word through the device's address space
};

struct ConvGeom;
struct EpiArgs;
int qnn_mfma_prepare_weights(qnn_weights* w, hipStream_t s);
int qnn_try_launch_mfma(const ConvGeom& g, const EpiArgs& e, int x_store, const void* x,
                        const qnn_weights* w, void* y, hipStream_t s, char* name, size_t name_len);
int qnn_conv_impl_pref();
int qnn_head_prepare(qnn_weights* w, hipStream_t s);
int qnn_try_launch_stem(const ConvGeom& g, const EpiArgs& e, const void* x, const float* wq, void* y, hipStream_t s);
int qnn_try_launch_first_fixed(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w, void* y,
                               hipStream_t s);
int qnn_try_launch_first_u8(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w, void* y,
                            hipStream_t s, bool f32in);

// ---- division by a launch-constant via multiply-high (dividends < 2^31) ----------
struct FastDiv {
    uint32_t d, m;
    int s;
};
static inline FastDiv qnn_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d; f.m = 0; f.s = 0;
    if (d > 1) {
        int s = 0;
        while ((1ull << s) < d) ++s;
        f.s = s;
        f.m = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
    }
    return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t qnn_div(uint32_t q, const FastDiv& f) {
    return f.d == 1 ? q : (__umulhi(q, f.m) >> (f.s - 1));
}
#endif

// ---- geometry shared by every conv kernel ------------------------------------
struct ConvGeom {
    int N, H, W, Ho, Wo;       // Ho/Wo: conv output (before pooling)
    int cin, cout, kh, kw, stride;
    int pt, pl;                // SAME padding before (top/left)
    int cw, kwords;
    int pool;                  // 1 or 2
    int Hp, Wp;                // stored output size (Ho/pool, Wo/pool)
    FastDiv fd_wp, fd_hp;      // q -> (n, py, px) without integer division
};

struct EpiArgs {
    const float* bias;         // or nullptr
    const float* bn_inv;       // or nullptr
    const float* bn_shift;
    float scale;               // 2^-(wshift+xshift); QNN_STORE_U8 input: the divisor 255 * 2^wshift
    float act_m;               // 2^(act_bits-1) for quantized_tanh
    int fn;
    int out_store;
    int ocw;                   // words per stored output pixel (packed) or cout (f32)
    const void* res;           // residual tensor (same pixels x cout) or nullptr
    int res_store;
    int res_cw;                // words (packed) or floats (f32) per residual pixel
    float res_scale;           // 2^-(res_bits-1)
    float post_scale;
    float trick_c, trick_s;    // output-side identity trick of the reference ("faithful" mode); trick_s == 0: off
    const float* fold_a;       // qnn_fold_t of this layer + epilogue (qnn_fold.h): per-channel slope, or nullptr = evaluate
    const int32_t* fold_b;     // the float32 chain; per-channel accumulator offset in units of acc / 256
    uint32_t flags;            // QNN_EPI_* kernel-selection bits of this call (qnn_epilogue_t.flags)
    int first_mode;            // domain declared for this call's float32 input: 0 none, 1 image bytes / 255, 2 [0, 1]
    uint32_t* dom_flag;        // the caller's domain-flag word for this call (qnn_epilogue_t.domain_flag) or nullptr = the handle's
    const float* fold_c;       // "bits" form of the fold (fold_b carries 0x4B400000, u = fma(as_float(acc), a, c)), or nullptr
    // projection shortcut computed inside the launch (qnn_projection_t, qnn_abi.h), or proj_x == nullptr
    const uint8_t* proj_x;     // block input: N x proj_H x proj_W x proj_cin int4 codes
    const uint8_t* proj_w;     // [cout][proj_cin] bytes code * 16 (the 1x1 kernel's d_mfma image)
    const float* proj_bias;    // or nullptr
    float proj_scale;          // 2^-(wshift + x_bits - 1) of the projection
    int proj_cin, proj_H, proj_W;
};

#ifdef __HIPCC__
#define QNN_HD __host__ __device__
#else
#define QNN_HD
#endif

QNN_HD static inline int qnn_per_word(int store) {
    return store == QNN_STORE_BIN ? 32 : store == QNN_STORE_I4 ? 8 : store == QNN_STORE_I8 ? 4 : 1;
}
QNN_HD static inline int qnn_words(int store, int channels) {
    if (store == QNN_STORE_T2) return 2 * ((channels + 31) / 32);     // (mask, sign) word pairs
    int pw = qnn_per_word(store);
    return (channels + pw - 1) / pw;
}

// TF 'SAME' padding (documented TF semantics; oracle: same_padding()).
static inline void qnn_same_pad(int in, int k, int s, int same, int* out, int* before) {
    if (same) {
        *out = (in + s - 1) / s;
        int total = (*out - 1) * s + k - in;
        if (total < 0) total = 0;
        *before = total / 2;
    } else {
        *out = (in - k) / s + 1;
        *before = 0;
    }
}

#ifdef __HIPCC__
// ---- float32 activation clips, replayed op by op ------------------------------
// All arithmetic uses the explicitly rounded intrinsics so that no FMA
// contraction can change a rounding the reference performs.

// binary_ops.py:16-24,8-13,37-51: 2*round_through(clip(0.5x+0.5,0,1)) - 1
__device__ __forceinline__ float qnn_binary_tanh(float x) {
    float h = __fadd_rn(__fmul_rn(0.5f, x), 0.5f);
    h = fminf(fmaxf(h, 0.0f), 1.0f);
    float r = rintf(h);                          // tf.round: half to even
    float rt = __fadd_rn(h, __fsub_rn(r, h));    // x + stop_gradient(round(x) - x)
    return __fsub_rn(__fmul_rn(2.0f, rt), 1.0f);
}
// the one bit of information in binary_tanh(x): 1 <=> +1
__device__ __forceinline__ uint32_t qnn_binary_bit(float x) {
    return qnn_binary_tanh(x) > 0.0f ? 1u : 0u;
}
// quantized_ops.py:49-66,87-100 in code units: clip(round_through(x*m), -m, m-1)
__device__ __forceinline__ float qnn_quant_code_f(float x, float m) {
    float t = __fmul_rn(x, m);
    float r = rintf(t);
    float rt = __fadd_rn(t, __fsub_rn(r, t));
    return fminf(fmaxf(rt, -m), __fsub_rn(m, 1.0f));
}
// m is always 2^(bits-1): dividing by it equals multiplying by its exact reciprocal (one integer
// subtraction on the exponent field) -- an IEEE divide costs ~10 instructions per value, which made
// every float32-output epilogue with a fused clip VALU-bound
__device__ __forceinline__ float qnn_quantized_tanh(float x, float m) {
    const float inv_m = __uint_as_float(0x7F000000u - __float_as_uint(m));
    return __fmul_rn(qnn_quant_code_f(x, m), inv_m);
}

// ---- dot products on packed words ----------------------------------------------
// popcount(a ^ w): number of channels whose signs differ.
__device__ __forceinline__ int qnn_dot_bin(uint32_t a, uint32_t w, int acc) {
    return acc + __popc(a ^ w);
}
// Same, with the accumulate form of v_bcnt_u32_b32 (D = popcount(S0) + S1) pinned: for
// callers that keep several independent chains themselves (the compiler otherwise emits
// bcnt(x, 0) plus a tree of v_add3, ~22 % more VALU instructions).
__device__ __forceinline__ int qnn_dot_bin_chain(uint32_t a, uint32_t w, int acc) {
    const uint32_t x = a ^ w;
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// 8 x (int4 * int4) + acc  -> v_dot8_i32_i4
__device__ __forceinline__ int qnn_dot_i4(uint32_t a, uint32_t w, int acc) {
    return __builtin_amdgcn_sdot8((int)a, (int)w, acc, false);
}
// 4 x (int8 * int8) + acc  -> v_dot4_i32_i8
__device__ __forceinline__ int qnn_dot_i8(uint32_t a, uint32_t w, int acc) {
    return __builtin_amdgcn_sdot4((int)a, (int)w, acc, false);
}

// ternary x ternary on (mask, sign) word pairs: popc(m) - 2 popc(m & (sa ^ sw)),  m = ma & mw
__device__ __forceinline__ int qnn_dot_t2(uint32_t ma, uint32_t sa, uint32_t mw, uint32_t sw, int acc) {
    const uint32_t m = ma & mw;
    return acc + __popc(m) - 2 * __popc(m & (sa ^ sw));
}

template <int STORE>
__device__ __forceinline__ int qnn_dot(uint32_t a, uint32_t w, int acc) {
    if constexpr (STORE == QNN_STORE_BIN) return qnn_dot_bin(a, w, acc);
    else if constexpr (STORE == QNN_STORE_I4) return qnn_dot_i4(a, w, acc);
    else return qnn_dot_i8(a, w, acc);
}

// ---- epilogue: acc -> float value in the reference's op order -------------------
__device__ __forceinline__ float qnn_epi_value(float v, int c, const EpiArgs& e) {
    // binary_layers.py:175-176: (o - (1 - 1/klm) * o) * klm, three float32 roundings ("faithful" mode only)
    if (e.trick_s != 0.0f) v = __fmul_rn(__fsub_rn(v, __fmul_rn(e.trick_c, v)), e.trick_s);
    if (e.bias) v = __fadd_rn(v, e.bias[c]);
    if (e.bn_inv) v = __fadd_rn(__fmul_rn(v, e.bn_inv[c]), e.bn_shift[c]);
    return v;
}
// residual merge of models/resnet.py:127-128: (x + y) * post_scale, float32, two roundings
__device__ __forceinline__ float qnn_epi_residual(float v, long q, int c, const EpiArgs& e) {
    if (!e.res) return v;
    float r;
    if (e.res_store == QNN_STORE_F32) {
        r = ((const float*)e.res)[q * e.res_cw + c];
    } else if (e.res_store == QNN_STORE_BIN) {
        const uint32_t w = ((const uint32_t*)e.res)[q * e.res_cw + (c >> 5)];
        r = ((w >> (c & 31)) & 1u) ? 1.0f : -1.0f;
    } else if (e.res_store == QNN_STORE_I4) {
        const uint32_t w = ((const uint32_t*)e.res)[q * e.res_cw + (c >> 3)];
        const int code = (int)(w << (28 - 4 * (c & 7))) >> 28;
        r = __fmul_rn((float)code, e.res_scale);
    } else {
        const uint32_t w = ((const uint32_t*)e.res)[q * e.res_cw + (c >> 2)];
        const int code = (int)(w << (24 - 8 * (c & 3))) >> 24;
        r = __fmul_rn((float)code, e.res_scale);
    }
    return __fmul_rn(__fadd_rn(r, v), e.post_scale);
}
// ---- QNN_STORE_U8 input: the affine map behind the exact integer sum (qnn_abi.h, qnn_conv2d_forward) ----
struct U8Affine {
    float A, B;
};
__device__ __forceinline__ U8Affine qnn_u8_affine(const EpiArgs& e, int c) {
    const double inv = e.bn_inv ? (double)e.bn_inv[c] : 1.0, shift = e.bn_inv ? (double)e.bn_shift[c] : 0.0;
    const double bias = e.bias ? (double)e.bias[c] : 0.0;
    const double m = e.fn == QNN_FN_QUANTIZED_TANH ? (double)e.act_m : 1.0;
    U8Affine a;
    a.A = (float)(inv * m / (double)e.scale);
    a.B = (float)((bias * inv + shift) * m);
    return a;
}
// t = fma(S, A, B) -> what is pooled: the integer code (quantized_tanh), +-1 (binary_tanh), or t itself
__device__ __forceinline__ float qnn_u8_value(float t, const EpiArgs& e) {
    if (e.fn == QNN_FN_BINARY_TANH) return t > 0x1p-24f ? 1.0f : -1.0f;
    if (e.fn == QNN_FN_QUANTIZED_TANH) return fminf(fmaxf(rintf(t), -e.act_m), e.act_m - 1.0f);
    return t;
}
// value -> stored code (BIN: 0/1; I4/I8: signed code) or float bits
__device__ __forceinline__ int qnn_epi_code(float v, const EpiArgs& e) {
    if (e.fn == QNN_FN_BINARY_TANH) {
        const int b = (int)qnn_binary_bit(v);
        return e.out_store == QNN_STORE_BIN ? b : 2 * b - 1;   // +-1 as a signed code
    }
    return (int)qnn_quant_code_f(v, e.act_m);
}
#endif  // __HIPCC__
