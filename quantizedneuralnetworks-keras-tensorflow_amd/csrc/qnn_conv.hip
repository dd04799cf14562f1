// Low-bit convolution / dense contractions for gfx950.
//
// Reference ops replaced: BinaryConv2D.call (layers/binary_layers.py:160-187),
// QuantizedConv2D.call (layers/quantized_layers.py:164-194), BinaryDense.call
// (binary_layers.py:78-85), QuantizedDense.call (quantized_layers.py:79-88) and the
// weight quantizers binarize / quantize (binary_ops.py:54-64, quantized_ops.py:49-66),
// plus, fused behind them, bias_add, inference BatchNormalization, the activation
// clip and MaxPooling2D of models/vgg.py:15-42.
//
// Two kernel families:
//   k_conv_ps       "pixel-stationary": one lane owns one output pixel and keeps
//                   that pixel's whole receptive field (kh*kw*cw packed words) in
//                   VGPRs; the wave walks the output channels and receives each
//                   filter through SCALAR loads (the filter address is wave-uniform),
//                   so the inner loop is v_xor+v_bcnt / v_dot8 / v_dot4 with an SGPR
//                   operand and no LDS or vector-memory traffic at all.  2x2 max-pool
//                   is a DPP quad reduction (lanes 4q..4q+3 hold one pool window);
//                   packed outputs are assembled in a register and stored once.
//   k_conv_generic  one thread per stored output element/word, runtime loops; any
//                   shape, any stride, float32 inputs; the correctness fallback.
#include <stdlib.h>

#include <type_traits>

#include "qnn_common.h"
#include "qnn_fold.h"
#include "qnn_mfma_common.h"

#ifndef QNN_XNOR_U
#define QNN_XNOR_U 16
#endif


namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// weights: quantize + pack on device (once per set_weights)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float quantize_weight(float w, int wkind, float H, float m, float cutoff) {
    if (wkind == QNN_W_BINARY) return __fmul_rn(H, qnn_binary_tanh(__fdiv_rn(w, H)));
    if (wkind == QNN_W_QUANT) return qnn_quantized_tanh(w, m);
    if (wkind == QNN_W_TERNARY) {
        // ternary_ops.py:21-27: W/H > cutoff -> 1, W/H <= -cutoff -> -1, else 0; times H.
        // ("exact" mode: the straight-through W + (Wt - W) of ternary_ops.py:41 is taken as Wt.)
        const float u = __fdiv_rn(w, H);
        const float t = u > cutoff ? 1.0f : (u <= -cutoff ? -1.0f : 0.0f);
        return __fmul_rn(t, H);
    }
    return w;
}

// ternary_ops.py:15-30: cutoff = 0.7 * mean(|W / H|) over the WHOLE kernel.  One block,
// fixed reduction order (deterministic), double accumulation.
__global__ __launch_bounds__(1024) void k_tern_cutoff(const float* __restrict__ kernel, int n, float H,
                                                      float* __restrict__ cutoff) {
    __shared__ double part[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)fabsf(__fdiv_rn(kernel[i], H));
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) cutoff[0] = __fmul_rn(0.7f, (float)(part[0] / (double)n));
}

// d_wq[c][t][ci] = quantizer(kernel_hwio[t][ci][c])
__global__ __launch_bounds__(kBlock) void k_prepack_float(const float* __restrict__ kernel,
                                                          float* __restrict__ wq, int taps,
                                                          int cin, int cout, int wkind, float H,
                                                          float m, const float* __restrict__ cutoff_p) {
    const int total = cout * taps * cin;
    const float cutoff = cutoff_p ? cutoff_p[0] : 0.0f;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const int ci = i % cin;
        const int t = (i / cin) % taps;
        const int c = i / (cin * taps);
        wq[i] = quantize_weight(kernel[((size_t)t * cin + ci) * cout + c], wkind, H, m, cutoff);
    }
}

// d_packed[c][t][w]: codes of the quantized values
template <int STORE>
__global__ __launch_bounds__(kBlock) void k_prepack_codes(const float* __restrict__ wq,
                                                          uint32_t* __restrict__ packed, int taps,
                                                          int cin, int cout, int cw, float code_m) {
    constexpr int PW = (STORE == QNN_STORE_BIN) ? 32 : (STORE == QNN_STORE_I4) ? 8 : 4;
    constexpr int BITS = 32 / PW;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int total = cout * taps * cw;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const int w = i % cw;
        const int ct = i / cw;   // c*taps + t
        const float* src = wq + (size_t)ct * cin + w * PW;
        const int left = cin - w * PW;
        uint32_t out = 0;
        for (int j = 0; j < PW && j < left; ++j) {
            uint32_t code;
            if constexpr (STORE == QNN_STORE_BIN) code = src[j] > 0.0f ? 1u : 0u;
            else code = (uint32_t)(int)__fmul_rn(src[j], code_m);
            out |= (code & MASK) << (j * BITS);
        }
        packed[i] = out;
    }
}

// ternary (or binary) weight values -> (mask, sign) word pairs per 32 input channels, QNN_STORE_T2
__global__ __launch_bounds__(kBlock) void k_prepack_t2(const float* __restrict__ wq, uint32_t* __restrict__ packed,
                                                       int taps, int cin, int cout, int pairs) {
    const int total = cout * taps * pairs;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const int w = i % pairs;
        const int ct = i / pairs;   // c*taps + t
        const float* src = wq + (size_t)ct * cin + w * 32;
        const int left = cin - w * 32;
        uint32_t mask = 0, sign = 0;
        for (int j = 0; j < 32 && j < left; ++j) {
            mask |= (src[j] != 0.0f ? 1u : 0u) << j;
            sign |= (src[j] > 0.0f ? 1u : 0u) << j;
        }
        packed[2 * i] = mask;
        packed[2 * i + 1] = sign;
    }
}

// BIN zero-padding corrections.  Out-of-image taps are fed as all-zero words
// (= every channel -1), so their spurious contribution sum_c (-1)*w_c must be
// removed: corr[rmask*8+cmask][c] = sum over taps (dy,dx) with dy in rmask or dx in
// cmask of sum_ci sign(w[dy][dx][ci][c]).
__global__ __launch_bounds__(kBlock) void k_corr_table(const float* __restrict__ wq,
                                                       int32_t* __restrict__ corr, int kh, int kw,
                                                       int cin, int cout) {
    const int total = 64 * cout;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const int c = i % cout;
        const int cls = i / cout;
        const int rmask = cls >> 3, cmask = cls & 7;
        int s = 0;
        for (int dy = 0; dy < kh; ++dy)
            for (int dx = 0; dx < kw; ++dx) {
                if (!(((rmask >> dy) & 1) | ((cmask >> dx) & 1))) continue;
                const float* src = wq + ((size_t)c * kh * kw + dy * kw + dx) * cin;
                for (int ci = 0; ci < cin; ++ci) s += src[ci] > 0.0f ? 1 : -1;
            }
        corr[i] = s;
    }
}

// d_wq[c][t][ci] -> HWIO float32 (tests)
__global__ __launch_bounds__(kBlock) void k_dequant_hwio(const float* __restrict__ wq,
                                                         float* __restrict__ kernel, int taps,
                                                         int cin, int cout) {
    const int total = cout * taps * cin;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const int ci = i % cin;
        const int t = (i / cin) % taps;
        const int c = i / (cin * taps);
        kernel[((size_t)t * cin + ci) * cout + c] = wq[i];
    }
}

// ---------------------------------------------------------------------------
// generic kernel
// ---------------------------------------------------------------------------
// conv result (before bias) of output channel c at conv pixel (n, oy, ox)
__device__ float conv_point(const ConvGeom& g, int x_store, const void* __restrict__ x,
                            const uint32_t* __restrict__ wp, const float* __restrict__ wq,
                            float scale, int n, int oy, int ox, int c) {
    if (x_store == QNN_STORE_F32) {
        const float* xf = (const float*)x;
        float acc = 0.0f;
        for (int dy = 0; dy < g.kh; ++dy) {
            const int iy = oy * g.stride + dy - g.pt;
            if ((unsigned)iy >= (unsigned)g.H) continue;
            for (int dx = 0; dx < g.kw; ++dx) {
                const int ix = ox * g.stride + dx - g.pl;
                if ((unsigned)ix >= (unsigned)g.W) continue;
                const float* a = xf + (((size_t)n * g.H + iy) * g.W + ix) * g.cin;
                const float* w = wq + ((size_t)c * g.kh * g.kw + dy * g.kw + dx) * g.cin;
                for (int ci = 0; ci < g.cin; ++ci) acc = fmaf(a[ci], w[ci], acc);
            }
        }
        return acc;
    }
    if (x_store == QNN_STORE_U8) {
        // image bytes x integer weight codes (scale = 255 * 2^wshift): the exact integer S, as a float (|S| < 2^24)
        const uint8_t* xb = (const uint8_t*)x;
        const float wscale = scale * (1.0f / 255.0f);            // 2^wshift, exact
        int acc = 0;
        for (int dy = 0; dy < g.kh; ++dy) {
            const int iy = oy * g.stride + dy - g.pt;
            if ((unsigned)iy >= (unsigned)g.H) continue;
            for (int dx = 0; dx < g.kw; ++dx) {
                const int ix = ox * g.stride + dx - g.pl;
                if ((unsigned)ix >= (unsigned)g.W) continue;
                const uint8_t* a = xb + (((size_t)n * g.H + iy) * g.W + ix) * g.cin;
                const float* w = wq + ((size_t)c * g.kh * g.kw + dy * g.kw + dx) * g.cin;
                for (int ci = 0; ci < g.cin; ++ci) acc += (int)a[ci] * (int)rintf(__fmul_rn(w[ci], wscale));
            }
        }
        return (float)acc;
    }
    const uint32_t* xa = (const uint32_t*)x;
    int acc = 0;
    for (int dy = 0; dy < g.kh; ++dy) {
        const int iy = oy * g.stride + dy - g.pt;
        if ((unsigned)iy >= (unsigned)g.H) continue;
        for (int dx = 0; dx < g.kw; ++dx) {
            const int ix = ox * g.stride + dx - g.pl;
            if ((unsigned)ix >= (unsigned)g.W) continue;
            const uint32_t* a = xa + (((size_t)n * g.H + iy) * g.W + ix) * g.cw;
            const uint32_t* w = wp + ((size_t)c * g.kh * g.kw + dy * g.kw + dx) * g.cw;
            if (x_store == QNN_STORE_BIN) {
                int p = 0;
                for (int j = 0; j < g.cw; ++j) p += __popc(a[j] ^ w[j]);
                acc += g.cin - 2 * p;   // pad bits are 0 in both operands
            } else if (x_store == QNN_STORE_T2) {
                for (int j = 0; j < g.cw; j += 2) acc = qnn_dot_t2(a[j], a[j + 1], w[j], w[j + 1], acc);
            } else if (x_store == QNN_STORE_I4) {
                for (int j = 0; j < g.cw; ++j) acc = qnn_dot_i4(a[j], w[j], acc);
            } else {
                for (int j = 0; j < g.cw; ++j) acc = qnn_dot_i8(a[j], w[j], acc);
            }
        }
    }
    return __fmul_rn((float)acc, scale);
}

// one thread per stored output slot: (stored pixel, cout) for float32 outputs,
// (stored pixel, word) for packed outputs
__global__ __launch_bounds__(kBlock) void k_conv_generic(ConvGeom g, EpiArgs e, int x_store,
                                                         const void* __restrict__ x,
                                                         const uint32_t* __restrict__ wp,
                                                         const float* __restrict__ wq,
                                                         void* __restrict__ y) {
    const int slots = e.ocw;
    const size_t total = (size_t)g.N * g.Hp * g.Wp * slots;
    const int pw = qnn_per_word(e.out_store);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const int slot = (int)(i % slots);
        const size_t q = i / slots;
        const int px = (int)(q % g.Wp);
        const int py = (int)((q / g.Wp) % g.Hp);
        const int n = (int)(q / ((size_t)g.Wp * g.Hp));
        if (x_store == QNN_STORE_U8) {
            // typed image input: one FMA behind the exact integer sum (qnn_abi.h, qnn_conv2d_forward)
            const int bits = e.out_store == QNN_STORE_F32 ? 32 : 32 / pw;
            uint32_t word = 0;
            for (int b = 0; b < (e.out_store == QNN_STORE_F32 ? 1 : pw); ++b) {
                const int c = e.out_store == QNN_STORE_F32 ? slot : slot * pw + b;
                if (c >= g.cout) break;
                const U8Affine af = qnn_u8_affine(e, c);
                float best = 0.0f;
                for (int s = 0; s < g.pool * g.pool; ++s) {
                    const int oy = py * g.pool + s / g.pool, ox = px * g.pool + s % g.pool;
                    const float t = __fmaf_rn(conv_point(g, x_store, x, wp, wq, e.scale, n, oy, ox, c), af.A, af.B);
                    const float v = qnn_u8_value(t, e);          // code (quantized_tanh), +-1 (binary_tanh) or t
                    best = (s == 0) ? v : fmaxf(best, v);
                }
                if (e.out_store == QNN_STORE_F32) {
                    if (e.fn == QNN_FN_QUANTIZED_TANH)
                        best = __fmul_rn(best, __uint_as_float(0x7F000000u - __float_as_uint(e.act_m)));
                    ((float*)y)[i] = best;
                } else {
                    const int code = e.out_store == QNN_STORE_BIN ? (best > 0.0f ? 1 : 0) : (int)best;
                    word |= ((uint32_t)code & ((1u << bits) - 1u)) << (b * bits);
                }
            }
            if (e.out_store != QNN_STORE_F32) ((uint32_t*)y)[i] = word;
        } else if (e.out_store == QNN_STORE_F32) {
            const int c = slot;
            float best = 0.0f;
            for (int s = 0; s < g.pool * g.pool; ++s) {
                const int oy = py * g.pool + s / g.pool, ox = px * g.pool + s % g.pool;
                float v = conv_point(g, x_store, x, wp, wq, e.scale, n, oy, ox, c);
                v = qnn_epi_value(v, c, e);
                v = qnn_epi_residual(v, (long)q, c, e);
                if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
                else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
                best = (s == 0) ? v : fmaxf(best, v);
            }
            ((float*)y)[i] = best;
        } else {
            const int bits = 32 / pw;
            const uint32_t mask = (1u << bits) - 1u;
            uint32_t word = 0;
            for (int b = 0; b < pw; ++b) {
                const int c = slot * pw + b;
                if (c >= g.cout) break;
                int best = 0;
                for (int s = 0; s < g.pool * g.pool; ++s) {
                    const int oy = py * g.pool + s / g.pool, ox = px * g.pool + s % g.pool;
                    float v = conv_point(g, x_store, x, wp, wq, e.scale, n, oy, ox, c);
                    v = qnn_epi_value(v, c, e);
                    v = qnn_epi_residual(v, (long)q, c, e);
                    const int code = qnn_epi_code(v, e);
                    best = (s == 0) ? code : max(best, code);
                }
                word |= ((uint32_t)best & mask) << (b * bits);
            }
            ((uint32_t*)y)[i] = word;
        }
    }
}

// ---------------------------------------------------------------------------
// dense layer on float32 features (the classifier behind a global average pool,
// models/resnet.py:138-142; the 'qnn' / 'bnn' / 'tnn' networks whose activations are
// LeakyReLU floats).  K.dot is a float32 matmul whose summation order TensorFlow does
// not specify; this kernel returns the CORRECTLY ROUNDED dot product: every float32
// product is exact in float64, the 64 lanes accumulate strided partial sums in float64
// (relative error ~1e-16), one butterfly reduction, one rounding to float32.  That is
// the oracle's `dot` (float64 accumulate, one rounding) and is within half an ulp of
// the exact value -- a float32 fmaf chain over K = 1024 unit-scale terms is already
// 1.8e-5 away from it (measured), outside the 1e-5 band of the north star.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dense_f32in(EpiArgs e, int N, int cin, int cout,
                                                    const float* __restrict__ x,
                                                    const float* __restrict__ wq,
                                                    float* __restrict__ y) {
    const int t = blockIdx.x;
    const int n = t / cout, c = t - n * cout;
    const int lane = threadIdx.x;
    const float* a = x + (size_t)n * cin;
    const float* w = wq + (size_t)c * cin;
    double acc0 = 0.0, acc1 = 0.0;
    if ((cin & 3) == 0) {
        for (int k = 4 * lane; k < cin; k += 256) {        // coalesced 16-byte loads of both vectors
            const float4 av = *reinterpret_cast<const float4*>(a + k);
            const float4 wv = *reinterpret_cast<const float4*>(w + k);
            acc0 += (double)av.x * (double)wv.x;
            acc1 += (double)av.y * (double)wv.y;
            acc0 += (double)av.z * (double)wv.z;
            acc1 += (double)av.w * (double)wv.w;
        }
    } else {
        for (int k = lane; k < cin; k += 64) acc0 += (double)a[k] * (double)w[k];
    }
    double sum = acc0 + acc1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float acc = (float)sum;
    if (lane == 0) {
        float v = qnn_epi_value(acc, c, e);
        if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
        else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
        y[t] = v;
    }
}

// ---------------------------------------------------------------------------
// 1x1 convolution of a packed int4 tensor with float32 output: the linear projection shortcuts of
// models/resnet.py:117-124 (kernel_size=1, strides=2, no BN, no activation; their value is added to a BN
// output, so it is kept in float32).  HBM-bound: the output is 8x the input.  Eight lanes share a pixel,
// each owns cout/8 consecutive output channels with their filters resident in VGPRs and stores them with
// 16-byte stores: one wave instruction writes 8 pixels x cout floats = a contiguous 1 or 2 KB.  (The
// pixel-stationary kernel below stores one float per lane and instruction, 64 different lines each:
// 119 us for the 224^2 -> 112^2 projection against 20 us of HBM time.)
// ---------------------------------------------------------------------------
template <int CW, int CPL>      // packed words per input pixel, output channels per lane
__global__ __launch_bounds__(kBlock) void k_conv_pw_f32(ConvGeom g, EpiArgs e, const uint32_t* __restrict__ x,
                                                        const uint32_t* __restrict__ wp,
                                                        float* __restrict__ y, long total_q) {
    const int sub = threadIdx.x & 7;                        // channel group of this lane
    const int c0 = sub * CPL;
    uint32_t wv[CPL][CW];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int k = 0; k < CW; ++k) wv[c][k] = wp[(size_t)(c0 + c) * CW + k];
    float bs[CPL], iv[CPL], sh[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        bs[c] = e.bias ? e.bias[c0 + c] : 0.0f;
        iv[c] = e.bn_inv ? e.bn_inv[c0 + c] : 1.0f;
        sh[c] = e.bn_inv ? e.bn_shift[c0 + c] : 0.0f;
    }
    const bool has_bias = e.bias != nullptr, has_bn = e.bn_inv != nullptr;
    const long stride_q = (long)gridDim.x * (kBlock / 8);
    for (long q = (long)blockIdx.x * (kBlock / 8) + (threadIdx.x >> 3); q < total_q; q += stride_q) {
        const uint32_t qrow = qnn_div((uint32_t)q, g.fd_wp);           // n*Ho + oy  (Wp == Wo, no pooling)
        const int ox = (int)((uint32_t)q - qrow * (uint32_t)g.Wp);
        const uint32_t n = qnn_div(qrow, g.fd_hp);
        const int oy = (int)(qrow - n * (uint32_t)g.Hp);
        const uint32_t* xp = x + (((size_t)n * g.H + (size_t)oy * g.stride) * g.W + (size_t)ox * g.stride) * CW;
        uint32_t xv[CW];
        if constexpr (CW == 2) { const uint2 t = *reinterpret_cast<const uint2*>(xp); xv[0] = t.x; xv[1] = t.y; }
        else if constexpr (CW == 4) { const uint4 t = *reinterpret_cast<const uint4*>(xp); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
        else {
#pragma unroll
            for (int k = 0; k < CW; ++k) xv[k] = xp[k];
        }
        float out[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < CW; ++k) acc = qnn_dot_i4(xv[k], wv[c][k], acc);
            float v = __fmul_rn((float)acc, e.scale);
            if (has_bias) v = __fadd_rn(v, bs[c]);
            if (has_bn) v = __fadd_rn(__fmul_rn(v, iv[c]), sh[c]);
            out[c] = v;
        }
        float* yp = y + (size_t)q * g.cout + c0;
#pragma unroll
        for (int c = 0; c < CPL; c += 4)
            *reinterpret_cast<float4*>(yp + c) = make_float4(out[c], out[c + 1], out[c + 2], out[c + 3]);
    }
}

// 0 = launched
int try_launch_pw_f32(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w, void* y,
                      hipStream_t s) {
    if (g.kh != 1 || g.kw != 1 || g.pool != 1 || e.res || e.out_store != QNN_STORE_F32 || e.fn != QNN_FN_NONE) return 1;
    if (w->store != QNN_STORE_I4 || g.pt != 0 || g.pl != 0) return 1;
    const int cw = g.cw, cpl = g.cout / 8;
    if (g.cout % 8 != 0 || g.cin % 8 != 0) return 1;
    const long total_q = (long)g.N * g.Ho * g.Wo;
    long blocks = (total_q + kBlock / 8 - 1) / (kBlock / 8);
    if (blocks > 256 * 16) blocks = 256 * 16;
#define PW_CASE(CW_, CPL_)                                                                                   \
    if (cw == CW_ && cpl == CPL_) {                                                                          \
        hipLaunchKernelGGL((k_conv_pw_f32<CW_, CPL_>), dim3((unsigned)blocks), dim3(kBlock), 0, s, g, e,      \
                           (const uint32_t*)x, w->d_packed, (float*)y, total_q);                             \
        return 0;                                                                                            \
    }
    PW_CASE(2, 4) PW_CASE(2, 8) PW_CASE(4, 4) PW_CASE(4, 8) PW_CASE(8, 8) PW_CASE(8, 16)
#undef PW_CASE
    return 1;
}

// ---------------------------------------------------------------------------
// pixel-stationary kernel
// ---------------------------------------------------------------------------
__device__ __forceinline__ int quad_max_i(int v) {
    int t = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = max(v, t);
    t = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    return max(v, t);
}
__device__ __forceinline__ float quad_max_f(float v) {
    float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v = fmaxf(v, t);
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    return fmaxf(v, t);
}

// XS  : storage of x (QNN_STORE_F32 = float32 input, CW = cin floats per tap)
// CW  : words per tap;  K : square kernel size;  OUT : storage of y
template <int XS, int CW, int K, int OUT>
__global__ __launch_bounds__(kBlock) void k_conv_ps(ConvGeom g, EpiArgs e,
                                                    const uint32_t* __restrict__ x,
                                                    const uint32_t* __restrict__ wts,
                                                    const int32_t* __restrict__ corr,
                                                    void* __restrict__ y) {
    constexpr int KWORDS = K * K * CW;
    constexpr int PWO = (OUT == QNN_STORE_BIN) ? 32 : (OUT == QNN_STORE_I4) ? 8 : 4;  // F32: 4 floats
    constexpr int OBITS = (OUT == QNN_STORE_F32) ? 32 : 32 / PWO;

    // ---- lane -> conv pixel (pool windows occupy aligned lane quads) ----
    const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total_q = (uint32_t)g.N * g.Hp * g.Wp;
    const uint32_t q0 = (g.pool == 2) ? (gl >> 2) : gl;
    const int sub = (g.pool == 2) ? (int)(gl & 3) : 0;
    const bool live = q0 < total_q;
    const uint32_t q = live ? q0 : total_q - 1;
    const uint32_t qrow = qnn_div(q, g.fd_wp);
    const int px = (int)(q - qrow * g.Wp);
    const int n = (int)qnn_div(qrow, g.fd_hp);
    const int py = (int)(qrow - (uint32_t)n * g.Hp);
    const int oy = py * g.pool + (sub >> 1);
    const int ox = px * g.pool + (sub & 1);

    // ---- this pixel's receptive field into registers ----
    uint32_t a[KWORDS];
    int rmask = 0, cmask = 0;
#pragma unroll
    for (int dy = 0; dy < K; ++dy) {
        const int iy = oy * g.stride + dy - g.pt;
        const bool rin = (unsigned)iy < (unsigned)g.H;
        if (!rin) rmask |= 1 << dy;
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const int ix = ox * g.stride + dx - g.pl;
            const bool cin_ = (unsigned)ix < (unsigned)g.W;
            if (dy == 0 && !cin_) cmask |= 1 << dx;
            const bool inb = rin && cin_;
            const uint32_t* p = x + (((long)n * g.H + (inb ? iy : 0)) * g.W + (inb ? ix : 0)) * CW;
            if constexpr (CW % 4 == 0) {
#pragma unroll
                for (int j = 0; j < CW; j += 4) {
                    uint4 v = inb ? *reinterpret_cast<const uint4*>(p + j) : make_uint4(0, 0, 0, 0);
                    a[(dy * K + dx) * CW + j + 0] = v.x;
                    a[(dy * K + dx) * CW + j + 1] = v.y;
                    a[(dy * K + dx) * CW + j + 2] = v.z;
                    a[(dy * K + dx) * CW + j + 3] = v.w;
                }
            } else if constexpr (CW % 2 == 0) {
#pragma unroll
                for (int j = 0; j < CW; j += 2) {
                    uint2 v = inb ? *reinterpret_cast<const uint2*>(p + j) : make_uint2(0, 0);
                    a[(dy * K + dx) * CW + j + 0] = v.x;
                    a[(dy * K + dx) * CW + j + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < CW; ++j) a[(dy * K + dx) * CW + j] = inb ? p[j] : 0u;
            }
        }
    }
    const int cls = rmask * 8 + cmask;
    const int ktot = K * K * g.cin;

    // ---- walk the output channels; filters arrive through scalar loads ----
    for (int c0 = 0; c0 < g.cout; c0 += PWO) {
        uint32_t word = 0;
        float fv[4];
#pragma unroll(OUT == QNN_STORE_F32 ? 4 : 2)
        for (int b = 0; b < PWO; ++b) {
            const int c = c0 + b;
            if (OUT != QNN_STORE_F32 && c >= g.cout) break;      // partial last word (wave-uniform)
            const uint32_t* w = wts + (long)c * KWORDS;
            float v;
            if constexpr (XS == QNN_STORE_F32) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < KWORDS; ++k)
                    acc = fmaf(__uint_as_float(a[k]), __uint_as_float(w[k]), acc);
                v = acc;
            } else {
                int acc = 0;
                if constexpr (XS == QNN_STORE_T2) {
#pragma unroll
                    for (int k = 0; k < KWORDS; k += 2) acc = qnn_dot_t2(a[k], a[k + 1], w[k], w[k + 1], acc);
                } else {
#pragma unroll
                    for (int k = 0; k < KWORDS; ++k) acc = qnn_dot<XS>(a[k], w[k], acc);
                }
                if constexpr (XS == QNN_STORE_BIN) {
                    acc = ktot - 2 * acc;
                    if (corr) acc += corr[cls * g.cout + c];
                }
                v = __fmul_rn((float)acc, e.scale);
            }
            v = qnn_epi_value(v, c, e);
            v = qnn_epi_residual(v, (long)q, c, e);
            if constexpr (OUT == QNN_STORE_F32) {
                if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
                else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
                if (g.pool == 2) v = quad_max_f(v);
                fv[b] = v;
            } else {
                int code = qnn_epi_code(v, e);
                if (g.pool == 2) code = quad_max_i(code);
                word |= ((uint32_t)code & ((OBITS == 32) ? 0xffffffffu : ((1u << OBITS) - 1u)))
                        << (b * OBITS);
            }
        }
        if (live && sub == 0) {
            if constexpr (OUT == QNN_STORE_F32)
                *reinterpret_cast<float4*>((float*)y + (long)q * g.cout + c0) =
                    make_float4(fv[0], fv[1], fv[2], fv[3]);
            else
                ((uint32_t*)y)[(long)q * e.ocw + c0 / PWO] = word;
        }
    }
}

// ---------------------------------------------------------------------------
// dense layer on packed activations (BinaryDense.call / QuantizedDense.call,
// binary_layers.py:78-85, quantized_layers.py:79-88): thread = (image, unit), the
// UP unit-slots of one image sit in adjacent lanes so the activation words are a
// broadcast load; weights (<= a few KB) stay in L1.  float32 output, full epilogue.
// ---------------------------------------------------------------------------
template <int XS, int UP>
__global__ __launch_bounds__(kBlock) void k_dense_packed(const uint32_t* __restrict__ x,
                                                         const uint32_t* __restrict__ wp, EpiArgs e,
                                                         float* __restrict__ y, int N, int cin,
                                                         int kwords, int units) {
    constexpr int IPB = kBlock / UP;                 // images per block
    const int u = threadIdx.x % UP;
    const int img = blockIdx.x * IPB + threadIdx.x / UP;
    if (img >= N || u >= units) return;
    const uint32_t* xr = x + (size_t)img * kwords;
    const uint32_t* wr = wp + (size_t)u * kwords;
    int acc = 0;
    int k = 0;
    for (; k + 4 <= kwords; k += 4) {
        const uint4 a = *reinterpret_cast<const uint4*>(xr + k);
        const uint4 w = *reinterpret_cast<const uint4*>(wr + k);
        if constexpr (XS == QNN_STORE_T2) {          // two (mask, sign) pairs
            acc = qnn_dot_t2(a.x, a.y, w.x, w.y, acc);
            acc = qnn_dot_t2(a.z, a.w, w.z, w.w, acc);
        } else {
            acc = qnn_dot<XS>(a.x, w.x, acc);
            acc = qnn_dot<XS>(a.y, w.y, acc);
            acc = qnn_dot<XS>(a.z, w.z, acc);
            acc = qnn_dot<XS>(a.w, w.w, acc);
        }
    }
    if constexpr (XS == QNN_STORE_T2) {
        for (; k < kwords; k += 2) acc = qnn_dot_t2(xr[k], xr[k + 1], wr[k], wr[k + 1], acc);
    } else {
        for (; k < kwords; ++k) acc = qnn_dot<XS>(xr[k], wr[k], acc);
    }
    if constexpr (XS == QNN_STORE_BIN) acc = cin - 2 * acc;   // pad bits are 0 in both operands
    float v = __fmul_rn((float)acc, e.scale);
    v = qnn_epi_value(v, u, e);
    if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
    else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
    y[(size_t)img * units + u] = v;
}

// The same layer with the K loop of one (image, unit) cut over KP adjacent lanes (classifier heads: <= 16 units, so
// one thread per (image, unit) leaves a 4096-image batch with 256 workgroups of long serial loops: 9.8 us for the
// 4096 x 1024 -> 10 head of the CIFAR VGG against 2-3 us of memory time).  Integer partial sums: any order is exact.
template <int XS, int UP, int KP>
__global__ __launch_bounds__(kBlock) void k_dense_packed_split(const uint32_t* __restrict__ x,
                                                               const uint32_t* __restrict__ wp, EpiArgs e,
                                                               float* __restrict__ y, int N, int cin,
                                                               int kwords, int units) {
    constexpr int IPB = kBlock / (UP * KP);          // images per block
    const int kp = threadIdx.x % KP;
    const int u = (threadIdx.x / KP) % UP;
    const int img = blockIdx.x * IPB + threadIdx.x / (UP * KP);
    const bool live = img < N && u < units;
    const uint32_t* xr = x + (size_t)(live ? img : 0) * kwords;
    const uint32_t* wr = wp + (size_t)(live ? u : 0) * kwords;
    int acc = 0;
    for (int k = kp * 4; k + 4 <= kwords; k += 4 * KP) {
        const uint4 a = *reinterpret_cast<const uint4*>(xr + k);
        const uint4 w = *reinterpret_cast<const uint4*>(wr + k);
        if constexpr (XS == QNN_STORE_T2) {
            acc = qnn_dot_t2(a.x, a.y, w.x, w.y, acc);
            acc = qnn_dot_t2(a.z, a.w, w.z, w.w, acc);
        } else {
            acc = qnn_dot<XS>(a.x, w.x, acc);
            acc = qnn_dot<XS>(a.y, w.y, acc);
            acc = qnn_dot<XS>(a.z, w.z, acc);
            acc = qnn_dot<XS>(a.w, w.w, acc);
        }
    }
#pragma unroll
    for (int d = 1; d < KP; d <<= 1) acc += __shfl_xor(acc, d);     // every lane of the group takes part
    if (!live || kp != 0) return;
    if constexpr (XS == QNN_STORE_BIN) acc = cin - 2 * acc;   // pad bits are 0 in both operands
    float v = __fmul_rn((float)acc, e.scale);
    v = qnn_epi_value(v, u, e);
    if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
    else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
    y[(size_t)img * units + u] = v;
}

template <int XS>
int launch_dense(const void* x, const qnn_weights* w, const EpiArgs& e, void* y, int N, hipStream_t s) {
    const int units = w->cout;
    const uint32_t* xu = (const uint32_t*)x;
    if (units <= 16 && (w->kwords % 16) == 0) {
        constexpr int UP = 16, KP = 4;
        const int ipb = kBlock / (UP * KP);
        hipLaunchKernelGGL((k_dense_packed_split<XS, UP, KP>), dim3((N + ipb - 1) / ipb), dim3(kBlock), 0, s, xu,
                           w->d_packed, e, (float*)y, N, w->cin, w->kwords, units);
        return 0;
    }
#define DENSE_CASE(UP)                                                                           \
    if (units <= UP) {                                                                           \
        const int ipb = kBlock / UP;                                                             \
        hipLaunchKernelGGL((k_dense_packed<XS, UP>), dim3((N + ipb - 1) / ipb), dim3(kBlock), 0, s, xu, \
                           w->d_packed, e, (float*)y, N, w->cin, w->kwords, units);              \
        return 0;                                                                                \
    }
    DENSE_CASE(16)
    DENSE_CASE(64)
    DENSE_CASE(256)
#undef DENSE_CASE
    return 1;
}

// ---------------------------------------------------------------------------
// Layer-surface 1-bit conv (traffic model M0): float32 NHWC in, float32 NHWC out,
// BinaryConv2D.call (binary_layers.py:160-187) with the preceding binary_tanh
// (binary_ops.py:37-51) fused on load.  One workgroup owns a strip of rows of one image:
//   phase 1  all waves stream the strip's float32 input (lane = channel: a 256-byte
//            coalesced load per 64 channels), compare against 2^-24 and the wave's
//            64-bit lane mask -- which IS the packed pixel -- goes to LDS;
//   phase 2  lane = output channel with its whole 3x3xCin filter resident in VGPRs; the
//            wave walks its pixels (wave-uniform), reads each tap's words from LDS with
//            a broadcast ds_read, XNOR+popcounts, and stores 64 channels = 256 contiguous
//            bytes per pixel.  Out-of-image taps are skipped by uniform branches, so no
//            correction table is needed here.
// ---------------------------------------------------------------------------
// Lane SEL of (mlo, mhi) takes the wave's 64-bit mask of (v > thr); the other lanes keep their values.  One block of
// assembly: compare into VCC, v_writelane_b32 both halves with an immediate lane select (two SGPR operands would break
// the constant-bus limit; clang 22 has no writelane builtin).  The s_nop covers the VALU-writes-SGPR -> v_writelane
// wait states, which the compiler's hazard recogniser does not insert inside inline assembly (without it the masks
// were wrong on gfx950).
template <int SEL>
__device__ __forceinline__ void qnn_mask_to_lane(float v, float thr, uint32_t& mlo, uint32_t& mhi) {
    asm volatile("v_cmp_gt_f32 vcc, %2, %3\n\ts_nop 4\n\tv_writelane_b32 %0, vcc_lo, %4\n\tv_writelane_b32 %1, vcc_hi, %4"
                 : "+v"(mlo), "+v"(mhi) : "v"(v), "v"(thr), "n"(SEL) : "vcc");
}
// masks of U compares -> lane u holds the 64-bit mask of value u (compile-time unrolled)
template <int I, int U>
__device__ __forceinline__ void qnn_collect_masks(const float (&v)[U], float thr, uint32_t& mlo, uint32_t& mhi) {
    if constexpr (I < U) {
        qnn_mask_to_lane<I>(v[I], thr, mlo, mhi);
        qnn_collect_masks<I + 1, U>(v, thr, mlo, mhi);
    }
}

// phase 2 of the layer-surface XNOR kernels: every wave walks whole rows of the strip whose packed rows sit in `tile`.
// Validity of the 3x3 taps is wave-uniform: the row class (top / middle / bottom) is fixed along a row and the column
// class only differs for the first and last pixel, so each class gets its own straight-line code (compile-time tap
// masks): nine LDS broadcast reads issued up front, no branches.
template <int CW, bool HAS_BN>
__device__ __forceinline__ void xnor_f32_rows(const ConvGeom& g, const EpiArgs& e, const uint2* tile,
                                              const uint32_t (&wreg)[9 * CW], float bias, float inv, float shift,
                                              int n, int r0, int rows_out, int wave, int c, float* __restrict__ y) {
    constexpr int PAIRS = CW / 2;
    const float kf_cin = (float)g.cin;
    auto pixel = [&](auto rm_c, auto cm_c, const uint2* rowbase, float* yrow, int ox) {
        constexpr int RM = decltype(rm_c)::value;      // bit dy set = row dy of the window is outside
        constexpr int CM = decltype(cm_c)::value;      // bit dx set = column dx is outside
        constexpr int NVALID = (3 - ((RM & 1) + ((RM >> 2) & 1))) * (3 - ((CM & 1) + ((CM >> 2) & 1)));
        // rowbase = tile row (oy+0), column 0; rows are g.W*PAIRS apart; dx is an immediate offset
        const uint2* p0 = rowbase + (ox - 1) * PAIRS;
        uint2 a[3][3][PAIRS];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (!((RM >> dy) & 1) && !((CM >> dx) & 1)) {
#pragma unroll
                    for (int j = 0; j < PAIRS; ++j) a[dy][dx][j] = p0[(size_t)dy * g.W * PAIRS + dx * PAIRS + j];
                }
        int acc0 = 0, acc1 = 0;      // two independent accumulate chains
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (!((RM >> dy) & 1) && !((CM >> dx) & 1)) {
#pragma unroll
                    for (int j = 0; j < PAIRS; ++j) {
                        acc0 = qnn_dot_bin_chain(a[dy][dx][j].x, wreg[(dy * 3 + dx) * CW + 2 * j], acc0);
                        acc1 = qnn_dot_bin_chain(a[dy][dx][j].y, wreg[(dy * 3 + dx) * CW + 2 * j + 1], acc1);
                    }
                }
        const int acc = acc0 + acc1;
        // K - 2*acc: both integers < 2^24, so the float FMA below is exact
        float v = fmaf((float)acc, -2.0f, (float)NVALID * kf_cin);
        v = __fadd_rn(v, bias);
        // (compile time: as a run-time flag the BN multiply and add were executed for every value and then deselected)
        if constexpr (HAS_BN) v = __fadd_rn(__fmul_rn(v, inv), shift);
        if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
        else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
        // written once, read by a later kernel at the earliest: the non-temporal hint keeps the
        // 268 MB of output from displacing the input rows the neighbouring strips re-read from L2
        // (measured 56 % -> 62 % of the HBM roofline on the CIFAR B0 layer)
        __builtin_nontemporal_store(v, &yrow[(size_t)ox * g.cout]);
    };
    auto walk_row = [&](auto rm_c, int oy) {
        using std::integral_constant;
        const uint2* rowbase = tile + (size_t)oy * g.W * PAIRS;
        float* yrow = y + (((size_t)n * g.H + r0 + oy) * g.W) * g.cout + c;
        if (g.W == 1) { pixel(rm_c, integral_constant<int, 5>{}, rowbase, yrow, 0); return; }
        pixel(rm_c, integral_constant<int, 1>{}, rowbase, yrow, 0);
        for (int ox = 1; ox < g.W - 1; ++ox) pixel(rm_c, integral_constant<int, 0>{}, rowbase, yrow, ox);
        pixel(rm_c, integral_constant<int, 4>{}, rowbase, yrow, g.W - 1);
    };
    for (int oy = wave; oy < rows_out; oy += 4) {
        using std::integral_constant;
        const int gy = r0 + oy;
        const bool top = gy == 0, bot = gy == g.H - 1;
        if (top && bot) walk_row(integral_constant<int, 5>{}, oy);
        else if (top) walk_row(integral_constant<int, 1>{}, oy);
        else if (bot) walk_row(integral_constant<int, 4>{}, oy);
        else walk_row(integral_constant<int, 0>{}, oy);
    }
}

template <int CW>   // packed words per pixel (cin / 32), even
__global__ __launch_bounds__(kBlock) void k_conv_xnor_f32(ConvGeom g, EpiArgs e, int in_fn, int TR,
                                                          int strips, const float* __restrict__ x,
                                                          const uint32_t* __restrict__ wp,
                                                          float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    uint2* tile = reinterpret_cast<uint2*>(smem_x);            // [(TR+2) rows][W][CW/2] uint2
    constexpr int PAIRS = CW / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x / strips;
    const int r0 = (blockIdx.x - n * strips) * TR;               // first output row of the strip
    const int rows_out = min(TR, g.H - r0);
    const int cbase = blockIdx.y * 64;
    const float thr = (in_fn == QNN_FN_GRID) ? 0.0f : 0x1p-24f;  // binary_tanh(x) = +1 iff x > 2^-24

    // ---- phase 1: binarize rows r0-1 .. r0+rows_out into LDS ----
    const int row_lo = max(r0 - 1, 0), row_hi = min(r0 + rows_out, g.H - 1);   // inclusive, in-image
    const int groups = (row_hi - row_lo + 1) * g.W * PAIRS;     // 64-channel groups to convert
    const float* xin = x + ((size_t)n * g.H + row_lo) * g.W * g.cin;
    uint2* tdst = tile + (size_t)(row_lo - (r0 - 1)) * g.W * PAIRS;
    constexpr int U = QNN_XNOR_U;      // loads in flight per lane: phase 1 is latency-bound otherwise
    for (int gi = wave * U; gi < groups; gi += 4 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (gi + u < groups) ? xin[(size_t)(gi + u) * 64 + lane] : -1.0f;
        // the wave's 64-bit compare mask IS the packed pixel group: lane u collects the mask of group gi + u
        // (v_writelane, no exec games), then U lanes store U groups with one LDS instruction
        uint32_t mlo = 0, mhi = 0;
        qnn_collect_masks<0, U>(v, thr, mlo, mhi);
        if (lane < U && gi + lane < groups) tdst[gi + lane] = make_uint2(mlo, mhi);
    }
    // ---- this lane's filter ----
    uint32_t wreg[9 * CW];
    const uint32_t* wsrc = wp + (size_t)(cbase + lane) * (9 * CW);
#pragma unroll
    for (int k = 0; k < 9 * CW; ++k) wreg[k] = wsrc[k];
    const int c = cbase + lane;
    const float bias = e.bias ? e.bias[c] : 0.0f;
    const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
    const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
    __syncthreads();
    if (e.bn_inv) xnor_f32_rows<CW, true>(g, e, tile, wreg, bias, inv, shift, n, r0, rows_out, wave, c, y);
    else xnor_f32_rows<CW, false>(g, e, tile, wreg, bias, inv, shift, n, r0, rows_out, wave, c, y);
}

// (Measured dead end, round 2: the same layer as a PERSISTENT double-buffered pipeline -- workgroups walking strips, the
// next strip's float32 rows requested into registers before the current strip is computed, filters fetched once per
// workgroup -- ran 129 / 30.5 us on the CIFAR B0 / C0 layers against 100 / 27.5 us for this kernel: 78-151 VGPRs cut the
// occupancy to 3-6 waves per SIMD, and many small independent workgroups overlap their load and compute phases better.)

// ---------------------------------------------------------------------------
// Fused-pipeline 1-bit conv: packed bits in, packed bits out (BN + binary_tanh + optional
// 2x2 max-pool fused).  Same structure as k_conv_xnor_f32 (lane = output channel, filter in
// VGPRs, activations by LDS broadcast reads), but phase 1 is a plain copy of the packed rows
// and the output of a pixel is the wave's 64-bit lane mask of (BN(v) > 2^-24): the packed
// word pair itself.  Max-pooling: sign(max_i t_i) = OR_i sign(t_i), i.e. a scalar OR of the
// four masks.  Lane j parks the mask of stored pixel j; they are stored 64 pixels at a time.
// ---------------------------------------------------------------------------
template <int CW, int POOL>
__global__ __launch_bounds__(kBlock) void k_conv_xnor_pk(ConvGeom g, EpiArgs e, int TRP, int strips,
                                                         const uint32_t* __restrict__ x,
                                                         const uint32_t* __restrict__ wp,
                                                         uint32_t* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem_x[];
    uint2* tile = reinterpret_cast<uint2*>(smem_x);            // [(TR+2) rows][W][CW/2] uint2
    constexpr int PAIRS = CW / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x / strips;
    const int pr0 = (blockIdx.x - n * strips) * TRP;             // first stored (pooled) row of the strip
    const int prows = min(TRP, g.Hp - pr0);                     // stored rows in this strip
    const int r0 = pr0 * POOL;                                   // first conv row
    const int rows_out = prows * POOL;                           // conv rows computed
    const int cbase = blockIdx.y * 64;

    // ---- phase 1: copy packed rows r0-1 .. r0+rows_out (in-image ones) into LDS ----
    const int row_lo = max(r0 - 1, 0), row_hi = min(r0 + rows_out, g.H - 1);
    const int nent = (row_hi - row_lo + 1) * g.W * PAIRS;       // uint2 entries
    const uint2* xin = reinterpret_cast<const uint2*>(x) + ((size_t)n * g.H + row_lo) * g.W * PAIRS;
    uint2* tdst = tile + (size_t)(row_lo - (r0 - 1)) * g.W * PAIRS;
    for (int i = threadIdx.x; i < nent; i += kBlock) tdst[i] = xin[i];

    // filters through LDS (coalesced copy, odd row pitch), as in k_conv_xnor_f32
    constexpr int WROW = 9 * CW + 1;
    uint32_t* wfil = reinterpret_cast<uint32_t*>(tile + (size_t)(TRP * POOL + 2) * g.W * PAIRS);
    for (int i = threadIdx.x; i < 64 * 9 * CW; i += kBlock) {
        const int fc = i / (9 * CW);
        wfil[fc * WROW + (i - fc * (9 * CW))] = wp[(size_t)cbase * (9 * CW) + i];
    }
    const int c = cbase + lane;
    const float bias = e.bias ? e.bias[c] : 0.0f;
    const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
    const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
    const bool has_bn = e.bn_inv != nullptr;
    const float kf_cin = (float)g.cin;
    __syncthreads();
    uint32_t wreg[9 * CW];
#pragma unroll
    for (int k = 0; k < 9 * CW; ++k) wreg[k] = wfil[lane * WROW + k];

    // mask of one conv pixel (tile row oy, column ox) for border class (RM, CM)
    auto pixel_mask = [&](auto rm_c, auto cm_c, int oy, int ox) -> unsigned long long {
        constexpr int RM = decltype(rm_c)::value;
        constexpr int CM = decltype(cm_c)::value;
        constexpr int NVALID = (3 - ((RM & 1) + ((RM >> 2) & 1))) * (3 - ((CM & 1) + ((CM >> 2) & 1)));
        const uint2* p0 = tile + (size_t)oy * g.W * PAIRS + (ox - 1) * PAIRS;
        uint2 a[3][3][PAIRS];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (!((RM >> dy) & 1) && !((CM >> dx) & 1)) {
#pragma unroll
                    for (int j = 0; j < PAIRS; ++j) a[dy][dx][j] = p0[(size_t)dy * g.W * PAIRS + dx * PAIRS + j];
                }
        int acc0 = 0, acc1 = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (!((RM >> dy) & 1) && !((CM >> dx) & 1)) {
#pragma unroll
                    for (int j = 0; j < PAIRS; ++j) {
                        acc0 = qnn_dot_bin_chain(a[dy][dx][j].x, wreg[(dy * 3 + dx) * CW + 2 * j], acc0);
                        acc1 = qnn_dot_bin_chain(a[dy][dx][j].y, wreg[(dy * 3 + dx) * CW + 2 * j + 1], acc1);
                    }
                }
        float v = fmaf((float)(acc0 + acc1), -2.0f, (float)NVALID * kf_cin);   // exact: integers < 2^24
        v = __fadd_rn(v, bias);
        if (has_bn) v = __fadd_rn(__fmul_rn(v, inv), shift);
        return __ballot(v > 0x1p-24f);                        // binary_tanh(v) = +1 iff v > 2^-24
    };
    // one stored pixel: POOL x POOL conv pixels, OR of their masks.  Row classes of the (up
    // to two) conv rows are RMA, RMB; the column class depends on the position in the row.
    using std::integral_constant;
    auto stored_pixel = [&](auto rma_c, auto rmb_c, auto cma_c, auto cmb_c, int oy, int ox) {
        unsigned long long m = pixel_mask(rma_c, cma_c, oy, ox);
        if constexpr (POOL == 2) {
            m |= pixel_mask(rma_c, cmb_c, oy, ox + 1);
            m |= pixel_mask(rmb_c, cma_c, oy + 1, ox);
            m |= pixel_mask(rmb_c, cmb_c, oy + 1, ox + 1);
        }
        return m;
    };
    auto walk = [&](auto rma_c, auto rmb_c, int prow) {
        const int oy = prow * POOL;                              // tile row of the first conv row
        uint32_t keep_lo = 0, keep_hi = 0;                       // lane j parks the mask of stored pixel j
        uint32_t* yrow = y + (((size_t)n * g.Hp + pr0 + prow) * g.Wp) * e.ocw + (cbase >> 5);
        int pend0 = 0;
        auto flush = [&](int upto) {                             // store parked masks of pixels pend0..upto-1
            const int j = pend0 + lane;
            if (j < upto) {
                uint32_t* dst = yrow + (size_t)j * e.ocw;
                dst[0] = keep_lo;
                dst[1] = keep_hi;
            }
            pend0 = upto;
        };
        for (int pxs = 0; pxs < g.Wp; ++pxs) {
            const int ox = pxs * POOL;
            unsigned long long m;
            const bool first = ox == 0, last_a = ox == g.W - 1, last_b = (ox + 1) == g.W - 1;
            if constexpr (POOL == 2) {
                // conv columns ox (class A) and ox+1 (class B)
                if (first && last_b) m = stored_pixel(rma_c, rmb_c, integral_constant<int, 1>{}, integral_constant<int, 4>{}, oy, ox);
                else if (first) m = stored_pixel(rma_c, rmb_c, integral_constant<int, 1>{}, integral_constant<int, 0>{}, oy, ox);
                else if (last_b) m = stored_pixel(rma_c, rmb_c, integral_constant<int, 0>{}, integral_constant<int, 4>{}, oy, ox);
                else m = stored_pixel(rma_c, rmb_c, integral_constant<int, 0>{}, integral_constant<int, 0>{}, oy, ox);
            } else {
                if (first && last_a) m = stored_pixel(rma_c, rmb_c, integral_constant<int, 5>{}, integral_constant<int, 5>{}, oy, ox);
                else if (first) m = stored_pixel(rma_c, rmb_c, integral_constant<int, 1>{}, integral_constant<int, 1>{}, oy, ox);
                else if (last_a) m = stored_pixel(rma_c, rmb_c, integral_constant<int, 4>{}, integral_constant<int, 4>{}, oy, ox);
                else m = stored_pixel(rma_c, rmb_c, integral_constant<int, 0>{}, integral_constant<int, 0>{}, oy, ox);
            }
            const int slot = pxs - pend0;                        // 0..63, wave-uniform
            if (lane == slot) { keep_lo = (uint32_t)m; keep_hi = (uint32_t)(m >> 32); }
            if (slot == 63) flush(pxs + 1);
        }
        flush(g.Wp);
    };
    for (int prow = wave; prow < prows; prow += 4) {
        const int gy = r0 + prow * POOL;                         // image row of conv row A
        const bool top_a = gy == 0, bot_a = gy == g.H - 1;
        if constexpr (POOL == 2) {
            const bool bot_b = (gy + 1) == g.H - 1;              // row B is never the top row
            if (top_a && bot_b) walk(integral_constant<int, 1>{}, integral_constant<int, 4>{}, prow);
            else if (top_a) walk(integral_constant<int, 1>{}, integral_constant<int, 0>{}, prow);
            else if (bot_b) walk(integral_constant<int, 0>{}, integral_constant<int, 4>{}, prow);
            else walk(integral_constant<int, 0>{}, integral_constant<int, 0>{}, prow);
        } else {
            if (top_a && bot_a) walk(integral_constant<int, 5>{}, integral_constant<int, 5>{}, prow);
            else if (top_a) walk(integral_constant<int, 1>{}, integral_constant<int, 1>{}, prow);
            else if (bot_a) walk(integral_constant<int, 4>{}, integral_constant<int, 4>{}, prow);
            else walk(integral_constant<int, 0>{}, integral_constant<int, 0>{}, prow);
        }
    }
}

// returns 0 if launched
int try_launch_xnor_pk(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w,
                       void* y, hipStream_t s, char* name, size_t name_len) {
    if (w->store != QNN_STORE_BIN || !w->d_packed) return 1;
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || !w->same_pad) return 1;
    if (g.cin % 64 != 0 || g.cout % 64 != 0) return 1;
    if (e.out_store != QNN_STORE_BIN || e.fn != QNN_FN_BINARY_TANH) return 1;
    if (g.W < 2 || (g.pool == 2 && (g.W < 2 || g.H < 2))) return 1;
    const int cw = g.cin / 32;
    if (cw != 2 && cw != 4 && cw != 8) return 1;
    // strip = TRP stored rows; LDS tile = (TRP*pool + 2) conv rows
    int TRP = g.Hp;
    while ((size_t)(TRP * g.pool + 2) * g.W * cw * 4 > 32768 && TRP > 1) TRP = (TRP + 1) / 2;
    const int strips = (g.Hp + TRP - 1) / TRP;
    const size_t lds = (size_t)(TRP * g.pool + 2) * g.W * cw * 4 + (size_t)64 * (9 * cw + 1) * 4;   // rows + the block's filters
    const dim3 grid((unsigned)(g.N * strips), (unsigned)(g.cout / 64)), block(kBlock);
    snprintf(name, name_len, "xnor_pk_cw%d", cw);
#define XPK_CASE(CW_)                                                                              \
    if (cw == CW_) {                                                                               \
        if (g.pool == 2)                                                                           \
            hipLaunchKernelGGL((k_conv_xnor_pk<CW_, 2>), grid, block, lds, s, g, e, TRP, strips,   \
                               (const uint32_t*)x, w->d_packed, (uint32_t*)y);                     \
        else                                                                                       \
            hipLaunchKernelGGL((k_conv_xnor_pk<CW_, 1>), grid, block, lds, s, g, e, TRP, strips,   \
                               (const uint32_t*)x, w->d_packed, (uint32_t*)y);                     \
        return 0;                                                                                  \
    }
    XPK_CASE(2) XPK_CASE(4) XPK_CASE(8)
#undef XPK_CASE
    return 1;
}

// returns 0 if launched
int try_launch_xnor_f32(const ConvGeom& g, const EpiArgs& e, int in_fn, const float* x,
                        const qnn_weights* w, void* y, hipStream_t s, char* name, size_t name_len) {
    if (w->store != QNN_STORE_BIN || !w->d_packed) return 1;
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || !w->same_pad || g.pool != 1) return 1;
    if (g.cin % 64 != 0 || g.cout % 64 != 0 || e.out_store != QNN_STORE_F32) return 1;
    if (in_fn != QNN_FN_BINARY_TANH && in_fn != QNN_FN_GRID) return 1;
    const int cw = g.cin / 32;
    if (cw != 2 && cw != 4 && cw != 8) return 1;
    // strip height: whole image if it fits in ~32 KB of LDS, else as many rows as fit
    int TR = g.H;
    while ((size_t)(TR + 2) * g.W * cw * 4 > 32768 && TR > 1) TR = (TR + 1) / 2;
    static const int tr_env = QNN_ENV_INT("QNN_XNOR_TR", 0);
    if (tr_env > 0 && tr_env < TR) TR = tr_env;
    else if (tr_env == 0 && TR > 8 && (TR % 8) == 0) TR = 8;    // measured: more, smaller items overlap load and compute better
    const int strips = (g.H + TR - 1) / TR;
    const size_t lds = (size_t)(TR + 2) * g.W * cw * 4;
    snprintf(name, name_len, "xnor_f32_cw%d", cw);
    const dim3 grid((unsigned)(g.N * strips), (unsigned)(g.cout / 64)), block(kBlock);
    if (cw == 2) hipLaunchKernelGGL(k_conv_xnor_f32<2>, grid, block, lds, s, g, e, in_fn, TR, strips, x, w->d_packed, (float*)y);
    else if (cw == 4) hipLaunchKernelGGL(k_conv_xnor_f32<4>, grid, block, lds, s, g, e, in_fn, TR, strips, x, w->d_packed, (float*)y);
    else hipLaunchKernelGGL(k_conv_xnor_f32<8>, grid, block, lds, s, g, e, in_fn, TR, strips, x, w->d_packed, (float*)y);
    return 0;
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
template <int XS, int CW, int K>
int launch_ps_out(const ConvGeom& g, const EpiArgs& e, const void* x, const uint32_t* wts,
                  const int32_t* corr, void* y, hipStream_t s) {
    const long lanes = (long)g.N * g.Hp * g.Wp * (g.pool == 2 ? 4 : 1);
    const dim3 grid((unsigned)((lanes + kBlock - 1) / kBlock)), block(kBlock);
    const uint32_t* xu = (const uint32_t*)x;
    switch (e.out_store) {
        case QNN_STORE_F32:
            hipLaunchKernelGGL((k_conv_ps<XS, CW, K, QNN_STORE_F32>), grid, block, 0, s, g, e, xu, wts, corr, y);
            break;
        case QNN_STORE_BIN:
            hipLaunchKernelGGL((k_conv_ps<XS, CW, K, QNN_STORE_BIN>), grid, block, 0, s, g, e, xu, wts, corr, y);
            break;
        case QNN_STORE_I4:
            hipLaunchKernelGGL((k_conv_ps<XS, CW, K, QNN_STORE_I4>), grid, block, 0, s, g, e, xu, wts, corr, y);
            break;
        case QNN_STORE_I8:
            hipLaunchKernelGGL((k_conv_ps<XS, CW, K, QNN_STORE_I8>), grid, block, 0, s, g, e, xu, wts, corr, y);
            break;
        default:
            return 1;
    }
    return 0;
}

// returns 0 if a pixel-stationary instantiation was launched, 1 if none fits
int try_launch_ps(const ConvGeom& g, const EpiArgs& e, int x_store, const void* x,
                  const qnn_weights* w, void* y, hipStream_t s, char* name, size_t name_len) {
    if (g.kh != g.kw || (g.kh != 3 && g.kh != 1)) return 1;
    if (e.out_store == QNN_STORE_F32 && g.cout % 4 != 0) return 1;   // float4 stores
    if (g.pool != 1 && g.pool != 2) return 1;
    const uint32_t* wts = x_store == QNN_STORE_F32 ? (const uint32_t*)w->d_wq : w->d_packed;
    const int32_t* corr = (x_store == QNN_STORE_BIN && w->same_pad && g.kh == 3) ? w->d_corr : nullptr;
    const int cw = x_store == QNN_STORE_F32 ? g.cin : g.cw;
    const char* xs = x_store == QNN_STORE_F32 ? "f32" : x_store == QNN_STORE_BIN ? "bin"
                     : x_store == QNN_STORE_T2 ? "t2" : x_store == QNN_STORE_I4 ? "i4" : "i8";
    snprintf(name, name_len, "ps_%s_cw%d_k%d", xs, cw, g.kh);
#define PS_CASE(XS, CW, K)                                                     \
    if (x_store == XS && cw == CW && g.kh == K)                                \
        return launch_ps_out<XS, CW, K>(g, e, x, wts, corr, y, s);
    PS_CASE(QNN_STORE_F32, 1, 3)
    PS_CASE(QNN_STORE_F32, 3, 3)
    PS_CASE(QNN_STORE_BIN, 1, 3)
    PS_CASE(QNN_STORE_BIN, 2, 3)
    PS_CASE(QNN_STORE_BIN, 4, 3)
    PS_CASE(QNN_STORE_BIN, 8, 3)
    PS_CASE(QNN_STORE_BIN, 1, 1)
    PS_CASE(QNN_STORE_BIN, 2, 1)
    PS_CASE(QNN_STORE_T2, 2, 3)       // 2 words per 32 channels: Cin <= 32 / 64 / 128
    PS_CASE(QNN_STORE_T2, 4, 3)
    PS_CASE(QNN_STORE_T2, 8, 3)
    PS_CASE(QNN_STORE_T2, 2, 1)
    PS_CASE(QNN_STORE_T2, 4, 1)
    PS_CASE(QNN_STORE_I4, 2, 3)
    PS_CASE(QNN_STORE_I4, 4, 3)
    PS_CASE(QNN_STORE_I4, 8, 3)
    PS_CASE(QNN_STORE_I4, 16, 3)
    PS_CASE(QNN_STORE_I4, 2, 1)
    PS_CASE(QNN_STORE_I4, 4, 1)
    PS_CASE(QNN_STORE_I4, 8, 1)
    PS_CASE(QNN_STORE_I8, 4, 3)
    PS_CASE(QNN_STORE_I8, 8, 3)
    PS_CASE(QNN_STORE_I8, 16, 3)
    PS_CASE(QNN_STORE_I8, 4, 1)
    PS_CASE(QNN_STORE_I8, 8, 1)
    PS_CASE(QNN_STORE_I8, 16, 1)
#undef PS_CASE
    return 1;
}

int check_epilogue(const qnn_weights* w, const qnn_epilogue_t* epi, int xshift, EpiArgs* e, int x_store = -1, int first_mode = 0) {
    e->bias = w->d_bias;
    e->bn_inv = epi->bn_inv;
    e->bn_shift = epi->bn_shift;
    QNN_REQUIRE((epi->bn_inv == nullptr) == (epi->bn_shift == nullptr), QNN_EINVAL,
                "epilogue: bn_inv and bn_shift must both be set or both be NULL");
    e->scale = ldexpf(1.0f, -(w->wshift + xshift));
    e->fn = epi->fn;
    e->out_store = epi->out_store;
    e->act_m = 1.0f;
    QNN_REQUIRE(epi->fn == QNN_FN_NONE || epi->fn == QNN_FN_BINARY_TANH ||
                    epi->fn == QNN_FN_QUANTIZED_TANH,
                QNN_EINVAL, "epilogue: fn=%d cannot be fused", epi->fn);
    if (epi->fn == QNN_FN_QUANTIZED_TANH) {
        QNN_REQUIRE(epi->act_bits >= 2 && epi->act_bits <= 24, QNN_EINVAL,
                    "epilogue: act_bits=%d", epi->act_bits);
        e->act_m = (float)(1u << (epi->act_bits - 1));
    }
    e->res = epi->res;
    e->res_store = epi->res_store;
    e->res_cw = 0;
    e->res_scale = 1.0f;
    e->post_scale = (epi->res || epi->proj) ? epi->post_scale : 1.0f;
    e->trick_c = epi->trick_s != 0.0f ? epi->trick_c : 0.0f;
    e->trick_s = epi->trick_s;
    e->fold_a = nullptr;
    e->fold_b = nullptr;
    e->fold_c = nullptr;
    e->dom_flag = epi->domain_flag;
    e->flags = epi->flags;
    e->first_mode = 0;
    e->proj_x = nullptr; e->proj_w = nullptr; e->proj_bias = nullptr;
    e->proj_scale = 1.0f; e->proj_cin = 0; e->proj_H = 0; e->proj_W = 0;
    if (epi->proj) {
        // the shortcut as a 1x1 strides-2 convolution of the block input, computed inside the launch (qnn_projection_t)
        const qnn_projection_t* pj = epi->proj;
        QNN_REQUIRE(pj->w && pj->x, QNN_EINVAL, "epilogue: proj needs the 1x1 weights and the block input");
        QNN_REQUIRE(!epi->res && !epi->fold && epi->trick_s == 0.0f && epi->pool == 1, QNN_EINVAL,
                    "epilogue: proj excludes res, fold, the identity trick and pooling");
        const qnn_weights* pw = pj->w;
        QNN_REQUIRE(pw->kh == 1 && pw->kw == 1 && pw->stride == 2 && pw->store == QNN_STORE_I4 && pw->cout == w->cout,
                    QNN_EINVAL, "epilogue: proj weights must be a 1x1 strides-2 int4 kernel with the layer's cout");
        QNN_REQUIRE(pj->x_bits >= 1 && pj->x_bits <= 4 && pj->H > 0 && pj->W > 0, QNN_EINVAL,
                    "epilogue: proj x_bits=%d, H=%d, W=%d", pj->x_bits, pj->H, pj->W);
        QNN_REQUIRE(pw->d_mfma != nullptr, QNN_EUNSUPPORTED,
                    "epilogue: proj with %d -> %d channels is not supported (keep the two-launch form)", pw->cin, pw->cout);
        e->proj_x = (const uint8_t*)pj->x;
        e->proj_w = pw->d_mfma;
        e->proj_bias = pw->d_bias;
        e->proj_scale = ldexpf(1.0f, -(pw->wshift + pj->x_bits - 1));
        e->proj_cin = pw->cin; e->proj_H = pj->H; e->proj_W = pj->W;
    }
    if (epi->fold) {
        // the fold must have been prepared for exactly this layer and epilogue; a handle whose sweep found a differing
        // point on some channel (folded < cout) is accepted and ignored: the kernels evaluate the float32 chain
        const qnn_fold* f = epi->fold;
        const bool has_res = epi->res != nullptr;
        // (a fold of the image entry, mode 3, serves the QNN_STORE_U8 and the QNN_STORE_F32_IMAGE calls of its layer)
        const bool xs_ok = f->mode == 3 ? (x_store == QNN_STORE_U8 || (x_store == QNN_STORE_F32 && first_mode == 1))
                                        : (f->x_bits == xshift + 1 && x_store != QNN_STORE_U8 && x_store != QNN_STORE_F32);
        QNN_REQUIRE(f->w == w && f->bn_inv == epi->bn_inv && f->bn_shift == epi->bn_shift && f->fn == epi->fn &&
                        f->act_bits == epi->act_bits && f->out_store == epi->out_store && xs_ok &&
                        (f->has_res != 0) == has_res &&
                        (!has_res || (f->res_store == epi->res_store && f->res_bits == epi->res_bits &&
                                      f->post_scale == epi->post_scale)) &&
                        epi->trick_s == 0.0f,
                    QNN_EINVAL, "epilogue: the fold handle was prepared for another layer / epilogue (qnn_fold_prepare)");
        if (f->folded == f->cout) { e->fold_a = f->d_a; e->fold_b = f->d_b; e->fold_c = f->mode >= 2 ? f->d_c : nullptr; }
    }
    QNN_REQUIRE(epi->trick_s == 0.0f || (epi->trick_s > 0.0f && epi->trick_s < 1.0e6f), QNN_EINVAL,
                "epilogue: trick_s=%g (the layer's kernel_lr_multiplier, or 0)", (double)epi->trick_s);
    if (epi->res) {
        QNN_REQUIRE(epi->pool == 1, QNN_EINVAL, "epilogue: a residual input cannot be combined with pooling");
        switch (epi->res_store) {
            case QNN_STORE_F32: e->res_cw = w->cout; break;
            case QNN_STORE_BIN: e->res_cw = qnn_words(QNN_STORE_BIN, w->cout); break;
            case QNN_STORE_I4:
            case QNN_STORE_I8:
                QNN_REQUIRE(epi->res_bits >= 1 && epi->res_bits <= epi->res_store, QNN_EINVAL,
                            "epilogue: res_bits=%d does not fit %d-bit storage", epi->res_bits, epi->res_store);
                e->res_cw = qnn_words(epi->res_store, w->cout);
                e->res_scale = ldexpf(1.0f, -(epi->res_bits - 1));
                break;
            default:
                qnn_set_error("epilogue: res_store=%d", epi->res_store);
                return QNN_EINVAL;
        }
    }
    switch (epi->out_store) {
        case QNN_STORE_F32:
            e->ocw = w->cout;
            break;
        case QNN_STORE_BIN:
            QNN_REQUIRE(epi->fn == QNN_FN_BINARY_TANH, QNN_EINVAL,
                        "epilogue: BIN output needs fn=binary_tanh");
            e->ocw = qnn_words(QNN_STORE_BIN, w->cout);
            break;
        case QNN_STORE_I4:
        case QNN_STORE_I8:
            QNN_REQUIRE(epi->fn == QNN_FN_BINARY_TANH ||
                            (epi->fn == QNN_FN_QUANTIZED_TANH && epi->act_bits <= epi->out_store),
                        QNN_EINVAL, "epilogue: fn=%d act_bits=%d does not fit %d-bit output",
                        epi->fn, epi->act_bits, epi->out_store);
            e->ocw = qnn_words(epi->out_store, w->cout);
            break;
        default:
            qnn_set_error("epilogue: out_store=%d", epi->out_store);
            return QNN_EINVAL;
    }
    return QNN_OK;
}

int conv_forward(const qnn_weights* w, const void* x, int x_store, int x_bits, int N, int H,
                 int W, const qnn_epilogue_t* epi, void* y, void* stream, bool dense) {
    QNN_REQUIRE(w && x && y && epi, QNN_EINVAL, "conv_forward: null pointer");
    QNN_REQUIRE(N >= 0 && H > 0 && W > 0, QNN_EINVAL, "conv_forward: N=%d H=%d W=%d", N, H, W);
    // float32 input with a declared domain (the typed entry of this call): handed to the first-layer dispatch in the
    // epilogue arguments
    int first_mode = 0;
    if (x_store == QNN_STORE_F32_IMAGE || x_store == QNN_STORE_F32_UNIT) {
        first_mode = x_store == QNN_STORE_F32_IMAGE ? 1 : 2;
        x_store = QNN_STORE_F32;
    }
    int xshift = 0;
    if (x_store == QNN_STORE_F32) {
        // any float32 values; uses the float32 copy of the quantized kernel
    } else if (x_store == QNN_STORE_U8) {
        // image bytes, value = code / 255: exact integer sum against the weight CODES (float32 copy * 2^wshift)
        QNN_REQUIRE(!dense, QNN_EUNSUPPORTED, "dense_forward: no QNN_STORE_U8 input");
        QNN_REQUIRE(w->wkind == QNN_W_BINARY || w->wkind == QNN_W_TERNARY || (w->wkind == QNN_W_QUANT && w->wbits <= 8),
                    QNN_EUNSUPPORTED, "conv_forward: QNN_STORE_U8 input needs low-bit weights of <= 8 bits (wkind=%d wbits=%d)",
                    w->wkind, w->wbits);
        QNN_REQUIRE(w->H == 1.0f || w->wkind == QNN_W_QUANT, QNN_EUNSUPPORTED, "conv_forward: QNN_STORE_U8 input needs H = 1");
        QNN_REQUIRE(!epi->res, QNN_EUNSUPPORTED, "conv_forward: no residual input behind a QNN_STORE_U8 layer");
        QNN_REQUIRE(epi->trick_s == 0.0f, QNN_EUNSUPPORTED, "conv_forward: no faithful trick on the QNN_STORE_U8 entry");
        QNN_REQUIRE((double)w->kh * w->kw * w->cin * 255.0 * (double)(1 << w->wshift) < 16777216.0, QNN_EUNSUPPORTED,
                    "conv_forward: QNN_STORE_U8 sums of this layer exceed 2^24");
    } else {
        QNN_REQUIRE(x_store == w->store, QNN_EINVAL,
                    "conv_forward: x_store=%d but the weights were prepacked for store=%d",
                    x_store, w->store);
        if (x_store != QNN_STORE_BIN && x_store != QNN_STORE_T2) {      // BIN / T2: value = code
            QNN_REQUIRE(x_bits >= 1 && x_bits <= x_store, QNN_EINVAL,
                        "conv_forward: x_bits=%d does not fit %d-bit storage", x_bits, x_store);
            xshift = x_bits - 1;
        }
    }
    ConvGeom g;
    g.N = N; g.H = H; g.W = W;
    g.cin = w->cin; g.cout = w->cout; g.kh = w->kh; g.kw = w->kw; g.stride = w->stride;
    qnn_same_pad(H, w->kh, w->stride, w->same_pad, &g.Ho, &g.pt);
    qnn_same_pad(W, w->kw, w->stride, w->same_pad, &g.Wo, &g.pl);
    QNN_REQUIRE(g.Ho > 0 && g.Wo > 0, QNN_EINVAL, "conv_forward: empty output");
    g.cw = w->cw; g.kwords = w->kwords;
    g.pool = epi->pool;
    QNN_REQUIRE(g.pool == 1 || g.pool == 2, QNN_EINVAL, "conv_forward: pool=%d", g.pool);
    QNN_REQUIRE(!dense || g.pool == 1, QNN_EINVAL, "dense_forward: pool must be 1");
    g.Hp = g.Ho / g.pool; g.Wp = g.Wo / g.pool;   // MaxPooling2D 'valid' drops the remainder
    QNN_REQUIRE(g.Hp > 0 && g.Wp > 0, QNN_EINVAL, "conv_forward: pooled output is empty");
    QNN_REQUIRE((double)g.N * g.Ho * g.Wo < 2.0e9 && (double)g.N * g.H * g.W * (g.cw > g.cin ? g.cw : g.cin) < 9.0e18,
                QNN_EUNSUPPORTED, "conv_forward: more than 2^31 output pixels in one call");
    g.fd_wp = qnn_fastdiv((uint32_t)g.Wp);
    g.fd_hp = qnn_fastdiv((uint32_t)g.Hp);
    EpiArgs e;
    int rc = check_epilogue(w, epi, xshift, &e, x_store, first_mode);
    if (rc != QNN_OK) return rc;
    e.first_mode = first_mode;
    if (x_store == QNN_STORE_F32) e.scale = 1.0f;   // d_wq holds real values already
    if (x_store == QNN_STORE_U8) e.scale = 255.0f * (float)(1 << w->wshift);   // the divisor D of the affine map
    // a restricted-domain kernel (first_fixed) saw a value outside its domain in an earlier launch of this layer, and
    // the flag has reached the host: report it now (qnn_weights_check is the synchronising form)
    if (!epi->domain_flag && w->h_flag && *(volatile uint32_t*)w->h_flag) {
        *(volatile uint32_t*)w->h_flag = 0;
        qnn_set_error("conv_forward: an earlier launch of this layer's restricted-domain kernel met inputs outside its "
                      "domain (first_fixed: [0, 1]; first_image: image bytes / 255); its outputs are unspecified.  Use the "
                      "exact first layer or QNN_STORE_U8 input");
        return QNN_EINVAL;
    }
    if (N == 0) return QNN_OK;

    hipStream_t s = (hipStream_t)stream;
    char name[64];
    // the pixel-stationary kernel needs whole pool windows (even Ho/Wo are not
    // required: the remainder row/column is simply never produced)
    // "faithful" output-side trick: only the kernels whose epilogue is qnn_epi_value implement it
    const bool trick = e.trick_s != 0.0f;
    QNN_REQUIRE(!(trick && dense), QNN_EINVAL, "dense_forward: the reference's Dense layers have no identity trick");
    const int pref = trick ? 1 : qnn_conv_impl_pref();
    bool launched = false;
    if (e.proj_x) {
        // the in-launch projection shortcut: the row-walking strip kernel or nothing (the caller keeps two launches)
        const bool ok = !dense && qnn_try_launch_mfma(g, e, x_store, x, w, y, s, name, sizeof(name)) == 0;
        QNN_REQUIRE(ok, QNN_EUNSUPPORTED, "conv_forward: no kernel computes a projection shortcut for this layer "
                    "(3x3 stride-1 int4, cin = cout in {32, 64}, proj cin = cin / 2, output ceil(H/2) x ceil(W/2) of the block input)");
        qnn_set_kernel_name(name);
        QNN_HIP(hipGetLastError());
        return QNN_OK;
    }
    if (dense && !e.res && x_store != QNN_STORE_F32 && e.out_store == QNN_STORE_F32 && (w->kwords % 4) == 0) {
        int rc2 = x_store == QNN_STORE_BIN  ? launch_dense<QNN_STORE_BIN>(x, w, e, y, N, s)
                  : x_store == QNN_STORE_T2 ? launch_dense<QNN_STORE_T2>(x, w, e, y, N, s)
                  : x_store == QNN_STORE_I4 ? launch_dense<QNN_STORE_I4>(x, w, e, y, N, s)
                                            : launch_dense<QNN_STORE_I8>(x, w, e, y, N, s);
        if (rc2 == 0) {
            launched = true;
            snprintf(name, sizeof(name), "dense_%s", x_store == QNN_STORE_BIN ? "bin" : x_store == QNN_STORE_T2 ? "t2"
                                                     : x_store == QNN_STORE_I4 ? "i4" : "i8");
        }
    }
    if (!launched && dense && !e.res && x_store == QNN_STORE_F32 && e.out_store == QNN_STORE_F32 && w->d_wq &&
        (long)N * g.cout < 2000000000L) {
        const unsigned blocks = (unsigned)((long)N * g.cout);
        hipLaunchKernelGGL(k_dense_f32in, dim3(blocks), dim3(64), 0, s, e, N, g.cin, g.cout,
                           (const float*)x, w->d_wq, (float*)y);
        launched = true;
        snprintf(name, sizeof(name), "dense_f32");
    }
    if (!launched && !dense && !trick && x_store == QNN_STORE_I4 && try_launch_pw_f32(g, e, x, w, y, s) == 0) {
        launched = true;
        snprintf(name, sizeof(name), "pw_i4_f32");
    }
    if (x_store == QNN_STORE_U8) {
        if (pref != 1 && qnn_try_launch_first_u8(g, e, x, w, y, s, false) == 0) {
            qnn_set_kernel_name("mfma_i8_first_u8");
        } else {
            const size_t total = (size_t)g.N * g.Hp * g.Wp * e.ocw;
            size_t blocks = (total + kBlock - 1) / kBlock;
            if (blocks > 65535u * 16u) blocks = 65535u * 16u;
            hipLaunchKernelGGL(k_conv_generic, dim3((unsigned)blocks), dim3(kBlock), 0, s, g, e, x_store,
                               x, w->d_packed, w->d_wq, y);
            qnn_set_kernel_name("generic_u8");
        }
        QNN_HIP(hipGetLastError());
        return QNN_OK;
    }
    // opt-in: float32 images that are bytes / 255 on the byte kernels (qnn_first_u8.hip, F32IN; also the ResNet stem)
    if (!launched && pref != 1 && !dense && x_store == QNN_STORE_F32 && first_mode == 1 &&
        qnn_try_launch_first_u8(g, e, x, w, y, s, true) == 0) {
        launched = true;
        snprintf(name, sizeof(name), "mfma_i8_first_img255");
    }
    if (!launched && pref != 1 && !dense && x_store == QNN_STORE_F32 && qnn_try_launch_stem(g, e, x, w->d_wq, y, s) == 0) {
        launched = true;                           // float-input layer with few filters (ResNet stem)
        snprintf(name, sizeof(name), "mfma_f32_stem_cin%d", g.cin);
    }
    if (!launched && pref != 1 && !dense)          // residual epilogues: only where the MFMA kernel has one
        launched = qnn_try_launch_mfma(g, e, x_store, x, w, y, s, name, sizeof(name)) == 0;
    if (!launched && x_store == QNN_STORE_BIN && !dense && !e.res && !trick)
        launched = try_launch_xnor_pk(g, e, x, w, y, s, name, sizeof(name)) == 0;
    if (!launched) launched = try_launch_ps(g, e, x_store, x, w, y, s, name, sizeof(name)) == 0;
    if (launched) {
        qnn_set_kernel_name(name);
    } else {
        const size_t total = (size_t)g.N * g.Hp * g.Wp * e.ocw;
        size_t blocks = (total + kBlock - 1) / kBlock;
        if (blocks > 65535u * 16u) blocks = 65535u * 16u;
        hipLaunchKernelGGL(k_conv_generic, dim3((unsigned)blocks), dim3(kBlock), 0, s, g, e, x_store,
                           x, w->d_packed, w->d_wq, y);
        qnn_set_kernel_name("generic");
    }
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" int qnn_prepack_weights(int wkind, int wbits, float H, const float* kernel, int kh,
                                   int kw, int cin, int cout, const float* bias, int stride,
                                   int same_pad, int store, void* stream, qnn_weights_t** out) {
    QNN_REQUIRE(kernel && out, QNN_EINVAL, "qnn_prepack_weights: null pointer");
    QNN_REQUIRE(kh > 0 && kw > 0 && cin > 0 && cout > 0, QNN_EINVAL,
                "qnn_prepack_weights: kernel shape (%d,%d,%d,%d)", kh, kw, cin, cout);
    QNN_REQUIRE(kh <= 3 && kw <= 3, QNN_EUNSUPPORTED,
                "qnn_prepack_weights: kernel %dx%d larger than 3x3 is not supported", kh, kw);
    QNN_REQUIRE(stride >= 1, QNN_EINVAL, "qnn_prepack_weights: stride=%d", stride);
    QNN_REQUIRE(wkind == QNN_W_FLOAT || wkind == QNN_W_BINARY || wkind == QNN_W_QUANT ||
                    wkind == QNN_W_TERNARY,
                QNN_EINVAL, "qnn_prepack_weights: wkind=%d not supported", wkind);
    QNN_REQUIRE(H > 0.0f, QNN_EINVAL, "qnn_prepack_weights: H=%g", (double)H);
    int wshift = 0;
    float m = 1.0f;
    if (wkind == QNN_W_QUANT) {
        QNN_REQUIRE(wbits >= 2 && wbits <= 24, QNN_EINVAL, "qnn_prepack_weights: wbits=%d", wbits);
        wshift = wbits - 1;
        m = (float)(1u << wshift);
    }
    switch (store) {
        case QNN_STORE_F32:
            break;
        case QNN_STORE_BIN:
            QNN_REQUIRE(wkind == QNN_W_BINARY && H == 1.0f, QNN_EINVAL,
                        "qnn_prepack_weights: BIN storage needs binary weights with H=1");
            break;
        case QNN_STORE_I4:
        case QNN_STORE_I8:
            QNN_REQUIRE(((wkind == QNN_W_BINARY || wkind == QNN_W_TERNARY) && H == 1.0f) ||
                            (wkind == QNN_W_QUANT && wbits <= store),
                        QNN_EINVAL, "qnn_prepack_weights: wkind=%d wbits=%d does not fit %d-bit storage",
                        wkind, wbits, store);
            break;
        case QNN_STORE_T2:
            QNN_REQUIRE((wkind == QNN_W_BINARY || wkind == QNN_W_TERNARY) && H == 1.0f, QNN_EINVAL,
                        "qnn_prepack_weights: sign / mask storage needs ternary (or binary) weights with H=1");
            break;
        default:
            qnn_set_error("qnn_prepack_weights: store=%d", store);
            return QNN_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    qnn_weights* w = new (std::nothrow) qnn_weights();
    QNN_REQUIRE(w, QNN_ENOMEM, "qnn_prepack_weights: host allocation failed");
    *w = qnn_weights{};
    w->wkind = wkind; w->wbits = wbits; w->H = H;
    w->kh = kh; w->kw = kw; w->cin = cin; w->cout = cout;
    w->stride = stride; w->same_pad = same_pad ? 1 : 0;
    w->store = store; w->wshift = wshift;
    w->cw = store == QNN_STORE_F32 ? cin : qnn_words(store, cin);
    w->kwords = kh * kw * w->cw;
    const int taps = kh * kw;
    const size_t nq = (size_t)cout * taps * cin;
#define PREPACK_HIP(expr)                                                      \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) {                                                \
            qnn_set_error("%s failed: %s", #expr, hipGetErrorString(_e));      \
            qnn_free_weights(w);                                               \
            return _e == hipErrorOutOfMemory ? QNN_ENOMEM : QNN_EHIP;          \
        }                                                                      \
    } while (0)
    PREPACK_HIP(hipMalloc(&w->d_wq, nq * sizeof(float)));
    if (store == QNN_STORE_F32 && kh == 3 && kw == 3 && cin == 3) {
        // domain flag of the restricted-domain first-layer kernel: one word of pinned host memory the device can write
        PREPACK_HIP(hipHostMalloc((void**)&w->h_flag, sizeof(uint32_t), hipHostMallocMapped));
        *w->h_flag = 0;
        PREPACK_HIP(hipHostGetDevicePointer((void**)&w->d_flag, w->h_flag, 0));
    }
    int grid = (int)((nq + kBlock - 1) / kBlock);
    if (grid > 4096) grid = 4096;
    if (wkind == QNN_W_TERNARY) {
        PREPACK_HIP(hipMalloc(&w->d_aux, 16));
        hipLaunchKernelGGL(k_tern_cutoff, dim3(1), dim3(1024), 0, s, kernel, (int)nq, H, (float*)w->d_aux);
    }
    hipLaunchKernelGGL(k_prepack_float, dim3(grid), dim3(kBlock), 0, s, kernel, w->d_wq, taps, cin,
                       cout, wkind, H, m, (const float*)w->d_aux);
    if (bias) {
        PREPACK_HIP(hipMalloc(&w->d_bias, cout * sizeof(float)));
        PREPACK_HIP(hipMemcpyAsync(w->d_bias, bias, cout * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (store != QNN_STORE_F32) {
        const size_t nw = (size_t)cout * w->kwords;
        PREPACK_HIP(hipMalloc(&w->d_packed, nw * sizeof(uint32_t)));
        grid = (int)((nw + kBlock - 1) / kBlock);
        if (grid > 4096) grid = 4096;
        if (store == QNN_STORE_T2)
            hipLaunchKernelGGL(k_prepack_t2, dim3(grid), dim3(kBlock), 0, s, w->d_wq, w->d_packed, taps, cin, cout,
                               w->cw / 2);
        else if (store == QNN_STORE_BIN)
            hipLaunchKernelGGL(k_prepack_codes<QNN_STORE_BIN>, dim3(grid), dim3(kBlock), 0, s, w->d_wq,
                               w->d_packed, taps, cin, cout, w->cw, m);
        else if (store == QNN_STORE_I4)
            hipLaunchKernelGGL(k_prepack_codes<QNN_STORE_I4>, dim3(grid), dim3(kBlock), 0, s, w->d_wq,
                               w->d_packed, taps, cin, cout, w->cw, m);
        else
            hipLaunchKernelGGL(k_prepack_codes<QNN_STORE_I8>, dim3(grid), dim3(kBlock), 0, s, w->d_wq,
                               w->d_packed, taps, cin, cout, w->cw, m);
        if (store == QNN_STORE_BIN && w->same_pad && (kh > 1 || kw > 1)) {
            PREPACK_HIP(hipMalloc(&w->d_corr, (size_t)64 * cout * sizeof(int32_t)));
            grid = (64 * cout + kBlock - 1) / kBlock;
            hipLaunchKernelGGL(k_corr_table, dim3(grid), dim3(kBlock), 0, s, w->d_wq, w->d_corr, kh,
                               kw, cin, cout);
        }
    }
    PREPACK_HIP(hipGetLastError());
    if (qnn_mfma_prepare_weights(w, s) != QNN_OK || qnn_head_prepare(w, s) != QNN_OK) {
        qnn_free_weights(w);
        return QNN_EHIP;
    }
#undef PREPACK_HIP
    *out = w;
    return QNN_OK;
}

extern "C" int qnn_free_weights(qnn_weights_t* w) {
    if (!w) return QNN_OK;
    if (w->d_packed) (void)hipFree(w->d_packed);
    if (w->d_wq) (void)hipFree(w->d_wq);
    if (w->d_bias) (void)hipFree(w->d_bias);
    if (w->d_corr) (void)hipFree(w->d_corr);
    if (w->d_mfma_own) (void)hipFree(w->d_mfma_own);
    if (w->d_aux) (void)hipFree(w->d_aux);
    if (w->h_flag) (void)hipHostFree(w->h_flag);
    if (w->d_head) (void)hipFree(w->d_head);
    delete w;
    return QNN_OK;
}

extern "C" int qnn_weights_check(const qnn_weights_t* w, void* stream) {
    QNN_REQUIRE(w, QNN_EINVAL, "qnn_weights_check: null weights");
    QNN_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (w->h_flag && *(volatile uint32_t*)w->h_flag) {
        *(volatile uint32_t*)w->h_flag = 0;
        qnn_set_error("qnn_weights_check: a launch of this layer's restricted-domain kernel met inputs outside its domain "
                      "(first_fixed: [0, 1]; first_image: image bytes / 255; NaN included); its outputs are unspecified.  "
                      "Use the exact first layer or QNN_STORE_U8 input");
        return QNN_EINVAL;
    }
    return QNN_OK;
}

extern "C" int qnn_weights_dequant(const qnn_weights_t* w, float* kernel_hwio, void* stream) {
    QNN_REQUIRE(w && kernel_hwio, QNN_EINVAL, "qnn_weights_dequant: null pointer");
    const int taps = w->kh * w->kw;
    const size_t nq = (size_t)w->cout * taps * w->cin;
    int grid = (int)((nq + kBlock - 1) / kBlock);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_dequant_hwio, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, w->d_wq,
                       kernel_hwio, taps, w->cin, w->cout);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_conv2d_forward(const qnn_weights_t* w, const void* x, int x_store, int x_bits,
                                  int N, int H, int W, const qnn_epilogue_t* epi, void* y,
                                  void* stream) {
    return conv_forward(w, x, x_store, x_bits, N, H, W, epi, y, stream, false);
}

// float32 NHWC input with the preceding activation clip fused on load.  `workspace`
// (qnn_conv2d_workspace_bytes) is only touched when no fused kernel fits and the clip +
// pack has to run as a separate pass.
extern "C" size_t qnn_conv2d_workspace_bytes(const qnn_weights_t* w, int N, int H, int W) {
    if (!w || w->store == QNN_STORE_F32) return 0;
    return (size_t)N * H * W * w->cw * 4;
}

extern "C" int qnn_conv2d_forward_f32in(const qnn_weights_t* w, const float* x, int in_fn, int in_bits,
                                        int N, int H, int W, const qnn_epilogue_t* epi, void* y,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    QNN_REQUIRE(w && x && y && epi, QNN_EINVAL, "qnn_conv2d_forward_f32in: null pointer");
    QNN_REQUIRE(w->store != QNN_STORE_F32, QNN_EINVAL,
                "qnn_conv2d_forward_f32in: weights were prepacked for float32 inputs only");
    QNN_REQUIRE(in_fn == QNN_FN_BINARY_TANH || in_fn == QNN_FN_QUANTIZED_TANH || in_fn == QNN_FN_GRID,
                QNN_EINVAL, "qnn_conv2d_forward_f32in: in_fn=%d", in_fn);
    const int x_bits = w->store == QNN_STORE_BIN ? 1 : in_bits;
    if (w->store == QNN_STORE_BIN && epi->out_store == QNN_STORE_F32 && epi->pool == 1 && N > 0 &&
        !epi->res && epi->trick_s == 0.0f && qnn_conv_impl_pref() != 2) {
        ConvGeom g;
        g.N = N; g.H = H; g.W = W;
        g.cin = w->cin; g.cout = w->cout; g.kh = w->kh; g.kw = w->kw; g.stride = w->stride;
        qnn_same_pad(H, w->kh, w->stride, w->same_pad, &g.Ho, &g.pt);
        qnn_same_pad(W, w->kw, w->stride, w->same_pad, &g.Wo, &g.pl);
        g.cw = w->cw; g.kwords = w->kwords; g.pool = 1; g.Hp = g.Ho; g.Wp = g.Wo;
        g.fd_wp = qnn_fastdiv((uint32_t)g.Wp); g.fd_hp = qnn_fastdiv((uint32_t)g.Hp);
        EpiArgs e;
        int rc = check_epilogue(w, epi, 0, &e);
        if (rc != QNN_OK) return rc;
        char name[64];
        if (try_launch_xnor_f32(g, e, in_fn, x, w, y, (hipStream_t)stream, name, sizeof(name)) == 0) {
            qnn_set_kernel_name(name);
            QNN_HIP(hipGetLastError());
            return QNN_OK;
        }
    }
    const size_t need = qnn_conv2d_workspace_bytes(w, N, H, W);
    QNN_REQUIRE(workspace && workspace_bytes >= need, QNN_EINVAL,
                "qnn_conv2d_forward_f32in: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    int rc = qnn_pack_f32(x, workspace, (size_t)N * H * W, w->cin, in_fn, x_bits, w->store, stream);
    if (rc != QNN_OK) return rc;
    return conv_forward(w, workspace, w->store, x_bits, N, H, W, epi, y, stream, false);
}

// The last conv group of a VGG and the classifier behind it in ONE launch (see the header).  Same result, bit for bit, as
// qnn_conv2d_forward (packed int4 output) followed by qnn_dense_forward on the flattened tensor.
extern "C" int qnn_conv2d_dense_forward(const qnn_weights_t* wc, const qnn_weights_t* wd, const void* x, int x_store,
                                        int x_bits, int N, int H, int W, const qnn_epilogue_t* epi_conv,
                                        const qnn_epilogue_t* epi_dense, float* y, void* stream) {
    QNN_REQUIRE(wc && wd && x && y && epi_conv && epi_dense, QNN_EINVAL, "qnn_conv2d_dense_forward: null pointer");
    QNN_REQUIRE(N >= 0 && H > 0 && W > 0, QNN_EINVAL, "qnn_conv2d_dense_forward: N=%d H=%d W=%d", N, H, W);
    QNN_REQUIRE(wd->kh == 1 && wd->kw == 1, QNN_EINVAL, "qnn_conv2d_dense_forward: the second handle is not a dense layer");
    if (x_store != QNN_STORE_I4 || wc->store != QNN_STORE_I4 || wd->store != QNN_STORE_I4 || !wc->d_mfma || !wd->d_head ||
        qnn_conv_impl_pref() == 1 || epi_conv->pool != 2 || epi_conv->out_store != QNN_STORE_I4 || epi_conv->res ||
        epi_conv->trick_s != 0.0f || epi_dense->out_store != QNN_STORE_F32 || epi_dense->fn != QNN_FN_NONE ||
        epi_dense->res || epi_dense->pool != 1 || epi_dense->trick_s != 0.0f) {
        qnn_set_error("qnn_conv2d_dense_forward: no fused kernel for this pair of layers");
        return QNN_EUNSUPPORTED;
    }
    QNN_REQUIRE(x_bits >= 1 && x_bits <= 4, QNN_EINVAL, "qnn_conv2d_dense_forward: x_bits=%d", x_bits);
    ConvGeom g;
    g.N = N; g.H = H; g.W = W;
    g.cin = wc->cin; g.cout = wc->cout; g.kh = wc->kh; g.kw = wc->kw; g.stride = wc->stride;
    qnn_same_pad(H, wc->kh, wc->stride, wc->same_pad, &g.Ho, &g.pt);
    qnn_same_pad(W, wc->kw, wc->stride, wc->same_pad, &g.Wo, &g.pl);
    g.cw = wc->cw; g.kwords = wc->kwords; g.pool = 2;
    g.Hp = g.Ho / 2; g.Wp = g.Wo / 2;
    if (g.Hp <= 0 || g.Wp <= 0 || g.pt != 1 || g.pl != 1 || (g.cin % 64) != 0 || g.cin > 128 || g.cout != 64 ||
        g.Hp * g.Wp * g.cout != wd->cin) {
        qnn_set_error("qnn_conv2d_dense_forward: no fused kernel for this geometry");
        return QNN_EUNSUPPORTED;
    }
    g.fd_wp = qnn_fastdiv((uint32_t)g.Wp);
    g.fd_hp = qnn_fastdiv((uint32_t)g.Hp);
    EpiArgs e, ed;
    int rc = check_epilogue(wc, epi_conv, x_bits - 1, &e);
    if (rc != QNN_OK) return rc;
    const int abits = epi_conv->fn == QNN_FN_QUANTIZED_TANH ? epi_conv->act_bits : 1;     // codes the dense layer sees
    rc = check_epilogue(wd, epi_dense, abits - 1, &ed);
    if (rc != QNN_OK) return rc;
    if (N == 0) return QNN_OK;
    MfmaGeom mg;
    mg.g = g; mg.kc = g.cin / 64; mg.steps = 9 * mg.kc; mg.x_pix_bytes = g.cin / 2;
    mg.total_q = (long)N * g.Hp * g.Wp;
    const double xb = (double)N * H * W * mg.x_pix_bytes, wb = (double)g.cout * 9 * g.cin;
    if (xb >= 2.0e9 || mg.total_q * 4 >= 2000000000L) {
        qnn_set_error("qnn_conv2d_dense_forward: tensor too large for one launch");
        return QNN_EUNSUPPORTED;
    }
    mg.x_bytes = (uint32_t)xb; mg.w_bytes = (uint32_t)wb; mg.ablate = 0;
    e.scale = e.scale * (1.0f / 256.0f);                    // both conv operands carry *16 (as qnn_try_launch_mfma)
    const char* kname = "";
    if (qnn_launch_areg_head(mg, e, x, wc->d_mfma, wd, ed, y, (hipStream_t)stream, &kname) != 0) {
        qnn_set_error("qnn_conv2d_dense_forward: no fused kernel for this geometry");
        return QNN_EUNSUPPORTED;
    }
    qnn_set_kernel_name(kname);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_dense_forward(const qnn_weights_t* w, const void* x, int x_store, int x_bits,
                                 int N, const qnn_epilogue_t* epi, void* y, void* stream) {
    QNN_REQUIRE(w, QNN_EINVAL, "qnn_dense_forward: null weights");
    QNN_REQUIRE(w->kh == 1 && w->kw == 1, QNN_EINVAL,
                "qnn_dense_forward: weights were prepacked as a %dx%d conv", w->kh, w->kw);
    return conv_forward(w, x, x_store, x_bits, N, 1, 1, epi, y, stream, true);
}
