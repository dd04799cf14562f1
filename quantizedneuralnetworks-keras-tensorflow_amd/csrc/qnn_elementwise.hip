// Elementwise activation clips and pack/unpack kernels (HBM-bound streaming).
//
// Reference ops replaced: layers/binary_ops.py:37-51 (binary_tanh),
// layers/quantized_ops.py:87-100 (quantized_tanh), layers/ternary_ops.py:15-54.
// Each reference op is ~7 separate TF elementwise kernels (7 HBM round trips);
// here it is one pass, 16 B per lane, one access per thread.
#include "qnn_common.h"

namespace {

constexpr int kBlock = 256;
// One item (16 B or one word) per thread, blocks in address order: measured 6.5 TB/s (81-82 % of
// peak) for the float32 clips, against 5.2 TB/s for a grid-stride loop over 2048 blocks, whose
// concurrently running blocks touch addresses megabytes apart.
constexpr int kMaxBlocks = 1 << 22;

inline int grid_for(size_t items) {
    size_t b = (items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// ---------------------------------------------------------------------------
#ifndef QNN_ACT_UNROLL
#define QNN_ACT_UNROLL 1
#endif

inline int act_grid(size_t items) {
    size_t b = (items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

template <int FN>
__device__ __forceinline__ float4 act4(float4 v, float m) {
    float4 r;
    if constexpr (FN == QNN_FN_BINARY_TANH) {
        r.x = qnn_binary_tanh(v.x); r.y = qnn_binary_tanh(v.y);
        r.z = qnn_binary_tanh(v.z); r.w = qnn_binary_tanh(v.w);
    } else {
        r.x = qnn_quantized_tanh(v.x, m); r.y = qnn_quantized_tanh(v.y, m);
        r.z = qnn_quantized_tanh(v.z, m); r.w = qnn_quantized_tanh(v.w, m);
    }
    return r;
}

// One pass, 16 B per lane and access, QNN_ACT_UNROLL independent loads in flight per lane (a
// single load per lane leaves the HBM pipe half empty at 8 blocks per CU); the data is touched
// once, so loads and stores carry the non-temporal hint.
template <int FN>
__global__ __launch_bounds__(kBlock) void k_act_f32(const float* __restrict__ x,
                                                    float* __restrict__ y, size_t n, float m) {
    constexpr int U = QNN_ACT_UNROLL;
    const size_t n4 = n / 4;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* x4 = reinterpret_cast<const v4f*>(x);
    v4f* y4 = reinterpret_cast<v4f*>(y);
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    auto f4 = [&](v4f v) {
        const float4 r = act4<FN>(make_float4(v.x, v.y, v.z, v.w), m);
        return (v4f){r.x, r.y, r.z, r.w};
    };
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&x4[i + u * stride]);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(f4(v[u]), &y4[i + u * stride]);
    }
    for (; i < n4; i += stride) y4[i] = f4(x4[i]);
    // tail (n % 4 elements)
    const size_t t = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) {
        if constexpr (FN == QNN_FN_BINARY_TANH) y[t] = qnn_binary_tanh(x[t]);
        else y[t] = qnn_quantized_tanh(x[t], m);
    }
}

// ternary_ops.py:52-54 + 15-30: pass 1 = sum |clip(x,-1,1)| in double
__global__ __launch_bounds__(kBlock) void k_tern_sum(const float* __restrict__ x, size_t n,
                                                     double* __restrict__ sum) {
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        s += (double)fabsf(fminf(fmaxf(x[i], -1.0f), 1.0f));
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[kBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < kBlock / 64; ++i) t += part[i];
        atomicAdd(sum, t);
        if (blockIdx.x == 0) atomicAdd(sum + 1, (double)n);      // the element count (the workspace was zeroed)
    }
}
// pass 2: W > cutoff -> 1, W <= -cutoff -> -1, else 0; then W + (Wt - W)
__global__ __launch_bounds__(kBlock) void k_tern_apply(const float* __restrict__ x,
                                                       float* __restrict__ y, size_t n,
                                                       const double* __restrict__ sum) {
    const float mean_abs = (float)(sum[0] / sum[1]);     // (sum |clip(x)|, element count): all-reduced when sharded
    const float cutoff = __fmul_rn(0.7f, mean_abs);
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float w = fminf(fmaxf(x[i], -1.0f), 1.0f);
        float wt = w > cutoff ? 1.0f : (w <= -cutoff ? -1.0f : 0.0f);
        y[i] = __fadd_rn(w, __fsub_rn(wt, w));
    }
}

// ---------------------------------------------------------------------------
// pack: one thread per output word; the thread's channels are contiguous in
// NHWC so its loads are float4 when the channel count allows.
template <int STORE>
__device__ __forceinline__ uint32_t encode_one(float v, int fn, float m) {
    if constexpr (STORE == QNN_STORE_BIN) {
        if (fn == QNN_FN_GRID) return v > 0.0f ? 1u : 0u;
        return qnn_binary_bit(v);
    } else {
        float code;
        if (fn == QNN_FN_GRID) code = __fmul_rn(v, m);                 // already k/m
        else if (fn == QNN_FN_BINARY_TANH) code = qnn_binary_tanh(v);  // +-1 as a code
        else code = qnn_quant_code_f(v, m);
        return (uint32_t)(int)code;
    }
}

template <int STORE>
__global__ __launch_bounds__(kBlock) void k_pack(const float* __restrict__ x,
                                                 uint32_t* __restrict__ y, size_t pixels,
                                                 int channels, int cw, int fn, float m) {
    constexpr int PW = (STORE == QNN_STORE_BIN) ? 32 : (STORE == QNN_STORE_I4) ? 8 : 4;
    constexpr int BITS = 32 / PW;
    constexpr uint32_t MASK = (BITS == 32) ? 0xffffffffu : ((1u << BITS) - 1u);
    const size_t words = pixels * (size_t)cw;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const bool vec = (channels % PW) == 0;   // full words, 16-B aligned channel groups
    for (size_t wi = (size_t)blockIdx.x * kBlock + threadIdx.x; wi < words; wi += stride) {
        uint32_t out = 0;
        if (vec) {
            // channels == cw * PW: word wi packs the PW floats at x + wi * PW (no pixel decode)
            const float4* s4 = reinterpret_cast<const float4*>(x + wi * PW);
#pragma unroll
            for (int j = 0; j < PW / 4; ++j) {
                float4 v = s4[j];
                out |= (encode_one<STORE>(v.x, fn, m) & MASK) << ((4 * j + 0) * BITS);
                out |= (encode_one<STORE>(v.y, fn, m) & MASK) << ((4 * j + 1) * BITS);
                out |= (encode_one<STORE>(v.z, fn, m) & MASK) << ((4 * j + 2) * BITS);
                out |= (encode_one<STORE>(v.w, fn, m) & MASK) << ((4 * j + 3) * BITS);
            }
        } else {
            const size_t p = wi / cw;
            const int w = (int)(wi - p * cw);
            const float* src = x + p * (size_t)channels + (size_t)w * PW;
            const int left = channels - w * PW;
            for (int j = 0; j < PW && j < left; ++j)
                out |= (encode_one<STORE>(src[j], fn, m) & MASK) << (j * BITS);
        }
        y[wi] = out;
    }
}

// ternary grid values {-1, 0, +1} -> (mask, sign) word pairs; one thread per pair (32 channels of one pixel)
__global__ __launch_bounds__(kBlock) void k_pack_t2(const float* __restrict__ x, uint32_t* __restrict__ y,
                                                    size_t pixels, int channels, int pairs) {
    const size_t total = pixels * (size_t)pairs;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const size_t p = i / pairs;
        const int w = (int)(i - p * pairs);
        const float* src = x + p * (size_t)channels + (size_t)w * 32;
        const int left = channels - w * 32;
        uint32_t mask = 0, sign = 0;
        for (int j = 0; j < 32 && j < left; ++j) {
            const float v = src[j];
            mask |= (v != 0.0f ? 1u : 0u) << j;
            sign |= (v > 0.0f ? 1u : 0u) << j;
        }
        y[2 * i] = mask;
        y[2 * i + 1] = sign;
    }
}
__global__ __launch_bounds__(kBlock) void k_unpack_t2(const uint32_t* __restrict__ x, float* __restrict__ y,
                                                      size_t pixels, int channels, int pairs) {
    const size_t n = pixels * (size_t)channels;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const size_t p = i / channels;
        const int c = (int)(i - p * channels);
        const uint32_t mask = x[(p * pairs + c / 32) * 2], sign = x[(p * pairs + c / 32) * 2 + 1];
        y[i] = ((mask >> (c & 31)) & 1u) ? (((sign >> (c & 31)) & 1u) ? 1.0f : -1.0f) : 0.0f;
    }
}

// BIN pack, channels % 32 == 0: the packed bit index equals the flat element index, so
// lane = element: one coalesced dword load per lane, one compare, and the wave's 64-bit
// lane mask IS two packed words (v_cmp writes it straight into an SGPR pair).  Eight
// loads are kept in flight per lane to cover HBM latency.
__global__ __launch_bounds__(kBlock) void k_pack_bin_ballot(const float* __restrict__ x,
                                                            uint32_t* __restrict__ y, size_t n,
                                                            int fn) {
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
    const size_t chunks = n / (64 * U);          // full chunks of U*64 elements
    const float thr = (fn == QNN_FN_GRID) ? 0.0f : 0x1p-24f;   // binary_tanh(x) = +1 iff x > 2^-24
    for (size_t c = wave; c < chunks; c += nwaves) {
        const float* src = x + c * (64 * U) + lane;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[u * 64];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long m = __ballot(v[u] > thr);
            if (lane == 2 * u) mine = (uint32_t)m;
            if (lane == 2 * u + 1) mine = (uint32_t)(m >> 32);
        }
        if (lane < 2 * U) y[c * (2 * U) + lane] = mine;
    }
    // tail: remaining (< U*64) elements, 64 at a time, by the first wave
    if (wave == 0) {
        for (size_t i = chunks * (64 * U); i < n; i += 64) {
            const float val = (i + lane < n) ? x[i + lane] : -1.0f;
            const unsigned long long m = __ballot(val > thr);
            if (lane == 0) y[i / 32] = (uint32_t)m;
            if (lane == 1 && i + 32 < n) y[i / 32 + 1] = (uint32_t)(m >> 32);
        }
    }
}

template <int STORE>
__global__ __launch_bounds__(kBlock) void k_unpack(const uint32_t* __restrict__ x,
                                                   float* __restrict__ y, size_t pixels,
                                                   int channels, int cw, float inv_m) {
    constexpr int PW = (STORE == QNN_STORE_BIN) ? 32 : (STORE == QNN_STORE_I4) ? 8 : 4;
    constexpr int BITS = 32 / PW;
    const size_t n = pixels * (size_t)channels;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const size_t p = i / channels;
        const int c = (int)(i - p * channels);
        const uint32_t word = x[p * cw + c / PW];
        const int sh = (c % PW) * BITS;
        float v;
        if constexpr (STORE == QNN_STORE_BIN) {
            v = ((word >> sh) & 1u) ? 1.0f : -1.0f;
        } else {
            int code = (int)(word << (32 - BITS - sh)) >> (32 - BITS);   // sign-extend
            v = __fmul_rn((float)code, inv_m);
        }
        y[i] = v;
    }
}

// AveragePooling2D(pool_size) 'valid' on a packed tensor (models/resnet.py:134 behind the last activation):
// window sums of grid values are exact integers, so y = float(sum of codes) * 2^-(bits-1) / (size*size) has ONE
// rounding (the division), exactly what tf.nn.avg_pool's float32 sum-then-divide gives on such values.
// One lane per (output pixel, channel): the lanes of a wave read the same words (broadcast) of consecutive channels.
template <int STORE>
__global__ __launch_bounds__(kBlock) void k_avgpool_packed(const uint32_t* __restrict__ x, float* __restrict__ y,
                                                           int N, int H, int W, int C, int cw, int size, float inv_m) {
    constexpr int PW = (STORE == QNN_STORE_BIN) ? 32 : (STORE == QNN_STORE_I4) ? 8 : 4;
    constexpr int BITS = 32 / PW;
    const int Ho = H / size, Wo = W / size;
    const size_t total = (size_t)N * Ho * Wo * C;
    const float area = (float)(size * size);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        const size_t q = i / C;
        const int ox = (int)(q % Wo);
        const int oy = (int)((q / Wo) % Ho);
        const size_t n = q / ((size_t)Wo * Ho);
        const int sh = (c % PW) * BITS;
        int acc = 0;
        for (int dy = 0; dy < size; ++dy) {
            const uint32_t* row = x + ((n * H + (size_t)oy * size + dy) * W + (size_t)ox * size) * cw + c / PW;
            for (int dx = 0; dx < size; ++dx) {
                const uint32_t word = row[(size_t)dx * cw];
                if constexpr (STORE == QNN_STORE_BIN) acc += ((word >> sh) & 1u) ? 1 : -1;
                else acc += (int)(word << (32 - BITS - sh)) >> (32 - BITS);
            }
        }
        y[i] = __fdiv_rn(__fmul_rn((float)acc, inv_m), area);
    }
}

// The same average for int4 codes with whole pixels per lane group: a wave owns one (output pixel, 64-channel block); lane
// = (pixel slot p = lane >> 3, word j = lane & 7), so one load instruction fetches eight whole 32-byte pixel blocks (the
// generic kernel above reads 4 bytes per lane 32 bytes apart and needs size^2 loads per lane: 19.5 us for the 8 x 8 pool
// behind the ImageNet ResNet, models/resnet.py:131); eight integer partial sums per lane, three shuffles, the same one
// rounding.  C % 64 == 0.
__global__ __launch_bounds__(256) void k_avgpool_i4_wave(const uint32_t* __restrict__ x, float* __restrict__ y, long nout,
                                                         int Ho, int Wo, int W, int H, int C, int cw, int size, float inv_m) {
    const int lane = threadIdx.x & 63;
    const int p = lane >> 3, j = lane & 7;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int cblocks = C / 64;
    if (wave >= nout * cblocks) return;
    const int cb = (int)(wave % cblocks);
    const long q = wave / cblocks;
    const int ox = (int)(q % Wo), oy = (int)((q / Wo) % Ho);
    const long n = q / ((long)Wo * Ho);
    const int area = size * size;
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = p; k < area; k += 8) {
        const int dy = k / size, dx = k - dy * size;
        const uint32_t word = x[((n * H + (long)oy * size + dy) * W + (long)ox * size + dx) * cw + cb * 8 + j];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (int)(word << (28 - 4 * i)) >> 28;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] += __shfl_xor(acc[i], 8);
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    if (p == 0) {
        float* yo = y + q * C + cb * 64 + j * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) yo[i] = __fdiv_rn(__fmul_rn((float)acc[i], inv_m), (float)area);
    }
}

// softmax over the last axis in float64 (max-shifted), one 256-thread workgroup per row (a classifier has few rows and
// ~1000 columns: one wave per row left most of the chip idle behind 3 x 16 serial float64 exponentials per lane):
// the classifier activation of models/resnet.py:137.
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ x, float* __restrict__ y, int cols) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (size_t)blockIdx.x * cols;
    float* yr = y + (size_t)blockIdx.x * cols;
    double mx = -__builtin_huge_val();
    for (int c = tid; c < cols; c += 256) mx = fmax(mx, (double)xr[c]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    double sum = 0.0;
    for (int c = tid; c < cols; c += 256) sum += exp((double)xr[c] - mx);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int c = tid; c < cols; c += 256) yr[c] = (float)(exp((double)xr[c] - mx) / sum);
}

}  // namespace

extern "C" int qnn_avgpool_packed_f32(const void* x, int store, int bits, int N, int H, int W, int C, int size,
                                      float* y, void* stream) {
    QNN_REQUIRE(x && y, QNN_EINVAL, "qnn_avgpool_packed_f32: null pointer");
    QNN_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && size > 0 && size <= H && size <= W, QNN_EINVAL,
                "qnn_avgpool_packed_f32: shape (%d,%d,%d,%d) size %d", N, H, W, C, size);
    QNN_REQUIRE(store == QNN_STORE_BIN || ((store == QNN_STORE_I4 || store == QNN_STORE_I8) && bits >= 1 && bits <= store),
                QNN_EINVAL, "qnn_avgpool_packed_f32: store=%d bits=%d", store, bits);
    const size_t total = (size_t)N * (H / size) * (W / size) * C;
    if (total == 0) return QNN_OK;
    const int cw = qnn_words(store, C);
    const float inv_m = store == QNN_STORE_BIN ? 1.0f : 1.0f / (float)(1u << (bits - 1));
    size_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 65535u) blocks = 65535u;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t* xu = (const uint32_t*)x;
    if (store == QNN_STORE_I4 && C % 64 == 0 && size * size >= 8) {
        const long nout = (long)N * (H / size) * (W / size), nwaves = nout * (C / 64);
        hipLaunchKernelGGL(k_avgpool_i4_wave, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, xu, y, nout, H / size, W / size,
                           W, H, C, cw, size, inv_m);
        QNN_HIP(hipGetLastError());
        return QNN_OK;
    }
    if (store == QNN_STORE_BIN)
        hipLaunchKernelGGL(k_avgpool_packed<QNN_STORE_BIN>, dim3((unsigned)blocks), dim3(kBlock), 0, s, xu, y, N, H, W, C, cw, size, inv_m);
    else if (store == QNN_STORE_I4)
        hipLaunchKernelGGL(k_avgpool_packed<QNN_STORE_I4>, dim3((unsigned)blocks), dim3(kBlock), 0, s, xu, y, N, H, W, C, cw, size, inv_m);
    else
        hipLaunchKernelGGL(k_avgpool_packed<QNN_STORE_I8>, dim3((unsigned)blocks), dim3(kBlock), 0, s, xu, y, N, H, W, C, cw, size, inv_m);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

// ---------------------------------------------------------------------------
extern "C" int qnn_binary_tanh_f32(const float* x, float* y, size_t n, void* stream) {
    QNN_REQUIRE(x && y, QNN_EINVAL, "qnn_binary_tanh_f32: null pointer");
    if (n == 0) return QNN_OK;
    hipLaunchKernelGGL(k_act_f32<QNN_FN_BINARY_TANH>, dim3(act_grid((n + 3) / 4)), dim3(kBlock), 0,
                       (hipStream_t)stream, x, y, n, 1.0f);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_quantized_tanh_f32(const float* x, float* y, size_t n, int nb, void* stream) {
    QNN_REQUIRE(x && y, QNN_EINVAL, "qnn_quantized_tanh_f32: null pointer");
    QNN_REQUIRE(nb >= 1 && nb <= 24, QNN_EINVAL, "qnn_quantized_tanh_f32: nb=%d out of range", nb);
    if (n == 0) return QNN_OK;
    const float m = (float)(1u << (nb - 1));
    hipLaunchKernelGGL(k_act_f32<QNN_FN_QUANTIZED_TANH>, dim3(act_grid((n + 3) / 4)), dim3(kBlock),
                       0, (hipStream_t)stream, x, y, n, m);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_ternary_abs_sum_f32(const float* x, size_t n, void* workspace16, void* stream) {
    QNN_REQUIRE(x && workspace16, QNN_EINVAL, "qnn_ternary_abs_sum_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // a memset node, not a copy from host memory: this call is captured into hipGraphs (a pageable host source would be
    // read at replay time, long after this frame is gone); the kernel adds the count itself
    QNN_HIP(hipMemsetAsync(workspace16, 0, 16, s));
    if (n == 0) return QNN_OK;
    const int g = grid_for(n) < 1024 ? grid_for(n) : 1024;
    hipLaunchKernelGGL(k_tern_sum, dim3(g), dim3(kBlock), 0, s, x, n, (double*)workspace16);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_ternary_apply_f32(const float* x, float* y, size_t n, const void* workspace16, void* stream) {
    QNN_REQUIRE(x && y && workspace16, QNN_EINVAL, "qnn_ternary_apply_f32: null pointer");
    if (n == 0) return QNN_OK;
    hipLaunchKernelGGL(k_tern_apply, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, n,
                       (const double*)workspace16);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_ternary_tanh_f32(const float* x, float* y, size_t n, void* workspace16,
                                    void* stream) {
    QNN_REQUIRE(x && y && workspace16, QNN_EINVAL, "qnn_ternary_tanh_f32: null pointer");
    if (n == 0) return QNN_OK;
    int rc = qnn_ternary_abs_sum_f32(x, n, workspace16, stream);
    if (rc != QNN_OK) return rc;
    return qnn_ternary_apply_f32(x, y, n, workspace16, stream);
}

extern "C" size_t qnn_packed_bytes(int store, size_t pixels, int channels) {
    if (store == QNN_STORE_F32) return pixels * (size_t)channels * 4;
    if (store != QNN_STORE_BIN && store != QNN_STORE_I4 && store != QNN_STORE_I8 && store != QNN_STORE_T2) return 0;
    return pixels * (size_t)qnn_words(store, channels) * 4;
}

extern "C" int qnn_pack_f32(const float* x, void* y, size_t pixels, int channels, int fn, int nb,
                            int store, void* stream) {
    QNN_REQUIRE(x && y, QNN_EINVAL, "qnn_pack_f32: null pointer");
    QNN_REQUIRE(channels > 0, QNN_EINVAL, "qnn_pack_f32: channels=%d", channels);
    QNN_REQUIRE(fn == QNN_FN_BINARY_TANH || fn == QNN_FN_QUANTIZED_TANH || fn == QNN_FN_GRID,
                QNN_EINVAL, "qnn_pack_f32: fn=%d cannot be encoded", fn);
    if (pixels == 0) return QNN_OK;
    const int cw = qnn_words(store, channels);
    const size_t words = pixels * (size_t)cw;
    float m = 1.0f;
    if (store == QNN_STORE_BIN) {
        QNN_REQUIRE(fn != QNN_FN_QUANTIZED_TANH, QNN_EINVAL,
                    "qnn_pack_f32: BIN storage needs binary_tanh or grid input");
    } else if (store == QNN_STORE_I4 || store == QNN_STORE_I8) {
        if (fn != QNN_FN_BINARY_TANH) {
            QNN_REQUIRE(nb >= (fn == QNN_FN_GRID ? 1 : 2) && nb <= store, QNN_EINVAL,
                        "qnn_pack_f32: nb=%d does not fit %d-bit storage", nb, store);
            m = (float)(1u << (nb - 1));
        }
    } else if (store == QNN_STORE_T2) {
        QNN_REQUIRE(fn == QNN_FN_GRID, QNN_EINVAL,
                    "qnn_pack_f32: QNN_STORE_T2 packs values that are already {-1, 0, +1} (fn = QNN_FN_GRID): "
                    "ternary_tanh needs the batch-wide mean first (qnn_ternary_tanh_f32)");
        hipLaunchKernelGGL(k_pack_t2, dim3(grid_for(pixels * (size_t)(cw / 2))), dim3(kBlock), 0, (hipStream_t)stream, x,
                           (uint32_t*)y, pixels, channels, cw / 2);
        QNN_HIP(hipGetLastError());
        return QNN_OK;
    } else {
        qnn_set_error("qnn_pack_f32: store=%d is not a packed kind", store);
        return QNN_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_for(words);
    if (store == QNN_STORE_BIN && (channels % 32) == 0)
        hipLaunchKernelGGL(k_pack_bin_ballot, dim3(grid_for(pixels * (size_t)channels / 8)), dim3(kBlock),
                           0, s, x, (uint32_t*)y, pixels * (size_t)channels, fn);
    else if (store == QNN_STORE_BIN)
        hipLaunchKernelGGL(k_pack<QNN_STORE_BIN>, dim3(g), dim3(kBlock), 0, s, x, (uint32_t*)y,
                           pixels, channels, cw, fn, m);
    else if (store == QNN_STORE_I4)
        hipLaunchKernelGGL(k_pack<QNN_STORE_I4>, dim3(g), dim3(kBlock), 0, s, x, (uint32_t*)y,
                           pixels, channels, cw, fn, m);
    else
        hipLaunchKernelGGL(k_pack<QNN_STORE_I8>, dim3(g), dim3(kBlock), 0, s, x, (uint32_t*)y,
                           pixels, channels, cw, fn, m);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_unpack_f32(const void* x, float* y, size_t pixels, int channels, int store,
                              int nb, void* stream) {
    QNN_REQUIRE(x && y, QNN_EINVAL, "qnn_unpack_f32: null pointer");
    QNN_REQUIRE(channels > 0, QNN_EINVAL, "qnn_unpack_f32: channels=%d", channels);
    if (pixels == 0) return QNN_OK;
    const int cw = qnn_words(store, channels);
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_for(pixels * (size_t)channels);
    float inv_m = 1.0f;
    if (store == QNN_STORE_T2) {
        hipLaunchKernelGGL(k_unpack_t2, dim3(g), dim3(kBlock), 0, s, (const uint32_t*)x, y, pixels, channels, cw / 2);
        QNN_HIP(hipGetLastError());
        return QNN_OK;
    }
    if (store != QNN_STORE_BIN) {
        QNN_REQUIRE(nb >= 1 && nb <= store, QNN_EINVAL, "qnn_unpack_f32: nb=%d vs store=%d", nb, store);
        inv_m = 1.0f / (float)(1u << (nb - 1));
    }
    if (store == QNN_STORE_BIN)
        hipLaunchKernelGGL(k_unpack<QNN_STORE_BIN>, dim3(g), dim3(kBlock), 0, s,
                           (const uint32_t*)x, y, pixels, channels, cw, inv_m);
    else if (store == QNN_STORE_I4)
        hipLaunchKernelGGL(k_unpack<QNN_STORE_I4>, dim3(g), dim3(kBlock), 0, s,
                           (const uint32_t*)x, y, pixels, channels, cw, inv_m);
    else if (store == QNN_STORE_I8)
        hipLaunchKernelGGL(k_unpack<QNN_STORE_I8>, dim3(g), dim3(kBlock), 0, s,
                           (const uint32_t*)x, y, pixels, channels, cw, inv_m);
    else {
        qnn_set_error("qnn_unpack_f32: store=%d is not a packed kind", store);
        return QNN_EINVAL;
    }
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

extern "C" int qnn_softmax_f32(const float* x, float* y, size_t rows, int cols, void* stream) {
    QNN_REQUIRE(cols > 0, QNN_EINVAL, "qnn_softmax_f32: cols=%d", cols);
    if (rows == 0) return QNN_OK;
    QNN_REQUIRE(x && y, QNN_EINVAL, "qnn_softmax_f32: null pointer");
    QNN_REQUIRE(rows < 2000000000ul, QNN_EUNSUPPORTED, "qnn_softmax_f32: too many rows for one launch");
    hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, cols);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}
