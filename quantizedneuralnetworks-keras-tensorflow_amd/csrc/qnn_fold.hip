// qnn_fold_prepare / _free / _info / _constants / _eval (qnn_abi.h): the epilogue of a low-bit convolution as integer
// thresholds, PROVEN against the float32 chain on every point of the layer's accumulator domain.
//
// What is folded (reference): K.bias_add (binary_layers.py:179-180, quantized_layers.py:186-187), the inference
// BatchNormalization behind it (models/vgg.py:16, models/resnet.py:61), the residual merge keras.layers.add + Lambda(x * 0.5)
// (models/resnet.py:127-128) and quantized_tanh (quantized_ops.py:87-100, models/vgg.py:17, models/resnet.py:129).
// The un-folded kernels evaluate that chain in float32 with one rounding per operation (qnn_epi_value, qnn_epi_residual,
// qnn_epi_code in qnn_common.h); this file evaluates THE SAME inline functions on every accumulator value the layer can
// produce and searches, per channel, the two constants of the folded form (qnn_fold.h).
//
// One workgroup per output channel:
//   1. domain: acc in [sum_k min(w_k a_lo, w_k a_hi), sum_k max(..)] from the int8 weight image and the input codes' range;
//   2. the chain's thresholds T(k, sc) = first accumulator with code >= k, by bisection (every float32 step is monotone);
//   3. for a few float32 neighbours of the real-number slope: the folded form's thresholds X(k, sc) by bisection on the
//      very device function the kernels inline, and the interval of offsets beta for which X and T select the same
//      accumulators; the middle of a non-empty interval is the candidate;
//   4. PROOF: the candidate is compared with the chain on every (accumulator, shortcut code) point of the domain; only a
//      candidate with zero differing points is accepted.  Steps 2-3 merely find candidates; step 4 is what is relied on.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qnn_fold.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCandidates = 129;     // slope candidates: the rounded real-number slope and +-1 .. +-64 float32 neighbours

struct FoldProblem {
    EpiArgs e;                       // chain constants; e.scale = 2^-(wshift + xshift), e.res unused
    const int8_t* wq8;               // int8 weight image [cout][K], code * 16
    int K;
    int a_lo, a_hi;                  // range of the input codes
    int has_res;
    float res_scale, post_scale;     // shortcut value = code * res_scale; merged = (shortcut + y) * post_scale
    int mode;                        // 1 or 2 (qnn_fold.h)
    float* A;
    int32_t* beta;
    float* C;
    int32_t* status;                 // per channel: 1 folded, 0 not
    int32_t* dom;                    // per channel: lo, hi
    unsigned long long* points;      // per channel: points compared
};

// the chain, exactly as k_conv_generic evaluates it (qnn_conv.hip)
__device__ __forceinline__ int chain_code(const FoldProblem& p, int c, int acc, int sc) {
    float v = __fmul_rn((float)acc, p.e.scale);
    v = qnn_epi_value(v, c, p.e);
    if (p.has_res) v = __fmul_rn(__fadd_rn(__fmul_rn((float)sc, p.res_scale), v), p.post_scale);
    return qnn_epi_code(v, p.e);
}

__device__ __forceinline__ long long block_reduce(long long v, bool want_max, long long* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) sh[t] = want_max ? (sh[t] > sh[t + s] ? sh[t] : sh[t + s]) : (sh[t] < sh[t + s] ? sh[t] : sh[t + s]);
        __syncthreads();
    }
    const long long r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ long long block_sum(long long v, long long* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    const long long r = sh[0];
    __syncthreads();
    return r;
}

constexpr long long kInf = (1LL << 60);

__global__ __launch_bounds__(kThreads) void k_fold_prepare(FoldProblem p) {
    __shared__ long long sh[kThreads];
    const int c = blockIdx.x, t = threadIdx.x;
    const int m = (int)p.e.act_m;                    // 8: codes -m .. m-1
    // ---- 1. domain ----
    long long lo = 0, hi = 0;
    for (int k = t; k < p.K; k += kThreads) {
        const int w = (int)p.wq8[(size_t)c * p.K + k] / 16;       // the image holds code * 16 exactly
        const int x0 = w * p.a_lo, x1 = w * p.a_hi;
        lo += x0 < x1 ? x0 : x1;
        hi += x0 < x1 ? x1 : x0;
    }
    lo = block_sum(lo, sh);
    hi = block_sum(hi, sh);
    if (t == 0) { p.dom[2 * c] = (int)lo; p.dom[2 * c + 1] = (int)hi; p.status[c] = 0; p.A[c] = 0.0f; p.beta[c] = 0; p.C[c] = 0.0f; p.points[c] = 0; }
    const int mode = p.mode;
    const int magic = mode == 2 ? kFoldMagicBits : 0;
    // ---- real-number constants of the chain ----
    const double inv = p.e.bn_inv ? (double)p.e.bn_inv[c] : 1.0, shift = p.e.bn_inv ? (double)p.e.bn_shift[c] : 0.0;
    const double bias = p.e.bias ? (double)p.e.bias[c] : 0.0;
    const double post = p.has_res ? (double)p.post_scale : 1.0;
    const double g = (double)p.e.scale * inv * post * (double)m;              // codes per accumulator step
    const double b0 = (bias * inv + shift) * post * (double)m;                // code at accumulator 0 (shortcut 0)
    if (!(g == g) || !(b0 == b0) || g == 0.0 || fabs(g) > 1.0e6 || fabs(g) < 1.0e-9 || fabs(b0) > 1.0e9) return;   // uniform
    const int dir = g < 0.0 ? -1 : 1;
    const int nsc = p.has_res ? 2 * m : 1, sc0 = p.has_res ? -m : 0;
    const long long zlo = dir > 0 ? lo : -hi, zhi = dir > 0 ? hi : -lo;       // z = dir * acc: the chain is non-decreasing in z
    const double q = p.has_res ? 2048.0 : 4096.0;                             // Q11 / Q12
    const double a_nom = g * q / (32767.0 * 256.0);                           // per matrix-pipe accumulator unit (acc * 256)
    const double beta_nom = (b0 + 0.5 - (p.has_res ? 4.0 : 0.0)) * 256.0 / g;  // in those units
    if (fabs(beta_nom) > 1.0e9) return;
    const long long bnz = (long long)llrint(beta_nom) * dir;                  // nominal offset in the z domain
    // this thread's (k, sc)
    const int nk = 2 * m - 1;
    const bool active = t < nk * nsc;
    const int k = active ? (t % nk) - (m - 1) : 0;
    const int sc = active ? sc0 + t / nk : 0;
    // ---- 2. the chain's threshold: first z in [zlo, zhi] with code >= k (zhi + 1: none) ----
    long long Tz = zhi + 1;
    if (active) {
        long long a = zlo, b = zhi + 1;                                       // invariant: code(z) >= k for z >= b (or b = zhi + 1)
        while (a < b) {
            const long long mid = a + ((b - a) >> 1);
            if (chain_code(p, c, (int)(dir * mid), sc) >= k) b = mid; else a = mid + 1;
        }
        Tz = a;
    }
    // ---- 3. + 4. candidates ----
    const long long npts = (zhi - zlo + 1) * nsc;
    for (int cand = 0; cand < kCandidates; ++cand) {
        const int step = cand == 0 ? 0 : ((cand & 1) ? (cand + 1) / 2 : -(cand / 2));
        float A = (float)a_nom;
        A = __int_as_float(__float_as_int(A) + step);                          // float32 neighbours (same sign, same binade nearly always)
        const float C = mode == 2 ? (float)(-kFoldMagic * (double)A) : 0.0f;
        const bool res = p.has_res != 0;
        // folded thresholds in the z domain: accw = dir * z'
        long long blo = -kInf, bhi = kInf;
        if (active) {
            const long long Zlo = 256 * zlo + bnz - (1 << 20), Zhi = 256 * zhi + bnz + (1 << 20);
            long long X;
            if (qnn_fold_code((int)(dir * Zhi) + magic, A, C, mode, res, sc) < k) X = kInf;
            else if (qnn_fold_code((int)(dir * Zlo) + magic, A, C, mode, res, sc) >= k) X = -kInf;
            else {
                long long a = Zlo, b = Zhi;                                    // F(a) < k <= F(b)
                while (b - a > 1) {
                    const long long mid = a + ((b - a) >> 1);
                    if (qnn_fold_code((int)(dir * mid) + magic, A, C, mode, res, sc) >= k) b = mid; else a = mid;
                }
                X = b;
            }
            if (Tz > zlo && Tz <= zhi) {               // threshold inside the domain
                if (X == kInf || X == -kInf) { blo = kInf; bhi = -kInf; }
                else { blo = X - 256 * Tz; bhi = X - 256 * (Tz - 1) - 1; }
            } else if (Tz == zlo) {                    // the chain is >= k everywhere
                if (X == kInf) { blo = kInf; bhi = -kInf; }
                else if (X != -kInf) blo = X - 256 * zlo;
            } else {                                   // never
                if (X == -kInf) { blo = kInf; bhi = -kInf; }
                else if (X != kInf) bhi = X - 256 * zhi - 1;
            }
        }
        blo = block_reduce(blo, true, sh);
        bhi = block_reduce(bhi, false, sh);
        if (blo > bhi) continue;                       // uniform
        long long bz;
        if (blo > -kInf / 2 && bhi < kInf / 2) bz = blo + ((bhi - blo) >> 1);
        else bz = bnz < blo ? blo : bnz > bhi ? bhi : bnz;
        if (bz > (1LL << 30) || bz < -(1LL << 30)) continue;
        const int beta = (int)(dir * bz);
        // (mode 2: outside |accw + beta| < 2^22 the accumulator's bits leave the binade of the magic constant; positive floats
        // order like their bit patterns, so the folded form stays monotone there and merely saturates -- whether the
        // thresholds all lie inside is decided by the sweep below, like everything else)
        // ---- 4. proof: every point of the domain ----
        long long bad = 0;
        for (long long i = t; i < npts; i += kThreads) {
            const int s = sc0 + (int)(i % nsc);
            const int acc = (int)(dir * (zlo + i / nsc));
            bad += chain_code(p, c, acc, s) != qnn_fold_code(256 * acc + beta + magic, A, C, mode, res, s);
        }
        bad = block_sum(bad, sh);
        if (bad == 0) {
            if (t == 0) { p.A[c] = A; p.beta[c] = beta + magic; p.C[c] = C; p.status[c] = 1; p.points[c] = (unsigned long long)npts; }
            return;
        }
    }
}

// ---- mode 3: the image entry.  chain = the entry's own specification (qnn_abi.h, qnn_conv2d_forward):
// code = clip(rint(fma(float(S), A, B)), -8, 7) with A, B from qnn_u8_affine; domain S in 255 * [sum of the negative
// weight codes, sum of the positive ones] ----
struct FoldProblemU8 {
    EpiArgs e;                       // e.scale = 255 * 2^wshift (the divisor D), bias / BN pointers, act_m, fn
    const float* wq;                 // quantized weights [cout][K] as float32 values
    int K;
    float wscale;                    // 2^wshift: value -> integer code
    float* A;
    float* C;
    int32_t* status;
    int32_t* dom;
    unsigned long long* points;
};

__global__ __launch_bounds__(kThreads) void k_fold_prepare_u8(FoldProblemU8 p) {
    __shared__ long long sh[kThreads];
    const int c = blockIdx.x, t = threadIdx.x;
    long long lo = 0, hi = 0;
    for (int k = t; k < p.K; k += kThreads) {
        const int code = (int)rintf(__fmul_rn(p.wq[(size_t)c * p.K + k], p.wscale));
        lo += code < 0 ? 255 * code : 0;
        hi += code > 0 ? 255 * code : 0;
    }
    lo = block_sum(lo, sh);
    hi = block_sum(hi, sh);
    if (t == 0) { p.dom[2 * c] = (int)lo; p.dom[2 * c + 1] = (int)hi; p.status[c] = 0; p.A[c] = 0.0f; p.C[c] = 0.0f; p.points[c] = 0; }
    const U8Affine af = qnn_u8_affine(p.e, c);
    if (!(af.A == af.A) || !(af.B == af.B) || af.A == 0.0f) return;                    // uniform
    const double k12 = 4096.0 / 32767.0;
    const float a_nom = (float)((double)af.A * k12);
    // snorm16 rounds to nearest and the code is the floor of the Q12 value: the -1/8192 centres the thresholds on k - 1/2
    const float c_nom = (float)(((double)af.B + 0.5 - 1.0 / 8192.0) * k12);
    const long long npts = hi - lo + 1;
    for (int cand = 0; cand < 5 * 33; ++cand) {
        const int ia = cand / 33, ic = cand % 33;
        const int da = ia == 0 ? 0 : ((ia & 1) ? (ia + 1) / 2 : -(ia / 2));
        const int dc = ic == 0 ? 0 : ((ic & 1) ? (ic + 1) / 2 : -(ic / 2));
        const float A = __int_as_float(__float_as_int(a_nom) + da), C = __int_as_float(__float_as_int(c_nom) + dc);
        long long bad = 0;
        for (long long i = t; i < npts; i += kThreads) {
            const int S = (int)(lo + i);
            const int want = (int)qnn_u8_value(__fmaf_rn((float)S, af.A, af.B), p.e);
            bad += want != qnn_fold_code(S, A, C, 3, false, 0);
        }
        bad = block_sum(bad, sh);
        if (bad == 0) {
            if (t == 0) { p.A[c] = A; p.C[c] = C; p.status[c] = 1; p.points[c] = (unsigned long long)npts; }
            return;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_fold_eval(const float* __restrict__ A, const int32_t* __restrict__ beta,
                                                        const float* __restrict__ C, int mode, int c,
                                                        int has_res, const int32_t* __restrict__ acc,
                                                        const int32_t* __restrict__ sc, int32_t* __restrict__ codes, size_t n) {
    const float a = A[c], cc = C[c];
    const int b = beta[c];
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
        codes[i] = qnn_fold_code(mode == 3 ? acc[i] : 256 * acc[i] + b, a, cc, mode, has_res != 0, has_res ? sc[i] : 0);
}

}  // namespace

extern "C" int qnn_fold_prepare(const qnn_weights_t* w, int x_store, int x_bits, const qnn_epilogue_t* epi, void* stream,
                                qnn_fold_t** out) {
    QNN_REQUIRE(w && epi && out, QNN_EINVAL, "qnn_fold_prepare: null pointer");
    *out = nullptr;
    const bool image = x_store == QNN_STORE_U8 || x_store == QNN_STORE_F32_IMAGE;
    if (image) x_store = QNN_STORE_U8;
    QNN_REQUIRE(image || (w->store == QNN_STORE_I4 && w->d_mfma && x_store == QNN_STORE_I4), QNN_EUNSUPPORTED,
                "qnn_fold_prepare: needs weights prepacked for QNN_STORE_I4 with a matrix-pipe image and QNN_STORE_I4 input "
                "(store=%d x_store=%d)", w->store, x_store);
    if (image) {
        QNN_REQUIRE(w->d_wq && w->kh == 3 && w->kw == 3 && w->cin == 3 && w->stride == 1 &&
                        (w->wkind == QNN_W_BINARY || w->wkind == QNN_W_TERNARY || (w->wkind == QNN_W_QUANT && w->wbits <= 4)) &&
                        !epi->res, QNN_EUNSUPPORTED,
                    "qnn_fold_prepare: the image entry is folded for 3x3 stride-1 layers on 3 channels with weights of <= 4 bits");
        x_bits = 0;
    } else
    QNN_REQUIRE(x_bits >= 1 && x_bits <= 4, QNN_EINVAL, "qnn_fold_prepare: x_bits=%d", x_bits);
    QNN_REQUIRE(epi->fn == QNN_FN_QUANTIZED_TANH && epi->act_bits == 4 && epi->out_store == QNN_STORE_I4, QNN_EUNSUPPORTED,
                "qnn_fold_prepare: only quantized_tanh with 4-bit codes stored as QNN_STORE_I4 is folded (fn=%d act_bits=%d "
                "out_store=%d)", epi->fn, epi->act_bits, epi->out_store);
    QNN_REQUIRE(epi->trick_s == 0.0f, QNN_EUNSUPPORTED, "qnn_fold_prepare: the output-side trick is not folded");
    QNN_REQUIRE((epi->bn_inv == nullptr) == (epi->bn_shift == nullptr), QNN_EINVAL,
                "qnn_fold_prepare: bn_inv and bn_shift must both be set or both be NULL");
    const bool has_res = epi->res != nullptr;
    if (has_res)
        QNN_REQUIRE(epi->res_store == QNN_STORE_I4 && epi->res_bits == 4 && epi->post_scale == 0.5f, QNN_EUNSUPPORTED,
                    "qnn_fold_prepare: a shortcut must be 4-bit QNN_STORE_I4 codes merged with post_scale 0.5 (res_store=%d "
                    "res_bits=%d post_scale=%g)", epi->res_store, epi->res_bits, (double)epi->post_scale);
    hipStream_t s = (hipStream_t)stream;
    qnn_fold* f = (qnn_fold*)calloc(1, sizeof(qnn_fold));
    QNN_REQUIRE(f, QNN_ENOMEM, "qnn_fold_prepare: out of host memory");
    f->w = w; f->x_store = x_store; f->x_bits = x_bits;
    f->bn_inv = epi->bn_inv; f->bn_shift = epi->bn_shift;
    f->fn = epi->fn; f->act_bits = epi->act_bits; f->out_store = epi->out_store;
    f->has_res = has_res ? 1 : 0; f->res_store = has_res ? epi->res_store : 0; f->res_bits = has_res ? epi->res_bits : 0;
    f->post_scale = has_res ? epi->post_scale : 1.0f;
    f->cout = w->cout;
    int32_t *d_status = nullptr, *d_dom = nullptr;
    unsigned long long* d_points = nullptr;
    auto fail = [&](int code) {
        if (f->d_a) (void)hipFree(f->d_a);
        if (f->d_b) (void)hipFree(f->d_b);
        if (f->d_c) (void)hipFree(f->d_c);
        if (d_status) (void)hipFree(d_status);
        if (d_dom) (void)hipFree(d_dom);
        if (d_points) (void)hipFree(d_points);
        free(f);
        return code;
    };
#define FOLD_HIP(expr)                                                                                           \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) {                                                                                  \
            qnn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);            \
            return fail(QNN_EHIP);                                                                               \
        }                                                                                                        \
    } while (0)
    FOLD_HIP(hipMalloc(&f->d_a, sizeof(float) * w->cout));
    FOLD_HIP(hipMalloc(&f->d_b, sizeof(int32_t) * w->cout));
    FOLD_HIP(hipMalloc(&f->d_c, sizeof(float) * w->cout));
    FOLD_HIP(hipMalloc(&d_status, sizeof(int32_t) * w->cout));
    FOLD_HIP(hipMalloc(&d_dom, sizeof(int32_t) * 2 * w->cout));
    FOLD_HIP(hipMalloc(&d_points, sizeof(unsigned long long) * w->cout));
    FoldProblem p;
    memset((void*)&p, 0, sizeof(p));
    p.e.bias = w->d_bias;
    p.e.bn_inv = epi->bn_inv; p.e.bn_shift = epi->bn_shift;
    p.e.scale = ldexpf(1.0f, -(w->wshift + x_bits - 1));
    p.e.act_m = 8.0f;
    p.e.fn = epi->fn; p.e.out_store = epi->out_store;
    p.wq8 = (const int8_t*)w->d_mfma;
    p.K = w->kh * w->kw * w->cin;
    p.a_lo = x_bits == 1 ? -1 : -(1 << (x_bits - 1));
    p.a_hi = x_bits == 1 ? 1 : (1 << (x_bits - 1)) - 1;
    p.has_res = f->has_res;
    p.res_scale = ldexpf(1.0f, -(4 - 1));
    p.post_scale = f->post_scale;
    p.A = f->d_a; p.beta = f->d_b; p.C = f->d_c; p.status = d_status; p.dom = d_dom; p.points = d_points;
    int32_t* h_status = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)w->cout);
    unsigned long long* h_points = (unsigned long long*)malloc(sizeof(unsigned long long) * (size_t)w->cout);
    if (!h_status || !h_points) { free(h_status); free(h_points); qnn_set_error("qnn_fold_prepare: out of host memory"); return fail(QNN_ENOMEM); }
    FoldProblemU8 pu;
    memset((void*)&pu, 0, sizeof(pu));
    if (image) {
        pu.e = p.e;
        pu.e.scale = 255.0f * (float)(1 << w->wshift);
        pu.wq = w->d_wq; pu.K = w->kh * w->kw * w->cin; pu.wscale = (float)(1 << w->wshift);
        pu.A = f->d_a; pu.C = f->d_c; pu.status = d_status; pu.dom = d_dom; pu.points = d_points;
        (void)hipMemsetAsync(f->d_b, 0, sizeof(int32_t) * w->cout, s);
    }
    // the "bits" form first (one instruction fewer per value; needs the accumulators inside +-2^22 on every channel),
    // the conversion form if some channel has no fold in it; the image entry has its own single form (mode 3)
    for (int mode = image ? 3 : 2; mode >= (image ? 3 : 1); --mode) {
        p.mode = mode;
        if (image) hipLaunchKernelGGL(k_fold_prepare_u8, dim3((unsigned)w->cout), dim3(kThreads), 0, s, pu);
        else hipLaunchKernelGGL(k_fold_prepare, dim3((unsigned)w->cout), dim3(kThreads), 0, s, p);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipMemcpyAsync(h_status, d_status, sizeof(int32_t) * w->cout, hipMemcpyDeviceToHost, s);
        if (he == hipSuccess) he = hipMemcpyAsync(h_status + w->cout, d_dom, sizeof(int32_t) * 2 * w->cout, hipMemcpyDeviceToHost, s);
        if (he == hipSuccess) he = hipMemcpyAsync(h_points, d_points, sizeof(unsigned long long) * w->cout, hipMemcpyDeviceToHost, s);
        if (he == hipSuccess) he = hipStreamSynchronize(s);
        if (he != hipSuccess) {
            free(h_status); free(h_points);
            qnn_set_error("qnn_fold_prepare: %s", hipGetErrorString(he));
            return fail(QNN_EHIP);
        }
        f->mode = mode;
        f->folded = 0; f->points = 0;
        f->acc_lo = 0; f->acc_hi = 0;
        for (int c = 0; c < w->cout; ++c) {
            f->folded += h_status[c] != 0;
            f->points += (long long)h_points[c];
            const int lo = h_status[w->cout + 2 * c], hi = h_status[w->cout + 2 * c + 1];
            if (c == 0 || lo < f->acc_lo) f->acc_lo = lo;
            if (c == 0 || hi > f->acc_hi) f->acc_hi = hi;
        }
        if (f->folded == w->cout) break;
    }
    free(h_status); free(h_points);
    (void)hipFree(d_status); (void)hipFree(d_dom); (void)hipFree(d_points);
#undef FOLD_HIP
    *out = f;
    return QNN_OK;
}

extern "C" int qnn_fold_free(qnn_fold_t* f) {
    if (!f) return QNN_OK;
    if (f->d_a) (void)hipFree(f->d_a);
    if (f->d_b) (void)hipFree(f->d_b);
    if (f->d_c) (void)hipFree(f->d_c);
    free(f);
    return QNN_OK;
}

extern "C" int qnn_fold_info(const qnn_fold_t* f, qnn_fold_info_t* info) {
    QNN_REQUIRE(f && info, QNN_EINVAL, "qnn_fold_info: null pointer");
    info->channels = f->cout;
    info->folded = f->folded;
    info->usable = f->folded == f->cout ? 1 : 0;
    info->shortcut_codes = f->has_res ? 16 : 1;
    info->points = f->points;
    info->acc_lo = f->acc_lo;
    info->acc_hi = f->acc_hi;
    info->mode = f->mode;
    return QNN_OK;
}

extern "C" int qnn_fold_constants(const qnn_fold_t* f, float* A, int32_t* beta, float* C, void* stream) {
    QNN_REQUIRE(f && A && beta, QNN_EINVAL, "qnn_fold_constants: null pointer");
    if (C) QNN_HIP(hipMemcpyAsync(C, f->d_c, sizeof(float) * f->cout, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    QNN_HIP(hipMemcpyAsync(A, f->d_a, sizeof(float) * f->cout, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    QNN_HIP(hipMemcpyAsync(beta, f->d_b, sizeof(int32_t) * f->cout, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return QNN_OK;
}

extern "C" int qnn_fold_eval(const qnn_fold_t* f, int c, const int32_t* acc, const int32_t* sc, int32_t* codes, size_t n,
                             void* stream) {
    QNN_REQUIRE(f && acc && codes, QNN_EINVAL, "qnn_fold_eval: null pointer");
    QNN_REQUIRE(c >= 0 && c < f->cout, QNN_EINVAL, "qnn_fold_eval: channel %d of %d", c, f->cout);
    QNN_REQUIRE(!f->has_res || sc, QNN_EINVAL, "qnn_fold_eval: this fold merges a shortcut: sc must be given");
    if (n == 0) return QNN_OK;
    size_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(k_fold_eval, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, f->d_a, f->d_b, f->d_c,
                       f->mode, c, f->has_res, acc, sc, codes, n);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}
