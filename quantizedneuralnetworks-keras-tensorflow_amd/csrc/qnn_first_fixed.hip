// OPTIONAL fixed-point variant of the float-input first layer (qnn_set_option("first_fixed", 1); the exact FMA-chain kernel
// of qnn_first.hip stays the default).  Dispatch: qnn_try_launch_mfma (qnn_mfma.hip).
//
// The exact layer cannot run faster than the f32 matrix peak allows (~96 us for 4096 x 32^2 x 3 -> 64).  For inputs in
// [0, 1] (images / 255, utils/load_data.py:40) and weights of <= 4 bits the same convolution is an INTEGER problem:
//     X = rint(x * 2^23) in [0, 2^23] = sum_j 2^(8j) (s_j + 128)  with the signed bytes s_j = (byte j of X) ^ 0x80,
//     sum_k c_k X_k = sum_j 2^(8j) (sum_k c_k s_jk) + 128 (1 + 2^8 + 2^16) sum_k c_k     (c_k: integer weight codes)
// -- three int8 MFMA passes (v_mfma_i32_16x16x64_i8, a separate pipe from the VALU), exact int32 arithmetic, ONE rounding
// when the sum becomes a float.  A zero-padding tap is X = 0, i.e. digits (-128, -128, -128): the halo converts like any
// other value.  The result differs from the ideal (real-number) convolution by at most 27 * 2^-24 = 1.6e-6 (input rounding)
// plus one float32 rounding -- inside the north star's 1e-5 -- but it is NOT the oracle's float32 FMA chain: activation
// codes sitting within that distance of a rounding threshold can differ (tests measure how many).
//
// Layout: a wave walks a strip of 16 conv columns down the image two conv rows at a time.  MFMA rows = 16 positions =
// 4 pool windows x (2 x 2), columns = 16 filters, K-block kq = tap row dy: 3 taps x (3 channels + 1 zero byte) + 4 zero
// bytes.  The float rows are converted once (108 values per step, two per lane) into three wave-private LDS byte planes
// (ring of four input rows, row pitch 40 words so that the four rows of a step fall on disjoint banks); operands are three
// ds_read_b32 per (tile, digit).  In the C/D layout a lane holds the four positions of ONE window for one filter: pooling
// is an in-lane max on the combined integers (filters of channels with negative BN scale are negated, as in the exact
// kernel), and the epilogue / nibble transpose / packed store are the exact kernel's.
#include <stdlib.h>

#include <type_traits>

#include "qnn_mfma_common.h"

namespace {

constexpr int kRowPitch = 40;                 // words per LDS row (18 pixels used; 40 == 8 mod 32)
constexpr int kPlane = 4 * kRowPitch;         // words per digit plane (ring of 4 rows)
constexpr int kWaveLds = 3 * kPlane;          // words per wave

// x -> the three offset digits of X = rint(x * 2^23) as one word: byte j = (digit j of X) - 128 as a signed byte
__device__ __forceinline__ uint32_t to_digits(float x, bool& bad) {
    bad |= !(x >= 0.0f && x <= 1.0f);                             // outside the domain (NaN included): reported, see below
    const int X = (int)rintf(__fmul_rn(x, 8388608.0f));
    return (uint32_t)min(max(X, 0), 8388608) ^ 0x00808080u;
}
constexpr int kDigitBias = 128 * (1 + 256 + 65536);              // sum_j 128 * 2^(8j)

// (QNN_STORE_I4, 2, BIN): the fused pipeline, quantized_tanh / binary_tanh codes;  (QNN_STORE_F32, 1, false): the raw layer
template <int OUT, int POOL, bool BIN>
__global__ __launch_bounds__(256, 4) void k_conv_first_fixed(ConvGeom g, EpiArgs e, const float* __restrict__ x,
                                                              const float* __restrict__ wq, void* __restrict__ y,
                                                              int ntasks, int spr, FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                              int rc, uint32_t img_x, float wscale, float vscale, float rvscale,
                                                              uint32_t* __restrict__ domain_flag) {
    extern __shared__ __attribute__((aligned(16))) char smem_fx[];
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem_fx) + wave * kWaveLds;
    uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
    uint4* tab = reinterpret_cast<uint4*>(smem_fx + 4 * kWaveLds * 4);      // [filter block][lane][2]
    for (int i = lane; i < kWaveLds; i += 64) lds[i] = 0;        // the 4th byte of every pixel stays zero

    // ---- filters: B operand of block nt = filter nt*16 + r, k-block kq = tap row dy.  Wave nt prepares block nt for the
    // whole workgroup (the scalar setup is ~250 instructions; every wave doing all four blocks cost 10 % of the launch) ----
    const float mfold = (OUT == QNN_STORE_I4 && !BIN) ? e.act_m : 1.0f;
    {
        const int c = wave * 16 + r;
        const float bias = e.bias ? e.bias[c] : 0.0f;
        const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
        const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
        const bool flip = POOL == 2 && inv < 0.0f;               // pool with max only: negate the filter, fold the sign
        // this lane's tap row (k-block 3 is padding: it loads row 2 and keeps zeros)
        const float* wrow = wq + (size_t)c * 27 + (kq < 3 ? kq : 2) * 9;
        int part = 0;
        uint32_t wd[3] = {0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int code = (int)rintf(__fmul_rn(wrow[i], wscale));
            if (flip) code = -code;
            if (kq == 3) code = 0;
            part += code;
            wd[i / 3] |= (uint32_t)(code & 0xFF) << (8 * (i % 3));
        }
        part += __shfl_xor(part, 16);                            // sum of the 27 codes of filter c: the three tap rows
        part += __shfl_xor(part, 32);                            // sit in the lanes r, r + 16, r + 32
        const float nb_ = __fmul_rn(flip ? -bias : bias, rvscale);           // bias / vscale (a power of two)
        const float ninv_ = __fmul_rn(__fmul_rn(flip ? -inv : inv, vscale), mfold);
        const float nshift_ = __fmul_rn(shift, mfold);
        tab[(wave * 64 + lane) * 2] = make_uint4(wd[0], wd[1], wd[2], (uint32_t)(part * kDigitBias));   // |.| < 2^31
        tab[(wave * 64 + lane) * 2 + 1] = make_uint4(__float_as_uint(nb_), __float_as_uint(ninv_), __float_as_uint(nshift_), 0u);
    }
    __syncthreads();
    v4i bw[4];
    int c0[4];
    float nb[4], ninv[4], nshift[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const uint4 t0 = tab[(nt * 64 + lane) * 2], t1 = tab[(nt * 64 + lane) * 2 + 1];
        bw[nt] = __builtin_bit_cast(v4i, make_uint4(t0.x, t0.y, t0.z, 0u));
        c0[nt] = (int)t0.w;
        nb[nt] = __uint_as_float(t1.x); ninv[nt] = __uint_as_float(t1.y); nshift[nt] = __uint_as_float(t1.z);
    }
    LaneEpi ke;
    lane_epi_init<QNN_STORE_I4>(ke, e, r, r);
    constexpr int kMagicBits = 0x4B400008;                       // 1.5 * 2^23 + 8: see qnn_mfma_strip.hip
    const float magic = __int_as_float(kMagicBits);
    const int code_lo = kMagicBits - (int)e.act_m, code_hi = kMagicBits + (int)e.act_m - 1;
    // ---- operand addresses (dword index inside a digit plane): position m = r: window w = r >> 2, (py, px) = bits of r.
    // A step works on conv rows yy0 = 2*rp and yy0 + 1; with rp0 even (the launcher makes every chunk start on an even
    // pair) the ring slot of input row yy0 + py + dy - 1 is (2*(rp & 1) + py + kq) & 3: two lane constants ----
    const int py = (r >> 1) & 1, px = r & 1, w = r >> 2;
    const int ab0 = ((py + kq) & 3) * kRowPitch + 2 * w + px;            // + 8*t, + dx by the 4-dword read
    const int ab1 = ((2 + py + kq) & 3) * kRowPitch + 2 * w + px;
    // ---- staging: two input rows = 108 floats per step, elements lane and lane + 64 (lanes >= 44: a dummy pixel) ----
    const int e0row = lane >= 54 ? 1 : 0, e0rem = lane - 54 * e0row;
    const int e0px = e0rem / 3, e0ch = e0rem - 3 * e0px;
    const bool e1ok = lane < 44;
    const int e1rem = e1ok ? lane + 10 : 0;
    const int e1px = e1rem / 3, e1ch = e1rem - 3 * e1px;
    const int st0 = e0row * (kRowPitch * 4) + e0px * 4 + e0ch;           // + slot * 160 (+ plane * 640)
    const int st1 = e1ok ? kRowPitch * 4 + e1px * 4 + e1ch : kRowPitch * 4 + 30 * 4;     // pixel 30 of a row is never read
    const int rowf = g.W * 3 * 4;                                // bytes per input row
    bool bad = false;                                            // this lane staged a value outside [0, 1]

    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int rp0 = chunk * rc, rp1 = min(rp0 + rc, g.H / 2);        // rc is even or nch == 1: rp0 is even
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)const_cast<float*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        // byte offsets of the two staged elements for input row `row` (added below); columns outside the image and, through
        // the per-image descriptor, rows outside it read 0.0f = X 0
        const int c0col = xs - 1 + e0px, c1col = xs - 1 + e1px;
        const int v0 = (c0col >= 0 && c0col < g.W) ? (c0col * 3 + e0ch) * 4 + e0row * rowf : (int)0x80000000;
        const int v1 = (e1ok && c1col >= 0 && c1col < g.W) ? (c1col * 3 + e1ch) * 4 + rowf : (int)0x80000000;
        auto stage_load = [&](int row, float& f0, float& f1) {   // rows `row`, `row + 1`
            const int so = row * rowf;                           // may be negative: the sum wraps out of range
            f0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, v0 + so, 0, 0));
            f1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, v1 + so, 0, 0));
        };
        // rows `row`, `row + 1` into the ring slots SLOT, SLOT + 1
        auto stage_write = [&](auto slotc, float f0, float f1) {
            constexpr int SB = decltype(slotc)::value * (kRowPitch * 4);
            const uint32_t d0 = to_digits(f0, bad), d1 = to_digits(f1, bad);
            ldsb[st0 + SB] = (uint8_t)d0; ldsb[st0 + SB + kPlane * 4] = (uint8_t)(d0 >> 8);
            ldsb[st0 + SB + 2 * kPlane * 4] = (uint8_t)(d0 >> 16);
            ldsb[st1 + SB] = (uint8_t)d1; ldsb[st1 + SB + kPlane * 4] = (uint8_t)(d1 >> 8);
            ldsb[st1 + SB + 2 * kPlane * 4] = (uint8_t)(d1 >> 16);
        };
        const int yy_first = 2 * rp0;
        float fa0, fa1, fb0, fb1, fc0, fc1;
        stage_load(yy_first - 1, fa0, fa1);
        stage_load(yy_first + 1, fb0, fb1);
        stage_load(yy_first + 3, fc0, fc1);
        stage_write(std::integral_constant<int, 0>{}, fa0, fa1);         // rows yy_first - 1, yy_first: slots 0, 1

        auto step = [&](auto parc, int rp) {
            constexpr int PAR = decltype(parc)::value;           // rp & 1
            const int yy0 = 2 * rp;
            // rows yy0+1, yy0+2 complete the four rows of this step: slots (yy0 + 2) & 3 and the next
            stage_write(std::integral_constant<int, PAR ? 0 : 2>{}, fb0, fb1);
            fb0 = fc0; fb1 = fc1;
            stage_load(yy0 + 5, fc0, fc1);                       // two steps ahead: in flight during two matrix phases
            const uint32_t* abase = lds + (PAR ? ab1 : ab0);
            // 8 groups (tile t = g >> 2, filter block nt = g & 3) of three digit passes; the passes of group g + 1 are
            // issued before group g is combined, so the combine never waits for the matrix pipe
            v4i A[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint32_t* p = abase + j * kPlane + 8 * t;
                    // the 4th dword (the pixel after the three taps) meets zero filter bytes
                    A[t][j] = __builtin_bit_cast(v4i, make_uint4(p[0], p[1], p[2], p[3]));
                }
            v4i acc[2][3];
            int T[8];
            auto issue = [&](int gq, v4i (&a)[3]) {
                const v4i z = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 3; ++j) a[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[gq >> 2][j], bw[gq & 3], z, 0, 0, 0);
            };
            auto combine = [&](int gq, const v4i (&a)[3]) {
                int raw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // two v_lshl_add_u32; the empty asm only stops the reassociation into two shifts and an add3
                    // (real instructions stay visible to the compiler's MFMA -> VALU hazard handling)
                    int hi = (int)(((uint32_t)a[2][i] << 8) + (uint32_t)a[1][i]);
                    asm("" : "+v"(hi));
                    raw[i] = (int)(((uint32_t)hi << 8) + (uint32_t)a[0][i]);
                }
                const int t = gq >> 2, nt = gq & 3;
                if constexpr (POOL == 2) {
                    T[gq] = max(max(raw[0], raw[1]), max(raw[2], raw[3])) + c0[nt];
                } else {
                    // raw layer output (tests): position i of window kq -> conv pixel (yy0 + (i >> 1), xs + 8t + 2kq + (i & 1))
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = nt * 16 + r;
                        float v = __fmul_rn((float)(raw[i] + c0[nt]), vscale);
                        v = qnn_epi_value(v, c, e);
                        if (e.fn == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
                        else if (e.fn == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
                        const long q = ((long)n * g.H + yy0 + (i >> 1)) * g.W + xs + 8 * t + 2 * kq + (i & 1);
                        reinterpret_cast<float*>(y)[q * g.cout + c] = v;
                    }
                }
            };
            issue(0, acc[0]);
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) {
                if (gq + 1 < 8) issue(gq + 1, acc[(gq + 1) & 1]);
                combine(gq, acc[gq & 1]);
            }
            if constexpr (OUT == QNN_STORE_I4) {
                // lane (filter r, window kq): value j = 4*t + nt -> after the transpose lane (r & 7) holds the word of
                // value j = r & 7: pooled pixel (xs/2 + kq + 4*(j >> 2)), channels (j & 3)*16 + (r & 8) .. +7
                int cb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float u = __fadd_rn(__fmul_rn(__fadd_rn((float)T[j], nb[j & 3]), ninv[j & 3]), nshift[j & 3]);
                    if constexpr (BIN) {
                        cb[j] = u > 0x1p-24f ? kMagicBits + 1 : kMagicBits - 1;
                    } else {
                        // rint + clamp in one add and one integer median (the low bits of u + magic are rint(u) + 8)
                        const int bits = __float_as_int(__fadd_rn(u, magic));
                        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cb[j]) : "v"(bits), "v"(code_lo), "v"(code_hi));
                    }
                }
                // cb[j] = kMagicBits - 8 + (code + 8): the sum of the shifted words minus the shifted constants (mod 2^32)
                // leaves the eight offset nibbles
                uint32_t P = (uint32_t)cb[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) P += (uint32_t)cb[j] << (4 * j);
                P -= (uint32_t)(kMagicBits - 8) * 0x11111111u;
                const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                const int j = r & 7;
                const long q = ((long)n * g.Hp + rp) * g.Wp + (xs >> 1) + kq + 4 * (j >> 2);
                reinterpret_cast<uint32_t*>(y)[q * e.ocw + (j & 3) * 2 + (r >> 3)] = Wd;
            }
        };
        int rp = rp0;
        for (; rp + 2 <= rp1; rp += 2) {
            step(std::integral_constant<int, 0>{}, rp);
            step(std::integral_constant<int, 1>{}, rp + 1);
        }
        if (rp < rp1) step(std::integral_constant<int, 0>{}, rp);
    }
    // The domain is part of this kernel's contract: a value outside [0, 1] was clamped above, so the outputs it reaches
    // are not the layer's.  Raise the layer's flag (a host-visible word inside the weights handle) instead of staying
    // silent; qnn_weights_check() / the next qnn_conv2d_forward turn it into QNN_EINVAL.
    if (bad) *domain_flag = 1u;
}

}  // namespace

// 0 = launched.  3x3, stride 1, SAME, 3 input channels, 64 filters of <= 4 bits (or binary), W % 16 == 0, H even;
// fused pipeline form (pool 2, int4 codes out) or raw float32 output without pooling / activation.
int qnn_try_launch_first_fixed(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w, void* y,
                               hipStream_t s) {
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.pt != 1 || g.pl != 1 || g.cin != 3 || g.cout != 64 || e.res) return 1;
    if ((g.W % 16) != 0 || (g.H % 2) != 0 || !w->d_wq || !(e.dom_flag ? e.dom_flag : w->d_flag)) return 1;    // no flag word, no restricted-domain kernel
    // weight codes = value * 2^wshift: binary (+-1, H = 1) or quantized to <= 4 bits (|code| <= 8)
    if (w->wkind == QNN_W_BINARY ? (w->H != 1.0f || w->wshift != 0)
                               : (w->wkind != QNN_W_QUANT || w->wshift < 1 || w->wshift > 3)) return 1;
    const bool fused = g.pool == 2 && e.out_store == QNN_STORE_I4 &&
                       ((e.fn == QNN_FN_QUANTIZED_TANH && e.act_m <= 8.0f) || e.fn == QNN_FN_BINARY_TANH);
    const bool rawf = g.pool == 1 && e.out_store == QNN_STORE_F32 &&
                      (e.fn == QNN_FN_NONE || e.fn == QNN_FN_QUANTIZED_TANH || e.fn == QNN_FN_BINARY_TANH);
    if (!fused && !rawf) return 1;
    const float wscale = (float)(1 << w->wshift);
    const float vscale = 1.0f / (8388608.0f * wscale);           // 2^-(23 + wshift)
    const int spr = g.W / 16;
    const double img_x = (double)g.H * g.W * 3 * 4.0;
    if (img_x >= 1.0e9 || (double)g.N * g.Hp * g.Wp * 8.0 >= 2.0e9 * 4) return 1;
    const int hp2 = g.H / 2;
    static const int bpc = QNN_ENV_INT("QNN_FIXED_BPC", 4);   // A/B switch (experiment builds only)
    const int blocks_cap = 256 * (bpc >= 1 && bpc <= 5 ? bpc : 4);
    const long nwaves = (long)blocks_cap * 4;
    int best_rc = hp2, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = 2; rc <= hp2 + 1; rc += 2) {              // even: every chunk starts on an even row pair
        const int nch = (hp2 + rc - 1) / rc;
        const long rounds = ((long)g.N * spr * nch + nwaves - 1) / nwaves;
        const double cost = (double)rounds * (rc + 1.5);
        if (cost < best_cost) { best_cost = cost; best_rc = rc; best_nch = nch; }
    }
    const long ntasks_l = (long)g.N * spr * best_nch;
    if (ntasks_l >= 2000000000L) return 1;
    long blocks = (ntasks_l + 3) / 4;
    if (blocks > blocks_cap) blocks = blocks_cap;
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = (size_t)4 * kWaveLds * 4 + 4 * 64 * 2 * 16;
#define FIXED_LAUNCH(OUT_, POOL_, BIN_)                                                                                   \
    hipLaunchKernelGGL((k_conv_first_fixed<OUT_, POOL_, BIN_>), grid, block, lds, s, g, e, (const float*)x, w->d_wq, y,    \
                       (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc, \
                       (uint32_t)img_x, wscale, vscale, 1.0f / vscale, (e.dom_flag ? e.dom_flag : w->d_flag))
    if (!fused) FIXED_LAUNCH(QNN_STORE_F32, 1, false);
    else if (e.fn == QNN_FN_BINARY_TANH) FIXED_LAUNCH(QNN_STORE_I4, 2, true);
    else FIXED_LAUNCH(QNN_STORE_I4, 2, false);
#undef FIXED_LAUNCH
    return 0;
}
