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
#include "qnn_mfma_common.h"

namespace {

constexpr int kRowPitch = 40;                 // words per LDS row (18 pixels used; 40 == 8 mod 32)
constexpr int kPlane = 4 * kRowPitch;         // words per digit plane (ring of 4 rows)
constexpr int kWaveLds = 3 * kPlane;          // words per wave

// x -> the three offset digits of X = rint(x * 2^23) as one word: byte j = (digit j of X) - 128 as a signed byte
__device__ __forceinline__ uint32_t to_digits(float x) {
    const int X = (int)rintf(__fmul_rn(x, 8388608.0f));
    return (uint32_t)min(max(X, 0), 8388608) ^ 0x00808080u;       // inputs outside [0, 1] saturate
}
constexpr int kDigitBias = 128 * (1 + 256 + 65536);              // sum_j 128 * 2^(8j)

template <int OUT, int POOL>      // (QNN_STORE_I4, 2): the fused pipeline;  (QNN_STORE_F32, 1): the raw layer (tests)
__global__ __launch_bounds__(256, 4) void k_conv_first_fixed(ConvGeom g, EpiArgs e, const float* __restrict__ x,
                                                              const float* __restrict__ wq, void* __restrict__ y,
                                                              int ntasks, int spr, FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                              int rc, uint32_t img_x, float wscale, float vscale) {
    extern __shared__ __attribute__((aligned(16))) char smem_fx[];
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem_fx) + wave * kWaveLds;
    uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
    for (int i = lane; i < kWaveLds; i += 64) lds[i] = 0;        // the 4th byte of every pixel stays zero

    // ---- filters: B operand of block nt = filter nt*16 + r, k-block kq = tap row dy ----
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float mfold = (OUT == QNN_STORE_I4 && !binary) ? e.act_m : 1.0f;
    v4i bw[4];
    int c0[4];
    float nb[4], ninv[4], nshift[4];
    LaneEpi ke;
    lane_epi_init<QNN_STORE_I4>(ke, e, r, r);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = nt * 16 + r;
        const float bias = e.bias ? e.bias[c] : 0.0f;
        const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
        const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
        const bool flip = POOL == 2 && inv < 0.0f;               // pool with max only: negate the filter, fold the sign
        int sum = 0;
        uint32_t wd[4] = {0u, 0u, 0u, 0u};
        for (int k = 0; k < 27; ++k) {
            int code = (int)rintf(__fmul_rn(wq[(size_t)c * 27 + k], wscale));
            if (flip) code = -code;
            sum += code;
            const int dy = k / 9, dx = (k % 9) / 3, ch = k % 3;
            if (dy == kq) wd[dx] |= (uint32_t)(code & 0xFF) << (8 * ch);
        }
        bw[nt] = kq < 3 ? __builtin_bit_cast(v4i, make_uint4(wd[0], wd[1], wd[2], 0u)) : v4i{0, 0, 0, 0};
        c0[nt] = sum * kDigitBias;                               // |.| <= 27 * 8 * 8421504 < 2^31
        nb[nt] = __fdiv_rn(flip ? -bias : bias, vscale);
        ninv[nt] = __fmul_rn(__fmul_rn(flip ? -inv : inv, vscale), mfold);
        nshift[nt] = __fmul_rn(shift, mfold);
    }
    // ---- operand addresses (dword index inside a digit plane): position m = r: window w = r >> 2, (py, px) = bits of r ----
    const int py = (r >> 1) & 1, px = r & 1, w = r >> 2;
    const int acol = 2 * w + px;                                 // + 8*t + dx
    // ---- staging: two input rows = 108 floats per step, elements lane and lane + 64 ----
    const int e0row = lane >= 54 ? 1 : 0, e0rem = lane - 54 * e0row;
    const int e0px = e0rem / 3, e0ch = e0rem - 3 * e0px;
    const bool e1ok = lane < 44;
    const int e1rem = e1ok ? lane + 10 : 0;
    const int e1px = e1rem / 3, e1ch = e1rem - 3 * e1px;
    const int rowf = g.W * 3 * 4;                                // bytes per input row

    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int rp0 = chunk * rc, rp1 = min(rp0 + rc, g.H / 2);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)const_cast<float*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        // byte offsets of the two staged elements for input row `row` (added below); columns outside the image and, through
        // the per-image descriptor, rows outside it read 0.0f = X 0
        const int c0col = xs - 1 + e0px, c1col = xs - 1 + e1px;
        const int v0 = (c0col >= 0 && c0col < g.W) ? (c0col * 3 + e0ch) * 4 + e0row * rowf : (int)0x80000000;
        const int v1 = (e1ok && c1col >= 0 && c1col < g.W) ? (c1col * 3 + e1ch) * 4 + rowf : (int)0x80000000;
        float f0, f1;
        auto stage_load = [&](int row) {                         // rows `row`, `row + 1`
            const int so = row * rowf;                           // may be negative: the sum wraps out of range
            f0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, v0 + so, 0, 0));
            f1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, v1 + so, 0, 0));
        };
        auto stage_write = [&](int row) {
            const int s0 = ((row + 1 + e0row) & 3) * (kRowPitch * 4), s1 = ((row + 2) & 3) * (kRowPitch * 4);
            const uint32_t d0 = to_digits(f0);
            const int b0 = s0 + e0px * 4 + e0ch;
            ldsb[b0] = (uint8_t)d0; ldsb[b0 + kPlane * 4] = (uint8_t)(d0 >> 8); ldsb[b0 + 2 * kPlane * 4] = (uint8_t)(d0 >> 16);
            if (e1ok) {
                const uint32_t d1 = to_digits(f1);
                const int b1 = s1 + e1px * 4 + e1ch;
                ldsb[b1] = (uint8_t)d1; ldsb[b1 + kPlane * 4] = (uint8_t)(d1 >> 8); ldsb[b1 + 2 * kPlane * 4] = (uint8_t)(d1 >> 16);
            }
        };
        const int yy_first = 2 * rp0;
        stage_load(yy_first - 1);
        stage_write(yy_first - 1);
        stage_load(yy_first + 1);
        for (int rp = rp0; rp < rp1; ++rp) {
            const int yy0 = 2 * rp;
            stage_write(yy0 + 1);                                // rows yy0+1, yy0+2 complete the four rows of this step
            stage_load(yy0 + 3);                                 // next step's rows, in flight during the matrix phase
            // input row of this lane's operand: yy0 + py + dy - 1 -> ring slot (yy0 + py + kq) & 3
            const int abase = ((yy0 + py + kq) & 3) * kRowPitch + acol;
            int T[2][4];                                         // [tile][filter block]: pooled sums
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v4i A[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint32_t* p = lds + j * kPlane + abase + 8 * t;
                    A[j] = __builtin_bit_cast(v4i, make_uint4(p[0], p[1], p[2], 0u));   // k-block 3: times zero filters
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const v4i z = {0, 0, 0, 0};
                    const v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], bw[nt], z, 0, 0, 0);
                    const v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], bw[nt], z, 0, 0, 0);
                    const v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], bw[nt], z, 0, 0, 0);
                    int raw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // two v_lshl_add_u32; the empty asm only stops the reassociation into two shifts and an add3
                        // (real instructions stay visible to the compiler's MFMA -> VALU hazard handling)
                        int hi = (int)(((uint32_t)a2[i] << 8) + (uint32_t)a1[i]);
                        asm("" : "+v"(hi));
                        raw[i] = (int)(((uint32_t)hi << 8) + (uint32_t)a0[i]);
                    }
                    if constexpr (POOL == 2) {
                        T[t][nt] = max(max(raw[0], raw[1]), max(raw[2], raw[3])) + c0[nt];
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
                }
            }
            if constexpr (OUT == QNN_STORE_I4) {
                // lane (filter r, window kq): 8 values j = 4*t + nt -> after the transpose lane (r & 7) holds the word of
                // value j = r & 7: pooled pixel (xs/2 + kq + 4*(j >> 2)), channels (j & 3)*16 + (r & 8) .. +7
                float tv[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        tv[t * 4 + nt] = __fadd_rn(__fmul_rn(__fadd_rn((float)T[t][nt], nb[nt]), ninv[nt]), nshift[nt]);
                const uint32_t P = pack_scaled<4, 8>(tv, e.act_m, binary);
                const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                const int j = r & 7;
                const long q = ((long)n * g.Hp + rp) * g.Wp + (xs >> 1) + kq + 4 * (j >> 2);
                reinterpret_cast<uint32_t*>(y)[q * e.ocw + (j & 3) * 2 + (r >> 3)] = Wd;
            }
        }
    }
}

}  // namespace

// 0 = launched.  3x3, stride 1, SAME, 3 input channels, 64 filters of <= 4 bits (or binary), W % 16 == 0, H even;
// fused pipeline form (pool 2, int4 codes out) or raw float32 output without pooling / activation.
int qnn_try_launch_first_fixed(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w, void* y,
                               hipStream_t s) {
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.pt != 1 || g.pl != 1 || g.cin != 3 || g.cout != 64 || e.res) return 1;
    if ((g.W % 16) != 0 || (g.H % 2) != 0 || !w->d_wq) return 1;
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
    const int blocks_cap = 256 * 4;
    const long nwaves = (long)blocks_cap * 4;
    int best_rc = hp2, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = 1; rc <= hp2; ++rc) {
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
    const size_t lds = (size_t)4 * kWaveLds * 4;
    if (fused)
        hipLaunchKernelGGL((k_conv_first_fixed<QNN_STORE_I4, 2>), grid, block, lds, s, g, e, (const float*)x, w->d_wq, y,
                           (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,
                           (uint32_t)img_x, wscale, vscale);
    else
        hipLaunchKernelGGL((k_conv_first_fixed<QNN_STORE_F32, 1>), grid, block, lds, s, g, e, (const float*)x, w->d_wq, y,
                           (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,
                           (uint32_t)img_x, wscale, vscale);
    return 0;
}
