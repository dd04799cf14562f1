// Float-input 3x3 stride-1 layer with FEW output channels (the stem of models/resnet.py:104: 3 -> 16 at the input
// resolution) on v_mfma_f32_16x16x4_f32, packed int4 output.  Dispatch: qnn_conv2d_forward (qnn_conv.hip).
//
// Same arithmetic as the first-layer kernels of qnn_first.hip: per output value an exact float32 FMA chain over
// k = (dy, dx, c) in ascending order (the MFMA's documented internal order; out-of-image taps contribute fma(w, 0, acc)),
// then bias / BN / activation in the reference's op order.  Same structure as the strip kernel (qnn_mfma_strip.hip):
// a wave walks a 16-pixel column strip of one image, filters are the A operand (rows = output channels) so a lane ends
// up with four consecutive channels of one pixel = one 2-byte store; per-image buffer descriptors make the padding free.
// The pixel-stationary VALU kernel this replaces needs 88 us for 64 x 224^2 x 3 -> 16; the f32 matrix pipe does the
// 7 K-steps of a row segment in 224 cycles.
#include "qnn_mfma_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

template <int CIN, bool BIAS>
__global__ __launch_bounds__(256, 4) void k_conv_stem(ConvGeom g, EpiArgs e, const float* __restrict__ x,
                                                       const float* __restrict__ wq, void* __restrict__ y,
                                                       int ntasks, int spr, FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                       int rc, uint32_t img_x, uint32_t img_y) {
    constexpr int K = 9 * CIN;
    constexpr int KS = (K + 3) / 4;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int nbase = blockIdx.y * 16;

    // A operand: filter nbase + r, k = 4*s + kq;  B operand: the same k of pixel r
    float aw[KS];
    int kdx[KS], koff[KS];            // tap column, float offset of (dy, dx, c) relative to (row yy-1, pixel xs+r-1)
    bool kok[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + kq;
        kok[s] = k < K;
        const int kk = kok[s] ? k : 0;
        const int tap = kk / CIN, ch = kk - tap * CIN;
        kdx[s] = tap % 3;
        koff[s] = ((tap / 3) * g.W + tap % 3) * CIN + ch;
        aw[s] = kok[s] ? wq[(size_t)(nbase + r) * K + k] : 0.0f;
    }
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float cfold = binary ? 1.0f : e.act_m;
    float nb[4], ninv[4], nshift[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = nbase + 4 * kq + i;
        nb[i] = BIAS ? e.bias[c] : 0.0f;
        ninv[i] = __fmul_rn(e.bn_inv ? e.bn_inv[c] : 1.0f, cfold);
        nshift[i] = __fmul_rn(e.bn_inv ? e.bn_shift[c] : 0.0f, cfold);
    }
    constexpr float kMagic = 12582920.0f;              // 1.5 * 2^23 + 8: see qnn_mfma_strip.hip
    constexpr int kMagicBits = 0x4B400008;
    const int code_lo = kMagicBits - (int)e.act_m, code_hi = kMagicBits + (int)e.act_m - 1;
    const int rowb = g.W * CIN * 4;
    const int orowb = g.W * e.ocw * 4;

    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int y0 = chunk * rc;
        const int y1 = min(y0 + rc, g.H);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)const_cast<float*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)y + (size_t)n * img_y, 0, (int)img_y, 0x00020000);
        // byte offset of this lane's operand of K-step s for output row y0 (rows above / below the image are out of range
        // by themselves; pixels left / right of it get an out-of-range offset that stays out of range under the row steps)
        int voff[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int px = xs + r + kdx[s] - 1;
            voff[s] = (kok[s] && px >= 0 && px < g.W) ? (((y0 - 1) * g.W + xs + r - 1) * CIN + koff[s]) * 4
                                                      : (int)0x80000000;
        }
        const bool pvalid = xs + r < g.W;
        int ovoff = pvalid ? (y0 * g.W + xs + r) * e.ocw * 4 + nbase / 2 + kq * 2 : (int)0x80000000;

        float xb[2][KS];
        auto load_row = [&](float (&dst)[KS]) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                dst[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, voff[s], 0, 0));
                voff[s] += rowb;
            }
        };
        load_row(xb[0]);                               // operands of output row y0
        auto body = [&](float (&cur)[KS], float (&nxt)[KS]) {
            load_row(nxt);                             // operands of the next output row
            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], cur[s], acc, 0, 0, 0);
            int cb[4];
            float u4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[i];
                if constexpr (BIAS) v = __fadd_rn(v, nb[i]);
                u4[i] = __fadd_rn(__fmul_rn(v, ninv[i]), nshift[i]);
            }
            if (binary) {
                asm volatile("; binary_tanh codes");
#pragma unroll
                for (int i = 0; i < 4; ++i) cb[i] = u4[i] > 0x1p-24f ? kMagicBits + 1 : kMagicBits - 1;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bits = __float_as_int(__fadd_rn(u4[i], kMagic));
                    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cb[i]) : "v"(bits), "v"(code_lo), "v"(code_hi));
                }
            }
            uint32_t P = ((uint32_t)cb[1] << 4) | (uint32_t)cb[0];
            P = ((uint32_t)cb[2] << 8) | P;
            P = ((uint32_t)cb[3] << 12) | P;
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(P ^ 0x8888u), yr, ovoff, 0, 0);
            ovoff += orowb;
        };
        int yy = y0;
        for (; yy + 2 <= y1; yy += 2) {
            body(xb[0], xb[1]);
            body(xb[1], xb[0]);
        }
        if (yy < y1) body(xb[0], xb[1]);
    }
}

}  // namespace

// 0 = launched.  Eligibility (3x3, stride 1, SAME, cin 1 or 3, cout a multiple of 16 below 64, packed int4 output,
// quantized or binary activation, no pooling, no residual) is checked here.
int qnn_try_launch_stem(const ConvGeom& g, const EpiArgs& e, const void* x, const float* wq, void* y, hipStream_t s) {
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.pt != 1 || g.pl != 1 || g.pool != 1 || e.res) return 1;
    if ((g.cin != 1 && g.cin != 3) || (g.cout % 16) != 0 || g.cout >= 64 || !wq) return 1;
    if (e.out_store != QNN_STORE_I4 || (e.fn != QNN_FN_QUANTIZED_TANH && e.fn != QNN_FN_BINARY_TANH)) return 1;
    if (e.fn == QNN_FN_QUANTIZED_TANH && e.act_m > 8.0f) return 1;
    const int spr = (g.W + 15) / 16;
    const double img_x = (double)g.H * g.W * g.cin * 4.0, img_y = (double)g.H * g.W * e.ocw * 4.0;
    if (img_x >= 1.0e9 || img_y >= 1.0e9) return 1;
    const int ny = g.cout / 16;
    const int blocks_cap = 256 * 4 / ny > 0 ? 256 * 4 / ny : 1;
    const long nwaves = (long)blocks_cap * 4;
    int best_rc = g.H, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = g.H < 4 ? g.H : 4; rc <= g.H; ++rc) {
        const int nch = (g.H + rc - 1) / rc;
        const long rounds = ((long)g.N * spr * nch + nwaves - 1) / nwaves;
        const double cost = (double)rounds * (rc + 2);
        if (cost < best_cost) { best_cost = cost; best_rc = rc; best_nch = nch; }
    }
    const long ntasks_l = (long)g.N * spr * best_nch;
    if (ntasks_l >= 2000000000L) return 1;
    long blocks = (ntasks_l + 3) / 4;
    if (blocks > blocks_cap) blocks = blocks_cap;
    const dim3 grid((unsigned)blocks, (unsigned)ny), block(256);
#define STEM_CASE(CIN_, BIAS_)                                                                                       \
    if (g.cin == CIN_ && (e.bias != nullptr) == BIAS_) {                                                             \
        hipLaunchKernelGGL((k_conv_stem<CIN_, BIAS_>), grid, block, 0, s, g, e, (const float*)x, wq, y, (int)ntasks_l, \
                           spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,      \
                           (uint32_t)img_x, (uint32_t)img_y);                                                        \
        return 0;                                                                                                    \
    }
    STEM_CASE(3, false) STEM_CASE(3, true) STEM_CASE(1, false) STEM_CASE(1, true)
#undef STEM_CASE
    return 1;
}
