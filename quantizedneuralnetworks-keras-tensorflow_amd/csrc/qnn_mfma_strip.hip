// 3x3 stride-1 int4 layers with 16 or 32 channels (the 224x224 and 112x112 stages of the ImageNet ResNet,
// models/resnet.py:104-129) as a ROW-WALKING kernel on v_mfma_i32_16x16x64_i8.
// Dispatch: qnn_try_launch_mfma (qnn_mfma.hip).
//
// These layers are bound by instruction issue, not by the matrix pipe (16 x 16 outputs need 3 or 12 MFMAs of 16
// cycles) and not by HBM: QuantizedConv2D.call (quantized_layers.py:164-194) + BN + residual merge + quantized_tanh
// is ~11 float32 VALU operations per output value whatever the kernel does.  So everything else is removed:
//
//   * A wave owns a 16-pixel-wide column strip of ONE image and walks down its rows.  The widened operands of an
//     input row are built once and used by the three output rows that need it (register rotation, loop unrolled by
//     three): per output row one 8-byte load per lane and K-step instead of nine per tile.
//   * Operand roles are swapped: A = filters (rows = output channels), B = pixels (columns).  In the C/D layout a
//     lane then holds FOUR CONSECUTIVE CHANNELS OF ONE PIXEL -- 16 bits of the packed output word: the lane stores
//     them with one 2-byte store (64 lanes = one full 128-byte line) and fetches its shortcut codes with one 2-byte
//     load.  No nibble transposes, no cross-lane traffic at all.
//   * Zero padding costs nothing inside the loop: every tensor is addressed through a buffer descriptor of ONE image,
//     so the rows above the first and below the last are out of range (raw buffer loads return 0); the left / right
//     edge lanes of the first / last strip get an out-of-range offset once per strip.  No masks, no scalar decode per
//     tile (the previous kernel spent ~88 scalar and ~84 vector instructions per 16 pixels; this one ~50 vector).
//
// K order: one K-step (64 deep = four 16-byte k-blocks) per input row and 16-channel group pair:
//   Cin 16: k-block kq = tap dx (kq = 3: zero filter);  Cin 32: step 0 = (dx 0 lo, dx 0 hi, dx 1 lo, dx 1 hi),
//   step 1 = (dx 2 lo, dx 2 hi, zero, zero).  int4 codes are widened to code*16 in both operands (x256 folded
//   into the power-of-two output scale), exactly as in qnn_mfma_small.hip.
#include "qnn_mfma_common.h"

#ifndef QNN_STRIP16_WPS
#define QNN_STRIP16_WPS 6        // waves per SIMD (= persistent workgroups per CU): Cin 16 needs 51-67 VGPRs
#endif
#ifndef QNN_STRIP32_WPS
#define QNN_STRIP32_WPS 3        // Cin 32: 118-128 VGPRs, the float32-shortcut variant spilled at 4 per SIMD
#endif

namespace {

template <int CIN, int NT, int RES, bool BIAS>   // RES: 0 none, 1 packed int4 shortcut, 2 float32 shortcut
__global__ __launch_bounds__(256, (CIN == 16 ? QNN_STRIP16_WPS : QNN_STRIP32_WPS)) void k_conv_strip(MfmaGeom mg, EpiArgs e,
                                                                 const uint8_t* __restrict__ x,
                                                                 const uint8_t* __restrict__ wq8,
                                                                 void* __restrict__ y, int ntasks, int spr,
                                                                 FastDiv fd_spr, int nch, FastDiv fd_nch, int rc,
                                                                 uint32_t img_x, uint32_t img_y, uint32_t img_r) {
    constexpr int ST = CIN / 16;                      // K-steps per input row
    constexpr int PIXB = CIN / 2;                     // bytes per stored input pixel
    const ConvGeom& g = mg.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;

    // ---- k-block of this lane in every K-step ----
    int dxs[ST], hbs[ST];
    bool kok[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        if constexpr (CIN == 16) { dxs[st] = kq; hbs[st] = 0; kok[st] = kq < 3; }
        else if (st == 0) { dxs[st] = kq >> 1; hbs[st] = kq & 1; kok[st] = true; }
        else { dxs[st] = 2; hbs[st] = kq & 1; kok[st] = kq < 2; }
        if (!kok[st]) dxs[st] = 1;
    }
    // ---- filters: A operand, row = output channel nt*16 + r ----
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);
    v4i bw[3][ST][NT];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int woff = kok[st] ? ((nt * 16 + r) * 9 + dy * 3 + dxs[st]) * CIN + hbs[st] * 16 : (int)0x80000000;
                bw[dy][st][nt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, 0, 0));
            }
    // ---- epilogue constants: this lane's channels are nt*16 + 4*kq + i ----
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float mfold = (!binary && RES == 0) ? e.act_m : 1.0f;
    const float clate = RES != 0 ? e.post_scale * (binary ? 1.0f : e.act_m) : 1.0f;   // both powers of two (host check)
    float nb[NT][4], ninv[NT][4], nshift[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = nt * 16 + 4 * kq + i;
            const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
            const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
            nb[nt][i] = BIAS ? __fdiv_rn(e.bias[c], e.scale) : 0.0f;
            ninv[nt][i] = __fmul_rn(__fmul_rn(inv, e.scale), mfold);
            nshift[nt][i] = __fmul_rn(shift, mfold);
        }
    const int rowb = g.W * PIXB;                      // bytes per input row
    const int orowb = g.W * e.ocw * 4;                // bytes per output row
    const int rrowb = RES == 2 ? g.W * g.cout * 4 : orowb;

    auto widen = [&](const uint2& q) -> v4i {
        const uint4 v = make_uint4((q.x << 4) & 0xF0F0F0F0u, q.x & 0xF0F0F0F0u,
                                   (q.y << 4) & 0xF0F0F0F0u, q.y & 0xF0F0F0F0u);
        return __builtin_bit_cast(v4i, v);
    };

    for (int task = wid; task < ntasks; task += nw) {
        // ---- task = (image, strip, row chunk): scalar decode, once per ~rc rows ----
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int y0 = chunk * rc;
        const int y1 = min(y0 + rc, g.H);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)y + (size_t)n * img_y, 0, (int)img_y, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            RES != 0 ? (uint8_t*)const_cast<void*>(e.res) + (size_t)n * img_r : (uint8_t*)y, 0,
            RES != 0 ? (int)img_r : 0, 0x00020000);
        // byte offset (inside the image) of this lane's k-block in input row y0 - 1; the left / right edge lanes of the
        // first / last strip start out of range and stay there (0x80000000 + row increments < 2^32)
        int voff[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const bool edge = (xs == 0 && r == 0 && dxs[st] == 0) || (xs + 16 == g.W && r == 15 && dxs[st] == 2);
            voff[st] = (kok[st] && !edge) ? ((y0 - 1) * g.W + xs + r + dxs[st] - 1) * PIXB + hbs[st] * 8
                                          : (int)0x80000000;
        }
        int ovoff = (y0 * g.W + xs + r) * e.ocw * 4 + kq * 2;                   // + nt*8
        int rvoff = RES == 2 ? ((y0 * g.W + xs + r) * g.cout + 4 * kq) * 4 : ovoff;   // + nt*64 (f32) / nt*8

        v4i X[3][ST];
        uint2 raw[ST];
        auto load_row = [&]() {
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                raw[st] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xr, voff[st], 0, 0));
                voff[st] += rowb;
            }
        };
        load_row();                                    // row y0 - 1
#pragma unroll
        for (int st = 0; st < ST; ++st) X[0][st] = widen(raw[st]);
        load_row();                                    // row y0
#pragma unroll
        for (int st = 0; st < ST; ++st) X[1][st] = widen(raw[st]);
        load_row();                                    // row y0 + 1 stays in `raw`

        // one output row: slots a / b / c hold input rows yy-1 / yy / yy+1
        auto body = [&](v4i (&Xa)[ST], v4i (&Xb)[ST], v4i (&Xc)[ST]) {
#pragma unroll
            for (int st = 0; st < ST; ++st) Xc[st] = widen(raw[st]);
            load_row();                                // row yy + 2 for the next iteration
            // shortcut of this output row, requested before the matrix phase
            uint32_t rs[NT];
            float4 rf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (RES == 1) rs[nt] = __builtin_amdgcn_raw_buffer_load_b16(rr, rvoff + 8 * nt, 0, 0);
                if constexpr (RES == 2)
                    rf[nt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, rvoff + 64 * nt, 0, 0));
            }
            v4i acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const v4i z = {0, 0, 0, 0};
                acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[0][0][nt], Xa[0], z, 0, 0, 0);
#pragma unroll
                for (int st = 1; st < ST; ++st)
                    acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[0][st][nt], Xa[st], acc[nt], 0, 0, 0);
#pragma unroll
                for (int st = 0; st < ST; ++st)
                    acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[1][st][nt], Xb[st], acc[nt], 0, 0, 0);
#pragma unroll
                for (int st = 0; st < ST; ++st)
                    acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[2][st][nt], Xc[st], acc[nt], 0, 0, 0);
            }
            // ---- epilogue: reference op order, one rounding per operation ----
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float t4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = (float)acc[nt][i];
                    if constexpr (BIAS) v = __fadd_rn(v, nb[nt][i]);
                    float t = __fadd_rn(__fmul_rn(v, ninv[nt][i]), nshift[nt][i]);
                    if constexpr (RES == 1) {
                        const int code = (int)(rs[nt] << (28 - 4 * i)) >> 28;
                        // shortcut value = code * 2^-(bits-1), exact: fma(code, scale, t) IS the reference's x + y
                        t = __fmul_rn(__fmaf_rn((float)code, e.res_scale, t), clate);
                    }
                    if constexpr (RES == 2) {
                        const float rv = i == 0 ? rf[nt].x : i == 1 ? rf[nt].y : i == 2 ? rf[nt].z : rf[nt].w;
                        t = __fmul_rn(__fadd_rn(rv, t), clate);
                    }
                    t4[i] = t;
                }
                const uint32_t P = pack_scaled<4, 4>(t4, e.act_m, binary) ^ 0x8888u;
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)P, yr, ovoff + 8 * nt, 0, 0);
            }
            ovoff += orowb;
            rvoff += rrowb;
        };
        int yy = y0;
        for (; yy + 3 <= y1; yy += 3) {
            body(X[0], X[1], X[2]);
            body(X[1], X[2], X[0]);
            body(X[2], X[0], X[1]);
        }
        if (yy < y1) {
            body(X[0], X[1], X[2]);
            if (yy + 1 < y1) body(X[1], X[2], X[0]);
        }
    }
}

template <int CIN, int NT>
int launch_strip(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s) {
    const ConvGeom& g = mg.g;
    const int spr = g.W / 16;
    const double img_x = (double)g.H * g.W * (CIN / 2), img_y = (double)g.H * g.W * e.ocw * 4.0;
    const int res = !e.res ? 0 : e.res_store == QNN_STORE_F32 ? 2 : 1;
    const double img_r = res == 2 ? (double)g.H * g.W * g.cout * 4.0 : img_y;
    if (img_x >= 1.0e9 || img_y >= 1.0e9 || img_r >= 1.0e9) return 1;
    // persistent grid: WPS waves per SIMD; rows per task chosen so that the task count fills whole
    // rounds of that grid (a round costs rc output rows + 3 rows of pipeline fill)
    const int blocks_cap = 256 * (CIN == 16 ? QNN_STRIP16_WPS : QNN_STRIP32_WPS);
    const long nwaves = (long)blocks_cap * 4;
    int best_rc = g.H, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = 6; rc <= g.H; ++rc) {
        const int nch = (g.H + rc - 1) / rc;
        const long tasks = (long)g.N * spr * nch;
        const long rounds = (tasks + nwaves - 1) / nwaves;
        const double cost = (double)rounds * (rc + 3);
        if (cost < best_cost) { best_cost = cost; best_rc = rc; best_nch = nch; }
    }
    const long ntasks_l = (long)g.N * spr * best_nch;
    if (ntasks_l >= 2000000000L) return 1;
    const int ntasks = (int)ntasks_l;
    long blocks = (ntasks + 3) / 4;
    if (blocks > blocks_cap) blocks = blocks_cap;
    const dim3 grid((unsigned)blocks), block(256);
    const bool bias = e.bias != nullptr;
#define STRIP_CASE(RES_, BIAS_)                                                                               \
    if (res == RES_ && bias == BIAS_) {                                                                       \
        hipLaunchKernelGGL((k_conv_strip<CIN, NT, RES_, BIAS_>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, \
                           ntasks, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),   \
                           best_rc, (uint32_t)img_x, (uint32_t)img_y, (uint32_t)img_r);                        \
        return 0;                                                                                             \
    }
    STRIP_CASE(0, false) STRIP_CASE(0, true) STRIP_CASE(1, false) STRIP_CASE(1, true)
    STRIP_CASE(2, false) STRIP_CASE(2, true)
#undef STRIP_CASE
    return 1;
}

}  // namespace

// cin == cout in {16, 32}; eligibility is checked by the caller
int qnn_launch_strip(int cin, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                     void* y, hipStream_t s) {
    return cin == 16 ? launch_strip<16, 1>(mg, e, x, w, y, s) : launch_strip<32, 2>(mg, e, x, w, y, s);
}
