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
// K order: an input row contributes 3*BP sixteen-channel k-blocks (BP = Cin/16 blocks per pixel), block j = (tap
// dx = j / BP, channel group j % BP); K-step st (64 deep = four k-blocks) takes j = 4*st + kq:
//   Cin 16: one step, kq = dx (kq = 3: zero filter);  Cin 32: (dx0 lo, dx0 hi, dx1 lo, dx1 hi), (dx2 lo, dx2 hi, 0, 0);
//   Cin 64: one step per tap.  int4 codes are widened to code*16 in both operands (x256 folded into the power-of-two
//   output scale), exactly as in qnn_mfma_small.hip.
// (Round 3 measured the other cut -- lane group kq owns tap dx = kq and fetches its whole pixel with 16-byte loads, BP
// K-steps three quarters full -- on the stride-1 kernel: Cin 32 unchanged (18.7 / 16.4 us), Cin 64 slower (16.6 against
// 14.8 us: a fourth K-step and 30 more registers); the stride-2 kernel below keeps it, 14.0 -> 12.2 us at Cin 32.)
// Filter rows are dealt to the NT channel tiles so that a lane's 4*NT results are CONSECUTIVE channels (tile nt, row
// 4*kq + i  <->  channel nbase + 4*NT*kq + 4*nt + i): with NT = 2 the two 16-bit fields of a lane are one 32-bit word,
// i.e. ONE store and ONE shortcut load per row instead of two each (round 3: the 32- and 64-channel layers ran with the
// texture-address FIFOs full, SQ_VMEM_TA_ADDR_FIFO_FULL 2.4-2.9 M against 0.3 M on the 16-channel layers).
// A wave computes 16*NT output channels (blockIdx.y selects the block): Cin 64 -> 64 runs as two 32-channel halves so
// that the 18 filter fragments stay in registers.  Any width: the last strip of a row is partly out of the image (its
// loads and stores get out-of-range offsets).
#include "qnn_mfma_common.h"
#include "qnn_fold.h"

#ifndef QNN_STRIP16_WPS
#define QNN_STRIP16_WPS 6        // waves per SIMD (= persistent workgroups per CU): Cin 16 needs 51-67 VGPRs
#endif
#ifndef QNN_STRIP32_WPS
#define QNN_STRIP32_WPS 1        // Cin 32: 118-128 VGPRs.  Round 4: ONE wave per SIMD with a six-row ring (3 before): same layer time,
                                 // room for the other batches' kernels: ResNet-224 with three batches in flight 85.1 -> 87.0 K img/s
#endif
#ifndef QNN_STRIP64_WPS
#define QNN_STRIP64_WPS 1        // Cin 64 (two 32-channel halves per strip): 164-200 VGPRs, spills at 3 per SIMD.  Round 4: ONE
                                 // wave per SIMD with a six-row ring: the layer alone is as fast as with two (11.3 / 13.2 us),
                                 // and the other batch's kernels find room beside it: ResNet-224 end to end 77.4 -> 79.5 K img/s
#endif
#ifndef QNN_STRIP16_DEPTH
#define QNN_STRIP16_DEPTH 3      // input rows requested ahead of the row being computed (see the ring in k_conv_strip)
#endif
#ifndef QNN_STRIP32_DEPTH
#define QNN_STRIP32_DEPTH 6
#endif
#ifndef QNN_STRIP64_DEPTH
#define QNN_STRIP64_DEPTH 6
#endif

namespace {

// Workgroup -> (task stream, channel block) when a layer's channel blocks are separate workgroups (gridDim.y > 1): the
// blocks of one task stream are placed on ONE XCD (workgroups are dealt to the eight XCDs round robin in dispatch order)
// and start together, so the pieces of a stored pixel written by different blocks meet in the same L2 instead of leaving
// partial sectors in several (measured on the un-pooled int8 first layer, qnn_first_u8.hip: 0.49 -> 0.39 ms).
__device__ __forceinline__ void strip_block_map(int& xw, int& yblk) {
    xw = blockIdx.x; yblk = blockIdx.y;
    const int nsl = gridDim.y, bxw = gridDim.x;
    if (nsl > 1 && (bxw & 7) == 0) {
        const int bid = blockIdx.y * bxw + blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3;
        yblk = j % nsl;
        xw = (j / nsl) * 8 + xcd;
    }
}

// FOLD: the epilogue as integer thresholds (qnn_fold.h; e.fold_a / e.fold_b, proven equal to the float32 chain on the
// layer's whole accumulator domain by qnn_fold_prepare): the offset is the MFMA's initial accumulator, then per value
// cvt + mul, per pair one v_cvt_pknorm_i16_f32 (+ two v_pk_add_i16 clamp with a shortcut), and the nibbles of a lane's
// field are gathered with two v_perm_b32 + shift + v_bfi_b32 (NT = 2) instead of cvt, [add], mul, add, [bfe, cvt, fma],
// add, v_med3 and a shift-or per value.
// RES: 0 none, 1 packed int4 shortcut, 2 float32 shortcut, 3 projection shortcut computed here (qnn_projection_t: the
// 1x1 strides-2 convolution of the block input with CIN / 2 channels, one more MFMA per tile and row on the even pixels of
// the even input rows -- the float32 tensor RES = 2 reads is never formed); FOLD: 0 chain, 1 / 2 = fold modes (qnn_fold.h)
template <int CIN, int NT, int RES, bool BIAS, int FOLD>
__global__ __launch_bounds__(256, (CIN == 16 ? QNN_STRIP16_WPS : CIN == 32 ? (RES == 2 ? 2 : QNN_STRIP32_WPS) : QNN_STRIP64_WPS)) void k_conv_strip(MfmaGeom mg, EpiArgs e,
                                                                 const uint8_t* __restrict__ x,
                                                                 const uint8_t* __restrict__ wq8,
                                                                 void* __restrict__ y, int ntasks, int spr,
                                                                 FastDiv fd_spr, int nch, FastDiv fd_nch, int rc,
                                                                 uint32_t img_x, uint32_t img_y, uint32_t img_r) {
    constexpr int BP = CIN / 16;                      // 16-channel k-blocks per pixel
    constexpr int ST = (3 * BP + 3) / 4;              // K-steps per input row: 1 / 2 / 3
    constexpr int PIXB = CIN / 2;                     // bytes per stored input pixel
    const ConvGeom& g = mg.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int xw_, yb_;
    strip_block_map(xw_, yb_);
    const int wid = xw_ * 4 + wave, nw = gridDim.x * 4;
    const int nbase = yb_ * (16 * NT);                // first output channel of this wave

    // ---- k-block of this lane in every K-step ----
    int dxs[ST], hbs[ST];
    bool kok[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        const int j = 4 * st + kq;
        kok[st] = j < 3 * BP;
        dxs[st] = kok[st] ? j / BP : 1;
        hbs[st] = kok[st] ? j % BP : 0;
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
                const int ch = nbase + 4 * NT * (r >> 2) + 4 * nt + (r & 3);       // filter of A row r in tile nt
                const int woff = kok[st] ? (ch * 9 + dy * 3 + dxs[st]) * CIN + hbs[st] * 16 : (int)0x80000000;
                bw[dy][st][nt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, 0, 0));
            }
    // ---- epilogue constants: this lane's channels are nbase + 4*NT*kq + 4*nt + i ----
    // Everything behind the BN is scaled by powers of two only (activation code scale m, residual post-scale): those
    // factors commute with every float32 rounding, so they are folded into the per-channel constants and the epilogue
    // is cvt, [add bias], mul, add, [fma shortcut], round, clamp per value.
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float cfold = (RES != 0 ? e.post_scale : 1.0f) * (binary ? 1.0f : e.act_m);
    const float rcoef = RES == 1 ? e.res_scale * cfold : cfold;        // shortcut code (or float value) -> scaled sum
    // (the float32 steps run on channel PAIRS: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 round each half exactly like
    // their scalar forms and issue at the same rate -- tools/micro/pk_f32_rate.hip -- so the multiply, the two adds and
    // the shortcut FMA cost half an instruction per value)
    v2f nb[NT][2], ninv[NT][2], nshift[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = nbase + 4 * NT * kq + 4 * nt + i;
            const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
            const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
            nb[nt][i >> 1][i & 1] = BIAS ? __fdiv_rn(e.bias[c], e.scale) : 0.0f;
            ninv[nt][i >> 1][i & 1] = __fmul_rn(__fmul_rn(inv, e.scale), cfold);
            nshift[nt][i >> 1][i & 1] = __fmul_rn(shift, cfold);
        }
    const v2f rcoef2 = {rcoef, rcoef};
    // folded epilogue: per-channel slope and accumulator offset (the MFMA chain starts from the offset)
    static_assert(!FOLD || RES < 2, "the float32 shortcuts are not folded");
    static_assert(RES != 3 || (NT == 2 && CIN >= 32), "projection shortcut: 32 / 64 channel stages");
    // ---- projection shortcut (RES = 3): 1x1 filters as one more A operand per tile; K-blocks beyond CIN / 2 channels are
    // out of range = zeros, and so are the same lanes' pixel loads ----
    constexpr int P0B = CIN / 4;                      // bytes per stored pixel of the block input (CIN / 2 channels)
    constexpr int BPP = CIN / 32;                     // its 16-channel k-blocks: 1 / 2
    v4i bwp[NT];
    v2f pbias[NT][2];
    const v2f pscale2 = {e.proj_scale, e.proj_scale};
    if constexpr (RES == 3) {
        const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(e.proj_w), 0, g.cout * (CIN / 2), 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ch = nbase + 4 * NT * (r >> 2) + 4 * nt + (r & 3);
            const int woff = kq < BPP ? ch * (CIN / 2) + kq * 16 : (int)0x80000000;
            bwp[nt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(prsrc, woff, 0, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                pbias[nt][i >> 1][i & 1] = e.proj_bias ? e.proj_bias[nbase + 4 * NT * kq + 4 * nt + i] : 0.0f;
        }
    }
    const bool has_pbias = e.proj_bias != nullptr;
    float fa[NT][4], fc[NT][4];
    v4i binit[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = nbase + 4 * NT * kq + 4 * nt + i;
            fa[nt][i] = FOLD ? e.fold_a[c] : 0.0f;
            fc[nt][i] = FOLD == 2 ? e.fold_c[c] : 0.0f;
            binit[nt][i] = FOLD ? e.fold_b[c] : 0;
        }
    // round-half-even + clamp + offset code in the integer domain: as_int(u + (1.5*2^23 + 8)) = 0x4B400008 + rint(u)
    // for |u| < 2^22 and is monotone in u everywhere, so a signed integer med3 clamps it; the low nibble is code + 8
    constexpr float kMagic = 12582920.0f;
    constexpr int kMagicBits = 0x4B400008;
    v2f magic2 = {kMagic, kMagic};
    asm volatile("" : "+v"(magic2));                  // a register pair (v_pk_add_f32 takes no literal): keeps the add packed
    const int code_lo = kMagicBits - (int)e.act_m, code_hi = kMagicBits + (int)e.act_m - 1;
    const int rowb = g.W * PIXB;                      // bytes per input row
    const int orowb = g.W * e.ocw * 4;                // bytes per output row
    const int rrowb = RES == 2 ? g.W * g.cout * 4 : RES == 3 ? 2 * e.proj_W * P0B : orowb;   // (RES = 3: every other input row)

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
            RES == 3 ? const_cast<uint8_t*>(e.proj_x) + (size_t)n * img_r
                     : RES != 0 ? (uint8_t*)const_cast<void*>(e.res) + (size_t)n * img_r : (uint8_t*)y, 0,
            RES != 0 ? (int)img_r : 0, 0x00020000);
        // byte offset (inside the image) of this lane's k-block in input row y0 - 1; lanes whose pixel lies left or right of
        // the image start out of range and stay there (0x80000000 + row increments < 2^32)
        int voff[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const int px = xs + r + dxs[st] - 1;
            voff[st] = (kok[st] && px >= 0 && px < g.W) ? ((y0 - 1) * g.W + px) * PIXB + hbs[st] * 8 : (int)0x80000000;
        }
        const bool pvalid = xs + r < g.W;                                       // last strip of a ragged row
        int ovoff = pvalid ? (y0 * g.W + xs + r) * e.ocw * 4 + nbase / 2 + kq * 2 * NT : (int)0x80000000;   // 2*NT bytes
        int rvoff = RES == 2 ? (pvalid ? ((y0 * g.W + xs + r) * g.cout + nbase + 4 * NT * kq) * 4 : (int)0x80000000)
                    : RES == 3 ? ((pvalid && kq < BPP) ? (2 * y0 * e.proj_W + 2 * (xs + r)) * P0B + kq * 8 : (int)0x80000000)
                             : ovoff;                                           // + nt*16 (f32: the next four floats)

        // Loads are requested D rows ahead (a ring of D raw input-row register sets and D shortcut registers): vmcnt retires
        // in issue order, so a wave that waits for the oldest request keeps D - 1 rows of younger ones in flight.  Round 4:
        // with the folded epilogue the kernels are no longer bound by instruction issue but by the bytes they keep in
        // flight (6144 waves x 3 rows x 128 unique bytes = 2.4 MB against ~1 us of loaded HBM latency = the 2.4 TB/s they
        // reached): the ring depth is a compile-time constant per channel count (QNN_STRIP*_DEPTH, a multiple of 3).
        constexpr int D = RES == 2 ? 3          // (the float32 shortcut of a projection block: 16 bytes per lane, row and tile)
                          : CIN == 16 ? QNN_STRIP16_DEPTH : CIN == 32 ? QNN_STRIP32_DEPTH : QNN_STRIP64_DEPTH;
        static_assert(D % 3 == 0 && D >= 3, "the row loop is unrolled D times and rotates three operand sets");
        v4i X[3][ST];
        uint2 raw[D][ST];
        uint32_t rs[D][NT];                            // packed shortcut: [.][0] holds the lane's whole 2*NT-byte field
        float4 rf[D][NT];
        int abl_st = 0; uint32_t abl_acc = 0;
        int abl_rows = 0;                              // timing experiments only (QNN_STRIP_ABL, -DQNN_EXPERIMENTS builds)
        auto load_row = [&](uint2 (&dst)[ST]) {
#ifdef QNN_STRIP_ABL
            if ((QNN_STRIP_ABL & 2) && ++abl_rows > D + 2) return;              // 2: no input loads behind the preamble
#endif
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                dst[st] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xr, voff[st], 0, 0));
                voff[st] += rowb;
            }
        };
        auto load_res = [&](uint32_t (&ds)[NT], float4 (&df)[NT]) {
            if constexpr (RES == 1) {
                if constexpr (NT == 1) ds[0] = __builtin_amdgcn_raw_buffer_load_b16(rr, rvoff, 0, 0);
                else ds[0] = __builtin_amdgcn_raw_buffer_load_b32(rr, rvoff, 0, 0);        // both tiles' codes in one word
            }
            if constexpr (RES == 3) {                  // this lane's k-block of block-input pixel (2 yy, 2 (xs + r)): 16 codes
                const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rr, rvoff, 0, 0));
                ds[0] = t.x; ds[NT - 1] = t.y;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if constexpr (RES == 2)
                    df[nt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, rvoff + 16 * nt, 0, 0));
            rvoff += rrowb;
        };
        {
            uint2 t0[ST], t1[ST];
            load_row(t0);                              // row y0 - 1
            load_row(t1);                              // row y0
#pragma unroll
            for (int d = 0; d < D; ++d) load_row(raw[d]);                       // rows y0 + 1 .. y0 + D
            if constexpr (RES != 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) load_res(rs[d], rf[d]);             // rows y0 .. y0 + D - 1
            }
#pragma unroll
            for (int st = 0; st < ST; ++st) X[0][st] = widen(t0[st]);
#pragma unroll
            for (int st = 0; st < ST; ++st) X[1][st] = widen(t1[st]);
        }

        // one output row yy = y0 + j: X slots a / b / c = j, j+1, j+2 (mod 3) hold input rows yy-1 / yy / yy+1; ring slot
        // j mod D holds row yy+1 (consumed here, refilled with row yy+1+D); shortcut slot j mod D holds row yy (refilled
        // with row yy+D)
        auto body = [&](v4i (&Xa)[ST], v4i (&Xb)[ST], v4i (&Xc)[ST], uint2 (&rw)[ST], uint32_t (&rsc)[NT],
                        float4 (&rfc)[NT]) {
#pragma unroll
            for (int st = 0; st < ST; ++st) Xc[st] = widen(rw[st]);
            load_row(rw);                              // row yy + 1 + D
            uint32_t rcur[NT];
            float4 fcur[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { rcur[nt] = rsc[nt]; fcur[nt] = rfc[nt]; }
            if constexpr (RES != 0) load_res(rsc, rfc);   // shortcut of row yy + D
            v4i accp[NT];
            if constexpr (RES == 3) {
                const v4i Xp = widen(make_uint2(rcur[0], rcur[NT - 1]));
                const v4i z0 = {0, 0, 0, 0};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accp[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bwp[nt], Xp, z0, 0, 0, 0);
            }
            v4i acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const v4i z = binit[nt];               // zeros, or the fold's per-channel offsets
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
            if constexpr (FOLD) {
                // ---- folded epilogue (qnn_fold.h) ----
                if constexpr (NT == 1) {
                    // pairs (c0, c2), (c1, c3): codes in bits 12-15 / 28-31 of a pair
                    uint32_t pe = qnn_fold_pair_m<FOLD>(acc[0][0], acc[0][2], fa[0][0], fa[0][2], fc[0][0], fc[0][2]);
                    uint32_t po = qnn_fold_pair_m<FOLD>(acc[0][1], acc[0][3], fa[0][1], fa[0][3], fc[0][1], fc[0][3]);
                    if constexpr (RES == 1) {
                        // shortcut field (16 bits): nibbles -> offset codes (sc + 8) << 10 in the halves of a pair
                        const uint32_t w = rcur[0] ^ 0x8888u;
                        const uint32_t yy2 = __builtin_amdgcn_perm(0u, w, 0x0C010C00u);          // bytes (b0, 0, b1, 0)
                        pe = qnn_fold_merge(pe, (yy2 & 0x000F000Fu) << 10);
                        po = qnn_fold_merge(po, (yy2 & 0x00F000F0u) << 6);
                    }
                    const uint32_t mm = (po & 0xF000F000u) | ((pe >> 4) & ~0xF000F000u);          // v_bfi_b32: bytes 1, 3 = (c1:c0), (c3:c2)
                    const uint32_t o16 = __builtin_amdgcn_perm(0u, mm, 0x0C0C0301u);
#ifdef QNN_STRIP_ABL
                    if (QNN_STRIP_ABL & 1) {                                   // 1: one store per eight rows (results wrong)
                        static_assert(true, "");
                        abl_acc ^= o16;
                        if ((++abl_st & 7) == 0) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)abl_acc, yr, ovoff, 0, 0);
                    } else
#endif
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)o16, yr, ovoff, 0, 0);
                } else {
                    static_assert(NT <= 2, "a lane's fields must fit one word");
                    uint32_t tp[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        tp[j] = qnn_fold_pair_m<FOLD>(acc[0][j], acc[1][j], fa[0][j], fa[1][j], fc[0][j], fc[1][j]);   // (c_j, c_j+4)
                    if constexpr (RES == 1) {
                        const uint32_t w = rcur[0] ^ 0x88888888u;
                        tp[0] = qnn_fold_merge(tp[0], (w & 0x000F000Fu) << 10);
                        tp[1] = qnn_fold_merge(tp[1], (w & 0x00F000F0u) << 6);
                        tp[2] = qnn_fold_merge(tp[2], (w & 0x0F000F00u) << 2);
                        tp[3] = qnn_fold_merge(tp[3], (w & 0xF000F000u) >> 2);
                    }
                    const uint32_t uo = __builtin_amdgcn_perm(tp[3], tp[1], 0x07030501u);         // (c1, c3, c5, c7) in the high nibbles
                    const uint32_t ue = __builtin_amdgcn_perm(tp[2], tp[0], 0x07030501u);         // (c0, c2, c4, c6)
                    const uint32_t o32 = (uo & 0xF0F0F0F0u) | ((ue >> 4) & 0x0F0F0F0Fu);
                    __builtin_amdgcn_raw_buffer_store_b32(o32, yr, ovoff, 0, 0);
                }
                ovoff += orowb;
                return;
            }
            // ---- epilogue: the reference's op order, one rounding per operation ----
            uint32_t field[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                v2f u2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f v = {(float)acc[nt][2 * h], (float)acc[nt][2 * h + 1]};
                    if constexpr (BIAS) v = v + nb[nt][h];
                    v2f u = v * ninv[nt][h];                       // two roundings per value, as the reference's BN
                    u = u + nshift[nt][h];
                    if constexpr (RES == 1) {
                        // shortcut value = code * 2^-(bits-1), exact: fma(code, scale, t) IS the reference's x + y
                        const v2f cd = {(float)((int)(rcur[0] << (28 - 8 * h - 16 * nt)) >> 28),
                                        (float)((int)(rcur[0] << (24 - 8 * h - 16 * nt)) >> 28)};
                        u = __builtin_elementwise_fma(cd, rcoef2, u);
                    }
                    if constexpr (RES == 2) {
                        const v2f rv = h == 0 ? v2f{fcur[nt].x, fcur[nt].y} : v2f{fcur[nt].z, fcur[nt].w};
                        u = __builtin_elementwise_fma(rv, rcoef2, u);   // (x + y) * 2^k == x*2^k + y*2^k, one rounding either way
                    }
                    if constexpr (RES == 3) {
                        // the float32 value k_conv_pw_f32 stores for this shortcut: float(sum) * scale [+ bias], one rounding each
                        v2f rv = v2f{(float)accp[nt][2 * h], (float)accp[nt][2 * h + 1]} * pscale2;
                        if (has_pbias) rv = rv + pbias[nt][h];
                        u = __builtin_elementwise_fma(rv, rcoef2, u);
                    }
                    u2[h] = u;
                }
                int cb[4];
                if (binary) {
                    asm volatile("; binary_tanh codes");          // keeps this a real (uniform) branch
#pragma unroll
                    for (int i = 0; i < 4; ++i) cb[i] = u2[i >> 1][i & 1] > 0x1p-24f ? kMagicBits + 1 : kMagicBits - 1;
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const v2f t = u2[h] + magic2;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int bits = __float_as_int(t[j]);
                            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cb[2 * h + j]) : "v"(bits), "v"(code_lo), "v"(code_hi));
                        }
                    }
                }
                // low nibble of cb[i] = code + 8, bits 4..21 are zero: three shift-ors assemble the 16-bit field
                uint32_t P = ((uint32_t)cb[1] << 4) | (uint32_t)cb[0];
                P = ((uint32_t)cb[2] << 8) | P;
                P = ((uint32_t)cb[3] << 12) | P;
                field[nt] = P;                             // bits 16.. hold shifted bits of the magic constant
            }
            if constexpr (NT == 1) {
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(field[0] ^ 0x8888u), yr, ovoff, 0, 0);
            } else {
                static_assert(NT <= 2, "a lane's fields must fit one word");
                __builtin_amdgcn_raw_buffer_store_b32(((field[1] << 16) | (field[0] & 0xFFFFu)) ^ 0x88888888u, yr, ovoff, 0, 0);
            }
            ovoff += orowb;
        };
#define STRIP_BODY(J) body(X[(J) % 3], X[((J) + 1) % 3], X[((J) + 2) % 3], raw[(J) % D], rs[(J) % D], rf[(J) % D])
        int yy = y0;
        for (; yy + D <= y1; yy += D) {
            STRIP_BODY(0); STRIP_BODY(1); STRIP_BODY(2);
            if constexpr (D > 3) { STRIP_BODY(3); STRIP_BODY(4); STRIP_BODY(5); }
            if constexpr (D > 6) { STRIP_BODY(6); STRIP_BODY(7); STRIP_BODY(8); }
            if constexpr (D > 9) { STRIP_BODY(9); STRIP_BODY(10); STRIP_BODY(11); }
        }
        static_assert(D <= 12, "unrolled by hand up to twelve rows");
        const int rem = y1 - yy;
        if (rem > 0) STRIP_BODY(0);
        if (rem > 1) STRIP_BODY(1);
        if constexpr (D > 3) { if (rem > 2) STRIP_BODY(2); if (rem > 3) STRIP_BODY(3); if (rem > 4) STRIP_BODY(4); }
        if constexpr (D > 6) { if (rem > 5) STRIP_BODY(5); if (rem > 6) STRIP_BODY(6); if (rem > 7) STRIP_BODY(7); }
        if constexpr (D > 9) { if (rem > 8) STRIP_BODY(8); if (rem > 9) STRIP_BODY(9); if (rem > 10) STRIP_BODY(10); }
#undef STRIP_BODY
    }
}

// ---------------------------------------------------------------------------------------------------------
// The stride-2 3x3 layers that open a ResNet stage (models/resnet.py:108-112; no residual merge behind them): the same
// strip walk over OUTPUT rows; an output row needs input rows 2*yy - pt .. + 2, of which only one is shared with the next
// output row, so every output row simply requests its three input rows (one output row ahead) and widens them.
// B operand of lane (pixel r, k-block): input pixel 2*(xs + r) - pl + dx.
// ---------------------------------------------------------------------------------------------------------
template <int CIN, int NT, bool BIAS, int FOLD>
__global__ __launch_bounds__(256, (CIN == 16 ? 4 : 2)) void k_conv_strip_s2(MfmaGeom mg, EpiArgs e,
                                                                             const uint8_t* __restrict__ x,
                                                                             const uint8_t* __restrict__ wq8,
                                                                             void* __restrict__ y, int ntasks, int spr,
                                                                             FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                                             int rc, uint32_t img_x, uint32_t img_y) {
    constexpr int BP = CIN / 16;
    constexpr int ST = BP;                             // tap dx = kq, channel group st (see k_conv_strip)
    constexpr int PIXB = CIN / 2;
    const ConvGeom& g = mg.g;                          // g.H, g.W: input; g.Ho, g.Wo: output
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int xw_, yb_;
    strip_block_map(xw_, yb_);
    const int wid = xw_ * 4 + wave, nw = gridDim.x * 4;
    const int nbase = yb_ * (16 * NT);
    int dxs[ST], hbs[ST];
    bool kok[ST];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        kok[st] = kq < 3;
        dxs[st] = kok[st] ? kq : 1;
        hbs[st] = st;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);
    v4i bw[3][ST][NT];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = nbase + 4 * NT * (r >> 2) + 4 * nt + (r & 3);       // as in k_conv_strip: consecutive channels per lane
                const int woff = kok[st] ? (ch * 9 + dy * 3 + dxs[st]) * CIN + hbs[st] * 16 : (int)0x80000000;
                bw[dy][st][nt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, 0, 0));
            }
    static_assert(NT == 2, "the stride-2 kernel stores one 32-bit word per lane");
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float cfold = binary ? 1.0f : e.act_m;
    v2f nb[NT][2], ninv[NT][2], nshift[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = nbase + 4 * NT * kq + 4 * nt + i;
            nb[nt][i >> 1][i & 1] = BIAS ? __fdiv_rn(e.bias[c], e.scale) : 0.0f;
            ninv[nt][i >> 1][i & 1] = __fmul_rn(__fmul_rn(e.bn_inv ? e.bn_inv[c] : 1.0f, e.scale), cfold);
            nshift[nt][i >> 1][i & 1] = __fmul_rn(e.bn_inv ? e.bn_shift[c] : 0.0f, cfold);
        }
    float fa[NT][4], fc[NT][4];                        // folded epilogue (k_conv_strip): slope, constant and accumulator offset
    v4i binit[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = nbase + 4 * NT * kq + 4 * nt + i;
            fa[nt][i] = FOLD ? e.fold_a[c] : 0.0f;
            fc[nt][i] = FOLD == 2 ? e.fold_c[c] : 0.0f;
            binit[nt][i] = FOLD ? e.fold_b[c] : 0;
        }
    constexpr float kMagic = 12582920.0f;
    constexpr int kMagicBits = 0x4B400008;
    v2f magic2 = {kMagic, kMagic};
    asm volatile("" : "+v"(magic2));                  // a register pair (v_pk_add_f32 takes no literal): keeps the add packed
    const int code_lo = kMagicBits - (int)e.act_m, code_hi = kMagicBits + (int)e.act_m - 1;
    const int rowb2 = 2 * g.W * PIXB;                  // two input rows per output row
    const int orowb = g.Wo * e.ocw * 4;
    auto widen = [&](const uint2& q) -> v4i {
        const uint4 v = make_uint4((q.x << 4) & 0xF0F0F0F0u, q.x & 0xF0F0F0F0u,
                                   (q.y << 4) & 0xF0F0F0F0u, q.y & 0xF0F0F0F0u);
        return __builtin_bit_cast(v4i, v);
    };
    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int y0 = chunk * rc;
        const int y1 = min(y0 + rc, g.Ho);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)y + (size_t)n * img_y, 0, (int)img_y, 0x00020000);
        int voff[3];
        const int px = 2 * (xs + r) - g.pl + kq;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
            voff[dy] = (kq < 3 && px >= 0 && px < g.W && xs + r < g.Wo) ? ((2 * y0 - g.pt + dy) * g.W + px) * PIXB
                                                                        : (int)0x80000000;
        const bool pvalid = xs + r < g.Wo;
        int ovoff = pvalid ? (y0 * g.Wo + xs + r) * e.ocw * 4 + nbase / 2 + kq * 2 * NT : (int)0x80000000;
        uint2 raw[2][3][ST];
        auto load_rows = [&](uint2 (&dst)[3][ST]) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                if constexpr (BP == 1) {
                    dst[dy][0] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xr, voff[dy], 0, 0));
                } else {
#pragma unroll
                    for (int h = 0; h < BP / 2; ++h) {
                        const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[dy] + 16 * h, 0, 0));
                        dst[dy][2 * h] = make_uint2(v.x, v.y);
                        dst[dy][2 * h + 1] = make_uint2(v.z, v.w);
                    }
                }
                voff[dy] += rowb2;
            }
        };
        load_rows(raw[0]);
        auto body = [&](uint2 (&cur)[3][ST], uint2 (&nxt)[3][ST]) {
            load_rows(nxt);                            // the three input rows of the next output row
            v4i acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = binit[nt];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    const v4i xo = widen(cur[dy][st]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[dy][st][nt], xo, acc[nt], 0, 0, 0);
                }
            if constexpr (FOLD) {
                uint32_t tp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    tp[j] = qnn_fold_pair_m<FOLD>(acc[0][j], acc[1][j], fa[0][j], fa[1][j], fc[0][j], fc[1][j]);   // (c_j, c_j+4)
                const uint32_t uo = __builtin_amdgcn_perm(tp[3], tp[1], 0x07030501u);
                const uint32_t ue = __builtin_amdgcn_perm(tp[2], tp[0], 0x07030501u);
                __builtin_amdgcn_raw_buffer_store_b32((uo & 0xF0F0F0F0u) | ((ue >> 4) & 0x0F0F0F0Fu), yr, ovoff, 0, 0);
                ovoff += orowb;
                return;
            }
            uint32_t field[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                v2f u2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f v = {(float)acc[nt][2 * h], (float)acc[nt][2 * h + 1]};
                    if constexpr (BIAS) v = v + nb[nt][h];
                    u2[h] = v * ninv[nt][h];
                    u2[h] = u2[h] + nshift[nt][h];
                }
                int cb[4];
                if (binary) {
                    asm volatile("; binary_tanh codes");
#pragma unroll
                    for (int i = 0; i < 4; ++i) cb[i] = u2[i >> 1][i & 1] > 0x1p-24f ? kMagicBits + 1 : kMagicBits - 1;
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const v2f t = u2[h] + magic2;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int bits = __float_as_int(t[j]);
                            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cb[2 * h + j]) : "v"(bits), "v"(code_lo), "v"(code_hi));
                        }
                    }
                }
                uint32_t P = ((uint32_t)cb[1] << 4) | (uint32_t)cb[0];
                P = ((uint32_t)cb[2] << 8) | P;
                P = ((uint32_t)cb[3] << 12) | P;
                field[nt] = P;
            }
            __builtin_amdgcn_raw_buffer_store_b32(((field[1] << 16) | (field[0] & 0xFFFFu)) ^ 0x88888888u, yr, ovoff, 0, 0);
            ovoff += orowb;
        };
        int yy = y0;
        for (; yy + 2 <= y1; yy += 2) {
            body(raw[0], raw[1]);
            body(raw[1], raw[0]);
        }
        if (yy < y1) body(raw[0], raw[1]);
    }
}

template <int CIN, int NT>
int launch_strip_s2(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s) {
    const ConvGeom& g = mg.g;
    const int spr = (g.Wo + 15) / 16;
    const int ny = g.cout / (16 * NT);
    const double img_x = (double)g.H * g.W * (CIN / 2), img_y = (double)g.Ho * g.Wo * e.ocw * 4.0;
    if (img_x >= 1.0e9 || img_y >= 1.0e9 || ny < 1 || ny * 16 * NT != g.cout) return 1;
    const int wps = CIN == 16 ? 4 : 2;
    const int blocks_cap = 256 * wps / ny > 0 ? 256 * wps / ny : 1;
    const long nwaves = (long)blocks_cap * 4;
    int best_rc = g.Ho, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = g.Ho < 4 ? g.Ho : 4; rc <= g.Ho; ++rc) {
        const int nch = (g.Ho + rc - 1) / rc;
        const long rounds = ((long)g.N * spr * nch + nwaves - 1) / nwaves;
        const double cost = (double)rounds * (rc + 2);
        if (cost < best_cost) { best_cost = cost; best_rc = rc; best_nch = nch; }
    }
    const long ntasks_l = (long)g.N * spr * best_nch;
    if (ntasks_l >= 2000000000L) return 1;
    long blocks = (ntasks_l + 3) / 4;
    if (blocks > blocks_cap) blocks = blocks_cap;
    const dim3 grid((unsigned)blocks, (unsigned)ny), block(256);
    if (e.fold_a && e.fold_c)   // folded epilogue (the bias is inside the fold), "bits" form
        hipLaunchKernelGGL((k_conv_strip_s2<CIN, NT, false, 2>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, (int)ntasks_l,
                           spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,
                           (uint32_t)img_x, (uint32_t)img_y);
    else if (e.fold_a)
        hipLaunchKernelGGL((k_conv_strip_s2<CIN, NT, false, 1>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, (int)ntasks_l,
                           spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,
                           (uint32_t)img_x, (uint32_t)img_y);
    else if (e.bias)
        hipLaunchKernelGGL((k_conv_strip_s2<CIN, NT, true, 0>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, (int)ntasks_l,
                           spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,
                           (uint32_t)img_x, (uint32_t)img_y);
    else
        hipLaunchKernelGGL((k_conv_strip_s2<CIN, NT, false, 0>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, (int)ntasks_l,
                           spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc,
                           (uint32_t)img_x, (uint32_t)img_y);
    return 0;
}

template <int CIN, int NT>
int launch_strip(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s) {
    const ConvGeom& g = mg.g;
    const int spr = (g.W + 15) / 16;
    const int ny = g.cout / (16 * NT);
    const double img_x = (double)g.H * g.W * (CIN / 2), img_y = (double)g.H * g.W * e.ocw * 4.0;
    const int res = e.proj_x ? 3 : !e.res ? 0 : e.res_store == QNN_STORE_F32 ? 2 : 1;
    const double img_r = res == 2 ? (double)g.H * g.W * g.cout * 4.0
                         : res == 3 ? (double)e.proj_H * e.proj_W * (CIN / 4) : img_y;
    if (res == 3 && (NT != 2 || CIN < 32 || e.proj_cin * 2 != CIN)) return 1;
    if (img_x >= 1.0e9 || img_y >= 1.0e9 || img_r >= 1.0e9 || ny < 1 || ny * 16 * NT != g.cout) return 1;
    // persistent grid: WPS waves per SIMD; rows per task chosen so that the task count fills whole
    // rounds of that grid (a round costs rc output rows + 3 rows of pipeline fill)
    const int wps = CIN == 16 ? QNN_STRIP16_WPS : CIN == 32 ? QNN_STRIP32_WPS : QNN_STRIP64_WPS;
    const int blocks_cap = 256 * wps / ny > 0 ? 256 * wps / ny : 1;
    const long nwaves = (long)blocks_cap * 4;
    int best_rc = g.H, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = g.H < 4 ? g.H : 4; rc <= g.H; ++rc) {
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
    const dim3 grid((unsigned)blocks, (unsigned)ny), block(256);
    const bool bias = e.bias != nullptr && !(e.fold_a != nullptr && res < 2);   // a fold contains the bias
#define STRIP_CASE(RES_, BIAS_, FOLD_)                                                                        \
    if (res == RES_ && bias == BIAS_ && fold == FOLD_) {                                                      \
        hipLaunchKernelGGL((k_conv_strip<CIN, NT, RES_, BIAS_, FOLD_>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, \
                           ntasks, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),   \
                           best_rc, (uint32_t)img_x, (uint32_t)img_y, (uint32_t)img_r);                        \
        return 0;                                                                                             \
    }
    // folded epilogue: everything behind the accumulator (bias included) is inside the fold's two constants
    const int fold = (e.fold_a == nullptr || res >= 2) ? 0 : e.fold_c ? 2 : 1;
    STRIP_CASE(0, false, 2) STRIP_CASE(1, false, 2) STRIP_CASE(0, false, 1) STRIP_CASE(1, false, 1)
    STRIP_CASE(0, false, 0) STRIP_CASE(0, true, 0) STRIP_CASE(1, false, 0) STRIP_CASE(1, true, 0)
    STRIP_CASE(2, false, 0) STRIP_CASE(2, true, 0)
    if constexpr (NT == 2 && CIN >= 32) { STRIP_CASE(3, false, 0) STRIP_CASE(3, true, 0) }
#undef STRIP_CASE
    return 1;
}

}  // namespace

// cin in {16, 32, 64}, cout a multiple of 16 / 32 / 32; eligibility is checked by the caller
int qnn_launch_strip(int cin, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                     void* y, hipStream_t s) {
    if (mg.g.stride == 2) {
        if (e.res) return 1;
        return cin == 16 ? launch_strip_s2<16, 2>(mg, e, x, w, y, s) : cin == 32 ? launch_strip_s2<32, 2>(mg, e, x, w, y, s) : 1;
    }
    if (cin == 16) return launch_strip<16, 1>(mg, e, x, w, y, s);
    if (cin == 32) return launch_strip<32, 2>(mg, e, x, w, y, s);
    return launch_strip<64, 2>(mg, e, x, w, y, s);
}
