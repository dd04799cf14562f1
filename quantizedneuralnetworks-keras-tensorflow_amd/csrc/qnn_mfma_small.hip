// 3x3 int4 layers with 16 / 32 input channels on v_mfma_i32_16x16x64_i8, both operands in registers.
// Dispatch: qnn_try_launch_mfma (qnn_mfma.hip).
#include "qnn_mfma_common.h"

namespace {

// ---------------------------------------------------------------------------------
// Small-channel 3x3 layers (Cin = 16 or 32, int4 in, int4 out: the 224x224 and 112x112
// stages of the ResNet): v_mfma_i32_16x16x64_i8 with BOTH operands in registers, no LDS.
//
// A 64-deep K-step covers 64 / Cin taps of one 16-pixel row segment: lane (r = lane & 15,
// kq = lane >> 4) supplies the sixteen channels kq selects of pixel r -- one contiguous
// 8-byte chunk of the NHWC tensor (Cin = 16: the whole tap kq; Cin = 32: half (kq & 1) of
// tap (kq >> 1)), fetched with one buffer load and widened to int8 in registers.  The
// filters of the wave's 16 x NT outputs for ALL K-steps stay in VGPRs (12 / 40 registers).
// A wave owns four consecutive row segments (64 pixels), its operand registers are
// refilled for the next tile right after they have been consumed (prefetch distance =
// one tile), SAME padding = per-(segment, K-step) scalar lane masks assembled from five
// constant masks per K-step (tap exists / tap in the row above / below / left-edge lane /
// right-edge lane) and the segment's scalar border flags.
// Epilogue: BN -> [residual merge: the shortcut word of this lane's OUTPUT position is
// loaded and nibble-transposed back, so every lane gets its channel's eight shortcut
// codes from one load] -> clip -> code -> nibble transpose -> one word per lane.
template <int CIN, int NT>
__global__ __launch_bounds__(256, (CIN == 16 ? QNN_SMALL16_WPC : QNN_SMALL32_WPC)) void k_conv_mfma_small(MfmaGeom mg, EpiArgs e,
                                                            const uint8_t* __restrict__ x,
                                                            const uint8_t* __restrict__ wq8,
                                                            void* __restrict__ y, int nsegs,
                                                            int ntiles, FastDiv fd_spr, int spr,
                                                            uint32_t y_bytes, uint32_t res_bytes) {
    constexpr int TAPS = 9;
    constexpr int KS = (TAPS * CIN + 63) / 64;          // 3 (Cin 16), 5 (Cin 32)
    constexpr int TPS = 64 / CIN;                        // taps per K-step: 4 / 2
    constexpr int LPT = 4 / TPS;                         // 16-lane groups per tap: 1 / 2
    constexpr int MT = 4;                                // row segments per wave tile
    constexpr int PIXB = CIN / 2;                        // bytes per pixel (int4)
    const ConvGeom& g = mg.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nbase = blockIdx.y * (16 * NT);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(x), 0, (int)mg.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(e.res ? e.res : (const void*)y), 0, (int)res_bytes, 0x00020000);

    // ---- per-lane K-slot constants, filters, constant lane masks ----
    int loff[KS];                                       // byte offset from the segment's first pixel
    v4i bw[KS][NT];
    // SAME padding: Cin 16 keeps five constant 64-bit lane masks per K-step in SGPRs and assembles a
    // load's mask with scalar ops; with five K-steps (Cin 32) those 50 SGPRs made the compiler spill
    // scalars into VGPR lanes, so there the same facts sit in one per-lane bit word per K-step
    // (bit 1 tap above, 2 below, 3 left-edge lane, 4 right-edge lane, 5 no such tap, 6 always) that
    // is ANDed with the segment's scalar flag word
    constexpr bool SMASK = CIN == 16;      // (VALU-side masks for Cin 16 too: measured 2 % slower)
    unsigned long long m_ok[KS], m_dy0[KS], m_dy2[KS], m_el[KS], m_er[KS];
    int lbits[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tap = ks * TPS + kq / LPT;
        const int sub = kq % LPT;
        const bool tok = tap < TAPS;
        const int dy = tok ? tap / 3 : 1, dx = tok ? tap % 3 : 1;
        loff[ks] = ((dy - 1) * g.W + (dx - 1) + r) * PIXB + sub * 8;
        if constexpr (SMASK) {
            m_ok[ks] = __ballot(tok);
            m_dy0[ks] = __ballot(tok && dy == 0);
            m_dy2[ks] = __ballot(tok && dy == 2);
            m_el[ks] = __ballot(tok && dx == 0 && r == 0);
            m_er[ks] = __ballot(tok && dx == 2 && r == 15);
        } else {
            lbits[ks] = (tok && dy == 0 ? 2 : 0) | (tok && dy == 2 ? 4 : 0) | (tok && dx == 0 && r == 0 ? 8 : 0) |
                        (tok && dx == 2 && r == 15 ? 16 : 0) | (tok ? 0 : 32) | 64;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int woff = tok ? ((nbase + nt * 16 + r) * TAPS + tap) * CIN + sub * 16 : (int)0x80000000;
            bw[ks][nt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, 0, 0));
        }
    }

    // ---- epilogue constants ----
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const bool has_res = e.res != nullptr;
    const bool res_f32 = has_res && e.res_store == QNN_STORE_F32;   // a float32 projection shortcut
    const float mfold = (!binary && !has_res) ? e.act_m : 1.0f;
    const float mlate = (!binary && has_res) ? e.act_m : 1.0f;
    LaneEpi ke;
    lane_epi_init<QNN_STORE_I4>(ke, e, nbase + r, r);
    FoldEpi fe[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        LaneEpi kb;
        lane_epi_init<QNN_STORE_I4>(kb, e, nbase + nt * 16 + r, r);
        fe[nt].nb = __fdiv_rn(kb.bias, e.scale);
        fe[nt].ninv = __fmul_rn(__fmul_rn(kb.inv, e.scale), mfold);
        fe[nt].nshift = __fmul_rn(kb.shift, mfold);
    }
    // after the nibble transpose this lane holds the word of value j = r & 7: segment pair member
    // (j >> 2), pixel 4*kq + (j & 3) of that segment, channels (r & 8) .. +7 of its 16-column tile
    const int jv = r & 7;
    const int out_px = 4 * kq + (jv & 3);
    const int out_cw = (nbase + (r & 8)) >> 3;            // + 2*nt

    // ---- tiles of this wave (XCD-contiguous ranges, waves interleaved) ----
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t_stride = (gridDim.x >> 3) * 4;
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_end = min((xcd + 1) * per_xcd, ntiles);
    int t = xcd * per_xcd + idx * 4 + wave;
    if (t >= t_end) return;

    // segment decode (all scalar): first pixel index and border flags packed into one word
    // (bit 0 valid, 1 top row, 2 bottom row, 3 left edge, 4 right edge) -- few live SGPRs matter
    // here: with one struct of booleans per segment the compiler spilled scalars into VGPR lanes
    auto decode = [&](int tile, int mt, int& px0) -> int {
        const int seg = tile * MT + mt;
        const bool ok = tile < t_end && seg < nsegs;
        const uint32_t row = qnn_div((uint32_t)seg, fd_spr);          // n*H + y
        const int xs = (seg - (int)row * spr) * 16;
        const int n = (int)qnn_div(row, g.fd_hp);                      // Hp == H (no pooling)
        const int yy = (int)row - n * g.H;
        px0 = (int)row * g.W + xs;
        return (ok ? 1 : 0) | (yy == 0 ? 2 : 0) | (yy == g.H - 1 ? 4 : 0) | (xs == 0 ? 8 : 0) |
               (xs + 16 == g.W ? 16 : 0);
    };
    uint2 R[KS][MT];
    auto issue = [&](int px0, int fl, int ks, int mt) {
        bool ok;
        if constexpr (SMASK) {
            unsigned long long m = m_ok[ks];
            m &= ~(((fl & 2) ? m_dy0[ks] : 0ull) | ((fl & 4) ? m_dy2[ks] : 0ull) |
                   ((fl & 8) ? m_el[ks] : 0ull) | ((fl & 16) ? m_er[ks] : 0ull));
            if (!(fl & 1)) m = 0ull;
            ok = __builtin_amdgcn_inverse_ballot_w64(m);
        } else {
            const int sbits = (fl & 0x1E) | 32 | ((fl & 1) ? 0 : 64);
            ok = (lbits[ks] & sbits) == 0;
        }
        const int voff = ok ? loff[ks] + px0 * PIXB : (int)0x80000000;   // out of range -> zeros
        R[ks][mt] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, voff, 0, 0));
    };
    auto operand = [&](const uint2& q) -> v4i {
        const uint4 v = make_uint4((q.x << 4) & 0xF0F0F0F0u, q.x & 0xF0F0F0F0u,
                                   (q.y << 4) & 0xF0F0F0F0u, q.y & 0xF0F0F0F0u);
        return __builtin_bit_cast(v4i, v);
    };
    auto bn = [&](int v, const FoldEpi& f) {
        return __fadd_rn(__fmul_rn(__fadd_rn((float)v, f.nb), f.ninv), f.nshift);
    };

    int cur_px0[MT], cur_fl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        cur_fl[mt] = decode(t, mt, cur_px0[mt]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) issue(cur_px0[mt], cur_fl[mt], ks, mt);
    }

    for (; t < t_end; t += t_stride) {
        // output word of this lane per segment pair (also the address of its shortcut word, which is
        // requested now so that the round trip hides behind the MFMA phase)
        int woff[MT / 2];
        uint32_t rw[NT][MT / 2];
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
            const int px0 = (jv >> 2) ? cur_px0[2 * mp + 1] : cur_px0[2 * mp];
            const bool sok = ((jv >> 2) ? cur_fl[2 * mp + 1] : cur_fl[2 * mp]) & 1;
            woff[mp] = sok ? ((px0 + out_px) * e.ocw + out_cw) * 4 : (int)0x80000000;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                rw[nt][mp] = (has_res && !res_f32) ? __builtin_amdgcn_raw_buffer_load_b32(rrsrc, woff[mp] + 8 * nt, 0, 0) : 0u;
        }
        // float32 shortcut: byte offset of (first pixel of the lane's row group, its channel) per segment
        int foff[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            foff[mt] = (res_f32 && (cur_fl[mt] & 1)) ? ((cur_px0[mt] + 4 * kq) * g.cout + nbase + r) * 4 : (int)0x80000000;
        v4i acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int npx0;
            const int nfl = decode(t + t_stride, mt, npx0);     // this segment of the NEXT tile
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i fa = operand(R[ks][mt]);
                issue(npx0, nfl, ks, mt);                 // the registers are free again: next tile
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (ks == 0) {
                        const v4i z = {0, 0, 0, 0};
                        acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, bw[ks][nt], z, 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, bw[ks][nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
            cur_px0[mt] = npx0;                            // (the epilogue below uses woff, computed above)
            cur_fl[mt] = nfl;
        }
        // ---- epilogue: C/D layout of 16x16: column r = output channel, rows 4*kq + i = pixels ----
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mp = 0; mp < MT / 2; ++mp) {
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = bn(acc[2 * mp + (j >> 2)][nt][j & 3], fe[nt]);
                if (res_f32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float rv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rrsrc, foff[2 * mp + (j >> 2)] + ((j & 3) * g.cout + nt * 16) * 4, 0, 0));
                        t8[j] = __fmul_rn(__fmul_rn(__fadd_rn(rv, t8[j]), e.post_scale), mlate);
                    }
                } else if (has_res) {
                    const uint32_t rt = transpose_nib8(rw[nt][mp], ke);   // nibble k = shortcut code of value k, this channel
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int code = (int)(rt << (28 - 4 * j)) >> 28;
                        const float rv = __fmul_rn((float)code, e.res_scale);
                        t8[j] = __fmul_rn(__fmul_rn(__fadd_rn(rv, t8[j]), e.post_scale), mlate);
                    }
                }
                const uint32_t P = pack_scaled<4, 8>(t8, e.act_m, binary);
                const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                __builtin_amdgcn_raw_buffer_store_b32(Wd, yrsrc, woff[mp] + 8 * nt, 0, 0);   // out of range: dropped
            }
    }
}

template <int CIN, int NT>
int launch_small(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
                 hipStream_t s) {
    const ConvGeom& g = mg.g;
    const int spr = g.W / 16;
    const long nsegs_l = (long)g.N * g.H * spr;
    const double ybytes = (double)g.N * g.H * g.W * e.ocw * 4.0;
    if (nsegs_l >= 2000000000L || ybytes >= 2.0e9) return 1;
    const int nsegs = (int)nsegs_l;
    const int ntiles = (nsegs + 3) / 4;
    const int ny = g.cout / (16 * NT);
    int gx = (((ntiles + 3) / 4 + 7) / 8) * 8;
    const int wpc = CIN == 16 ? QNN_SMALL16_WPC : QNN_SMALL32_WPC;   // resident workgroups per CU (register budget)
    const int cap = ((256 * wpc / ny + 7) / 8) * 8;
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)ny), block(256);
    const double rbytes = e.res && e.res_store == QNN_STORE_F32 ? (double)g.N * g.H * g.W * g.cout * 4.0 : ybytes;
    if (rbytes >= 2.0e9) return 1;
    hipLaunchKernelGGL((k_conv_mfma_small<CIN, NT>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y,
                       nsegs, ntiles, qnn_fastdiv((uint32_t)spr), spr, (uint32_t)ybytes, (uint32_t)rbytes);
    return 0;
}

}  // namespace

int qnn_launch_small(int cin, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                     void* y, hipStream_t s) {
    return cin == 16 ? launch_small<16, 1>(mg, e, x, w, y, s) : launch_small<32, 2>(mg, e, x, w, y, s);
}
