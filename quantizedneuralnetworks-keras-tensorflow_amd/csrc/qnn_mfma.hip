// int8 MFMA implicit-GEMM convolution for gfx950 (v_mfma_i32_32x32x32_i8).
//
// Replaces the float32 Conv2D that QuantizedConv2D.call emits
// (layers/quantized_layers.py:171-177) for layers whose activations and weights are
// stored as int8 codes (wbits/abits <= 8) or packed int4 codes (<= 4 bits): the int4
// operands are widened to int8 while they are staged into LDS (code*16 in the high
// nibble of each byte: two VALU ops per 8 codes, no sign-extension needed; the 2^8
// factor is folded into the power-of-two output scale), so both widths share one
// MFMA main loop.
//
// GEMM view: M = output pixels (N*Ho*Wo), N = cout, K = kh*kw*cin, walked one
// (tap, 64-channel chunk) per K-step.  A workgroup of WM x WN waves owns a
// (64*WM) x (64*WN) output tile; every wave a 64x64 sub-tile = 2x2 MFMA tiles, i.e.
// 4 ds_read_b128 per 4 MFMAs per 32-deep k-step.  A/B tiles are rows of 64 bytes in
// LDS, 16-byte chunks XOR-swizzled by (row>>2)&3 so that the four 16-lane groups of
// a ds_read_b128 hit disjoint banks.  Out-of-image taps are staged as zero bytes,
// which is exactly TF 'SAME' zero padding in the code domain.
//
// With max-pooling fused, the M index is ordered "pool window major, 2x2 position
// minor": rows 4g..4g+3 of the tile are one window and land in four consecutive
// accumulator registers of ONE lane (C/D layout row = (r&3) + 8*(r>>2) + 4*(lane>>5)),
// so pooling is an in-lane max -- no cross-lane traffic.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "qnn_common.h"
#ifndef QNN_SMALL16_WPC
#define QNN_SMALL16_WPC 4
#endif
#ifndef QNN_SMALL32_WPC
#define QNN_SMALL32_WPC 2
#endif
#ifndef QNN_FIRST_WPS
#define QNN_FIRST_WPS 3
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

int qnn_conv_impl_pref();   // 0 auto, 1 valu, 2 mfma (qnn_api.hip)

namespace {

struct MfmaGeom {
    ConvGeom g;
    int kc;            // 64-channel chunks per tap
    int steps;         // kh*kw*kc
    int x_pix_bytes;   // bytes per input pixel as stored
    long total_q;      // stored output pixels
    uint32_t x_bytes, w_bytes;   // sizes of the x tensor / int8 weight image (buffer descriptors)
    int ablate;                  // timing experiments only (QNN_MFMA_ABLATE): 1 = no A traffic, 2 = no B traffic
};

// ---------------------------------------------------------------------------------
// Epilogue shared by the MFMA kernels.
//
// A lane owns ONE output channel c and, per 32x32 MFMA tile, 16 rows (pixels) in
// groups of four consecutive accumulator registers.  With pooling the four registers
// of a group are one 2x2 window.  Per value the reference computes
//     t = ((v + bias) * inv) + shift ; code = clip(round(t * m))        (or sign bit)
// which is monotone in v (non-decreasing for inv >= 0, non-increasing for inv < 0),
// so max-pooling is done on the RAW conv value with max or min chosen by sign(inv):
// exact, and 4x less epilogue arithmetic.
//
// Packed outputs: a lane first packs its own codes (different pixels, same channel)
// into a register, then an in-register transpose across the lanes that share an
// output word (8 lanes for int4, 4 for int8) leaves every lane holding one complete
// word, so the tile is written with one dword store per lane.
struct LaneEpi {
    float bias, inv, shift;
    bool neg;                 // inv < 0: pool with min
    uint32_t selA, selB;      // v_perm selectors of the transposes
    uint32_t maskC, rotC;     // nibble stage (int4 only)
};

template <int OUT>
__device__ __forceinline__ void lane_epi_init(LaneEpi& k, const EpiArgs& e, int c, int li) {
    k.bias = e.bias ? e.bias[c] : 0.0f;
    k.inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
    k.shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
    k.neg = k.inv < 0.0f;
    if constexpr (OUT == QNN_STORE_I4) {
        k.selA = (li & 4) ? 0x03020706u : 0x05040100u;
        k.selB = (li & 2) ? 0x03070105u : 0x06020400u;
        k.maskC = (li & 1) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
        k.rotC = (li & 1) ? 4u : 28u;
    } else {
        k.selA = (li & 2) ? 0x03020706u : 0x05040100u;
        k.selB = (li & 1) ? 0x03070105u : 0x06020400u;
        k.maskC = 0; k.rotC = 0;
    }
}

// BN on one value, reference op order (two roundings for the BN, one for the bias)
__device__ __forceinline__ float bn_apply(float v, const LaneEpi& k) {
    return __fadd_rn(__fmul_rn(__fadd_rn(v, k.bias), k.inv), k.shift);
}
// pool a 2x2 window on raw values (see header comment)
__device__ __forceinline__ float pool_raw(const float (&v)[4], const LaneEpi& k) {
    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    const float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
    return k.neg ? mn : mx;
}
// post-BN value -> unsigned offset code (code + 2^(bits-1)); XOR-ed back to two's
// complement after packing.  rint == round_through for finite values.
template <int OBITS, bool BIN>
__device__ __forceinline__ uint32_t ucode(float t, const EpiArgs& e) {
    constexpr int OFF = 1 << (OBITS - 1);
    if constexpr (BIN) return (t > 0x1p-24f) ? (uint32_t)(OFF + 1) : (uint32_t)(OFF - 1);   // +1 iff x > 2^-24
    const float r = __builtin_amdgcn_fmed3f(rintf(__fmul_rn(t, e.act_m)), -e.act_m, e.act_m - 1.0f);
    return (uint32_t)((int)r + OFF);
}
template <int OBITS, int N>
__device__ __forceinline__ uint32_t pack_own(const float* t, const EpiArgs& e) {
    uint32_t P = 0;
    if (e.fn == QNN_FN_BINARY_TANH) {
#pragma unroll
        for (int j = 0; j < N; ++j) P |= ucode<OBITS, true>(t[j], e) << (OBITS * j);
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) P |= ucode<OBITS, false>(t[j], e) << (OBITS * j);
    }
    return P;
}

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);
}
// 8x8 nibble transpose across the 8 lanes of an octet: in: lane i holds nibbles
// M[i][0..7]; out: lane j holds M[0..7][j]
__device__ __forceinline__ uint32_t transpose_nib8(uint32_t P, const LaneEpi& k) {
    uint32_t Q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)P, 0x101F);   // lane ^ 4
    P = __builtin_amdgcn_perm(Q, P, k.selA);
    Q = dpp_xor2(P);
    P = __builtin_amdgcn_perm(Q, P, k.selB);
    Q = dpp_xor1(P);
    const uint32_t R = __builtin_amdgcn_alignbit(Q, Q, k.rotC);
    return (P & k.maskC) | (R & ~k.maskC);
}
// 4x4 byte transpose across the 4 lanes of a quad
__device__ __forceinline__ uint32_t transpose_byte4(uint32_t P, const LaneEpi& k) {
    uint32_t Q = dpp_xor2(P);
    P = __builtin_amdgcn_perm(Q, P, k.selA);
    Q = dpp_xor1(P);
    return __builtin_amdgcn_perm(Q, P, k.selB);
}

// Store NV finished (post-pool, post-BN) values of one lane.  Value j belongs to stored
// pixel qof(j) and output channel cof(j); within one call all cof(j) agree modulo 32
// with the lane index li, so the nibble/byte/bit position inside a word is li's.
template <int OUT, int NV, typename QF, typename CF>
__device__ __forceinline__ void store_values(const float (&t)[NV], const LaneEpi& k,
                                             const EpiArgs& e, int li, QF qof, CF cof,
                                             long total_q, int cout, void* __restrict__ y) {
    if constexpr (OUT == QNN_STORE_F32) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float r = t[j];
            if (e.fn == QNN_FN_BINARY_TANH) r = qnn_binary_tanh(r);
            else if (e.fn == QNN_FN_QUANTIZED_TANH) r = qnn_quantized_tanh(r, e.act_m);
            const long q = qof(j);
            // float32 surfaces are written once and are far larger than L2: non-temporal
            if (q < total_q) __builtin_nontemporal_store(r, &((float*)y)[q * cout + cof(j)]);
        }
    } else if constexpr (OUT == QNN_STORE_BIN) {
        // one ballot per value: bits of lanes 0-31 / 32-63 are the 32 channels of the
        // two pixel rows; lane (j mod 32) of each half keeps word j and stores it later
        static_assert(NV <= 32, "at most 32 values per call");
        const bool hi = (threadIdx.x & 32) != 0;
        uint32_t mine = 0;
        long myq = total_q;
        int myc = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const unsigned long long m = __ballot(t[j] > 0x1p-24f);   // binary_tanh = +1 iff x > 2^-24
            const uint32_t w = hi ? (uint32_t)(m >> 32) : (uint32_t)m;
            if (li == j) { mine = w; myq = qof(j); myc = cof(j); }
        }
        if (myq < total_q) ((uint32_t*)y)[myq * e.ocw + (myc >> 5)] = mine;
    } else if constexpr (OUT == QNN_STORE_I4) {
        static_assert(NV % 8 == 0, "int4 packing works on 8 values per lane");
#pragma unroll
        for (int g = 0; g < NV / 8; ++g) {
            uint32_t P = pack_own<4, 8>(&t[8 * g], e);
            P = transpose_nib8(P, k) ^ 0x88888888u;
            const int jl = 8 * g + (li & 7);       // after the transpose lane (li&7) holds word jl
            const long q = qof(jl);
            const int c = cof(jl);
            if (q < total_q) ((uint32_t*)y)[q * e.ocw + (c >> 3)] = P;
        }
    } else {
        static_assert(NV % 4 == 0, "int8 packing works on 4 values per lane");
#pragma unroll
        for (int g = 0; g < NV / 4; ++g) {
            uint32_t P = pack_own<8, 4>(&t[4 * g], e);
            P = transpose_byte4(P, k) ^ 0x80808080u;
            const int jl = 4 * g + (li & 3);
            const long q = qof(jl);
            const int c = cof(jl);
            if (q < total_q) ((uint32_t*)y)[q * e.ocw + (c >> 2)] = P;
        }
    }
}

// XS: QNN_STORE_I8 or QNN_STORE_I4 (storage of x; weights are always int8 bytes here)
template <int XS, int WM, int WN, int OUT, int POOL>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : WM * WN >= 8 ? 2 : 3)) void k_conv_mfma(MfmaGeom mg, EpiArgs e,
                                                           const uint8_t* __restrict__ x,
                                                           const uint8_t* __restrict__ wq8,
                                                           void* __restrict__ y) {
    constexpr int T = 64 * WM * WN;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int RPP = T / 4;               // rows staged per pass
    constexpr int NA = BM / RPP, NB = BN / RPP;
    constexpr int XCH = (XS == QNN_STORE_I8) ? 16 : 8;     // stored bytes per 16-channel chunk
    static_assert(NA >= 1 && NB >= 1, "tile too small for the workgroup");
    const ConvGeom& g = mg.g;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BUF = BM * 64, B_BUF = BN * 64;
    constexpr int B_BASE = 2 * A_BUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share
    // an L2), so give each XCD a contiguous range of M-tiles: neighbouring tiles share their
    // halo rows (and all of them the weights) in that XCD's L2.  Bijective for any grid size.
    long tile;
    {
        const unsigned nb_ = gridDim.x, b_ = blockIdx.x;
        const unsigned q_ = nb_ / 8, r_ = nb_ % 8, xcd = b_ % 8, idx = b_ / 8;
        tile = (mg.ablate & 4) ? (long)b_
                               : (long)((xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + idx);
    }
    const int nbase = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(x), 0, (int)mg.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);

    // ---- per-thread staging rows: byte offset of the receptive field's top-left
    // pixel (+ this thread's chunk) and a 9-bit "tap is inside the image" mask --------
    const int srow = tid >> 2, sch = tid & 3;
    int a_voff[NA];
    uint32_t a_mask[NA];
    int a_lds[NA], b_lds[NB], b_voff[NB];
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        const int R = srow + p * RPP;
        long q;
        int sub = 0;
        if constexpr (POOL == 2) { q = tile * (BM / 4) + (R >> 2); sub = R & 3; }
        else q = tile * BM + R;
        a_mask[p] = 0;
        a_voff[p] = 0;
        if (q < mg.total_q) {
            const uint32_t qrow = qnn_div((uint32_t)q, g.fd_wp);
            const int px = (int)((uint32_t)q - qrow * g.Wp);
            const int n = (int)qnn_div(qrow, g.fd_hp);
            const int py = (int)(qrow - (uint32_t)n * g.Hp);
            const int iy0 = (py * POOL + (sub >> 1)) * g.stride - g.pt;
            const int ix0 = (px * POOL + (sub & 1)) * g.stride - g.pl;
            a_voff[p] = ((n * g.H + iy0) * g.W + ix0) * mg.x_pix_bytes + sch * XCH;
            for (int dy = 0; dy < g.kh; ++dy)
                for (int dx = 0; dx < g.kw; ++dx)
                    if ((unsigned)(iy0 + dy) < (unsigned)g.H && (unsigned)(ix0 + dx) < (unsigned)g.W)
                        a_mask[p] |= 1u << (dy * g.kw + dx);
        }
        a_lds[p] = R * 64 + ((sch ^ ((R >> 2) & 3)) << 4);
    }
    const int w_row_bytes = g.kh * g.kw * g.cin;   // int8 bytes per cout
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int R = srow + p * RPP;
        b_voff[p] = (nbase + R) * w_row_bytes + sch * 16;
        b_lds[p] = B_BASE + R * 64 + ((sch ^ ((R >> 2) & 3)) << 4);
    }

    // ---- fragment read addresses (constant per lane) -----------------------------------
    const int li = lane & 31, lh = lane >> 5;
    int fa_addr[2][2], fb_addr[2][2];   // [tile t][kk]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ra_ = wm * 64 + t * 32 + li;
            fa_addr[t][kk] = ra_ * 64 + (((kk * 2 + lh) ^ ((ra_ >> 2) & 3)) << 4);
            const int rb_ = wn * 64 + t * 32 + li;
            fb_addr[t][kk] = B_BASE + rb_ * 64 + (((kk * 2 + lh) ^ ((rb_ >> 2) & 3)) << 4);
        }

    // uniform K-step state, advanced incrementally (no divisions in the loop)
    int s_tap = 0, s_kc = 0, s_dy = 0, s_dx = 0;
    // two staging register sets (2-deep prefetch); int4 activations stay packed (8 B per
    // 16-channel chunk) until they are written to LDS
    using araw_t = typename std::conditional<XS == QNN_STORE_I8, uint4, uint2>::type;
    araw_t raA[NA], raB[NA];
    uint4 rbA[NB], rbB[NB];
    auto stage_load = [&](araw_t (&ra)[NA], uint4 (&rb)[NB]) {
        const int xoff = (s_dy * g.W + s_dx) * mg.x_pix_bytes + s_kc * (4 * XCH);
        const int woff = (s_tap < g.kh * g.kw && !(mg.ablate & 2)) ? s_tap * g.cin + s_kc * 64
                                                                    : (int)0x40000000;   // past the end -> zeros
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const bool ok = ((a_mask[p] >> s_tap) & 1u) && !(mg.ablate & 1);
            const int voff = ok ? a_voff[p] + xoff : (int)0x80000000;   // out of range -> zeros
            if constexpr (XS == QNN_STORE_I8)
                ra[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
            else
                ra[p] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, voff, 0, 0));
        }
#pragma unroll
        for (int p = 0; p < NB; ++p)
            rb[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_voff[p], woff, 0));
        // advance (tap, kc)
        if (++s_kc == mg.kc) {
            s_kc = 0; ++s_tap;
            if (++s_dx == g.kw) { s_dx = 0; ++s_dy; }
        }
    };

    v16i acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

    auto stage_write = [&](const araw_t (&ra)[NA], const uint4 (&rb)[NB], int bufoff_a, int bufoff_b) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            uint4 v;
            if constexpr (XS == QNN_STORE_I8) v = ra[p];
            else v = make_uint4((ra[p].x << 4) & 0xF0F0F0F0u, ra[p].x & 0xF0F0F0F0u,
                                (ra[p].y << 4) & 0xF0F0F0F0u, ra[p].y & 0xF0F0F0F0u);
            *reinterpret_cast<uint4*>(smem + a_lds[p] + bufoff_a) = v;
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) *reinterpret_cast<uint4*>(smem + b_lds[p] + bufoff_b) = rb[p];
    };
    auto compute = [&](int bufoff_a, int bufoff_b) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v4i fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const v4i*>(smem + fa_addr[t][kk] + bufoff_a);
                fb[t] = *reinterpret_cast<const v4i*>(smem + fb_addr[t][kk] + bufoff_b);
            }
            if (!(mg.ablate & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
            if (!(mg.ablate & 8)) __builtin_amdgcn_s_setprio(0);
        }
    };

    // ---- main loop: one barrier per K-step; the loads of step k+2 are issued before the
    // MFMAs of step k and only waited for (counted vmcnt) after the MFMAs of step k+1 ----
    // Loads are issued UNCONDITIONALLY (a step past the end has tap >= kh*kw, whose mask
    // bit is 0 -> out-of-range offset -> the buffer load returns zeros without touching
    // memory): conditional loads make the compiler fall back to vmcnt(0).
    const int S = mg.steps;
    stage_load(raA, rbA);                          // step 0
    stage_load(raB, rbB);                          // step 1
    stage_write(raA, rbA, 0, 0);
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < S; ks += 2) {
        stage_load(raA, rbA);                      // step ks+2 -> set A
        compute(0, 0);                             // step ks   (buffer 0)
        stage_write(raB, rbB, A_BUF, B_BUF);       // step ks+1 -> buffer 1
        __syncthreads();
        stage_load(raB, rbB);                      // step ks+3 -> set B
        compute(A_BUF, B_BUF);                     // step ks+1 (buffer 1)
        stage_write(raA, rbA, 0, 0);               // step ks+2 -> buffer 0
        __syncthreads();
    }
    if (ks < S) compute(0, 0);                     // odd step count: last step sits in buffer 0

    // ---- epilogue -------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int c = nbase + wn * 64 + b * 32 + li;
        LaneEpi k;
        lane_epi_init<OUT>(k, e, c, li);
        if constexpr (POOL == 2) {
            float t[8];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // int -> float -> *2^-s is monotone: pool on the integer accumulators
                    const int i0 = acc[a][b][4 * g4], i1 = acc[a][b][4 * g4 + 1];
                    const int i2 = acc[a][b][4 * g4 + 2], i3 = acc[a][b][4 * g4 + 3];
                    const int mx = max(max(i0, i1), max(i2, i3));
                    const int mn = min(min(i0, i1), min(i2, i3));
                    t[a * 4 + g4] = bn_apply(__fmul_rn((float)(k.neg ? mn : mx), e.scale), k);
                }
            store_values<OUT, 8>(t, k, e, li,
                [&](int j) { return tile * (BM / 4) + ((wm * 64 + (j >> 2) * 32 + 8 * (j & 3) + 4 * lh) >> 2); },
                [&](int) { return c; }, mg.total_q, g.cout, y);
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = bn_apply(__fmul_rn((float)acc[a][b][r], e.scale), k);
                store_values<OUT, 16>(t, k, e, li,
                    [&](int j) { return tile * BM + wm * 64 + a * 32 + (j & 3) + 8 * (j >> 2) + 4 * lh; },
                    [&](int) { return c; }, mg.total_q, g.cout, y);
            }
        }
    }
}

// 16x16x64 variant, defined below (it uses the folded-epilogue helpers)
template <int XS, int WM, int WN, int OUT, int POOL>
__global__ void k_conv_mfma16(MfmaGeom mg, EpiArgs e, const uint8_t* __restrict__ x,
                              const uint8_t* __restrict__ wq8, void* __restrict__ y);

template <int XS, int WM, int WN, int OUT>
void launch_pool(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
                 hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const long rows = mg.total_q * (mg.g.pool == 2 ? 4 : 1);
    const dim3 grid((unsigned)((rows + BM - 1) / BM), (unsigned)(mg.g.cout / BN));
    const dim3 block(64 * WM * WN);
    const size_t lds = 2 * (BM + BN) * 64;
    // the large tiles (16 / 8 waves per workgroup, four waves per SIMD) use the 16x16x64 shape
    if constexpr (OUT != QNN_STORE_BIN && WM == 4 && WN >= 2) {
        static const int shape = getenv("QNN_MFMA_SHAPE") ? atoi(getenv("QNN_MFMA_SHAPE")) : 16;
        if (shape == 16) {
            if (mg.g.pool == 2)
                hipLaunchKernelGGL((k_conv_mfma16<XS, WM, WN, OUT, 2>), grid, block, lds, s, mg, e,
                                   (const uint8_t*)x, w, y);
            else
                hipLaunchKernelGGL((k_conv_mfma16<XS, WM, WN, OUT, 1>), grid, block, lds, s, mg, e,
                                   (const uint8_t*)x, w, y);
            return;
        }
    }
    if (mg.g.pool == 2)
        hipLaunchKernelGGL((k_conv_mfma<XS, WM, WN, OUT, 2>), grid, block, lds, s, mg, e,
                           (const uint8_t*)x, w, y);
    else
        hipLaunchKernelGGL((k_conv_mfma<XS, WM, WN, OUT, 1>), grid, block, lds, s, mg, e,
                           (const uint8_t*)x, w, y);
}

template <int XS, int WM, int WN>
int launch_out(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
               hipStream_t s) {
    switch (e.out_store) {
        case QNN_STORE_F32: launch_pool<XS, WM, WN, QNN_STORE_F32>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_BIN: launch_pool<XS, WM, WN, QNN_STORE_BIN>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_I4: launch_pool<XS, WM, WN, QNN_STORE_I4>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_I8: launch_pool<XS, WM, WN, QNN_STORE_I8>(mg, e, x, w, y, s); return 0;
    }
    return 1;
}

// folded per-lane epilogue constants and exact float packing (used by the persistent kernels)
struct FoldEpi {
    float nb, ninv, nshift;   // (+-)bias, (+-)inv * m, shift * m
};

__device__ __forceinline__ float max4(float a, float b, float c, float d) {
    // v_maximum3_f32 x2 (NaN-propagating, no canonicalisation moves)
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b),
                                         __builtin_elementwise_maximum(c, d));
}

// NV pre-scaled post-BN values (t * m) of one lane -> offset-coded fields of OBITS bits
template <int OBITS, int NV>
__device__ __forceinline__ uint32_t pack_scaled(const float* tm, float m, bool binary) {
    constexpr int FPER = 16 / OBITS;                 // fields per exact 16-bit half
    static_assert(NV % FPER == 0 && NV * OBITS <= 32, "fields must fill whole halves of one word");
    constexpr int OFFSUM = (1 << (OBITS - 1)) * (OBITS == 4 ? 0x1111 : 0x0101);
    float c[NV];
    if (binary) {
        asm volatile("; binary_tanh codes");         // keeps this a real (uniform) branch
#pragma unroll
        for (int j = 0; j < NV; ++j) c[j] = tm[j] > 0x1p-24f ? 1.0f : -1.0f;
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) c[j] = __builtin_amdgcn_fmed3f(rintf(tm[j]), -m, m - 1.0f);
    }
    uint32_t word = 0;
#pragma unroll
    for (int h = 0; h < NV / FPER; ++h) {
        float S = (float)OFFSUM;
#pragma unroll
        for (int j = 0; j < FPER; ++j) S = __fmaf_rn(c[h * FPER + j], (float)(1 << (OBITS * j)), S);
        word |= (uint32_t)S << (16 * h);
    }
    return word;
}

// ---------------------------------------------------------------------------------
// Same implicit GEMM on v_mfma_i32_16x16x64_i8.  On random operands the chip holds a lower
// clock under 32x32x32 at four waves per SIMD than under 16x16x64 (tools/ubench_shape.hip:
// 1 474 vs 1 693 T MAC/s for a bare register loop), and the large tiles of the 8-bit VGG
// layers run exactly at that occupancy.  Same staging, same LDS traffic (4 + 4 fragment reads
// of 16 bytes per step and wave), 16 MFMAs of 16 cycles instead of 8 of 32; the 64-byte rows
// use a chunk permutation that is conflict-free for this fragment shape.  The 2x2 pool window
// is still the four accumulator registers of one lane.  Outputs: float32, int4, int8.
template <int XS, int WM, int WN, int OUT, int POOL>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : WM * WN >= 8 ? 2 : 3)) void k_conv_mfma16(MfmaGeom mg, EpiArgs e,
                                                           const uint8_t* __restrict__ x,
                                                           const uint8_t* __restrict__ wq8,
                                                           void* __restrict__ y) {
    constexpr int T = 64 * WM * WN;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int RPP = T / 4;               // rows staged per pass
    constexpr int NA = BM / RPP, NB = BN / RPP;
    constexpr int XCH = (XS == QNN_STORE_I8) ? 16 : 8;     // stored bytes per 16-channel chunk
    static_assert(NA >= 1 && NB >= 1, "tile too small for the workgroup");
    const ConvGeom& g = mg.g;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // chunk swizzle of a 64-byte row: conflict-free for the 16x16x64 fragment reads (16 rows of one
    // chunk per 16 lanes; ds_read_b128 lane groups {0-3,12-15,20-27} ...) and for the staging writes
    auto swz = [](int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; };      // {0,2,3,1}[(row>>2)&3]
    constexpr int A_BUF = BM * 64, B_BUF = BN * 64;
    constexpr int B_BASE = 2 * A_BUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share
    // an L2), so give each XCD a contiguous range of M-tiles: neighbouring tiles share their
    // halo rows (and all of them the weights) in that XCD's L2.  Bijective for any grid size.
    long tile;
    {
        const unsigned nb_ = gridDim.x, b_ = blockIdx.x;
        const unsigned q_ = nb_ / 8, r_ = nb_ % 8, xcd = b_ % 8, idx = b_ / 8;
        tile = (mg.ablate & 4) ? (long)b_
                               : (long)((xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + idx);
    }
    const int nbase = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(x), 0, (int)mg.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);

    // ---- per-thread staging rows: byte offset of the receptive field's top-left
    // pixel (+ this thread's chunk) and a 9-bit "tap is inside the image" mask --------
    const int srow = tid >> 2, sch = tid & 3;
    int a_voff[NA];
    uint32_t a_mask[NA];
    int a_lds[NA], b_lds[NB], b_voff[NB];
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        const int R = srow + p * RPP;
        long q;
        int sub = 0;
        if constexpr (POOL == 2) { q = tile * (BM / 4) + (R >> 2); sub = R & 3; }
        else q = tile * BM + R;
        a_mask[p] = 0;
        a_voff[p] = 0;
        if (q < mg.total_q) {
            const uint32_t qrow = qnn_div((uint32_t)q, g.fd_wp);
            const int px = (int)((uint32_t)q - qrow * g.Wp);
            const int n = (int)qnn_div(qrow, g.fd_hp);
            const int py = (int)(qrow - (uint32_t)n * g.Hp);
            const int iy0 = (py * POOL + (sub >> 1)) * g.stride - g.pt;
            const int ix0 = (px * POOL + (sub & 1)) * g.stride - g.pl;
            a_voff[p] = ((n * g.H + iy0) * g.W + ix0) * mg.x_pix_bytes + sch * XCH;
            for (int dy = 0; dy < g.kh; ++dy)
                for (int dx = 0; dx < g.kw; ++dx)
                    if ((unsigned)(iy0 + dy) < (unsigned)g.H && (unsigned)(ix0 + dx) < (unsigned)g.W)
                        a_mask[p] |= 1u << (dy * g.kw + dx);
        }
        a_lds[p] = R * 64 + ((sch ^ swz(R)) << 4);
    }
    const int w_row_bytes = g.kh * g.kw * g.cin;   // int8 bytes per cout
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int R = srow + p * RPP;
        b_voff[p] = (nbase + R) * w_row_bytes + sch * 16;
        b_lds[p] = B_BASE + R * 64 + ((sch ^ swz(R)) << 4);
    }

    // ---- fragment read addresses (constant per lane): 16x16x64 operands = row (lane & 15),
    // 16-byte chunk (lane >> 4) of a 16-row tile; tile t of the wave's 64 rows is +1024 bytes ----
    const int cq = lane & 15, rg = lane >> 4;
    int fa_addr, fb_addr;
    {
        const int ra_ = wm * 64 + cq;
        fa_addr = ra_ * 64 + ((rg ^ swz(ra_)) << 4);
        const int rb_ = wn * 64 + cq;
        fb_addr = B_BASE + rb_ * 64 + ((rg ^ swz(rb_)) << 4);
    }

    // uniform K-step state, advanced incrementally (no divisions in the loop)
    int s_tap = 0, s_kc = 0, s_dy = 0, s_dx = 0;
    // two staging register sets (2-deep prefetch); int4 activations stay packed (8 B per
    // 16-channel chunk) until they are written to LDS
    using araw_t = typename std::conditional<XS == QNN_STORE_I8, uint4, uint2>::type;
    araw_t raA[NA], raB[NA];
    uint4 rbA[NB], rbB[NB];
    auto stage_load = [&](araw_t (&ra)[NA], uint4 (&rb)[NB]) {
        const int xoff = (s_dy * g.W + s_dx) * mg.x_pix_bytes + s_kc * (4 * XCH);
        const int woff = (s_tap < g.kh * g.kw && !(mg.ablate & 2)) ? s_tap * g.cin + s_kc * 64
                                                                    : (int)0x40000000;   // past the end -> zeros
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const bool ok = ((a_mask[p] >> s_tap) & 1u) && !(mg.ablate & 1);
            const int voff = ok ? a_voff[p] + xoff : (int)0x80000000;   // out of range -> zeros
            if constexpr (XS == QNN_STORE_I8)
                ra[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
            else
                ra[p] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, voff, 0, 0));
        }
#pragma unroll
        for (int p = 0; p < NB; ++p)
            rb[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_voff[p], woff, 0));
        // advance (tap, kc)
        if (++s_kc == mg.kc) {
            s_kc = 0; ++s_tap;
            if (++s_dx == g.kw) { s_dx = 0; ++s_dy; }
        }
    };

    v4i acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4i){0, 0, 0, 0};

    auto stage_write = [&](const araw_t (&ra)[NA], const uint4 (&rb)[NB], int bufoff_a, int bufoff_b) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            uint4 v;
            if constexpr (XS == QNN_STORE_I8) v = ra[p];
            else v = make_uint4((ra[p].x << 4) & 0xF0F0F0F0u, ra[p].x & 0xF0F0F0F0u,
                                (ra[p].y << 4) & 0xF0F0F0F0u, ra[p].y & 0xF0F0F0F0u);
            *reinterpret_cast<uint4*>(smem + a_lds[p] + bufoff_a) = v;
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) *reinterpret_cast<uint4*>(smem + b_lds[p] + bufoff_b) = rb[p];
    };
    auto compute = [&](int bufoff_a, int bufoff_b) {
        v4i fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // swz() only depends on bits 2-3 of the row, which adding 16*t does not change
            fa[t] = *reinterpret_cast<const v4i*>(smem + fa_addr + bufoff_a + t * 1024);
            fb[t] = *reinterpret_cast<const v4i*>(smem + fb_addr + bufoff_b + t * 1024);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- main loop: one barrier per K-step; the loads of step k+2 are issued before the
    // MFMAs of step k and only waited for (counted vmcnt) after the MFMAs of step k+1 ----
    // Loads are issued UNCONDITIONALLY (a step past the end has tap >= kh*kw, whose mask
    // bit is 0 -> out-of-range offset -> the buffer load returns zeros without touching
    // memory): conditional loads make the compiler fall back to vmcnt(0).
    const int S = mg.steps;
    stage_load(raA, rbA);                          // step 0
    stage_load(raB, rbB);                          // step 1
    stage_write(raA, rbA, 0, 0);
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < S; ks += 2) {
        stage_load(raA, rbA);                      // step ks+2 -> set A
        compute(0, 0);                             // step ks   (buffer 0)
        stage_write(raB, rbB, A_BUF, B_BUF);       // step ks+1 -> buffer 1
        __syncthreads();
        stage_load(raB, rbB);                      // step ks+3 -> set B
        compute(A_BUF, B_BUF);                     // step ks+1 (buffer 1)
        stage_write(raA, rbA, 0, 0);               // step ks+2 -> buffer 0
        __syncthreads();
    }
    if (ks < S) compute(0, 0);                     // odd step count: last step sits in buffer 0

    // ---- epilogue: C/D layout of 16x16: lane holds column (lane & 15) and rows 4*(lane >> 4) + r
    // of each tile -> the four registers of a tile are one 2x2 pool window of one channel ----
    static_assert(OUT != QNN_STORE_BIN, "1-bit outputs take the 32x32x32 kernel");
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    constexpr bool PACKED = OUT == QNN_STORE_I4 || OUT == QNN_STORE_I8;
    const float mfold = (PACKED && !binary) ? e.act_m : 1.0f;
    LaneEpi ke;                               // selector constants of the transposes (lane & 7 / lane & 3)
    lane_epi_init<OUT>(ke, e, nbase + wn * 64 + cq, cq);
    FoldEpi fe[4];
    bool neg[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        LaneEpi kb;
        lane_epi_init<OUT>(kb, e, nbase + wn * 64 + b * 16 + cq, cq);
        neg[b] = kb.neg;
        fe[b].nb = __fdiv_rn(kb.bias, e.scale);
        fe[b].ninv = __fmul_rn(__fmul_rn(kb.inv, e.scale), mfold);
        fe[b].nshift = __fmul_rn(kb.shift, mfold);
    }
    auto bn = [&](int v, const FoldEpi& f) {
        return __fadd_rn(__fmul_rn(__fadd_rn((float)v, f.nb), f.ninv), f.nshift);
    };
    auto finish = [&](float t) {              // float32 outputs: the activation on the unscaled value
        if (e.fn == QNN_FN_BINARY_TANH) return qnn_binary_tanh(t);
        if (e.fn == QNN_FN_QUANTIZED_TANH) return qnn_quantized_tanh(t, e.act_m);
        return t;
    };
    const int cb = nbase + wn * 64 + cq;       // channel of tile column b: cb + 16*b
    if constexpr (POOL == 2) {
        // window (a, rg): stored pixel q = tile*(BM/4) + wm*16 + a*4 + rg
        float tv[4][4];                         // [b][a]
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const v4i v = acc[a][b];
                const int mx = max(max(v[0], v[1]), max(v[2], v[3]));
                const int mn = min(min(v[0], v[1]), min(v[2], v[3]));
                tv[b][a] = bn(neg[b] ? mn : mx, fe[b]);
            }
        const long q0 = tile * (BM / 4) + wm * 16 + rg;
        if constexpr (OUT == QNN_STORE_F32) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const long q = q0 + a * 4;
                    if (q < mg.total_q) ((float*)y)[q * g.cout + cb + 16 * b] = finish(tv[b][a]);   // 64-byte segments: no nt hint
                }
        } else if constexpr (OUT == QNN_STORE_I4) {
            // one nibble transpose per pair of tile columns: value j = (b & 1) * 4 + a
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = tv[2 * bp + (j >> 2)][j & 3];
                const uint32_t P = pack_scaled<4, 8>(t8, e.act_m, binary);
                const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                const int j = cq & 7;            // after the transpose this lane holds value j of its octet
                const long q = q0 + (j & 3) * 4;
                const int c = cb + 16 * (2 * bp + (j >> 2));
                if (q < mg.total_q) ((uint32_t*)y)[q * e.ocw + (c >> 3)] = Wd;
            }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t P = pack_scaled<8, 4>(tv[b], e.act_m, binary);
                const uint32_t Wd = transpose_byte4(P, ke) ^ 0x80808080u;
                const long q = q0 + (cq & 3) * 4;
                if (q < mg.total_q) ((uint32_t*)y)[q * e.ocw + ((cb + 16 * b) >> 2)] = Wd;
            }
        }
    } else {
        // row (a, rg, r): stored pixel q = tile*BM + wm*64 + a*16 + rg*4 + r
        const long q0 = tile * BM + wm * 64 + rg * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float tv[4][4];                     // [b][r]
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) tv[b][r] = bn(acc[a][b][r], fe[b]);
            if constexpr (OUT == QNN_STORE_F32) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long q = q0 + a * 16 + r;
                        if (q < mg.total_q) ((float*)y)[q * g.cout + cb + 16 * b] = finish(tv[b][r]);
                    }
            } else if constexpr (OUT == QNN_STORE_I4) {
#pragma unroll
                for (int bp = 0; bp < 2; ++bp) {
                    float t8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t8[j] = tv[2 * bp + (j >> 2)][j & 3];
                    const uint32_t P = pack_scaled<4, 8>(t8, e.act_m, binary);
                    const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                    const int j = cq & 7;
                    const long q = q0 + a * 16 + (j & 3);
                    const int c = cb + 16 * (2 * bp + (j >> 2));
                    if (q < mg.total_q) ((uint32_t*)y)[q * e.ocw + (c >> 3)] = Wd;
                }
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t P = pack_scaled<8, 4>(tv[b], e.act_m, binary);
                    const uint32_t Wd = transpose_byte4(P, ke) ^ 0x80808080u;
                    const long q = q0 + a * 16 + (cq & 3);
                    if (q < mg.total_q) ((uint32_t*)y)[q * e.ocw + ((cb + 16 * b) >> 2)] = Wd;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Weight-resident persistent variant for short-K layers (Cout slice of 64, K = kh*kw*cin
// small enough that the slice's whole int8 weight image fits in LDS beside two A buffers).
//
// k_conv_mfma above pays, per 256x64 output tile, a cold prologue (two K-steps of global
// latency) and an epilogue nothing overlaps with; at K = 576 (9 steps) that is most of a
// tile's lifetime (measured: 22 % of the int8 matrix peak on the CIFAR B0 layer).  Here a
// workgroup stays resident, loads its 64 filters into LDS ONCE (all K-steps, same
// swizzled 64-byte rows), and walks its M-tiles as ONE continuous K-step stream: the
// two-deep register prefetch runs across tile boundaries, so the loads of the next tile
// are in flight while the current tile finishes and is stored, and a step stages only the
// A tile.  Two workgroups per CU (2 x (32 KB A + S*4 KB B)) interleave: one's epilogue
// VALU runs under the other's MFMAs.
//
// Streams: the LOAD stream (l_*) is two K-steps ahead of the COMPUTE stream (c_*); each has
// its own (tile, step) position; the per-row offsets / tap masks belong to the load stream
// and are recomputed when it enters a new tile.  Tiles are dealt so that every XCD (L2)
// owns a contiguous range of M-tiles.
template <int XS, int OUT, int POOL>
__global__ __launch_bounds__(256, 2) void k_conv_mfma_wres(MfmaGeom mg, EpiArgs e,
                                                           const uint8_t* __restrict__ x,
                                                           const uint8_t* __restrict__ wq8,
                                                           void* __restrict__ y, int ntiles) {
    constexpr int BM = 256, RPP = 64, NA = 4;
    constexpr int XCH = (XS == QNN_STORE_I8) ? 16 : 8;     // stored bytes per 16-channel chunk
    constexpr int A_BUF = BM * 64;
    constexpr int ROWTAB = 2 * A_BUF;                       // two tables of 256 x (offset, mask)
    constexpr int B_BASE = ROWTAB + 2 * BM * 8;
    constexpr int B_STEP = 64 * 64;                         // one K-step of the 64-filter slice
    constexpr int MAXS = 12;
    constexpr bool PACKED = OUT == QNN_STORE_I4 || OUT == QNN_STORE_I8;
    const ConvGeom& g = mg.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = mg.steps;
    const int ntaps = g.kh * g.kw;
    const int nbase = blockIdx.y * 64;

    // tiles of this workgroup: XCD x owns [x*per_xcd, (x+1)*per_xcd), its workgroups interleave
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t_stride = gridDim.x >> 3;                    // grid.x is a multiple of 8
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_begin = xcd * per_xcd + idx;
    const int t_end = min((xcd + 1) * per_xcd, ntiles);
    if (t_begin >= t_end) return;                           // uniform per workgroup
    const int my_tiles = (t_end - t_begin + t_stride - 1) / t_stride;
    const int total = my_tiles * S;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(x), 0, (int)mg.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);

    const int srow = tid >> 2, sch = tid & 3;
    // ---- the slice's filters -> LDS, all K-steps (issued together, written below) ----
    uint4 wreg[MAXS];
    {
        const int w_row_bytes = ntaps * g.cin;
        const int wv = (nbase + srow) * w_row_bytes + sch * 16;
#pragma unroll
        for (int st = 0; st < MAXS; ++st)
            if (st < S) {
                const int tap = st / mg.kc, kcc = st - tap * mg.kc;
                wreg[st] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrsrc, wv, tap * g.cin + kcc * 64, 0));
            }
    }

    // ---- per-lane epilogue constants (power-of-two factors folded, see k_conv_first_lds):
    //      t*m = (v + bias/scale) * (inv*scale*m) + shift*m ----
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float mfold = (PACKED && !binary) ? e.act_m : 1.0f;
    LaneEpi ke[2];
    FoldEpi fe[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        lane_epi_init<OUT>(ke[b], e, nbase + b * 32 + li, li);
        fe[b].nb = __fdiv_rn(ke[b].bias, e.scale);
        fe[b].ninv = __fmul_rn(__fmul_rn(ke[b].inv, e.scale), mfold);
        fe[b].nshift = __fmul_rn(ke[b].shift, mfold);
    }
    int lane_off = 0, lane_row = 0;       // packed outputs: word offset / local row of this lane's word
    if constexpr (OUT == QNN_STORE_I4) {
        const int jl = li & 7;
        lane_row = (POOL == 2) ? wm * 16 + 2 * (jl & 3) + lh + 8 * (jl >> 2)
                               : wm * 64 + (jl & 3) + 8 * (jl >> 2) + 4 * lh;
        lane_off = lane_row * e.ocw + ((nbase + li) >> 3);
    } else if constexpr (OUT == QNN_STORE_I8) {
        const int jl = li & 3;
        lane_row = (POOL == 2) ? wm * 16 + 2 * jl + lh : wm * 64 + jl + 4 * lh;
        lane_off = lane_row * e.ocw + ((nbase + li) >> 2);
    }

    // ---- row table: thread r computes (byte offset of the receptive field's top-left pixel,
    // 9-bit "tap inside the image" mask) of tile row r once; the four threads that stage
    // a row read it back from LDS ----
    auto row_compute = [&](int tile, int par) {
        const int R = tid;
        long q;
        int sub = 0;
        if constexpr (POOL == 2) { q = (long)tile * (BM / 4) + (R >> 2); sub = R & 3; }
        else q = (long)tile * BM + R;
        uint32_t m = 0;
        int voff = 0;
        if (tile < t_end && q < mg.total_q) {
            const uint32_t qrow = qnn_div((uint32_t)q, g.fd_wp);
            const int px = (int)((uint32_t)q - qrow * g.Wp);
            const int n = (int)qnn_div(qrow, g.fd_hp);
            const int py = (int)(qrow - (uint32_t)n * g.Hp);
            const int iy0 = (py * POOL + (sub >> 1)) * g.stride - g.pt;
            const int ix0 = (px * POOL + (sub & 1)) * g.stride - g.pl;
            voff = ((n * g.H + iy0) * g.W + ix0) * mg.x_pix_bytes;
            // taps [lo, hi) of a row / column lie inside the image; mask = outer product
            const int xlo = max(0, -ix0), xhi = min(g.kw, g.W - ix0);
            const int ylo = max(0, -iy0), yhi = min(g.kh, g.H - iy0);
            const uint32_t cm = xhi > xlo ? ((1u << xhi) - 1u) & ~((1u << xlo) - 1u) : 0u;
            for (int dy = ylo; dy < yhi; ++dy) m |= cm << (dy * g.kw);
        }
        *reinterpret_cast<uint2*>(smem + ROWTAB + par * (BM * 8) + R * 8) = make_uint2((uint32_t)voff, m);
    };
    int a_voff[NA];
    uint32_t a_mask[NA];
    auto row_fetch = [&](int par) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const uint2 v = *reinterpret_cast<const uint2*>(smem + ROWTAB + par * (BM * 8) + (srow + p * RPP) * 8);
            a_voff[p] = (int)v.x + sch * XCH;
            a_mask[p] = v.y;
        }
    };
    int a_lds[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        const int R = srow + p * RPP;
        a_lds[p] = R * 64 + ((sch ^ ((R >> 2) & 3)) << 4);
    }

    // ---- load stream ----
    int l_tile = t_begin, l_par = 0, l_tap = 0, l_kc = 0, l_dy = 0, l_dx = 0;
    using araw_t = typename std::conditional<XS == QNN_STORE_I8, uint4, uint2>::type;
    araw_t raA[NA], raB[NA];
    auto stage_load = [&](araw_t (&ra)[NA]) {
        const int xoff = (l_dy * g.W + l_dx) * mg.x_pix_bytes + l_kc * (4 * XCH);
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const bool ok = (a_mask[p] >> l_tap) & 1u;
            const int voff = ok ? a_voff[p] + xoff : (int)0x80000000;   // out of range -> zeros
            if constexpr (XS == QNN_STORE_I8)
                ra[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
            else
                ra[p] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, voff, 0, 0));
        }
        if (++l_kc == mg.kc) {
            l_kc = 0; ++l_tap;
            if (++l_dx == g.kw) { l_dx = 0; ++l_dy; }
            if (l_tap == ntaps) {                  // the load stream enters the next tile:
                l_tap = 0; l_dy = 0; l_dx = 0;     // its rows were tabulated one tile ago (at least
                l_tile += t_stride;                // one barrier back); tabulate the one after it
                l_par ^= 1;
                row_fetch(l_par);
                row_compute(l_tile + t_stride, l_par ^ 1);
            }
        }
    };
    auto stage_write = [&](const araw_t (&ra)[NA], int bufoff) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            uint4 v;
            if constexpr (XS == QNN_STORE_I8) v = ra[p];
            else v = make_uint4((ra[p].x << 4) & 0xF0F0F0F0u, ra[p].x & 0xF0F0F0F0u,
                                (ra[p].y << 4) & 0xF0F0F0F0u, ra[p].y & 0xF0F0F0F0u);
            *reinterpret_cast<uint4*>(smem + a_lds[p] + bufoff) = v;
        }
    };

    // ---- fragment read addresses: [kk]; the second 32-row tile is +2048 bytes ----
    int fa_addr[2], fb_addr[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ra_ = wm * 64 + li;
        fa_addr[kk] = ra_ * 64 + (((kk * 2 + lh) ^ ((ra_ >> 2) & 3)) << 4);
        fb_addr[kk] = B_BASE + li * 64 + (((kk * 2 + lh) ^ ((li >> 2) & 3)) << 4);
    }

    v16i acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

    // ---- compute stream ----
    int c_tile = t_begin, c_ks = 0;
    auto bn = [&](int v, const FoldEpi& f) {
        return __fadd_rn(__fmul_rn(__fadd_rn((float)v, f.nb), f.ninv), f.nshift);
    };
    auto epilogue = [&]() {
        const long tile = c_tile;
        const long row0 = tile * (POOL == 2 ? BM / 4 : BM);          // first stored pixel of the tile
        const long rem_l = mg.total_q - row0;
        const int rem = rem_l > BM ? BM : (int)rem_l;                // stored pixels left from row0
        uint32_t* ytile = reinterpret_cast<uint32_t*>(y) + row0 * e.ocw;   // packed outputs only
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = nbase + b * 32 + li;
            if constexpr (POOL == 2) {
                float t[8];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        // int -> float -> affine map is monotone: pool on the integer accumulators
                        const int i0 = acc[a][b][4 * g4], i1 = acc[a][b][4 * g4 + 1];
                        const int i2 = acc[a][b][4 * g4 + 2], i3 = acc[a][b][4 * g4 + 3];
                        const int mx = max(max(i0, i1), max(i2, i3));
                        const int mn = min(min(i0, i1), min(i2, i3));
                        t[a * 4 + g4] = bn(ke[b].neg ? mn : mx, fe[b]);
                    }
                if constexpr (OUT == QNN_STORE_I4) {
                    const uint32_t P = pack_scaled<4, 8>(t, e.act_m, binary);
                    const uint32_t Wd = transpose_nib8(P, ke[0]) ^ 0x88888888u;
                    if (lane_row < rem) ytile[lane_off + b * 4] = Wd;
                } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const uint32_t P = pack_scaled<8, 4>(&t[4 * a], e.act_m, binary);
                        const uint32_t Wd = transpose_byte4(P, ke[0]) ^ 0x80808080u;
                        if (lane_row + 8 * a < rem) ytile[lane_off + 8 * a * e.ocw + b * 8] = Wd;
                    }
                } else {
                    store_values<OUT, 8>(t, ke[b], e, li,
                        [&](int j) { return row0 + (wm * 16 + (j >> 2) * 8 + 2 * (j & 3) + lh); },
                        [&](int) { return c; }, mg.total_q, g.cout, y);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float t[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = bn(acc[a][b][r], fe[b]);
                    if constexpr (OUT == QNN_STORE_I4) {
#pragma unroll
                        for (int gq = 0; gq < 2; ++gq) {
                            const uint32_t P = pack_scaled<4, 8>(&t[8 * gq], e.act_m, binary);
                            const uint32_t Wd = transpose_nib8(P, ke[0]) ^ 0x88888888u;
                            const int dr = a * 32 + 16 * gq;
                            if (lane_row + dr < rem) ytile[lane_off + dr * e.ocw + b * 4] = Wd;
                        }
                    } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const uint32_t P = pack_scaled<8, 4>(&t[4 * gq], e.act_m, binary);
                            const uint32_t Wd = transpose_byte4(P, ke[0]) ^ 0x80808080u;
                            const int dr = a * 32 + 8 * gq;
                            if (lane_row + dr < rem) ytile[lane_off + dr * e.ocw + b * 8] = Wd;
                        }
                    } else {
                        store_values<OUT, 16>(t, ke[b], e, li,
                            [&](int j) { return row0 + wm * 64 + a * 32 + (j & 3) + 8 * (j >> 2) + 4 * lh; },
                            [&](int) { return c; }, mg.total_q, g.cout, y);
                    }
                }
            }
        }
    };
    auto compute = [&](int bufoff) {
        const int boff = c_ks * B_STEP;
        const bool first = c_ks == 0;              // first K-step of a tile: C = 0
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v4i fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const v4i*>(smem + fa_addr[kk] + bufoff + t * 2048);
                fb[t] = *reinterpret_cast<const v4i*>(smem + fb_addr[kk] + boff + t * 2048);
            }
            __builtin_amdgcn_s_setprio(1);
            if (kk == 0 && first) {
                const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], z, 0, 0, 0);
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (++c_ks == S) {                         // tile finished: store it, start the next
            epilogue();
            c_ks = 0;
            c_tile += t_stride;
        }
    };

    // ---- prologue: row tables of the first two tiles, filters into LDS, steps 0 and 1 in flight ----
    row_compute(t_begin, 0);
    row_compute(t_begin + t_stride, 1);
#pragma unroll
    for (int st = 0; st < MAXS; ++st)
        if (st < S)
            *reinterpret_cast<uint4*>(smem + B_BASE + st * B_STEP + srow * 64 +
                                      ((sch ^ ((srow >> 2) & 3)) << 4)) = wreg[st];
    __syncthreads();
    row_fetch(0);
    stage_load(raA);
    stage_load(raB);
    stage_write(raA, 0);
    __syncthreads();
    int gs = 0;
    for (; gs + 1 < total; gs += 2) {
        stage_load(raA);                           // step gs+2 -> set A
        compute(0);                                // step gs   (buffer 0)
        stage_write(raB, A_BUF);                   // step gs+1 -> buffer 1
        __syncthreads();
        stage_load(raB);                           // step gs+3 -> set B
        compute(A_BUF);                            // step gs+1 (buffer 1)
        stage_write(raA, 0);                       // step gs+2 -> buffer 0
        __syncthreads();
    }
    if (gs < total) compute(0);                    // odd number of steps: the last sits in buffer 0
}

// ---------------------------------------------------------------------------------
// 3x3, Cin = 64*KC (KC <= 2): activations straight from global memory into MFMA operand
// registers, filters resident in LDS, no barrier in the main loop.
//
// In v_mfma_i32_32x32x32_i8 lane l supplies A[row l&31][k = 16*(l>>5) .. +15]: sixteen
// consecutive channels of ONE pixel = one 8-byte (int4) / 16-byte (int8) chunk of the
// NHWC tensor.  So every lane can fetch exactly its own operand bytes with one buffer
// load per (32-row tile, 32-deep k-block): the A tile never visits LDS, nothing is
// shared between waves, and the only workgroup barrier is the one after the filter
// slice has been written to LDS.  A wave owns a 64-row x 64-filter output tile (2x2
// MFMA tiles) and walks its tiles as a continuous stream: K-steps fully unrolled (one
// step = one tap x 64 channels), three rotating operand register sets, the loads of
// step s+2 issued before the MFMAs of step s (also across the tile boundary: the next
// tile's rows are decoded at step S-2), C = 0 on a tile's first step.
// Zero padding: per tile and per (tap, 32-row tile) ONE 64-bit lane mask (ballot of "tap
// inside the image" at row-decode time) kept in SGPRs and applied with a single
// v_cndmask on the byte offset (out of range -> the buffer load returns zeros).
// Three workgroups (12 waves) per CU: one wave's staging / epilogue VALU runs under
// the other waves' MFMAs (int8 MFMA and VALU co-issue on gfx950).
template <int XS, int OUT, int POOL, int KC>
__global__ __launch_bounds__(256, (OUT == QNN_STORE_F32 ? 2 : 3)) void k_conv_mfma_areg(MfmaGeom mg, EpiArgs e,
                                                           const uint8_t* __restrict__ x,
                                                           const uint8_t* __restrict__ wq8,
                                                           void* __restrict__ y, int ntiles) {
    constexpr int TAPS = 9, S = TAPS * KC;
    static_assert(S % 3 == 0, "operand register sets rotate with period 3");
    constexpr int XCH = (XS == QNN_STORE_I8) ? 16 : 8;     // stored bytes per 16-channel chunk
    constexpr int B_STEP = 64 * 64;                         // one K-step of the 64-filter slice
    constexpr int TM = 64;                                  // rows per wave tile
    constexpr int TQ = (POOL == 2) ? TM / 4 : TM;           // stored pixels per wave tile
    constexpr bool PACKED = OUT == QNN_STORE_I4 || OUT == QNN_STORE_I8;
    const ConvGeom& g = mg.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbase = blockIdx.y * 64;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(x), 0, (int)mg.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);

    // ---- the slice's filters: all K-steps, loads issued now, written to LDS further down so
    // that the first tile's row decode and operand loads overlap their latency ----
    const int srow = tid >> 2, sch = tid & 3;
    uint4 wreg[S];
    {
        const int w_row_bytes = TAPS * g.cin;
        const int wv = (nbase + srow) * w_row_bytes + sch * 16;
#pragma unroll
        for (int st = 0; st < S; ++st)
            wreg[st] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                     wrsrc, wv, (st / KC) * g.cin + (st % KC) * 64, 0));
    }

    // tiles of this wave: XCD x owns [x*per_xcd, (x+1)*per_xcd), its waves interleave
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t_stride = (gridDim.x >> 3) * 4;              // grid.x is a multiple of 8
    const int per_xcd = (ntiles + 7) >> 3;
    const int t_end = min((xcd + 1) * per_xcd, ntiles);
    int t = xcd * per_xcd + idx * 4 + wave;

    // ---- per-lane epilogue constants (power-of-two factors folded, see k_conv_first_lds) ----
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    // residual merge (models/resnet.py:127-128, un-pooled layers only): needs the unscaled
    // post-BN value, so the code scale is applied after the merge instead of being folded
    const bool has_res = POOL == 1 && e.res != nullptr;
    const float mfold = (PACKED && !binary && !has_res) ? e.act_m : 1.0f;
    const float mlate = (PACKED && !binary && has_res) ? e.act_m : 1.0f;
    LaneEpi ke[2];
    FoldEpi fe[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        lane_epi_init<OUT>(ke[b], e, nbase + b * 32 + li, li);
        fe[b].nb = __fdiv_rn(ke[b].bias, e.scale);
        fe[b].ninv = __fmul_rn(__fmul_rn(ke[b].inv, e.scale), mfold);
        fe[b].nshift = __fmul_rn(ke[b].shift, mfold);
    }
    int lane_off = 0, lane_row = 0;       // packed outputs: word offset / local row of this lane's word
    if constexpr (OUT == QNN_STORE_I4) {
        const int jl = li & 7;
        lane_row = (POOL == 2) ? 2 * (jl & 3) + lh + 8 * (jl >> 2) : (jl & 3) + 8 * (jl >> 2) + 4 * lh;
        lane_off = lane_row * e.ocw + ((nbase + li) >> 3);
    } else if constexpr (OUT == QNN_STORE_I8) {
        const int jl = li & 3;
        lane_row = (POOL == 2) ? 2 * jl + lh : jl + 4 * lh;
        lane_off = lane_row * e.ocw + ((nbase + li) >> 2);
    }

    // ---- load stream: rows li and li+32 of the tile being fetched ----
    int a_voff[2];
    unsigned long long okm[TAPS][2];      // lanes whose tap is inside the image (SGPR pairs)
    auto row_setup = [&](int tile) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int R = mt * 32 + li;
            long q;
            int sub = 0;
            if constexpr (POOL == 2) { q = (long)tile * TQ + (R >> 2); sub = R & 3; }
            else q = (long)tile * TQ + R;
            uint32_t m = 0;
            int voff = 0;
            if (tile < t_end && q < mg.total_q) {
                const uint32_t qrow = qnn_div((uint32_t)q, g.fd_wp);
                const int px = (int)((uint32_t)q - qrow * g.Wp);
                const int n = (int)qnn_div(qrow, g.fd_hp);
                const int py = (int)(qrow - (uint32_t)n * g.Hp);
                const int iy0 = (py * POOL + (sub >> 1)) * g.stride - g.pt;
                const int ix0 = (px * POOL + (sub & 1)) * g.stride - g.pl;
                voff = ((n * g.H + iy0) * g.W + ix0) * mg.x_pix_bytes + lh * (2 * XCH);
                const int xlo = max(0, -ix0), xhi = min(3, g.W - ix0);
                const int ylo = max(0, -iy0), yhi = min(3, g.H - iy0);
                const uint32_t cm = xhi > xlo ? ((1u << xhi) - 1u) & ~((1u << xlo) - 1u) : 0u;
                const uint32_t rm = yhi > ylo ? ((1u << yhi) - 1u) & ~((1u << ylo) - 1u) : 0u;
                m = cm * ((rm & 1u) | ((rm & 2u) << 2) | ((rm & 4u) << 4));     // outer product
            }
            a_voff[mt] = voff;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) okm[tap][mt] = __ballot((m >> tap) & 1u);
        }
    };
    // one operand register set = both k-blocks of both 32-row tiles.  Lane half lh owns the
    // contiguous chunks 2*lh, 2*lh+1 of its pixel (k-block kk <-> chunk 2*lh + kk; the filter
    // fragments below use the same order): int4 -> ONE 16-byte load per 32-row tile and step
    struct aset_t { uint4 v[2][XS == QNN_STORE_I8 ? 2 : 1]; };
    aset_t R[3];
    auto issue = [&](int st, aset_t& r) {                  // st = step within the tile (compile time)
        const int tap = st / KC, kc = st % KC;
        const int xoff = ((tap / 3) * g.W + (tap % 3)) * mg.x_pix_bytes + kc * (4 * XCH);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bool ok = __builtin_amdgcn_inverse_ballot_w64(okm[tap][mt]);
            const int voff = ok ? a_voff[mt] + xoff : (int)0x80000000;   // out of range -> zeros
            r.v[mt][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
            if constexpr (XS == QNN_STORE_I8)
                r.v[mt][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff + 16, 0, 0));
        }
    };
    auto operand = [&](const aset_t& r, int mt, int kk) -> v4i {
        if constexpr (XS == QNN_STORE_I8) return __builtin_bit_cast(v4i, r.v[mt][kk]);
        else {
            const uint32_t lo = kk ? r.v[mt][0].z : r.v[mt][0].x, hi = kk ? r.v[mt][0].w : r.v[mt][0].y;
            const uint4 v = make_uint4((lo << 4) & 0xF0F0F0F0u, lo & 0xF0F0F0F0u,
                                       (hi << 4) & 0xF0F0F0F0u, hi & 0xF0F0F0F0u);
            return __builtin_bit_cast(v4i, v);
        }
    };

    // B fragment addresses: [kk]; the second 32-filter tile is +2048 bytes, a K-step +4096
    int fb_addr[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb_addr[kk] = li * 64 + (((lh * 2 + kk) ^ ((li >> 2) & 3)) << 4);

    v16i acc[2][2];
    auto bn = [&](int v, const FoldEpi& f) {
        return __fadd_rn(__fmul_rn(__fadd_rn((float)v, f.nb), f.ninv), f.nshift);
    };
    auto epilogue = [&](int tile) {
        const long row0 = (long)tile * TQ;                          // first stored pixel of the tile
        const long rem_l = mg.total_q - row0;
        const int rem = rem_l > TM ? TM : (int)rem_l;               // stored pixels left from row0
        uint32_t* ytile = reinterpret_cast<uint32_t*>(y) + row0 * e.ocw;   // packed outputs only
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = nbase + b * 32 + li;
            if constexpr (POOL == 2) {
                float tv[8];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        // int -> float -> affine map is monotone: pool on the integer accumulators
                        const int i0 = acc[a][b][4 * g4], i1 = acc[a][b][4 * g4 + 1];
                        const int i2 = acc[a][b][4 * g4 + 2], i3 = acc[a][b][4 * g4 + 3];
                        const int mx = max(max(i0, i1), max(i2, i3));
                        const int mn = min(min(i0, i1), min(i2, i3));
                        tv[a * 4 + g4] = bn(ke[b].neg ? mn : mx, fe[b]);
                    }
                if constexpr (OUT == QNN_STORE_I4) {
                    const uint32_t P = pack_scaled<4, 8>(tv, e.act_m, binary);
                    const uint32_t Wd = transpose_nib8(P, ke[0]) ^ 0x88888888u;
                    if (lane_row < rem) ytile[lane_off + b * 4] = Wd;
                } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const uint32_t P = pack_scaled<8, 4>(&tv[4 * a], e.act_m, binary);
                        const uint32_t Wd = transpose_byte4(P, ke[0]) ^ 0x80808080u;
                        if (lane_row + 8 * a < rem) ytile[lane_off + 8 * a * e.ocw + b * 8] = Wd;
                    }
                } else {
                    store_values<OUT, 8>(tv, ke[b], e, li,
                        [&](int j) { return row0 + ((j >> 2) * 8 + 2 * (j & 3) + lh); },
                        [&](int) { return c; }, mg.total_q, g.cout, y);
                }
            } else if constexpr (OUT == QNN_STORE_F32) {
                // float32 surface.  Straight-line code: the activation is chosen once per tile (three
                // copies of the loop) and only the last, partial tile guards its stores -- with a branch per
                // value the compiler spilled 348 bytes per lane and this path ran 4x slower than the
                // packed ones.  32-bit offsets from the tile's first pixel, non-temporal stores.
                float* yt = reinterpret_cast<float*>(y) + row0 * g.cout;
                const int lbase = 4 * lh * g.cout + c;
                const int lrem = rem - 4 * lh;
                auto emit = [&](auto fn_c, auto full_c) {
                    constexpr int FN = decltype(fn_c)::value;
                    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int srow = a * 32 + (r & 3) + 8 * (r >> 2);      // wave-uniform
                            float v = bn(acc[a][b][r], fe[b]);
                            if (has_res) {
                                const long q = row0 + srow + 4 * lh;
                                if (FULL || srow < lrem) v = __fmul_rn(qnn_epi_residual(v, q, c, e), mlate);
                            }
                            if constexpr (FN == QNN_FN_BINARY_TANH) v = qnn_binary_tanh(v);
                            else if constexpr (FN == QNN_FN_QUANTIZED_TANH) v = qnn_quantized_tanh(v, e.act_m);
                            if (FULL || srow < lrem) __builtin_nontemporal_store(v, &yt[lbase + srow * g.cout]);
                            // keep the scheduler from hoisting all 64 conversions and addresses of a tile
                            // in front of the first store (256 VGPRs and spills otherwise)
                            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                        }
                };
                using std::integral_constant;
                using std::true_type;
                using std::false_type;
                if (rem >= TM) {
                    if (e.fn == QNN_FN_BINARY_TANH) emit(integral_constant<int, QNN_FN_BINARY_TANH>{}, true_type{});
                    else if (e.fn == QNN_FN_QUANTIZED_TANH) emit(integral_constant<int, QNN_FN_QUANTIZED_TANH>{}, true_type{});
                    else emit(integral_constant<int, QNN_FN_NONE>{}, true_type{});
                } else {
                    if (e.fn == QNN_FN_BINARY_TANH) emit(integral_constant<int, QNN_FN_BINARY_TANH>{}, false_type{});
                    else if (e.fn == QNN_FN_QUANTIZED_TANH) emit(integral_constant<int, QNN_FN_QUANTIZED_TANH>{}, false_type{});
                    else emit(integral_constant<int, QNN_FN_NONE>{}, false_type{});
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = bn(acc[a][b][r], fe[b]);
                    if (has_res) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long q = row0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            if (q < mg.total_q) tv[r] = __fmul_rn(qnn_epi_residual(tv[r], q, c, e), mlate);
                        }
                    }
                    if constexpr (OUT == QNN_STORE_I4) {
#pragma unroll
                        for (int gq = 0; gq < 2; ++gq) {
                            const uint32_t P = pack_scaled<4, 8>(&tv[8 * gq], e.act_m, binary);
                            const uint32_t Wd = transpose_nib8(P, ke[0]) ^ 0x88888888u;
                            const int dr = a * 32 + 16 * gq;
                            if (lane_row + dr < rem) ytile[lane_off + dr * e.ocw + b * 4] = Wd;
                        }
                    } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const uint32_t P = pack_scaled<8, 4>(&tv[4 * gq], e.act_m, binary);
                            const uint32_t Wd = transpose_byte4(P, ke[0]) ^ 0x80808080u;
                            const int dr = a * 32 + 8 * gq;
                            if (lane_row + dr < rem) ytile[lane_off + dr * e.ocw + b * 8] = Wd;
                        }
                    } else {
                        store_values<OUT, 16>(tv, ke[b], e, li,
                            [&](int j) { return row0 + a * 32 + (j & 3) + 8 * (j >> 2) + 4 * lh; },
                            [&](int) { return c; }, mg.total_q, g.cout, y);
                    }
                }
            }
        }
    };

    // When a wave's tile stride covers whole images (and no tile is partial) the decoded rows
    // are the same for all of its tiles up to the image base: advance the offsets by a constant
    // instead of decoding again.
    const int img_q = g.Hp * g.Wp;
    const bool periodic = ((long)t_stride * TQ) % img_q == 0 && (long)ntiles * TQ == mg.total_q;
    const int voff_step = (int)(((long)t_stride * TQ) / img_q) * g.H * g.W * mg.x_pix_bytes;
    auto next_rows = [&](int tile) {
        if (!periodic) row_setup(tile);
        else if (tile < t_end) { a_voff[0] += voff_step; a_voff[1] += voff_step; }
        else {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) { okm[tap][0] = 0; okm[tap][1] = 0; }
        }
    };

    // ---- main stream ----
    row_setup(t);                                           // past this wave's range: all masks 0
    issue(0, R[0]);
    issue(1, R[1]);
    {
#pragma unroll
        for (int st = 0; st < S; ++st)
            *reinterpret_cast<uint4*>(smem + st * B_STEP + srow * 64 + ((sch ^ ((srow >> 2) & 3)) << 4)) = wreg[st];
    }
    __syncthreads();                                        // the only barrier
    if (t >= t_end) return;
    for (; t < t_end; t += t_stride) {
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (st == S - 2) next_rows(t + t_stride);       // the load stream enters the next tile
            issue((st + 2) % S, R[(st + 2) % 3]);
            v4i fa[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = operand(R[st % 3], mt, kk);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                v4i fb[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    fb[b] = *reinterpret_cast<const v4i*>(smem + fb_addr[kk] + st * B_STEP + b * 2048);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        if (st == 0 && kk == 0) {
                            const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                            acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a][kk], fb[b], z, 0, 0, 0);
                        } else {
                            acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a][kk], fb[b], acc[a][b], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        epilogue(t);
    }
}

template <int XS, int OUT, int KC>
void launch_areg_pool(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
                      hipStream_t s) {
    const long rows = mg.total_q * (mg.g.pool == 2 ? 4 : 1);
    const int ntiles = (int)((rows + 63) / 64);              // 64-row wave tiles
    const int ny = mg.g.cout / 64;
    int gx = (((ntiles + 3) / 4 + 7) / 8) * 8;
    // three resident workgroups per CU (two for float32 outputs: their 64 stores per tile need more registers)
    const int cap = (((OUT == QNN_STORE_F32 ? 512 : 768) / ny + 7) / 8) * 8;
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)ny), block(256);
    const size_t lds = (size_t)9 * KC * 64 * 64;
    static const bool lds_ok = [] {
        (void)hipFuncSetAttribute((const void*)k_conv_mfma_areg<XS, OUT, 2, KC>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)k_conv_mfma_areg<XS, OUT, 1, KC>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        return true;
    }();
    (void)lds_ok;
    if (mg.g.pool == 2)
        hipLaunchKernelGGL((k_conv_mfma_areg<XS, OUT, 2, KC>), grid, block, lds, s, mg, e, (const uint8_t*)x, w, y, ntiles);
    else
        hipLaunchKernelGGL((k_conv_mfma_areg<XS, OUT, 1, KC>), grid, block, lds, s, mg, e, (const uint8_t*)x, w, y, ntiles);
}

template <int XS, int KC>
int launch_areg(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
                hipStream_t s) {
    switch (e.out_store) {
        case QNN_STORE_F32: launch_areg_pool<XS, QNN_STORE_F32, KC>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_BIN: launch_areg_pool<XS, QNN_STORE_BIN, KC>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_I4: launch_areg_pool<XS, QNN_STORE_I4, KC>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_I8: launch_areg_pool<XS, QNN_STORE_I8, KC>(mg, e, x, w, y, s); return 0;
    }
    return 1;
}

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

template <int XS, int OUT>
void launch_wres_pool(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
                      hipStream_t s) {
    const long rows = mg.total_q * (mg.g.pool == 2 ? 4 : 1);
    const int ntiles = (int)((rows + 255) / 256);
    const int ny = mg.g.cout / 64;
    int gx = ((ntiles + 7) / 8) * 8;
    const int cap = ((512 / ny + 7) / 8) * 8;               // two resident workgroups per CU
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)gx, (unsigned)ny), block(256);
    const size_t lds = 2 * 256 * 64 + 2 * 256 * 8 + (size_t)mg.steps * 64 * 64;
    // more than 64 KB of dynamic LDS has to be allowed per kernel once
    static const bool lds_ok = [] {
        (void)hipFuncSetAttribute((const void*)k_conv_mfma_wres<XS, OUT, 2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)k_conv_mfma_wres<XS, OUT, 1>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        return true;
    }();
    (void)lds_ok;
    if (mg.g.pool == 2)
        hipLaunchKernelGGL((k_conv_mfma_wres<XS, OUT, 2>), grid, block, lds, s, mg, e, (const uint8_t*)x, w, y, ntiles);
    else
        hipLaunchKernelGGL((k_conv_mfma_wres<XS, OUT, 1>), grid, block, lds, s, mg, e, (const uint8_t*)x, w, y, ntiles);
}

template <int XS>
int launch_wres(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y,
                hipStream_t s) {
    switch (e.out_store) {
        case QNN_STORE_F32: launch_wres_pool<XS, QNN_STORE_F32>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_BIN: launch_wres_pool<XS, QNN_STORE_BIN>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_I4: launch_wres_pool<XS, QNN_STORE_I4>(mg, e, x, w, y, s); return 0;
        case QNN_STORE_I8: launch_wres_pool<XS, QNN_STORE_I8>(mg, e, x, w, y, s); return 0;
    }
    return 1;
}

// ---------------------------------------------------------------------------------
// Float-input first layer on the float32 matrix pipe (v_mfma_f32_32x32x2_f32).
// gfx950's f32 MFMA is bit-for-bit a k-ordered fmaf chain (one rounding per product,
// no wider accumulation), i.e. exactly the (dy,dx,c)-ordered chain the VALU kernel
// and the oracle's conv2d_device_order evaluate -- but it runs beside the VALU, which
// is left to the epilogue.  M = pixels (32 per tile), N = cout (32 per MFMA tile),
// K = 9*CIN padded to even.  Each wave keeps ALL its filters in VGPRs and walks the
// pixel tiles; no LDS, no barriers.
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CIN, int NT, int OUT, int POOL>   // NT = cout / 32
__global__ __launch_bounds__(256, (NT <= 2 ? 3 : 2)) void k_conv_first_mfma(ConvGeom g, EpiArgs e,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ wq,
                                                         void* __restrict__ y, long total_q,
                                                         long tiles, uint32_t x_bytes) {
    constexpr int K = 9 * CIN;
    constexpr int KS = (K + 1) / 2;          // MFMA k-steps of 2
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int cbase = blockIdx.y * (NT * 32);      // this block's slice of output channels

    // B operand: lane (li, lh) holds w[k = 2s+lh][cout = cbase + nt*32 + li]
    float wb[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 2 * s + lh;
            wb[nt][s] = k < K ? wq[(long)(cbase + nt * 32 + li) * K + k] : 0.0f;
        }
    LaneEpi ke[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) lane_epi_init<OUT>(ke[nt], e, cbase + nt * 32 + li, li);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
    // A operand of tile `t`: lane (li, lh) supplies x[pixel li][k = 2s+lh]; only the
    // address offset differs between the two lane halves, and both candidates are
    // wave-uniform, so the gather is KS predicated dword loads per lane.
    auto load_tile = [&](long t, float (&av)[KS]) {
        long q;
        int sub = 0;
        if constexpr (POOL == 2) { q = t * 8 + (li >> 2); sub = li & 3; }
        else q = t * 32 + li;
        const uint32_t qq = (uint32_t)(q < total_q ? q : total_q - 1);
        const uint32_t qrow = qnn_div(qq, g.fd_wp);
        const int px = (int)(qq - qrow * g.Wp);
        const int n = (int)qnn_div(qrow, g.fd_hp);
        const int py = (int)(qrow - (uint32_t)n * g.Hp);
        const int iy0 = (py * POOL + (sub >> 1)) * g.stride - g.pt;
        const int ix0 = (px * POOL + (sub & 1)) * g.stride - g.pl;
        bool inb[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                inb[dy * 3 + dx] = (unsigned)(iy0 + dy) < (unsigned)g.H && (unsigned)(ix0 + dx) < (unsigned)g.W;
        // byte offset of the receptive field's top-left pixel; taps outside the image
        // get an offset past the end of the buffer, which a raw buffer load returns as 0
        const int base4 = (((n * g.H + iy0) * g.W + ix0) * CIN) * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ke = 2 * s, ko = 2 * s + 1;
            const int te = ke / CIN, ce = ke % CIN;
            const int to = (ko < K) ? ko / CIN : 0, co = (ko < K) ? ko % CIN : 0;
            const int off_e = (((te / 3) * g.W + (te % 3)) * CIN + ce) * 4;
            const int off_o = (((to / 3) * g.W + (to % 3)) * CIN + co) * 4;
            const bool ok = lh ? (ko < K && inb[to]) : inb[te];
            const int off = lh ? off_o : off_e;
            const uint32_t voff = ok ? (uint32_t)(base4 + off) : 0xFFFFFFF0u;
            av[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)voff, 0, 0));
        }
    };

    float cur[KS], nxt[KS];
    if (wave_id < tiles) load_tile(wave_id, cur);
    for (long tile = wave_id; tile < tiles; tile += nwaves) {
        const bool more = tile + nwaves < tiles;
        if (more) load_tile(tile + nwaves, nxt);
        // ---- K-ordered MFMA chains, two 32-channel blocks at a time ----
#pragma unroll
        for (int nc = 0; nc < NT; nc += 2) {
            v16f acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], wb[nc + u][s], acc[u], 0, 0, 0);
            }
            // ---- epilogue ----
            if constexpr (POOL == 2) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float w[4] = {acc[u][4 * g4], acc[u][4 * g4 + 1], acc[u][4 * g4 + 2],
                                            acc[u][4 * g4 + 3]};
                        t[u * 4 + g4] = bn_apply(pool_raw(w, ke[nc + u]), ke[nc + u]);
                    }
                store_values<OUT, 8>(t, ke[0], e, li,
                    [&](int j) { return tile * 8 + 2 * (j & 3) + lh; },
                    [&](int j) { return cbase + (nc + (j >> 2)) * 32 + li; }, total_q, g.cout, y);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float t[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = bn_apply(acc[u][r], ke[nc + u]);
                    store_values<OUT, 16>(t, ke[nc + u], e, li,
                        [&](int j) { return tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * lh; },
                        [&](int) { return cbase + (nc + u) * 32 + li; }, total_q, g.cout, y);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
        }
    }
}

// ---------------------------------------------------------------------------------
// Same layer, operands through LDS: every wave stages the float32 patch of its 32-pixel
// tile (plus a zero halo) into a wave-private LDS tile with coalesced buffer loads, and
// the lanes fetch their MFMA A operands with ds_read_b32 at loop-invariant addresses.
//
// The f32 MFMA shares the FMA datapath with the VALU (measured: the two do not overlap,
// DESIGN.md 3.1), so every VALU instruction in this loop is paid in full.  Hence:
//   * everything that depends only on the tile index lives in SGPRs (the wave index is
//     read with readfirstlane, the tile decode is s_mul_hi arithmetic);
//   * the zero halo is a scalar 64-bit lane mask per staging load: OR of the per-border
//     masks (built once with ballots) selected by the tile's border flags, applied with
//     one v_cndmask on the buffer offset (out-of-range offset -> the load returns 0.0f);
//   * filters of channels with a negative BN scale are negated on load (exactly negating
//     the FMA chain) so pooling is v_maximum3 only; the sign is folded back into the BN
//     constants: ((-m + b) * inv) == ((m + (-b)) * (-inv)) bit for bit;
//   * for packed outputs the power-of-two code scale 2^(bits-1) is folded into inv and
//     shift (exact scaling), and the codes of one lane are assembled as an exact float
//     sum  S = sum (code_j + off) * 2^(bits*j)  (< 2^16) with one v_fma per code and one
//     v_cvt_u32 per 16 bits instead of cvt + shift + or per code.
// Tiling: POOL==2: 8 pool windows in a row = conv rows 2*py..2*py+1 x 16 columns (needs
// Wp % 8 == 0); POOL==1: 32 pixels in a row (needs W % 32 == 0): tiles never straddle
// the image edge, so no store needs a bounds check.  No barriers: the LDS tile is
// private to the wave.
template <int CIN, int NT, int OUT, int POOL>
__global__ __launch_bounds__(256, QNN_FIRST_WPS) void k_conv_first_lds(ConvGeom g, EpiArgs e,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ wq,
                                                           void* __restrict__ y, long total_q,
                                                           int tiles, int tiles_per_row,
                                                           FastDiv fd_tpr, uint32_t x_bytes) {
    constexpr int K = 9 * CIN;
    constexpr int KS = (K + 1) / 2;
    constexpr int TROWS = (POOL == 2) ? 4 : 3;        // conv rows + halo
    constexpr int TCOLS = (POOL == 2) ? 18 : 34;      // conv cols + halo
    constexpr int TE = TROWS * TCOLS * CIN;           // floats per tile
    constexpr int NJ = (TE + 63) / 64;                // staging loads per lane
    constexpr bool PACKED = OUT == QNN_STORE_I4 || OUT == QNN_STORE_I8;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* lds = reinterpret_cast<float*>(smem_f) + wv * TE;          // wave-private tile
    const int wave_id = blockIdx.x * 4 + wv;
    const int nwaves = gridDim.x * 4;
    const int cbase = blockIdx.y * (NT * 32);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float mfold = (PACKED && !binary) ? e.act_m : 1.0f;

    // ---- per-lane constants ----
    LaneEpi ke[NT];
    FoldEpi fe[NT];
    float wb[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        lane_epi_init<OUT>(ke[nt], e, cbase + nt * 32 + li, li);
        const bool flip = POOL == 2 && ke[nt].neg;
        fe[nt].nb = flip ? -ke[nt].bias : ke[nt].bias;
        fe[nt].ninv = __fmul_rn(flip ? -ke[nt].inv : ke[nt].inv, mfold);
        fe[nt].nshift = __fmul_rn(ke[nt].shift, mfold);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 2 * s + lh;
            float w = k < K ? wq[(long)(cbase + nt * 32 + li) * K + k] : 0.0f;
            wb[nt][s] = flip ? -w : w;
        }
    }
    // staging: element ej = lane + 64*j of the [TROWS][TCOLS][CIN] tile
    int st_goff[NJ];
    unsigned long long mX[NJ], mT[NJ], mB[NJ], mL[NJ], mR[NJ];   // lanes outside the tile / on each halo edge
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ej = lane + 64 * j;
        const int r = ej / (TCOLS * CIN), rem = ej - r * (TCOLS * CIN);
        const int col = rem / CIN, ch = rem - col * CIN;
        st_goff[j] = ((r * g.W + col) * CIN + ch) * 4;
        mX[j] = __ballot(ej >= TE);
        mT[j] = __ballot(r == 0);
        mB[j] = __ballot(r == TROWS - 1);
        mL[j] = __ballot(col == 0);
        mR[j] = __ballot(col == TCOLS - 1);
    }
    // operand k = 2s+lh of this lane's pixel: LDS word index relative to the tile
    int lrow, lcol;
    if constexpr (POOL == 2) { lrow = (li & 3) >> 1; lcol = 2 * (li >> 2) + (li & 1); }
    else { lrow = 0; lcol = li; }
    int op_idx[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lh;
        const int kk = k < K ? k : 0;
        const int tap = kk / CIN, ch = kk - tap * CIN;
        op_idx[s] = ((lrow + tap / 3) * TCOLS + (lcol + tap % 3)) * CIN + ch;
    }
    const bool kpad = (K & 1) && lh == 1;          // lane half 1 of the last k-step is padding
    // packed outputs: after the in-register transpose lane (li & 7) / (li & 3) of an octet /
    // quad holds one finished word; its word offset from the tile's first stored pixel
    int lane_off = 0;
    if constexpr (OUT == QNN_STORE_I4) {
        const int jl = li & 7;
        lane_off = (POOL == 2) ? (2 * (jl & 3) + lh) * e.ocw + ((cbase + (jl >> 2) * 32 + li) >> 3)
                               : ((jl & 3) + 8 * (jl >> 2) + 4 * lh) * e.ocw + ((cbase + li) >> 3);
    } else if constexpr (OUT == QNN_STORE_I8) {
        const int jl = li & 3;
        lane_off = (POOL == 2) ? (2 * jl + lh) * e.ocw + ((cbase + li) >> 2)
                               : (jl + 4 * lh) * e.ocw + ((cbase + li) >> 2);
    }

    // all scalar: t is wave-uniform
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        const uint32_t trow = qnn_div((uint32_t)t, fd_tpr);           // = n*rows + row
        const int tb = t - (int)trow * tiles_per_row;
        const int rows_per_img = (POOL == 2) ? g.Hp : g.H;
        const FastDiv& fdh = g.fd_hp;                                  // Hp == H when POOL == 1
        n = (int)qnn_div(trow, fdh);
        const int rr = (int)trow - n * rows_per_img;
        oy0 = rr * POOL;
        ox0 = tb * ((POOL == 2) ? 16 : 32);
    };
    float stg[NJ];
    auto stage_load = [&](int t) {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
        const int base4 = (((n * g.H + (oy0 - 1)) * g.W + (ox0 - 1)) * CIN) * 4;
        const bool top = oy0 == 0, bot = oy0 + (TROWS - 2) == g.H;
        const bool left = ox0 == 0, right = ox0 + (TCOLS - 2) == g.W;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned long long m = mX[j] | (top ? mT[j] : 0ull) | (bot ? mB[j] : 0ull) |
                                         (left ? mL[j] : 0ull) | (right ? mR[j] : 0ull);
            const bool halo = __builtin_amdgcn_inverse_ballot_w64(m);
            const int voff = halo ? (int)0x80000000 : base4 + st_goff[j];   // out of range -> 0.0f
            stg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, 0, 0));
        }
    };
    auto stage_write = [&](int) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + 64 * j < TE) lds[lane + 64 * j] = stg[j];
    };
    // When the wave stride is a whole number of images, a wave sees the same tile position
    // (hence the same halo lanes) in every image: the per-lane byte offsets (or the out-of-range
    // marker, which stays out of range under the additions) just advance by a constant, and no
    // tile is decoded inside the loop.  Loads past the tensor end return zeros.
    const int tiles_per_img = ((POOL == 2) ? g.Hp : g.H) * tiles_per_row;
    const bool periodic = (nwaves % tiles_per_img) == 0;
    const int img_step = nwaves / tiles_per_img;                    // images per wave stride
    const int x_step = img_step * g.H * g.W * CIN * 4;
    const long q_step = (long)img_step * ((POOL == 2) ? g.Hp * g.Wp : g.H * g.W);
    int pvoff[NJ];
    auto stage_load_next = [&]() {                                  // periodic mode: the next tile of this wave
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            stg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, pvoff[j], 0, 0));
            pvoff[j] += x_step;
        }
    };

    // Order inside one iteration (tile i): MFMAs on the operands fetched during the previous
    // iteration -> hand tile i+1 from the staging registers to LDS, fetch its operands, start
    // the global loads of tile i+2 -> epilogue and store of tile i.  The s_waitcnt vmcnt(0)
    // in front of the LDS hand-over (loads and stores share the counter on gfx9) then sits
    // AFTER a whole MFMA phase, so neither the previous store's write acknowledge nor the
    // load latency is exposed, and the operand fetch hides behind the epilogue.
    int t = wave_id;
    if (t >= tiles) return;
    float av[KS];
    auto fetch_operands = [&]() {
#pragma unroll
        for (int s = 0; s < KS; ++s) av[s] = lds[op_idx[s]];
    };
    long q_run;
    {
        int n, oy0, ox0;
        tile_origin(t, n, oy0, ox0);
        q_run = (POOL == 2) ? ((long)n * g.Hp + (oy0 >> 1)) * g.Wp + (ox0 >> 1) : ((long)n * g.H + oy0) * g.W + ox0;
        const int base4 = (((n * g.H + (oy0 - 1)) * g.W + (ox0 - 1)) * CIN) * 4;
        const bool top = oy0 == 0, bot = oy0 + (TROWS - 2) == g.H;
        const bool left = ox0 == 0, right = ox0 + (TCOLS - 2) == g.W;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned long long m = mX[j] | (top ? mT[j] : 0ull) | (bot ? mB[j] : 0ull) |
                                         (left ? mL[j] : 0ull) | (right ? mR[j] : 0ull);
            pvoff[j] = __builtin_amdgcn_inverse_ballot_w64(m) ? (int)0x80000000 : base4 + st_goff[j];
        }
    }
    if (periodic) stage_load_next(); else stage_load(t);
    stage_write(0);
    fetch_operands();
    if (periodic) stage_load_next(); else stage_load(min(t + nwaves, tiles - 1));   // unconditional (clamped)
    for (; t < tiles; t += nwaves) {
        long q_base;
        if (periodic) { q_base = q_run; q_run += q_step; }
        else {
            int n, oy0, ox0;
            tile_origin(t, n, oy0, ox0);
            // stored-pixel index of this tile's first window / pixel
            q_base = (POOL == 2) ? ((long)n * g.Hp + (oy0 >> 1)) * g.Wp + (ox0 >> 1)
                                 : ((long)n * g.H + oy0) * g.W + ox0;
        }
        uint32_t* ytile = reinterpret_cast<uint32_t*>(y) + q_base * e.ocw;   // packed outputs only
        if (kpad) av[KS - 1] = 0.0f;
#pragma unroll
        for (int nc = 0; nc < NT; nc += 2) {
            v16f acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wb[nc + u][s], acc[u], 0, 0, 0);
            if (nc + 2 >= NT) {
                __builtin_amdgcn_sched_barrier(0);
                stage_write(0);
                fetch_operands();
                if (periodic) stage_load_next(); else stage_load(min(t + 2 * nwaves, tiles - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
            auto bn = [&](float v, const FoldEpi& f) {
                return __fadd_rn(__fmul_rn(__fadd_rn(v, f.nb), f.ninv), f.nshift);
            };
            if constexpr (POOL == 2) {
                float tv[8];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        tv[u * 4 + g4] = bn(max4(acc[u][4 * g4], acc[u][4 * g4 + 1], acc[u][4 * g4 + 2],
                                                 acc[u][4 * g4 + 3]), fe[nc + u]);
                // tile row R = 8*g4 + 4*lh + s is window R/4 = 2*g4 + lh
                if constexpr (OUT == QNN_STORE_I4) {
                    const uint32_t P = pack_scaled<4, 8>(tv, e.act_m, binary);
                    ytile[lane_off + nc * 4] = transpose_nib8(P, ke[0]) ^ 0x88888888u;
                } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const uint32_t P = pack_scaled<8, 4>(&tv[4 * u], e.act_m, binary);
                        ytile[lane_off + (nc + u) * 8] = transpose_byte4(P, ke[0]) ^ 0x80808080u;
                    }
                } else {
                    store_values<OUT, 8>(tv, ke[0], e, li,
                        [&](int j) { return q_base + 2 * (j & 3) + lh; },
                        [&](int j) { return cbase + (nc + (j >> 2)) * 32 + li; }, total_q, g.cout, y);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = bn(acc[u][r], fe[nc + u]);
                    if constexpr (OUT == QNN_STORE_I4) {
#pragma unroll
                        for (int gq = 0; gq < 2; ++gq) {
                            const uint32_t P = pack_scaled<4, 8>(&tv[8 * gq], e.act_m, binary);
                            ytile[lane_off + 16 * gq * e.ocw + (nc + u) * 4] =
                                transpose_nib8(P, ke[0]) ^ 0x88888888u;
                        }
                    } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const uint32_t P = pack_scaled<8, 4>(&tv[4 * gq], e.act_m, binary);
                            ytile[lane_off + 8 * gq * e.ocw + (nc + u) * 8] =
                                transpose_byte4(P, ke[0]) ^ 0x80808080u;
                        }
                    } else {
                        store_values<OUT, 16>(tv, ke[nc + u], e, li,
                            [&](int j) { return q_base + (j & 3) + 8 * (j >> 2) + 4 * lh; },
                            [&](int) { return cbase + (nc + u) * 32 + li; }, total_q, g.cout, y);
                    }
                }
            }
        }
    }
}

template <int CIN, int NT>
int launch_first(const ConvGeom& g, const EpiArgs& e, const void* x, const float* wq, void* y,
                 hipStream_t s) {
    const long total_q = (long)g.N * g.Hp * g.Wp;
    const long rows = total_q * (g.pool == 2 ? 4 : 1);
    const long tiles = (rows + 31) / 32;
    long blocks = (tiles + 3) / 4;
    const int ny = g.cout / (NT * 32);          // channel slices (blockIdx.y)
    const long max_blocks = 256 * 4 / ny;       // persistent: ~4 blocks (16 waves) per CU in total
    if (blocks > max_blocks) blocks = max_blocks;
    const dim3 grid((unsigned)blocks, (unsigned)ny), block(256);
    const float* xf = (const float*)x;
    const double xb = (double)g.N * g.H * g.W * CIN * 4.0;
    if (xb >= 2.0e9) return 1;                  // 31-bit buffer offsets
    const uint32_t x_bytes = (uint32_t)xb;
    static const int no_lds = getenv("QNN_FIRST_GATHER") ? atoi(getenv("QNN_FIRST_GATHER")) : 0;
    const bool lds_ok = !no_lds && NT == 2 && g.stride == 1 && g.pt == 1 && g.pl == 1 &&
                        ((g.pool == 2 && (g.Wp % 8) == 0 && (g.H % 2) == 0 && (g.W % 2) == 0) ||
                         (g.pool == 1 && (g.W % 32) == 0));
    if (lds_ok) {
        const int tpr = g.pool == 2 ? g.Wp / 8 : g.W / 32;
        const int rows = g.pool == 2 ? g.Hp : g.H;
        const long ntiles = (long)g.N * rows * tpr;
        if (ntiles < 2.0e9) {
            long lblocks = (ntiles + 3) / 4;
            const long lmax = 256 * QNN_FIRST_WPS / ny;   // persistent: QNN_FIRST_WPS waves per SIMD
            if (lblocks > lmax) lblocks = lmax;
            const dim3 lgrid((unsigned)lblocks, (unsigned)ny);
            const FastDiv fd_tpr = qnn_fastdiv((uint32_t)tpr);
            const size_t lds_bytes = (size_t)4 * ((g.pool == 2 ? 4 * 18 : 3 * 34) * CIN) * 4;   // one tile per wave
#define FIRST_LDS_CASE(OUT)                                                                      \
            if (e.out_store == OUT) {                                                            \
                if (g.pool == 2)                                                                 \
                    hipLaunchKernelGGL((k_conv_first_lds<CIN, NT, OUT, 2>), lgrid, block, lds_bytes, s, g, e, xf, wq, y, total_q, (int)ntiles, tpr, fd_tpr, x_bytes); \
                else                                                                             \
                    hipLaunchKernelGGL((k_conv_first_lds<CIN, NT, OUT, 1>), lgrid, block, lds_bytes, s, g, e, xf, wq, y, total_q, (int)ntiles, tpr, fd_tpr, x_bytes); \
                return 0;                                                                        \
            }
            FIRST_LDS_CASE(QNN_STORE_F32)
            FIRST_LDS_CASE(QNN_STORE_BIN)
            FIRST_LDS_CASE(QNN_STORE_I4)
            FIRST_LDS_CASE(QNN_STORE_I8)
#undef FIRST_LDS_CASE
        }
    }
#define FIRST_CASE(OUT)                                                                      \
    if (e.out_store == OUT) {                                                                \
        if (g.pool == 2)                                                                     \
            hipLaunchKernelGGL((k_conv_first_mfma<CIN, NT, OUT, 2>), grid, block, 0, s, g, e, xf, wq, y, total_q, tiles, x_bytes); \
        else                                                                                 \
            hipLaunchKernelGGL((k_conv_first_mfma<CIN, NT, OUT, 1>), grid, block, 0, s, g, e, xf, wq, y, total_q, tiles, x_bytes); \
        return 0;                                                                            \
    }
    FIRST_CASE(QNN_STORE_F32)
    FIRST_CASE(QNN_STORE_BIN)
    FIRST_CASE(QNN_STORE_I4)
    FIRST_CASE(QNN_STORE_I8)
#undef FIRST_CASE
    return 1;
}

// int8 weight image for the I4 path: every packed word (8 nibbles) -> two words of
// (code*16) bytes in the same even/odd order the activation staging produces
__global__ __launch_bounds__(256) void k_expand_i4_weights(const uint32_t* __restrict__ packed,
                                                           uint32_t* __restrict__ out, int words) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) {
        const uint32_t p = packed[i];
        out[2 * i] = (p << 4) & 0xF0F0F0F0u;
        out[2 * i + 1] = p & 0xF0F0F0F0u;
    }
}

}  // namespace

// Build the int8 weight image the MFMA kernel reads (called from qnn_prepack_weights).
int qnn_mfma_prepare_weights(qnn_weights* w, hipStream_t s) {
    w->d_mfma = nullptr;
    const bool small = w->store == QNN_STORE_I4 && (w->cin == 16 || w->cin == 32) && (w->cout % 16) == 0 &&
                       w->kh == 3 && w->kw == 3;
    if (!small && (w->cin % 64 != 0 || w->cout % 64 != 0)) return QNN_OK;
    if (w->store == QNN_STORE_I8) {
        w->d_mfma = (uint8_t*)w->d_packed;      // int8 codes, natural channel order
        return QNN_OK;
    }
    if (w->store != QNN_STORE_I4) return QNN_OK;
    const int words = w->cout * w->kwords;
    QNN_HIP(hipMalloc(&w->d_mfma_own, (size_t)words * 8));
    int grid = (words + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_expand_i4_weights, dim3(grid), dim3(256), 0, s, w->d_packed,
                       (uint32_t*)w->d_mfma_own, words);
    w->d_mfma = (uint8_t*)w->d_mfma_own;
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

// returns 0 if launched, 1 if this shape is not eligible
int qnn_try_launch_mfma(const ConvGeom& g, const EpiArgs& e, int x_store, const void* x,
                        const qnn_weights* w, void* y, hipStream_t s, char* name, size_t name_len) {
    if (x_store == QNN_STORE_F32) {
        // float-input first layer on the f32 matrix pipe
        if (e.res) return 1;
        if (g.kh != 3 || g.kw != 3 || (g.cin != 1 && g.cin != 3)) return 1;
        if (g.cout != 64 && g.cout != 128 && g.cout != 256) return 1;
        const int pw = e.out_store == QNN_STORE_F32 ? 1 : qnn_per_word(e.out_store);
        if (g.cout % pw != 0) return 1;
        snprintf(name, name_len, "mfma_f32_first_cin%d", g.cin);
        // the LDS-staged kernel works on 64-filter slices (blockIdx.y): prefer it whenever its tiling applies
        const bool lds_shape = g.stride == 1 && g.pt == 1 && g.pl == 1 &&
                               ((g.pool == 2 && (g.Wp % 8) == 0 && (g.H % 2) == 0 && (g.W % 2) == 0) ||
                                (g.pool == 1 && (g.W % 32) == 0));
        if (lds_shape && !getenv("QNN_FIRST_GATHER"))
            return g.cin == 3 ? launch_first<3, 2>(g, e, x, w->d_wq, y, s) : launch_first<1, 2>(g, e, x, w->d_wq, y, s);
        if (g.cout == 64)
            return g.cin == 3 ? launch_first<3, 2>(g, e, x, w->d_wq, y, s) : launch_first<1, 2>(g, e, x, w->d_wq, y, s);
        if (g.cout == 128)
            return g.cin == 3 ? launch_first<3, 4>(g, e, x, w->d_wq, y, s) : launch_first<1, 4>(g, e, x, w->d_wq, y, s);
        return g.cin == 3 ? launch_first<3, 4>(g, e, x, w->d_wq, y, s) : launch_first<1, 4>(g, e, x, w->d_wq, y, s);
    }
    if (!w->d_mfma) return 1;
    if (x_store != QNN_STORE_I8 && x_store != QNN_STORE_I4) return 1;
    // small-channel 3x3 int4 layers: both operands in registers
    if (x_store == QNN_STORE_I4 && w->store == QNN_STORE_I4 && (g.cin == 16 || g.cin == 32) && g.kh == 3 &&
        g.kw == 3 && g.stride == 1 && g.pt == 1 && g.pl == 1 && g.pool == 1 && (g.W % 16) == 0 &&
        e.out_store == QNN_STORE_I4 && (g.cout % (g.cin == 16 ? 16 : 32)) == 0 &&
        (!e.res || (e.res_store == QNN_STORE_I4 && e.res_cw == e.ocw) ||
         (e.res_store == QNN_STORE_F32 && e.res_cw == g.cout)) && !getenv("QNN_MFMA_SMALL_OFF")) {
        MfmaGeom ms;
        ms.g = g; ms.kc = 1; ms.steps = 0; ms.x_pix_bytes = g.cin / 2;
        ms.total_q = (long)g.N * g.H * g.W;
        const double xb_ = (double)g.N * g.H * g.W * ms.x_pix_bytes, wb_ = (double)g.cout * 9 * g.cin;
        if (xb_ < 2.0e9 && wb_ < 2.0e9) {
            ms.x_bytes = (uint32_t)xb_; ms.w_bytes = (uint32_t)wb_; ms.ablate = 0;
            EpiArgs es = e;
            es.scale = e.scale * (1.0f / 256.0f);            // both operands carry *16
            snprintf(name, name_len, "mfma_i4_small_c%d", g.cin);
            const int rc_ = g.cin == 16 ? launch_small<16, 1>(ms, es, x, w->d_mfma, y, s)
                                        : launch_small<32, 2>(ms, es, x, w->d_mfma, y, s);
            if (rc_ == 0) return 0;
        }
    }
    if (g.cin % 64 != 0 || g.cout % 64 != 0) return 1;
    const int pw = e.out_store == QNN_STORE_F32 ? 1 : qnn_per_word(e.out_store);
    if (g.cout % pw != 0) return 1;
    MfmaGeom mg;
    mg.g = g;
    mg.kc = g.cin / 64;
    mg.steps = g.kh * g.kw * mg.kc;
    mg.x_pix_bytes = x_store == QNN_STORE_I8 ? g.cin : g.cin / 2;
    mg.total_q = (long)g.N * g.Hp * g.Wp;
    const double xb = (double)g.N * g.H * g.W * mg.x_pix_bytes;
    const double wb = (double)g.cout * g.kh * g.kw * g.cin;
    if (xb >= 2.0e9 || wb >= 2.0e9) return 1;          // 31-bit buffer offsets
    mg.x_bytes = (uint32_t)xb;
    mg.w_bytes = (uint32_t)wb;
    static const int ablate = getenv("QNN_MFMA_ABLATE") ? atoi(getenv("QNN_MFMA_ABLATE")) : 0;
    mg.ablate = ablate;
    EpiArgs e2 = e;
    if (x_store == QNN_STORE_I4) e2.scale = e.scale * (1.0f / 256.0f);   // both operands carry *16
    // tile shape: waves along M x waves along N (64x64 per wave)
    static const char* tile_env = getenv("QNN_MFMA_TILE");
    int wm_ = 4, wn_ = 1;
    if ((g.cout % 256) == 0) { wm_ = 4; wn_ = 4; }          // measured: 256x256 > 128x256 > 256x128 > 128x128
    else if ((g.cout % 128) == 0) { wm_ = 4; wn_ = 2; }
    if (tile_env && strlen(tile_env) == 3) {
        const int em = tile_env[0] - '0', en = tile_env[2] - '0';
        if (em >= 1 && en >= 1 && (g.cout % (64 * en)) == 0) { wm_ = em; wn_ = en; }
    }
    // short-K layers with one 64-filter slice: weight-resident persistent kernel
    static const int wres_env = getenv("QNN_MFMA_WRES") ? atoi(getenv("QNN_MFMA_WRES")) : -1;
    const long rows_ = mg.total_q * (g.pool == 2 ? 4 : 1);
    const bool wres_fit = mg.steps <= 12 && rows_ < 2000000000L;
    const bool wres = wres_env == 0 ? false : wres_env == 1 ? wres_fit : (wres_fit && g.cout == 64 && !tile_env);
    // 3x3, Cin <= 128: operands straight into registers (QNN_MFMA_AREG=0 disables, =1 also for Cout > 64)
    static const int areg_env = getenv("QNN_MFMA_AREG") ? atoi(getenv("QNN_MFMA_AREG")) : -1;
    const bool areg_fit = g.kh == 3 && g.kw == 3 && mg.kc <= 2 && rows_ < 2000000000L;
    const bool areg = areg_env == 0 ? false : areg_env == 1 ? areg_fit : (areg_fit && g.cout == 64 && !tile_env && wres_env < 0);
    if (e.res && !(areg && g.pool == 1)) return 1;          // the other MFMA kernels have no residual epilogue
    if (areg) {
        snprintf(name, name_len, "mfma_%s_areg64x64", x_store == QNN_STORE_I8 ? "i8" : "i4");
        if (x_store == QNN_STORE_I8)
            return mg.kc == 1 ? launch_areg<QNN_STORE_I8, 1>(mg, e2, x, w->d_mfma, y, s)
                              : launch_areg<QNN_STORE_I8, 2>(mg, e2, x, w->d_mfma, y, s);
        return mg.kc == 1 ? launch_areg<QNN_STORE_I4, 1>(mg, e2, x, w->d_mfma, y, s)
                          : launch_areg<QNN_STORE_I4, 2>(mg, e2, x, w->d_mfma, y, s);
    }
    if (wres) {
        snprintf(name, name_len, "mfma_%s_wres256x64", x_store == QNN_STORE_I8 ? "i8" : "i4");
        return x_store == QNN_STORE_I8 ? launch_wres<QNN_STORE_I8>(mg, e2, x, w->d_mfma, y, s)
                                       : launch_wres<QNN_STORE_I4>(mg, e2, x, w->d_mfma, y, s);
    }
    snprintf(name, name_len, "mfma_%s_%dx%d", x_store == QNN_STORE_I8 ? "i8" : "i4", 64 * wm_, 64 * wn_);
#define TILE_CASE(XS_, M_, N_) \
    if (x_store == XS_ && wm_ == M_ && wn_ == N_) return launch_out<XS_, M_, N_>(mg, e2, x, w->d_mfma, y, s);
    TILE_CASE(QNN_STORE_I8, 4, 1) TILE_CASE(QNN_STORE_I8, 2, 2) TILE_CASE(QNN_STORE_I8, 4, 2)
    TILE_CASE(QNN_STORE_I8, 2, 4) TILE_CASE(QNN_STORE_I8, 4, 4)
    TILE_CASE(QNN_STORE_I4, 4, 1) TILE_CASE(QNN_STORE_I4, 2, 2) TILE_CASE(QNN_STORE_I4, 4, 2)
    TILE_CASE(QNN_STORE_I4, 2, 4) TILE_CASE(QNN_STORE_I4, 4, 4)
#undef TILE_CASE
    return 1;
}
