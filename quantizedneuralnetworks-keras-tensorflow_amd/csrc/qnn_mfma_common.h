// Shared by the MFMA translation units (qnn_mfma*.hip, qnn_first.hip): geometry, per-lane epilogue
// constants, exact float packing and the in-register transposes of packed outputs.
#pragma once
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
#ifndef QNN_DMA_NBUF
#define QNN_DMA_NBUF 3          // LDS buffers of the LDS-DMA implicit GEMM (qnn_mfma.hip): loads run NBUF-1 K-steps ahead
#endif
#ifndef QNN_DMA_PREFETCH
#define QNN_DMA_PREFETCH 1      // LDS-DMA implicit GEMM: fill the next K-step's operand registers under this step's MFMAs
#endif
#ifndef QNN_FIRST_WPS
#define QNN_FIRST_WPS 3
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));   // operand pair of the packed float32 VALU operations
typedef int v16i __attribute__((ext_vector_type(16)));

int qnn_conv_impl_pref();   // 0 auto, 1 valu, 2 mfma (qnn_api.hip)

// (outside the anonymous namespace: it appears in the signatures of the cross-TU launchers below)
struct MfmaGeom {
    ConvGeom g;
    int kc;            // 64-channel chunks per tap
    int steps;         // kh*kw*kc
    int x_pix_bytes;   // bytes per input pixel as stored
    long total_q;      // stored output pixels
    uint32_t x_bytes, w_bytes;   // sizes of the x tensor / int8 weight image (buffer descriptors)
    int ablate;                  // timing experiments only (QNN_MFMA_ABLATE): 1 = no A traffic, 2 = no B traffic
};

namespace {



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

// folded per-lane epilogue constants and exact float packing (used by the persistent kernels)
struct FoldEpi {
    float nb, ninv, nshift;   // (+-)bias, (+-)inv * m, shift * m
};

__device__ __forceinline__ float max4(float a, float b, float c, float d) {
    // v_maximum3_f32 x2 (NaN-propagating, no canonicalisation moves)
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b),
                                         __builtin_elementwise_maximum(c, d));
}

// NV pre-scaled post-BN values (t * m) of one lane -> offset-coded fields of OBITS bits.
// rint + clamp in one float add and one integer median: t + (1.5 * 2^23 + OFF) rounds to an integer (ties to even, as
// rint: the constant is even) whose float bits are kMagic + rint(t); v_med3_i32 clamps the bits to [kMagic - m,
// kMagic + m - 1] (values beyond +-2^22, infinities included, order like their bits), so the low OBITS bits are
// code + OFF.  The fields are assembled with shift-adds; the shifted copies of the constant cancel modulo 2^32.
template <int OBITS, int NV>
__device__ __forceinline__ uint32_t pack_scaled(const float* tm, float m, bool binary) {
    static_assert(NV * OBITS <= 32, "fields must fit one word");
    constexpr int OFF = 1 << (OBITS - 1);
    constexpr int kMagic = 0x4B400000 + OFF;                       // float bits of 1.5 * 2^23 + OFF
    int cb[NV];
    if (binary) {
        asm volatile("; binary_tanh codes");         // keeps this a real (uniform) branch
#pragma unroll
        for (int j = 0; j < NV; ++j) cb[j] = tm[j] > 0x1p-24f ? kMagic + 1 : kMagic - 1;
    } else {
        const int lo = kMagic - (int)m, hi = kMagic + (int)m - 1;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int bits = __float_as_int(__fadd_rn(tm[j], __int_as_float(kMagic)));
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cb[j]) : "v"(bits), "v"(lo), "v"(hi));
        }
    }
    uint32_t word = (uint32_t)cb[0], bias = (uint32_t)(kMagic - OFF);
#pragma unroll
    for (int j = 1; j < NV; ++j) {
        word += (uint32_t)cb[j] << (OBITS * j);
        bias += (uint32_t)(kMagic - OFF) << (OBITS * j);
    }
    return word - bias;
}

}  // namespace

// launchers living in their own translation units (0 = launched, 1 = shape not eligible)
int qnn_launch_first(int cin, int nt, const ConvGeom& g, const EpiArgs& e, const void* x, const float* wq,
                     void* y, hipStream_t s);
int qnn_launch_areg(int x_store, int kc, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                    void* y, hipStream_t s);
int qnn_launch_wres(int x_store, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                    void* y, hipStream_t s);
int qnn_launch_halo(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s);
int qnn_launch_areg_head(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, const qnn_weights* wd,
                         const EpiArgs& ed, float* y, hipStream_t s, const char** kname);
int qnn_launch_small(int cin, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                     void* y, hipStream_t s);
int qnn_launch_strip(int cin, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                     void* y, hipStream_t s);
int qnn_launch_strip16_lds(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s);
