// The epilogue as integer thresholds (qnn_abi.h, qnn_fold_prepare): device functions shared by the kernels that use a
// fold and by the prepare / verify / eval kernels of qnn_fold.hip -- ONE definition, so what the sweep proves is what
// the kernels execute.
//
// Per output channel the reference's chain  bias_add -> BatchNormalization -> [+ shortcut, * 0.5] -> quantized_tanh
// (models/vgg.py:16-17, models/resnet.py:59-63,127-129; float32, one rounding per operation) is a monotone step function
// of the integer accumulator.  Folded form, on the accumulator in matrix-pipe units accw = 256 * acc (both int4 operands
// are widened to code * 16) with the per-channel offset beta already added (it is the MFMA's initial accumulator):
//     u  = float(accw) * A                          v_cvt_f32_i32, v_mul_f32            (exact conversion: |accw| < 2^24)
//     T  = snorm16(u) = rint(clamp(u, -1, 1) * 32767)   v_cvt_pknorm_i16_f32, two values per instruction
//   no shortcut:   T is Q12 (code * 4096 + fraction), saturated to the 4-bit code range by the conversion itself;
//   shortcut sc:   T is Q11;  W = sat16(T + (sc + 8) * 1024);  T' = sat16(W + W)   two v_pk_add_i16 clamp per pair
//     code = T >> 12 (arithmetic): the top nibble of each 16-bit half IS the two's-complement code.
// Pairs of values live in the two halves of one register from the conversion on.
//
// Second form ("bits", mode 2; tried first by qnn_fold_prepare -- it needs the thresholds, not the whole domain, inside
// |accw + beta| < 2^22: beyond that the bit pattern leaves the constant's binade, which keeps the order and only saturates): the stored
// offset is beta + 0x4B400000, so the integer accumulator IS the bit pattern of the float 12582912 + (accw + beta), and
//     u = fma(as_float(acc), A, C),   C = float(-12582912 * A)                    one v_fma_f32, no conversion
// replaces the conversion and the multiply (the residue 12582912 * A + C is a constant the offset search absorbs).
#pragma once
#include "qnn_common.h"

#ifdef __HIPCC__
typedef short qnn_s2 __attribute__((ext_vector_type(2)));

constexpr int kFoldMagicBits = 0x4B400000;          // float bits of 12582912 = 1.5 * 2^23 (ulp 1)
constexpr double kFoldMagic = 12582912.0;

// two accumulators (offset included) -> two saturated Q12 (no shortcut) / Q11 (shortcut) values
__device__ __forceinline__ uint32_t qnn_fold_pair(int accw0, int accw1, float a0, float a1) {
    const float u0 = __fmul_rn((float)accw0, a0), u1 = __fmul_rn((float)accw1, a1);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(u0, u1));
}
// the same on accumulators that carry the magic constant ("bits" form)
__device__ __forceinline__ uint32_t qnn_fold_pair_bits(int acc0, int acc1, float a0, float a1, float c0, float c1) {
    const float u0 = __fmaf_rn(__int_as_float(acc0), a0, c0), u1 = __fmaf_rn(__int_as_float(acc1), a1, c1);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(u0, u1));
}
// Third form (mode 3): the typed image entry (QNN_STORE_U8 / QNN_STORE_F32_IMAGE first layers), whose accumulator is the
// exact integer S = sum byte * code in units of ONE (no widening, so no sub-step offset to play with): the folded form
// keeps the conversion and takes its offset in the FMA,  u = fma(float(S), A, C),  and only the rounding add, the
// integer median and the shift-adds of the packing go (v_cvt_pknorm_i16_f32 + v_perm_b32 / v_bfi_b32 instead).
__device__ __forceinline__ uint32_t qnn_fold_pair_fma(int s0, int s1, float a0, float a1, float c0, float c1) {
    const float u0 = __fmaf_rn((float)s0, a0, c0), u1 = __fmaf_rn((float)s1, a1, c1);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(u0, u1));
}
template <int MODE>
__device__ __forceinline__ uint32_t qnn_fold_pair_m(int acc0, int acc1, float a0, float a1, float c0, float c1) {
    if constexpr (MODE == 2) return qnn_fold_pair_bits(acc0, acc1, a0, a1, c0, c1);
    else return qnn_fold_pair(acc0, acc1, a0, a1);
}
// shortcut merge: s = ((sc + 8) << 10) in both halves (offset-coded shortcut codes at Q11 / 2)
__device__ __forceinline__ uint32_t qnn_fold_merge(uint32_t t, uint32_t s) {
    const qnn_s2 w = __builtin_elementwise_add_sat(__builtin_bit_cast(qnn_s2, t), __builtin_bit_cast(qnn_s2, s));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(w, w));
}
// the 4-bit codes of a pair (tests / verifier; the kernels pick the nibbles up with v_perm_b32 / v_bfi_b32)
__device__ __forceinline__ int qnn_fold_code_lo(uint32_t t) { return (int)(short)(t & 0xFFFFu) >> 12; }
__device__ __forceinline__ int qnn_fold_code_hi(uint32_t t) { return (int)t >> 28; }
// one value: accw = accumulator + stored offset; mode 1 (conversion + multiply) or 2 (bits); sc ignored without a shortcut
__device__ __forceinline__ int qnn_fold_code(int accw, float a, float c, int mode, bool res, int sc) {
    uint32_t t = mode == 2 ? qnn_fold_pair_bits(accw, accw, a, a, c, c)
                 : mode == 3 ? qnn_fold_pair_fma(accw, accw, a, a, c, c) : qnn_fold_pair(accw, accw, a, a);
    if (res) t = qnn_fold_merge(t, (uint32_t)((sc + 8) << 10) * 0x00010001u);
    return qnn_fold_code_lo(t);
}
#endif

// host side of the handle
struct qnn_fold {
    const qnn_weights* w;          // the layer it was built for
    int x_store, x_bits;
    const float* bn_inv;           // the epilogue it was built for (pointer identity is checked at launch)
    const float* bn_shift;
    int fn, act_bits, out_store;
    int has_res, res_store, res_bits;
    float post_scale;
    int cout;
    int mode;                      // 1: u = float(accw + beta) * A;  2 ("bits"): u = fma(as_float(accw + beta'), A, C);
                                   // 3 (image entry, x_store = QNN_STORE_U8): u = fma(float(S), A, C), no offset
    float* d_a;                    // [cout] slope
    int32_t* d_b;                  // [cout] offset, units of acc / 256 (mode 2: + 0x4B400000)
    float* d_c;                    // [cout] mode 2: C = float(-12582912 * A); mode 1: zeros
    int folded;                    // channels whose fold reproduced the chain on the whole domain
    long long points;
    int acc_lo, acc_hi;
};
