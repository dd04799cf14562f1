// First layer on the dataset's own bytes: x_store = QNN_STORE_U8, value = code / 255 (utils/load_data.py:40 forms the
// reference's float32 images exactly so).  Dispatch: conv_forward (qnn_conv.hip); any other shape with U8 input takes
// k_conv_generic, which implements the same arithmetic.
//
// With the integer weight codes k = w * 2^wshift the layer is an INTEGER problem with one offset:
//     s = code - 128  (one XOR with 0x80: a signed byte),     sum_k k s  +  128 sum_k k  =  sum_k k code  =  S
// -- ONE pass of v_mfma_i32_16x16x64_i8 per 16 positions x 16 filters.  A zero-padding tap is code 0, i.e. the byte
// -128, so the halo converts like any other pixel and sum_k k runs over all 27 taps regardless of the border: the
// constant 128 sum k is produced by the MFMA itself, in the K-block a 3x3x(3+1) window leaves empty (A bytes -128 from a
// constant LDS block, B bytes summing to -sum k).  The accumulator IS S.
// Everything the launch fuses behind the sum is one float32 FMA per value, t = fma(S, A[c], B[c]) (qnn_abi.h,
// qnn_conv2d_forward): the three digit passes, the 64 shift-adds and the four-rounding epilogue of the float-input
// fixed-point variant (qnn_first_fixed.hip) are gone, and the input is a quarter of the bytes.
//
// Layout (as qnn_first_fixed.hip): a wave walks a strip of 16 conv columns down the image two conv rows at a time.
// MFMA rows = 16 positions = 4 pool windows x (2 x 2), columns = 16 filters, K-block kq = tap row dy: 3 taps x
// (3 channels + 1 pad byte) + 4 bytes meeting zero weights; K-block 3 = the offset term.  Bytes are staged once
// (108 per step, two per lane) into a wave-private LDS ring of four input rows of 4-byte pixels.  In the C/D layout a
// lane holds the four positions of ONE window for one filter: pooling is an in-lane max on the integers (filters of
// channels with negative BN scale are negated, A[c] with them).
#include "qnn_mfma_common.h"
#include "qnn_fold.h"

namespace {

constexpr int kRowPitchU = 40;                 // words per LDS row (19 pixels touched; 40 == 8 mod 32)
constexpr int kRingU = 4 * kRowPitchU;         // ring of four input rows
constexpr int kConstU = kRingU;                // four words 0x80808080: the A operand of K-block 3
constexpr int kWaveLdsU = kRingU + 4;          // words per wave (a multiple of four: 16-byte aligned)

// F32IN (opt-in, qnn_set_option("first_image", 1)): the same kernel for float32 inputs that ARE image bytes / 255
// (utils/load_data.py:40).  A staged value x is read as the byte k = rint(255 x) when |255 x - k| <= 2^-15 and
// 0 <= k <= 255 -- true for every float32 quotient k/255 (the product is off by < 2^-16) -- and raises the layer's
// domain flag otherwise (qnn_weights_check), exactly like the fixed-point variant.  An accepted x is within 1.5e-7 of
// k/255, so the result is within 27 * 1.5e-7 + one rounding of the real convolution of the floats: inside the 1e-5
// contract for anything the check lets through, and the typed entry's exact result for real images.
// Five instructions per element: t + 2^23 rounds t to an integer k (ties to even, as rint) and leaves k in the low bits of
// the sum; the distance is taken to k CLAMPED to [0, 255], so an integer outside the byte range fails the same test as
// a fraction does (NaN and infinities fail it too: every comparison with NaN is false).
__device__ __forceinline__ uint32_t image_byte(float x, bool& bad) {
    const float t = __fmul_rn(x, 255.0f);
    const float u = __fadd_rn(t, 8388608.0f);
    const float r = __builtin_amdgcn_fmed3f(__fsub_rn(u, 8388608.0f), 0.0f, 255.0f);
    bad |= !(fabsf(__fsub_rn(t, r)) <= 0x1p-15f);
    return __float_as_uint(u);                      // low byte = k for every accepted x; only that byte is stored
}

// (QNN_STORE_I4, 2, BIN): the fused pipeline, quantized_tanh / binary_tanh codes;  (QNN_STORE_F32, 1, false): the layer
// behind the float32 surface (any fn)
// FOLD (round 4): the epilogue of the pooled int4 form as qnn_fold.h's mode 3 -- u = fma(float(S), A2, C2), one
// v_cvt_pknorm_i16_f32 per pair, two v_perm_b32 + shift + v_bfi_b32 per eight values -- with A2 / C2 from a fold handle that
// qnn_fold_prepare accepted only after it reproduced  clip(rint(fma(float(S), A, B)))  on every S the filter can produce.
template <int OUT, int POOL, bool BIN, bool F32IN, bool FOLD = false>
__global__ __launch_bounds__(256, 4) void k_conv_first_u8(ConvGeom g, EpiArgs e, const void* __restrict__ x,
                                                           const float* __restrict__ wq, void* __restrict__ y,
                                                           int ntasks, int spr, FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                           int rc, uint32_t img_x, float wscale, float D,
                                                           uint32_t* __restrict__ domain_flag) {
    extern __shared__ __attribute__((aligned(16))) char smem_u8[];
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem_u8) + wave * kWaveLdsU;
    uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
    uint4* tab = reinterpret_cast<uint4*>(smem_u8 + 4 * kWaveLdsU * 4);      // [filter block][lane][2]
    for (int i = lane; i < kWaveLdsU; i += 64) lds[i] = 0x80808080u;        // code 0 everywhere; the constant block

    // ---- filters: B operand of block nt = filter nt*16 + r, K-block kq.  Wave nt prepares block nt for the workgroup ----
    {
        const int c = wave * 16 + r;
        const float bias = e.bias ? e.bias[c] : 0.0f;
        const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
        const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
        const bool flip = POOL == 2 && inv < 0.0f;               // pool with max only: negate the filter and A[c]
        const float* wrow = wq + (size_t)c * 27 + (kq < 3 ? kq : 2) * 9;
        int part = 0;
        uint32_t wd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int code = (int)rintf(__fmul_rn(wrow[i], wscale));
            if (flip) code = -code;
            if (kq == 3) code = 0;
            part += code;
            wd[i / 3] |= (uint32_t)(code & 0xFF) << (8 * (i % 3));
        }
        part += __shfl_xor(part, 16);                            // sum of the 27 codes of filter c (all four K-block
        part += __shfl_xor(part, 32);                            // lanes end up with it)
        if (kq == 3) {
            // sixteen bytes that sum to -part: against A bytes of -128 they contribute +128 * sum k
            int rem = -part;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int b = min(max(rem, -127), 127);
                rem -= b;
                wd[j >> 2] |= (uint32_t)(b & 0xFF) << (8 * (j & 3));
            }
        }
        // the affine map behind S (qnn_abi.h): float64 from the float32 constants, one rounding each
        const double m = e.fn == QNN_FN_QUANTIZED_TANH ? (double)e.act_m : 1.0;
        float A = (float)((double)inv * m / (double)D);
        float B = (float)(((double)bias * (double)inv + (double)shift) * m);
        if constexpr (FOLD) { A = e.fold_a[c]; B = e.fold_c[c]; }
        if (flip) A = -A;
        tab[(wave * 64 + lane) * 2] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        tab[(wave * 64 + lane) * 2 + 1] = make_uint4(__float_as_uint(A), __float_as_uint(B), 0u, 0u);
    }
    __syncthreads();
    v4i bw[4];
    float fa[4], fb[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const uint4 t0 = tab[(nt * 64 + lane) * 2], t1 = tab[(nt * 64 + lane) * 2 + 1];
        bw[nt] = __builtin_bit_cast(v4i, t0);
        fa[nt] = __uint_as_float(t1.x); fb[nt] = __uint_as_float(t1.y);
    }
    LaneEpi ke;
    lane_epi_init<QNN_STORE_I4>(ke, e, r, r);
    constexpr int kMagicBits = 0x4B400008;                       // 1.5 * 2^23 + 8: see qnn_mfma_strip.hip
    const float magic = __int_as_float(kMagicBits);
    const int code_lo = kMagicBits - (int)e.act_m, code_hi = kMagicBits + (int)e.act_m - 1;
    // ---- operand addresses (dword index): position m = r: window w = r >> 2, (py, px) = bits of r.  A step works on
    // conv rows yy0 = 2*rp and yy0 + 1; with rp0 even the ring slot of input row yy0 + py + dy - 1 is
    // (2*(rp & 1) + py + kq) & 3.  K-block 3 reads the constant block for every tile ----
    const int py = (r >> 1) & 1, px = r & 1, w = r >> 2;
    const bool kconst = kq == 3;
    const int ab0 = kconst ? kConstU : ((py + kq) & 3) * kRowPitchU + 2 * w + px;            // + 8*t, + dx by the 4-dword read
    const int ab1 = kconst ? kConstU : ((2 + py + kq) & 3) * kRowPitchU + 2 * w + px;
    const int tstep = kconst ? 0 : 8;
    // ---- staging: two input rows = 108 bytes per step, elements lane and lane + 64 (lanes >= 44: a dummy pixel) ----
    const int e0row = lane >= 54 ? 1 : 0, e0rem = lane - 54 * e0row;
    const int e0px = e0rem / 3, e0ch = e0rem - 3 * e0px;
    const bool e1ok = lane < 44;
    const int e1rem = e1ok ? lane + 10 : 0;
    const int e1px = e1rem / 3, e1ch = e1rem - 3 * e1px;
    const int st0 = e0row * (kRowPitchU * 4) + e0px * 4 + e0ch;           // + slot * 160
    const int st1 = e1ok ? kRowPitchU * 4 + e1px * 4 + e1ch : kRowPitchU * 4 + 30 * 4;     // pixel 30 of a row is never read
    constexpr int EB = F32IN ? 4 : 1;                            // bytes per input element
    const int rowb = g.W * 3 * EB;                               // bytes per input row
    bool bad = false;                                            // F32IN: this lane staged a value off the byte grid

    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int rp0 = chunk * rc, rp1 = min(rp0 + rc, g.H / 2);        // rc is even or nch == 1: rp0 is even
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)const_cast<void*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        // byte offsets of the two staged elements in input row 0 (the row offset is added below); columns outside the
        // image and, through the per-image descriptor, rows outside it read 0 = code 0
        // this lane's packed output word of a step (fused form): byte offset inside the image's output, the step's row
        // goes into the scalar offset of the store -- no per-step address arithmetic on the vector unit
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<uint32_t*>(y) + (size_t)n * g.Hp * g.Wp * e.ocw, 0, g.Hp * g.Wp * e.ocw * 4, 0x00020000);
        const int ylane = (((xs >> 1) + kq + 4 * ((r & 7) >> 2)) * e.ocw + (r & 3) * 2 + (r >> 3)) * 4;
        const int yrow = g.Wp * e.ocw * 4;
        const int c0col = xs - 1 + e0px, c1col = xs - 1 + e1px;
        const int v0 = (c0col >= 0 && c0col < g.W) ? (c0col * 3 + e0ch) * EB + e0row * rowb : (int)0x80000000;
        const int v1 = (e1ok && c1col >= 0 && c1col < g.W) ? (c1col * 3 + e1ch) * EB + rowb : (int)0x80000000;
        auto stage_load = [&](int row, uint32_t& b0, uint32_t& b1) {   // rows `row`, `row + 1`
            const int so = row * rowb;                           // may be negative: the sum wraps out of range
            if constexpr (F32IN) {
                b0 = __builtin_amdgcn_raw_buffer_load_b32(xr, v0 + so, 0, 0);      // float bits; converted at the write
                b1 = __builtin_amdgcn_raw_buffer_load_b32(xr, v1 + so, 0, 0);
            } else {
                b0 = __builtin_amdgcn_raw_buffer_load_b8(xr, v0 + so, 0, 0);
                b1 = __builtin_amdgcn_raw_buffer_load_b8(xr, v1 + so, 0, 0);
            }
        };
        auto stage_write = [&](auto slotc, uint32_t b0, uint32_t b1) {   // into the ring slots SLOT, SLOT + 1
            constexpr int SB = decltype(slotc)::value * (kRowPitchU * 4);
            if constexpr (F32IN) {
                b0 = image_byte(__uint_as_float(b0), bad);
                b1 = image_byte(__uint_as_float(b1), bad);
            }
            ldsb[st0 + SB] = (uint8_t)(b0 ^ 0x80u);
            ldsb[st1 + SB] = (uint8_t)(b1 ^ 0x80u);
        };
        const int yy_first = 2 * rp0;
        uint32_t fa0, fa1, fb0, fb1, fc0, fc1;
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
            stage_load(yy0 + 5, fc0, fc1);                       // two steps ahead
            const uint32_t* abase = lds + (PAR ? ab1 : ab0);
            v4i A[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t* p = abase + tstep * t;
                // the 4th dword (the pixel after the three taps) meets zero filter bytes
                A[t] = __builtin_bit_cast(v4i, make_uint4(p[0], p[1], p[2], p[3]));
            }
            // 8 groups (tile t = gq >> 2, filter block nt = gq & 3): the MFMA of group gq + 1 is issued before group gq
            // is consumed
            v4i acc[2];
            int T[8];
            const v4i z = {0, 0, 0, 0};
            auto consume = [&](int gq, const v4i& a) {
                const int t = gq >> 2, nt = gq & 3;
                if constexpr (POOL == 2) {
                    T[gq] = max(max(a[0], a[1]), max(a[2], a[3]));
                } else {
                    // position i of window kq -> conv pixel (yy0 + (i >> 1), xs + 8t + 2kq + (i & 1))
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = nt * 16 + r;
                        float v = __fmaf_rn((float)a[i], fa[nt], fb[nt]);
                        if (e.fn == QNN_FN_BINARY_TANH) v = v > 0x1p-24f ? 1.0f : -1.0f;
                        else if (e.fn == QNN_FN_QUANTIZED_TANH)
                            v = __fmul_rn(fminf(fmaxf(rintf(v), -e.act_m), e.act_m - 1.0f),
                                          __uint_as_float(0x7F000000u - __float_as_uint(e.act_m)));
                        const long q = ((long)n * g.H + yy0 + (i >> 1)) * g.W + xs + 8 * t + 2 * kq + (i & 1);
                        reinterpret_cast<float*>(y)[q * g.cout + c] = v;
                    }
                }
            };
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], bw[0], z, 0, 0, 0);
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) {
                if (gq + 1 < 8)
                    acc[(gq + 1) & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[(gq + 1) >> 2], bw[(gq + 1) & 3], z, 0, 0, 0);
                consume(gq, acc[gq & 1]);
            }
            if constexpr (OUT == QNN_STORE_I4 && FOLD) {
                static_assert(!FOLD || (!BIN && POOL == 2), "the fold covers the pooled quantized_tanh form");
                uint32_t tp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)             // pairs (value j, value j + 4): same filter block, tiles 0 and 1
                    tp[j] = qnn_fold_pair_fma(T[j], T[j + 4], fa[j], fa[j], fb[j], fb[j]);
                const uint32_t uo = __builtin_amdgcn_perm(tp[3], tp[1], 0x07030501u);
                const uint32_t ue = __builtin_amdgcn_perm(tp[2], tp[0], 0x07030501u);
                const uint32_t P = (uo & 0xF0F0F0F0u) | ((ue >> 4) & 0x0F0F0F0Fu);      // nibble j = code of value j
                __builtin_amdgcn_raw_buffer_store_b32(transpose_nib8(P, ke), yr, ylane, rp * yrow, 0);
            } else if constexpr (OUT == QNN_STORE_I4) {
                // lane (filter r, window kq): value j = 4*t + nt -> after the transpose lane (r & 7) holds the word of
                // value j = r & 7: pooled pixel (xs/2 + kq + 4*(j >> 2)), channels (j & 3)*16 + (r & 8) .. +7
                int cb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float u = __fmaf_rn((float)T[j], fa[j & 3], fb[j & 3]);
                    if constexpr (BIN) {
                        cb[j] = u > 0x1p-24f ? kMagicBits + 1 : kMagicBits - 1;
                    } else {
                        // rint + clamp in one add and one integer median (the low bits of u + magic are rint(u) + 8)
                        const int bits = __float_as_int(__fadd_rn(u, magic));
                        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cb[j]) : "v"(bits), "v"(code_lo), "v"(code_hi));
                    }
                }
                // sum_j cb[j] << 4j by Horner: seven v_lshl_add_u32 (the compiler's form was 7 shifts + 5 three-way adds)
                uint32_t P = (uint32_t)cb[7];
#pragma unroll
                for (int j = 6; j >= 0; --j) asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(P) : "v"(P), "v"(cb[j]));
                P -= (uint32_t)(kMagicBits - 8) * 0x11111111u;
                const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                __builtin_amdgcn_raw_buffer_store_b32(Wd, yr, ylane, rp * yrow, 0);
            }
        };
        int rp = rp0;
        for (; rp + 2 <= rp1; rp += 2) {
            step(std::integral_constant<int, 0>{}, rp);
            step(std::integral_constant<int, 1>{}, rp + 1);
        }
        if (rp < rp1) step(std::integral_constant<int, 0>{}, rp);
    }
    if constexpr (F32IN) {
        if (bad) *domain_flag = 1u;          // reported by qnn_weights_check / the next qnn_conv2d_forward, never silent
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The same layer WITHOUT pooling and with packed output: the first conv of a deeper stage (VGG with nla > 1: 3 -> 64 ...
// 256 filters, int4 or int8 codes) and the ResNet stem (3 -> 16, models/resnet.py:96-99).  Same strip walk, staging and
// MFMA; every conv position is stored, so the epilogue runs on all 2048 results of a step (not on 512 pooled ones) and
// the offset term 128 * sum k is added as an integer (an 8-bit filter's sum does not fit the sixteen spare bytes).
// A wave computes NBLK blocks of 16 filters (blockIdx.y selects the slice of 16 * NBLK).  Packing: int8 -- a 4 x 4 byte
// transpose over a lane quad leaves lane (r & 3) = position with four consecutive filters; int4 -- an 8 x 8 nibble
// transpose over a lane octet pairs two filter blocks (NBLK = 4) or the step's two tiles (NBLK = 1) so that every lane
// ends with one pixel's eight consecutive channels.  One dword store per lane and group.
// (16 filters per wave = two MFMAs per step: little to overlap inside a wave, so that form runs six workgroups per CU)
template <int OUT, int NBLK, bool BIN, bool F32IN>
__global__ __launch_bounds__(256, (NBLK == 1 ? 6 : 3)) void k_conv_first_u8_full(ConvGeom g, EpiArgs e, const void* __restrict__ x,
                                                                const float* __restrict__ wq, void* __restrict__ y,
                                                                int ntasks, int spr, FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                                int rc, uint32_t img_x, float wscale, float D,
                                                                uint32_t* __restrict__ domain_flag) {
    static_assert(OUT == QNN_STORE_I4 || OUT == QNN_STORE_I8, "packed outputs");
    static_assert(NBLK == 1 || NBLK == 4, "16 or 64 filters per wave");
    extern __shared__ __attribute__((aligned(16))) char smem_u8[];
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroup -> (task stream xw, filter slice): the slices of one task stream sit on ONE XCD (workgroups are dealt to
    // the eight XCDs round robin) and start together, so the 64-byte pieces of a stored pixel that different slices write
    // meet in the same L2 instead of leaving four partial lines in four of them.  gridDim.y = slices, gridDim.x % 8 == 0
    // (otherwise the plain mapping).
    const int nsl = gridDim.y, bxw = gridDim.x;
    const int bid = blockIdx.y * bxw + blockIdx.x;
    int xw = blockIdx.x, slice = blockIdx.y;
    if ((bxw & 7) == 0 && nsl > 1) {
        const int xcd = bid & 7, j = bid >> 3;
        slice = j % nsl;
        xw = (j / nsl) * 8 + xcd;
    }
    const int wid = xw * 4 + wave, nw = bxw * 4;
    const int cbase = slice * (16 * NBLK);
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem_u8) + wave * kWaveLdsU;
    uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
    uint4* tab = reinterpret_cast<uint4*>(smem_u8 + 4 * kWaveLdsU * 4);      // [filter block][lane][2]
    for (int i = lane; i < kWaveLdsU; i += 64) lds[i] = 0x80808080u;

    // int8 output of 64 filters per workgroup: the MFMA operands swap roles (rows = filters, columns = positions), and the
    // rows of filter block nt are the filters 16 * (row >> 2) + 4 * nt + (row & 3): a lane then ends with the SIXTEEN
    // CONSECUTIVE filters cbase + 16 kq .. + 15 of one position = one 16-byte store, no lane transposes.  (The 4-byte
    // stores of the other form, 16-byte runs 256 bytes apart, kept this layer at 0.68 ms for 1.07 GB on VGG-large.)
    constexpr bool SWAP = OUT == QNN_STORE_I8 && NBLK == 4;
    if (wave < NBLK) {
        const int c = SWAP ? cbase + 16 * (r >> 2) + 4 * wave + (r & 3) : cbase + wave * 16 + r;
        const float bias = e.bias ? e.bias[c] : 0.0f;
        const float inv = e.bn_inv ? e.bn_inv[c] : 1.0f;
        const float shift = e.bn_inv ? e.bn_shift[c] : 0.0f;
        const float* wrow = wq + (size_t)c * 27 + (kq < 3 ? kq : 2) * 9;
        int part = 0;
        uint32_t wd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int code = (int)rintf(__fmul_rn(wrow[i], wscale));
            if (kq == 3) code = 0;
            part += code;
            wd[i / 3] |= (uint32_t)(code & 0xFF) << (8 * (i % 3));
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const double m = e.fn == QNN_FN_QUANTIZED_TANH ? (double)e.act_m : 1.0;
        const float A = (float)((double)inv * m / (double)D);
        const float B = (float)(((double)bias * (double)inv + (double)shift) * m);
        tab[(wave * 64 + lane) * 2] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        tab[(wave * 64 + lane) * 2 + 1] = make_uint4(__float_as_uint(A), __float_as_uint(B), (uint32_t)(128 * part), 0u);
    }
    __syncthreads();
    v4i bw[NBLK];
    float fa[NBLK], fb[NBLK];
    int c0[NBLK];
#pragma unroll
    for (int nt = 0; nt < NBLK; ++nt) {
        const uint4 t0 = tab[(nt * 64 + lane) * 2], t1 = tab[(nt * 64 + lane) * 2 + 1];
        bw[nt] = __builtin_bit_cast(v4i, t0);
        fa[nt] = __uint_as_float(t1.x); fb[nt] = __uint_as_float(t1.y); c0[nt] = (int)t1.z;
    }
    float fa16[SWAP ? 16 : 1], fb16[SWAP ? 16 : 1];
    int c16[SWAP ? 16 : 1];
    if constexpr (SWAP) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {                 // filter cbase + 16 kq + j = row 4 kq + (j & 3) of block j >> 2
            const uint4 t1 = tab[((j >> 2) * 64 + 4 * kq + (j & 3)) * 2 + 1];
            fa16[j] = __uint_as_float(t1.x); fb16[j] = __uint_as_float(t1.y); c16[j] = (int)t1.z;
        }
    }
    LaneEpi ke;
    lane_epi_init<OUT>(ke, e, r, r);
    const int py = (r >> 1) & 1, px = r & 1, w = r >> 2;
    const bool kconst = kq == 3;
    const int ab0 = kconst ? kConstU : ((py + kq) & 3) * kRowPitchU + 2 * w + px;
    const int ab1 = kconst ? kConstU : ((2 + py + kq) & 3) * kRowPitchU + 2 * w + px;
    const int tstep = kconst ? 0 : 8;
    const int e0row = lane >= 54 ? 1 : 0, e0rem = lane - 54 * e0row;
    const int e0px = e0rem / 3, e0ch = e0rem - 3 * e0px;
    const bool e1ok = lane < 44;
    const int e1rem = e1ok ? lane + 10 : 0;
    const int e1px = e1rem / 3, e1ch = e1rem - 3 * e1px;
    const int st0 = e0row * (kRowPitchU * 4) + e0px * 4 + e0ch;
    const int st1 = e1ok ? kRowPitchU * 4 + e1px * 4 + e1ch : kRowPitchU * 4 + 30 * 4;
    constexpr int EB = F32IN ? 4 : 1;
    const int rowb = g.W * 3 * EB;
    bool bad = false;
    // ---- where this lane's packed word of a group goes: position jj of window kq (tile t: + 8 columns) ----
    const int jj = (OUT == QNN_STORE_I8) ? (r & 3) : (r & 3);          // after the transposes a lane holds position r & 3
    const int pixb = e.ocw * 4;                                        // bytes per stored pixel
    // int8: four consecutive filters from r & ~3;  int4: eight from r & 8, block (or tile) selected by bit 2 of r
    const int chan_b = (OUT == QNN_STORE_I8) ? (r & ~3) : ((r & 8) >> 1);
    const int hi = (r >> 2) & 1;                                       // int4: second block / second tile of the pair

    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int rp0 = chunk * rc, rp1 = min(rp0 + rc, g.H / 2);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (uint8_t*)const_cast<void*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<uint8_t*>(y) + (size_t)n * g.H * g.W * pixb, 0, g.H * g.W * pixb, 0x00020000);
        // byte offset of this lane's word for tile 0 of conv row pair 0 (NBLK = 1, int4: the lane's tile is `hi`)
        const int ycol = xs + 2 * kq + (jj & 1) + ((OUT == QNN_STORE_I4 && NBLK == 1) ? 8 * hi : 0);
        const int ylane = ((jj >> 1) * g.W + ycol) * pixb + cbase * (OUT == QNN_STORE_I8 ? 1 : 0) + (OUT == QNN_STORE_I4 ? cbase / 2 : 0) + chan_b;
        const int yrow2 = 2 * g.W * pixb;                               // two conv rows per step
        const int c0col = xs - 1 + e0px, c1col = xs - 1 + e1px;
        const int v0 = (c0col >= 0 && c0col < g.W) ? (c0col * 3 + e0ch) * EB + e0row * rowb : (int)0x80000000;
        const int v1 = (e1ok && c1col >= 0 && c1col < g.W) ? (c1col * 3 + e1ch) * EB + rowb : (int)0x80000000;
        auto stage_load = [&](int row, uint32_t& b0, uint32_t& b1) {
            const int so = row * rowb;
            if constexpr (F32IN) {
                b0 = __builtin_amdgcn_raw_buffer_load_b32(xr, v0 + so, 0, 0);
                b1 = __builtin_amdgcn_raw_buffer_load_b32(xr, v1 + so, 0, 0);
            } else {
                b0 = __builtin_amdgcn_raw_buffer_load_b8(xr, v0 + so, 0, 0);
                b1 = __builtin_amdgcn_raw_buffer_load_b8(xr, v1 + so, 0, 0);
            }
        };
        auto stage_write = [&](auto slotc, uint32_t b0, uint32_t b1) {
            constexpr int SB = decltype(slotc)::value * (kRowPitchU * 4);
            if constexpr (F32IN) {
                b0 = image_byte(__uint_as_float(b0), bad);
                b1 = image_byte(__uint_as_float(b1), bad);
            }
            ldsb[st0 + SB] = (uint8_t)(b0 ^ 0x80u);
            ldsb[st1 + SB] = (uint8_t)(b1 ^ 0x80u);
        };
        const int yy_first = 2 * rp0;
        uint32_t fa0, fa1, fb0, fb1, fc0, fc1;
        stage_load(yy_first - 1, fa0, fa1);
        stage_load(yy_first + 1, fb0, fb1);
        stage_load(yy_first + 3, fc0, fc1);
        stage_write(std::integral_constant<int, 0>{}, fa0, fa1);

        auto step = [&](auto parc, int rp) {
            constexpr int PAR = decltype(parc)::value;
            const int yy0 = 2 * rp;
            stage_write(std::integral_constant<int, PAR ? 0 : 2>{}, fb0, fb1);
            fb0 = fc0; fb1 = fc1;
            stage_load(yy0 + 5, fc0, fc1);
            const uint32_t* abase = lds + (PAR ? ab1 : ab0);
            v4i A[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t* p = abase + tstep * t;
                A[t] = __builtin_bit_cast(v4i, make_uint4(p[0], p[1], p[2], p[3]));
            }
            const v4i z = {0, 0, 0, 0};
            const int srow = rp * yrow2;                            // scalar offset of the step's rows
            // t = fma(S, A, B) of the four positions of window kq for filter block nt (pre-scaled by the code scale)
            auto values = [&](const v4i& a, int nt, float* tv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tv[i] = __fmaf_rn((float)(a[i] + c0[nt]), fa[nt], fb[nt]);
            };
            if constexpr (SWAP) {
                // position r of tile t: conv row 2 rp + py, column xs + 8 t + 2 w + px
                const int ypos = (py * g.W + xs + 2 * w + px) * pixb + cbase + 16 * kq;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint32_t Pq[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const v4i a = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[nt], A[t], z, 0, 0, 0);
                        float tv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            tv[i] = __fmaf_rn((float)(a[i] + c16[4 * nt + i]), fa16[4 * nt + i], fb16[4 * nt + i]);
                        Pq[nt] = pack_scaled<8, 4>(tv, e.act_m, BIN) ^ 0x80808080u;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, make_uint4(Pq[0], Pq[1], Pq[2], Pq[3])), yr,
                                                           ypos + (8 * t) * pixb, srow, 0);
                }
            } else if constexpr (OUT == QNN_STORE_I8) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int nt = 0; nt < NBLK; ++nt) {
                        const v4i a = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], bw[nt], z, 0, 0, 0);
                        float tv[4];
                        values(a, nt, tv);
                        const uint32_t P = pack_scaled<8, 4>(tv, e.act_m, BIN);
                        const uint32_t Wd = transpose_byte4(P, ke) ^ 0x80808080u;
                        __builtin_amdgcn_raw_buffer_store_b32(Wd, yr, ylane + (8 * t) * pixb + nt * 16, srow, 0);
                    }
            } else if constexpr (NBLK == 4) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int np = 0; np < 2; ++np) {                // filter blocks 2*np, 2*np + 1 share a word
                        const v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], bw[2 * np], z, 0, 0, 0);
                        const v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], bw[2 * np + 1], z, 0, 0, 0);
                        float tv[8];
                        values(a0, 2 * np, tv);
                        values(a1, 2 * np + 1, tv + 4);
                        const uint32_t P = pack_scaled<4, 8>(tv, e.act_m, BIN);
                        const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                        __builtin_amdgcn_raw_buffer_store_b32(Wd, yr, ylane + (8 * t) * pixb + (2 * np + hi) * 8, srow, 0);
                    }
            } else {
                const v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], bw[0], z, 0, 0, 0);      // tile 0, tile 1
                const v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], bw[0], z, 0, 0, 0);
                float tv[8];
                values(a0, 0, tv);
                values(a1, 0, tv + 4);
                const uint32_t P = pack_scaled<4, 8>(tv, e.act_m, BIN);
                const uint32_t Wd = transpose_nib8(P, ke) ^ 0x88888888u;
                __builtin_amdgcn_raw_buffer_store_b32(Wd, yr, ylane, srow, 0);
            }
        };
        int rp = rp0;
        for (; rp + 2 <= rp1; rp += 2) {
            step(std::integral_constant<int, 0>{}, rp);
            step(std::integral_constant<int, 1>{}, rp + 1);
        }
        if (rp < rp1) step(std::integral_constant<int, 0>{}, rp);
    }
    if constexpr (F32IN) {
        if (bad) *domain_flag = 1u;
    }
}

}  // namespace

// 0 = launched.  3x3, stride 1, SAME, 3 input channels, 64 filters of <= 7 bits (or binary), W % 16 == 0, H even;
// fused pipeline form (pool 2, int4 codes out) or float32 output without pooling.
int qnn_try_launch_first_u8(const ConvGeom& g, const EpiArgs& e, const void* x, const qnn_weights* w, void* y,
                            hipStream_t s, bool f32in) {
    if (f32in && !(e.dom_flag ? e.dom_flag : w->d_flag)) return 1;                           // no flag word, no restricted-domain kernel
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.pt != 1 || g.pl != 1 || g.cin != 3 || e.res) return 1;
    if ((g.W % 16) != 0 || (g.H % 2) != 0 || !w->d_wq) return 1;
    // weight codes = value * 2^wshift: binary (+-1, H = 1) or quantized (pooled form: <= 7 bits, |code| <= 64 -- the negated
    // filter still fits a signed byte and sixteen bytes hold -sum k; un-pooled packed form: <= 8 bits)
    if (w->wkind == QNN_W_BINARY ? (w->H != 1.0f || w->wshift != 0)
                               : (w->wkind != QNN_W_QUANT || w->wshift < 1 || w->wshift > 7)) return 1;
    const bool small_codes = w->wshift <= 6;
    const bool ok_fused = g.cout == 64 && small_codes && g.pool == 2 && e.out_store == QNN_STORE_I4 &&
                          ((e.fn == QNN_FN_QUANTIZED_TANH && e.act_m <= 8.0f) || e.fn == QNN_FN_BINARY_TANH);
    const bool rawf = g.cout == 64 && small_codes && g.pool == 1 && e.out_store == QNN_STORE_F32 &&
                      (e.fn == QNN_FN_NONE || e.fn == QNN_FN_QUANTIZED_TANH || e.fn == QNN_FN_BINARY_TANH);
    // un-pooled packed output (k_conv_first_u8_full): 16 filters (ResNet stem) or a multiple of 64
    const bool ok_full = g.pool == 1 && (e.out_store == QNN_STORE_I4 || e.out_store == QNN_STORE_I8) &&
                         (g.cout == 16 || (g.cout % 64) == 0) &&
                         ((e.fn == QNN_FN_QUANTIZED_TANH && e.act_m <= (e.out_store == QNN_STORE_I4 ? 8.0f : 128.0f)) ||
                          e.fn == QNN_FN_BINARY_TANH);
    if (!ok_fused && !rawf && !ok_full) return 1;
    const float wscale = (float)(1 << w->wshift);
    const int spr = g.W / 16;
    const double img_x = (double)g.H * g.W * 3 * (f32in ? 4 : 1);
    const float D = 255.0f * wscale;                             // the divisor of the affine map (qnn_abi.h)
    if (img_x >= 1.0e9 || (double)g.N * g.Hp * g.Wp * 8.0 >= 2.0e9 * 4) return 1;
    const int hp2 = g.H / 2;
    const bool fused = g.cout == 64 && w->wshift <= 6 && g.pool == 2 && e.out_store == QNN_STORE_I4 &&
                       ((e.fn == QNN_FN_QUANTIZED_TANH && e.act_m <= 8.0f) || e.fn == QNN_FN_BINARY_TANH);
    const bool full = !fused && g.pool == 1 && e.out_store != QNN_STORE_F32;
    // persistent grid = what is resident: four workgroups per CU (three for the un-pooled packed form, launch bounds)
    const int bpc = QNN_ENV_INT("QNN_U8_BPC", full ? (g.cout == 16 ? 6 : 3) : 4);   // (override: experiment builds only)
    const int blocks_cap = 256 * (bpc >= 1 && bpc <= 8 ? bpc : 4);
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
    const size_t lds = (size_t)4 * kWaveLdsU * 4 + 4 * 64 * 2 * 16;
    if (full && !fused) {
        if ((double)g.H * g.W * e.ocw * 4.0 >= 2.0e9) return 1;
        const int nblk = g.cout == 16 ? 1 : 4;
        const int gy = g.cout / (16 * nblk);
        long bx = blocks_cap / gy > 0 ? blocks_cap / gy : 1;
        if (bx > (ntasks_l + 3) / 4) bx = (ntasks_l + 3) / 4;
        const dim3 fgrid((unsigned)bx, (unsigned)gy);
        const bool bin = e.fn == QNN_FN_BINARY_TANH;
#define U8_FULL(OUT_, NB_, BIN_, F32_)                                                                                      \
        hipLaunchKernelGGL((k_conv_first_u8_full<OUT_, NB_, BIN_, F32_>), fgrid, block, lds, s, g, e, x, w->d_wq, y,         \
                           (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),       \
                           best_rc, (uint32_t)img_x, wscale, D, (e.dom_flag ? e.dom_flag : w->d_flag))
#define U8_FULL_B(OUT_, NB_)                                                                                                \
        do {                                                                                                               \
            if (bin) { if (f32in) U8_FULL(OUT_, NB_, true, true); else U8_FULL(OUT_, NB_, true, false); }                   \
            else { if (f32in) U8_FULL(OUT_, NB_, false, true); else U8_FULL(OUT_, NB_, false, false); }                     \
        } while (0)
        if (e.out_store == QNN_STORE_I8) { if (nblk == 1) U8_FULL_B(QNN_STORE_I8, 1); else U8_FULL_B(QNN_STORE_I8, 4); }
        else { if (nblk == 1) U8_FULL_B(QNN_STORE_I4, 1); else U8_FULL_B(QNN_STORE_I4, 4); }
#undef U8_FULL_B
#undef U8_FULL
        return 0;
    }
#define U8_LAUNCH(OUT_, POOL_, BIN_)                                                                                       \
    do {                                                                                                                   \
        if (f32in)                                                                                                         \
            hipLaunchKernelGGL((k_conv_first_u8<OUT_, POOL_, BIN_, true>), grid, block, lds, s, g, e, x, w->d_wq, y,        \
                               (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),  \
                               best_rc, (uint32_t)img_x, wscale, D, (e.dom_flag ? e.dom_flag : w->d_flag));                                            \
        else                                                                                                               \
            hipLaunchKernelGGL((k_conv_first_u8<OUT_, POOL_, BIN_, false>), grid, block, lds, s, g, e, x, w->d_wq, y,       \
                               (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),  \
                               best_rc, (uint32_t)img_x, wscale, D, (e.dom_flag ? e.dom_flag : w->d_flag));                                            \
    } while (0)
    if (!fused) U8_LAUNCH(QNN_STORE_F32, 1, false);
    else if (e.fn == QNN_FN_BINARY_TANH) U8_LAUNCH(QNN_STORE_I4, 2, true);
    else if (e.fold_a && e.fold_c && e.act_m == 8.0f) {       // folded epilogue (mode 3 handle of this layer, qnn_fold.h)
        if (f32in)
            hipLaunchKernelGGL((k_conv_first_u8<QNN_STORE_I4, 2, false, true, true>), grid, block, lds, s, g, e, x, w->d_wq, y,
                               (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),
                               best_rc, (uint32_t)img_x, wscale, D, (e.dom_flag ? e.dom_flag : w->d_flag));
        else
            hipLaunchKernelGGL((k_conv_first_u8<QNN_STORE_I4, 2, false, false, true>), grid, block, lds, s, g, e, x, w->d_wq, y,
                               (int)ntasks_l, spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch),
                               best_rc, (uint32_t)img_x, wscale, D, (e.dom_flag ? e.dom_flag : w->d_flag));
    }
    else U8_LAUNCH(QNN_STORE_I4, 2, false);
#undef U8_LAUNCH
    return 0;
}
