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
#include <math.h>

#include "qnn_mfma_common.h"



namespace {

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
        tile = QNN_ABLATE(mg.ablate, 4) ? (long)b_
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
        const int woff = (s_tap < g.kh * g.kw && !QNN_ABLATE(mg.ablate, 2)) ? s_tap * g.cin + s_kc * 64
                                                                    : (int)0x40000000;   // past the end -> zeros
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const bool ok = ((a_mask[p] >> s_tap) & 1u) && !QNN_ABLATE(mg.ablate, 1);
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
            if (!QNN_ABLATE(mg.ablate, 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
            if (!QNN_ABLATE(mg.ablate, 8)) __builtin_amdgcn_s_setprio(0);
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

// launch of the LDS-DMA variant (defined after the kernel: it takes the kernel's address)
template <int XS, int WM, int WN, int OUT>
void launch_dma16(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s,
                  dim3 grid, dim3 block);

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
        static const int shape = QNN_ENV_INT("QNN_MFMA_SHAPE", 16);
        // int8 activations on the 256 x 256 tile: both operands go global -> LDS by LDS-DMA (see k_conv_mfma16)
        if constexpr (XS == QNN_STORE_I8 && WM == 4 && (WN == 4 || WN == 2)) {
            static const int dma = QNN_ENV_INT("QNN_MFMA_DMA", 1);   // A/B switch (experiment builds only)
            if (shape == 16 && dma) {
                launch_dma16<XS, WM, WN, OUT>(mg, e, x, w, y, s, grid, block);
                return;
            }
        }
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

// ---------------------------------------------------------------------------------
// Same implicit GEMM on v_mfma_i32_16x16x64_i8.  On random operands the chip holds a lower
// clock under 32x32x32 at four waves per SIMD than under 16x16x64 (tools/ubench_shape.hip:
// 1 474 vs 1 693 T MAC/s for a bare register loop), and the large tiles of the 8-bit VGG
// layers run exactly at that occupancy.  Same staging, same LDS traffic (4 + 4 fragment reads
// of 16 bytes per step and wave), 16 MFMAs of 16 cycles instead of 8 of 32; the 64-byte rows
// use a chunk permutation that is conflict-free for this fragment shape.  The 2x2 pool window
// is still the four accumulator registers of one lane.  Outputs: float32, int4, int8.
//
// DMA variant (int8 activations, 16 waves): both operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds`
// (no staging registers, no ds_write: the VGPR -> LDS store path costs 13 cycles per KiB and was busy 40 % of a
// K-step), three LDS buffers, one raw s_barrier per K-step and a counted vmcnt, so two steps of loads stay in flight
// across the barrier.  An LDS-DMA wave-instruction writes 64 x 16 bytes LINEARLY from its wave-uniform base: the
// chunk swizzle moves to the source side (lane (row, slot) fetches chunk slot ^ swz(row) of its row); out-of-range
// offsets (padding taps, steps past the end) write zeros.
template <int XS, int WM, int WN, int OUT, int POOL, bool DMA>
__device__ __forceinline__ void conv_mfma16_body(const MfmaGeom& mg, const EpiArgs& e, const uint8_t* __restrict__ x,
                                                 const uint8_t* __restrict__ wq8, void* __restrict__ y) {
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
    constexpr int NBUF = DMA ? QNN_DMA_NBUF : 2;
    constexpr int B_BASE = NBUF * A_BUF;
    static_assert(!DMA || XS == QNN_STORE_I8, "LDS-DMA staging: int8 rows (int4 rows are widened in registers)");
    constexpr int NLD = NA + NB;             // LDS-DMA loads per thread and K-step

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
        tile = QNN_ABLATE(mg.ablate, 4) ? (long)b_
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
            a_voff[p] = ((n * g.H + iy0) * g.W + ix0) * mg.x_pix_bytes + (DMA ? (sch ^ swz(R)) : sch) * XCH;
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
        b_voff[p] = (nbase + R) * w_row_bytes + (DMA ? (sch ^ swz(R)) : sch) * 16;
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
        const int woff = (s_tap < g.kh * g.kw && !QNN_ABLATE(mg.ablate, 2)) ? s_tap * g.cin + s_kc * 64
                                                                    : (int)0x40000000;   // past the end -> zeros
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const bool ok = ((a_mask[p] >> s_tap) & 1u) && !QNN_ABLATE(mg.ablate, 1);
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

    const int S = mg.steps;
    if constexpr (DMA) {
        // step k lives in buffer k % NBUF (NBUF = 3: measured; see QNN_DMA_NBUF).  Per step: wait for this wave's own loads of step k (the loads of step k + 1
        // stay in flight), barrier (every wave's part of step k is in LDS, every wave is done reading step k - 1),
        // issue the loads of step k + NBUF - 1 into the buffer step k - 1 used, then the MFMAs of step k.
        using lds_ptr = __attribute__((address_space(3))) void*;
        auto dma_issue = [&](int buf) {
            const int xoff = (s_dy * g.W + s_dx) * mg.x_pix_bytes + s_kc * (4 * XCH);
            const int woff = (s_tap < g.kh * g.kw && !QNN_ABLATE(mg.ablate, 2)) ? s_tap * g.cin + s_kc * 64
                                                                        : (int)0x40000000;   // past the end -> zeros
            // pass p of a tile = rows [p * RPP, (p + 1) * RPP): wave w writes the 1 KiB at p * RPP * 64 + w * 1024
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                const bool ok = ((a_mask[p] >> s_tap) & 1u) && !QNN_ABLATE(mg.ablate, 1);                 // (ablate: timing experiments)
                const int voff = ok ? a_voff[p] + xoff : (int)0x80000000;                        // out of range -> zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    xrsrc, (lds_ptr)(smem + buf * A_BUF + p * (RPP * 64) + wave * 1024), 16, voff, 0, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < NB; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    wrsrc, (lds_ptr)(smem + B_BASE + buf * B_BUF + p * (RPP * 64) + wave * 1024), 16, b_voff[p], woff, 0, 0);
            if (++s_kc == mg.kc) {
                s_kc = 0; ++s_tap;
                if (++s_dx == g.kw) { s_dx = 0; ++s_dy; }
            }
        };
#if QNN_DMA_PREFETCH
        // Fragment prefetch: the operand registers of step k + 1 are filled WHILE the MFMAs of step k run (the B
        // fragments into a second register set, each A fragment into its own registers as soon as its row of MFMAs has
        // been issued), so a step starts on the matrix pipe right after its barrier instead of after 8 LDS reads that
        // all 16 waves issue at once.  The wait / barrier at the top of step k therefore covers the loads of step
        // k + 1, and the loads issued at step k go three steps ahead, into the buffer step k just vacated.
        static_assert(NBUF == 3, "the prefetching schedule rotates three buffers");
        if (true) {
            dma_issue(0); dma_issue(1); dma_issue(2);
            v4i fa[4], fb[2][4];
            auto read_a = [&](int t, int b_) { return *reinterpret_cast<const v4i*>(smem + fa_addr + b_ * A_BUF + t * 1024); };
            auto read_b = [&](int t, int b_) { return *reinterpret_cast<const v4i*>(smem + fb_addr + b_ * B_BUF + t * 1024); };
            __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NLD));   // vmcnt: step 0 has landed (two steps may be in flight)
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[t] = read_a(t, 0); fb[0][t] = read_b(t, 0); }
            int buf = 0;                               // buffer of step ks
            auto step = [&](auto curc) {
                constexpr int C = decltype(curc)::value;
                const int nb = buf == 2 ? 0 : buf + 1; // buffer of step ks + 1
                __builtin_amdgcn_s_waitcnt(0x0070 | NLD);   // vmcnt(NLD): step ks + 1 has landed;  lgkmcnt(0): this step's fragments are in
                __builtin_amdgcn_s_barrier();          // ... for every wave, and every wave has read buffer `buf` for the last time
                dma_issue(buf);                        // step ks + 3 (past the end: zeros, never used)
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[1 - C][t] = read_b(t, nb);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[C][b], acc[a][b], 0, 0, 0);
                    fa[a] = read_a(a, nb);
                }
                __builtin_amdgcn_s_setprio(0);
                buf = nb;
            };
            int ks = 0;
            for (; ks + 2 <= S; ks += 2) {
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
            }
            if (ks < S) step(std::integral_constant<int, 0>{});
            __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0), lgkmcnt(0): nothing may land after the kernel's LDS is gone
        } else
#endif
        {
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i) dma_issue(i);                   // steps 0 .. NBUF-2
        int buf = 0, nxt = NBUF - 1;                   // buffer of step ks, buffer of step ks + NBUF - 1
        for (int ks = 0; ks < S; ++ks) {
            // vmcnt(2 * (NBUF - 2)): step ks has landed, the later steps may be in flight
            __builtin_amdgcn_s_waitcnt(0x0F70 | (NLD * (NBUF - 2)));
            __builtin_amdgcn_s_barrier();
            dma_issue(nxt);                            // step ks + 2 (past the end: zeros, never read)
            compute(buf * A_BUF, buf * B_BUF);
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): the fragment reads of this step are done
            buf = buf == NBUF - 1 ? 0 : buf + 1;
            nxt = nxt == NBUF - 1 ? 0 : nxt + 1;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): no load may land after the kernel's LDS is gone
        }
    } else {
    // ---- main loop: one barrier per K-step; the loads of step k+2 are issued before the
    // MFMAs of step k and only waited for (counted vmcnt) after the MFMAs of step k+1 ----
    // Loads are issued UNCONDITIONALLY (a step past the end has tap >= kh*kw, whose mask
    // bit is 0 -> out-of-range offset -> the buffer load returns zeros without touching
    // memory): conditional loads make the compiler fall back to vmcnt(0).
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
    }

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
            // float32 BN on value PAIRS: v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar forms and issue
            // at the same rate (tools/micro/pk_f32_rate.hip), so the three operations cost 1.5 instructions per value instead
            // of 3 -- the un-pooled int8 layers of VGG-large spend a fifth of their time in this epilogue (64 values per lane
            // and tile, all 16 waves of the workgroup at once, nothing on the matrix pipe meanwhile)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const v2f nb2 = {fe[b].nb, fe[b].nb}, ninv2 = {fe[b].ninv, fe[b].ninv}, nshift2 = {fe[b].nshift, fe[b].nshift};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f v = {(float)acc[a][b][2 * h], (float)acc[a][b][2 * h + 1]};
                    v = v + nb2;
                    v = v * ninv2;
                    v = v + nshift2;
                    tv[b][2 * h] = v[0]; tv[b][2 * h + 1] = v[1];
                }
            }
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

template <int XS, int WM, int WN, int OUT, int POOL>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : WM * WN >= 8 ? 2 : 3)) void k_conv_mfma16(MfmaGeom mg, EpiArgs e,
                                                           const uint8_t* __restrict__ x,
                                                           const uint8_t* __restrict__ wq8,
                                                           void* __restrict__ y) {
    conv_mfma16_body<XS, WM, WN, OUT, POOL, false>(mg, e, x, wq8, y);
}
template <int XS, int WM, int WN, int OUT, int POOL>
__global__ __launch_bounds__(64 * WM * WN, 4) void k_conv_mfma16_dma(MfmaGeom mg, EpiArgs e, const uint8_t* __restrict__ x,
                                                                     const uint8_t* __restrict__ wq8, void* __restrict__ y) {
    conv_mfma16_body<XS, WM, WN, OUT, POOL, true>(mg, e, x, wq8, y);
}

template <int XS, int WM, int WN, int OUT>
void launch_dma16(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s,
                  dim3 grid, dim3 block) {
    if constexpr (XS == QNN_STORE_I8 && WM == 4 && (WN == 4 || WN == 2) && OUT != QNN_STORE_BIN) {
        const size_t lds3 = (size_t)QNN_DMA_NBUF * (64 * WM + 64 * WN) * 64;
        if (mg.g.pool == 2) {
            (void)hipFuncSetAttribute((const void*)k_conv_mfma16_dma<XS, WM, WN, OUT, 2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipLaunchKernelGGL((k_conv_mfma16_dma<XS, WM, WN, OUT, 2>), grid, block, lds3, s, mg, e,
                               (const uint8_t*)x, w, y);
        } else {
            (void)hipFuncSetAttribute((const void*)k_conv_mfma16_dma<XS, WM, WN, OUT, 1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipLaunchKernelGGL((k_conv_mfma16_dma<XS, WM, WN, OUT, 1>), grid, block, lds3, s, mg, e,
                               (const uint8_t*)x, w, y);
        }
    }
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
    // the 1x1 strides-2 projection of a ResNet stage (models/resnet.py:117-124): read by the strip kernel of the block's
    // second convolution when the shortcut is computed inside that launch (qnn_projection_t); no kernel of this file
    // runs such a layer on its own
    const bool proj = w->store == QNN_STORE_I4 && (w->cin == 16 || w->cin == 32) && w->cout == 2 * w->cin &&
                      w->kh == 1 && w->kw == 1 && w->stride == 2;
    if (!small && !proj && (w->cin % 64 != 0 || w->cout % 64 != 0)) return QNN_OK;
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
        // opt-in: float32 inputs that are image bytes / 255 on the uint8 entry's kernel (qnn_first_u8.hip, F32IN)
        if (e.first_mode == 1 && qnn_try_launch_first_u8(g, e, x, w, y, s, true) == 0) {
            snprintf(name, name_len, "mfma_i8_first_img255");
            return 0;
        }
        // opt-in fixed-point variant (qnn_first_fixed.hip): NOT the oracle's float32 chain, see its header
        if (e.first_mode == 2 && qnn_try_launch_first_fixed(g, e, x, w, y, s) == 0) {
            snprintf(name, name_len, "mfma_i8x3_first_fixed");
            return 0;
        }
        snprintf(name, name_len, "mfma_f32_first_cin%d", g.cin);
        // the LDS-staged kernel works on 64-filter slices (blockIdx.y): prefer it whenever its tiling applies
        const bool lds_shape = g.stride == 1 && g.pt == 1 && g.pl == 1 &&
                               ((g.pool == 2 && (g.Wp % 8) == 0 && (g.H % 2) == 0 && (g.W % 2) == 0) ||
                                (g.pool == 1 && (g.W % 32) == 0));
        // (QNN_FIRST_GATHER, read once in qnn_first.hip, selects the gather variant for A/B timing)
        if (lds_shape || g.cout == 64) return qnn_launch_first(g.cin, 2, g, e, x, w->d_wq, y, s);
        return qnn_launch_first(g.cin, 4, g, e, x, w->d_wq, y, s);
    }
    if (!w->d_mfma) return 1;
    if (x_store != QNN_STORE_I8 && x_store != QNN_STORE_I4) return 1;
    static const bool small_off = QNN_ENV_STR("QNN_MFMA_SMALL_OFF") != nullptr;   // A/B switch (experiment builds only)
    // 3x3 stride-1 int4 layers with 16 / 32 / 64 input channels: row-walking strip kernel (qnn_mfma_strip.hip).
    // The residual's post-scale (models/resnet.py:128: 0.5) must be a power of two so that it folds exactly into the
    // activation's code scale.
    {
        int pexp = 0;
        const bool pow2 = (!e.res && !e.proj_x) || (e.post_scale > 0.0f && frexpf(e.post_scale, &pexp) == 0.5f);
        const int cmul = g.cin == 16 ? 16 : 32;
        // stride 2 (the first conv of a stage: no residual, Cin 16 / 32, Cout a multiple of 32): the same walk over
        // output rows, three fresh input rows per output row
        const bool s1 = g.stride == 1 && g.pt == 1 && g.pl == 1 && (g.cout % cmul) == 0;
        const bool s2 = g.stride == 2 && (g.cin == 16 || g.cin == 32) && (g.cout % 32) == 0 && !e.res;
        const bool proj_ok = !e.proj_x || (s1 && (g.cin == 32 || g.cin == 64) && g.cout == g.cin && e.proj_cin * 2 == g.cin &&
                                           !e.res && !e.fold_a && (e.proj_H + 1) / 2 == g.H && (e.proj_W + 1) / 2 == g.W);
        const bool shape = x_store == QNN_STORE_I4 && w->store == QNN_STORE_I4 && proj_ok &&
                           (g.cin == 16 || g.cin == 32 || g.cin == 64) && g.kh == 3 && g.kw == 3 && (s1 || s2) &&
                           g.pool == 1 && e.out_store == QNN_STORE_I4 && pow2 &&
                           (!e.res || (e.res_store == QNN_STORE_I4 && e.res_cw == e.ocw) ||
                            (e.res_store == QNN_STORE_F32 && e.res_cw == g.cout));
        // Cin 64 (auto): every un-pooled layer.  Measured, 64 x 56^2 / 4096 x 16^2 pixels, round 3 (one 32-bit store and one
        // shortcut load per row): with the merge 13.9 us here against 36.4 us on the LDS-weight kernel, without it
        // 13.1 / 43.5 against 14.0 / 46.5 (round 2, two 16-bit accesses per row: 16.3 / 54.6 against 14.1 / 47.3).
        const bool want = g.cin == 64 ? !(e.flags & QNN_EPI_NO_STRIP64) : true;
        if (shape && want && !(e.flags & QNN_EPI_NO_STRIP)) {
            MfmaGeom ms;
            ms.g = g; ms.kc = 1; ms.steps = 0; ms.x_pix_bytes = g.cin / 2;
            ms.total_q = (long)g.N * g.H * g.W;
            const double wb_ = (double)g.cout * 9 * g.cin;
            if (wb_ < 2.0e9) {
                ms.x_bytes = 0; ms.w_bytes = (uint32_t)wb_; ms.ablate = 0;
                EpiArgs es = e;
                es.scale = e.scale * (1.0f / 256.0f);        // both operands carry *16
                es.proj_scale = e.proj_scale * (1.0f / 256.0f);
                // 16 -> 16 channels with a usable fold and an even width: the LDS-staged form (qnn_mfma_strip16.hip: a sixth
                // of the load and a quarter of the store instructions)
                static const bool lds16_off = QNN_ENV_STR("QNN_STRIP16_LDS_OFF") != nullptr;   // A/B switch (experiment builds only)
                if (g.cin == 16 && !lds16_off && !(e.flags & QNN_EPI_NO_LDS16) && qnn_launch_strip16_lds(ms, es, x, w->d_mfma, y, s) == 0) {
                    snprintf(name, name_len, "strip_i4_c16_lds");
                    return 0;
                }
                snprintf(name, name_len, g.stride == 2 ? "strip_i4_c%d_s2" : e.proj_x ? "strip_i4_c%d_proj" : "strip_i4_c%d", g.cin);
                if (qnn_launch_strip(g.cin, ms, es, x, w->d_mfma, y, s) == 0) return 0;
            }
        }
    }
    // small-channel 3x3 int4 layers on the tile kernel (both operands in registers)
    if (e.proj_x) return 1;      // the in-launch projection shortcut exists in the strip kernel only
    if (x_store == QNN_STORE_I4 && w->store == QNN_STORE_I4 && (g.cin == 16 || g.cin == 32) && g.kh == 3 &&
        g.kw == 3 && g.stride == 1 && g.pt == 1 && g.pl == 1 && g.pool == 1 && (g.W % 16) == 0 &&
        e.out_store == QNN_STORE_I4 && (g.cout % (g.cin == 16 ? 16 : 32)) == 0 &&
        (!e.res || (e.res_store == QNN_STORE_I4 && e.res_cw == e.ocw) ||
         (e.res_store == QNN_STORE_F32 && e.res_cw == g.cout)) && !small_off) {
        MfmaGeom ms;
        ms.g = g; ms.kc = 1; ms.steps = 0; ms.x_pix_bytes = g.cin / 2;
        ms.total_q = (long)g.N * g.H * g.W;
        const double xb_ = (double)g.N * g.H * g.W * ms.x_pix_bytes, wb_ = (double)g.cout * 9 * g.cin;
        if (xb_ < 2.0e9 && wb_ < 2.0e9) {
            ms.x_bytes = (uint32_t)xb_; ms.w_bytes = (uint32_t)wb_; ms.ablate = 0;
            EpiArgs es = e;
            es.scale = e.scale * (1.0f / 256.0f);            // both operands carry *16
            snprintf(name, name_len, "mfma_i4_small_c%d", g.cin);
            const int rc_ = qnn_launch_small(g.cin, ms, es, x, w->d_mfma, y, s);
            if (rc_ == 0) return 0;
        }
    }
    if (g.cin % 64 != 0 || g.cout % 64 != 0) return 1;
    // the tiled kernels keep the in-image taps of a pixel in a 32-bit mask and shift it by up to kh*kw + 2 (the
    // past-the-end LDS-DMA steps): larger windows (not reachable today, qnn_prepack_weights stops at 3x3) fall back
    if (g.kh * g.kw + 3 > 32) return 1;
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
    static const int ablate = QNN_ENV_INT("QNN_MFMA_ABLATE", 0);
    mg.ablate = ablate;
    EpiArgs e2 = e;
    if (x_store == QNN_STORE_I4) e2.scale = e.scale * (1.0f / 256.0f);   // both operands carry *16
    // tile shape: waves along M x waves along N (64x64 per wave)
    static const char* tile_env = QNN_ENV_STR("QNN_MFMA_TILE");
    int wm_ = 4, wn_ = 1;
    if ((g.cout % 256) == 0) { wm_ = 4; wn_ = 4; }          // measured: 256x256 > 128x256 > 256x128 > 128x128
    else if ((g.cout % 128) == 0) { wm_ = 4; wn_ = 2; }
    if (tile_env && strlen(tile_env) == 3) {
        const int em = tile_env[0] - '0', en = tile_env[2] - '0';
        if (em >= 1 && en >= 1 && (g.cout % (64 * en)) == 0) { wm_ = em; wn_ = en; }
    }
    // short-K layers with one 64-filter slice: weight-resident persistent kernel
    static const int wres_env = QNN_ENV_INT("QNN_MFMA_WRES", -1);
    const long rows_ = mg.total_q * (g.pool == 2 ? 4 : 1);
    const bool wres_fit = mg.steps <= 12 && rows_ < 2000000000L;
    const bool wres = wres_env == 0 ? false : wres_env == 1 ? wres_fit : (wres_fit && g.cout == 64 && !tile_env);
    // 3x3, Cin <= 128: operands straight into registers (QNN_MFMA_AREG=0 disables, =1 also for Cout > 64)
    static const int areg_env = QNN_ENV_INT("QNN_MFMA_AREG", -1);
    const bool areg_fit = g.kh == 3 && g.kw == 3 && mg.kc <= 2 && rows_ < 2000000000L;
    const bool areg = areg_env == 0 ? false : areg_env == 1 ? areg_fit : (areg_fit && g.cout == 64 && !tile_env && wres_env < 0);
    if (e.res && !(areg && g.pool == 1)) return 1;          // the other MFMA kernels have no residual epilogue
    // pooled int4 layers whose pooled map tiles into 8 x 2 / 4 x 4 rectangles: receptive field staged once through LDS
    // (k_conv_mfma_halo, qnn_mfma_areg.hip; qnn_set_option("halo", 0) keeps them on the per-tap kernel below)
    if (areg && x_store == QNN_STORE_I4 && !(e.flags & QNN_EPI_NO_HALO) &&
        qnn_launch_halo(mg, e2, x, w->d_mfma, y, s) == 0) {
        snprintf(name, name_len, "mfma_i4_halo64x64");
        return 0;
    }
    if (areg) {
        snprintf(name, name_len, "mfma_%s_areg64x64", x_store == QNN_STORE_I8 ? "i8" : "i4");
        return qnn_launch_areg(x_store, mg.kc, mg, e2, x, w->d_mfma, y, s);
    }
    if (wres) {
        snprintf(name, name_len, "mfma_%s_wres256x64", x_store == QNN_STORE_I8 ? "i8" : "i4");
        return qnn_launch_wres(x_store, mg, e2, x, w->d_mfma, y, s);
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
