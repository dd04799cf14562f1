// Persistent short-K int8 MFMA kernels: filter slice resident in LDS (k_conv_mfma_wres) and activations
// loaded straight into operand registers (k_conv_mfma_areg).  Dispatch: qnn_try_launch_mfma (qnn_mfma.hip).
#include "qnn_mfma_common.h"
#include "qnn_fold.h"

namespace {

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
// HEAD (round 3): the classifier behind the last conv of a VGG (Flatten + Dense + BN, models/vgg.py:38-42) inside this
// kernel's epilogue.  With a 4 x 4 pooled map a wave's 64-row tile IS one image, so the wave holds all 1024 activation
// codes of its image as 2 x 8 nibbles per lane; the dense layer is then 2 x `units` v_dot8_i32_i4 per lane against a
// per-lane weight table (k_head_table: the dense kernel's codes in this kernel's (pixel, channel) -> (lane, nibble)
// order, kept in LDS), a 17-shuffle transpose-reduction over the wave, and one float32 epilogue on 16 lanes.  Removes
// the dense launch (4 us of work, ~10 us of launch boundary) and the packed tensor in front of it.
struct HeadArgs {
    const uint32_t* tab;       // [16 units][64 lanes][2 channel halves] dwords of 8 int4 weight codes, zero past `units`
    const float* bias;         // dense bias, BN constants (or null), as qnn_epi_value reads them
    const float* bn_inv;
    const float* bn_shift;
    float scale;               // 2^-(wshift + xshift) of the dense layer
    int units;
    float* y;                  // (N, units) float32
};

// 16 per-lane accumulators -> their 16 wave-wide sums, one per lane: lane l ends with the sum of accumulator
// u = 8*b0 + 4*b1 + 2*b2 + b3 (b_i = bit i of l), every lane that shares the low four bits holds the same total
__device__ __forceinline__ int head_reduce16(const int (&a)[16], int lane, int& u_out) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    int v8[8], v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = (b0 ? a[i + 8] : a[i]) + __shfl_xor(b0 ? a[i] : a[i + 8], 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = (b1 ? v8[i + 4] : v8[i]) + __shfl_xor(b1 ? v8[i] : v8[i + 4], 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) v2[i] = (b2 ? v4[i + 2] : v4[i]) + __shfl_xor(b2 ? v4[i] : v4[i + 2], 4);
    int v = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    u_out = (b0 ? 8 : 0) + (b1 ? 4 : 0) + (b2 ? 2 : 0) + (b3 ? 1 : 0);
    return v;
}

template <int XS, int OUT, int POOL, int KC, bool HEAD = false>
__global__ __launch_bounds__(256, (OUT == QNN_STORE_F32 ? 2 : 3)) void k_conv_mfma_areg(MfmaGeom mg, EpiArgs e,
                                                           const uint8_t* __restrict__ x,
                                                           const uint8_t* __restrict__ wq8,
                                                           void* __restrict__ y, int ntiles, HeadArgs hd) {
    static_assert(!HEAD || (POOL == 2 && OUT == QNN_STORE_I4), "the fused classifier sits behind pooled int4 codes");
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
    // pooling takes the window's max or min by the sign of the channel's BN scale; when no channel of the slice has a
    // negative one (the usual case) the minimum is never needed: two instructions per pooled value instead of five
    const bool all_pos = !__any((int)(ke[0].neg || ke[1].neg));
    // ---- fused classifier: table -> LDS (behind the filters), this lane's unit and its float32 constants ----
    constexpr int HTAB = 9 * KC * B_STEP;                   // byte offset of the table in LDS
    int hu = 0;
    float hbias = 0.0f, hinv = 1.0f, hshift = 0.0f;
    if constexpr (HEAD) {
        for (int i = tid; i < 16 * 64 * 2; i += 256) reinterpret_cast<uint32_t*>(smem + HTAB)[i] = hd.tab[i];
        hu = ((lane & 1) ? 8 : 0) + ((lane & 2) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 8) ? 1 : 0);
        if (hu < hd.units) {
            hbias = hd.bias ? hd.bias[hu] : 0.0f;
            hinv = hd.bn_inv ? hd.bn_inv[hu] : 1.0f;
            hshift = hd.bn_inv ? hd.bn_shift[hu] : 0.0f;
        }
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
        int dacc[16];
        if constexpr (HEAD) {
#pragma unroll
            for (int u = 0; u < 16; ++u) dacc[u] = 0;
        }
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
                        if (all_pos) {                   // wave-uniform: every BN scale of this slice is >= 0
                            tv[a * 4 + g4] = bn(mx, fe[b]);
                        } else {
                            const int mn = min(min(i0, i1), min(i2, i3));
                            tv[a * 4 + g4] = bn(ke[b].neg ? mn : mx, fe[b]);
                        }
                    }
                if constexpr (HEAD) {
                    // this lane's codes of channel c at the pooled pixels 8*(j >> 2) + 2*(j & 3) + lh, j = nibble index:
                    // two's-complement nibbles straight into v_dot8_i32_i4 against the table's nibbles of every unit
                    const uint32_t Q = pack_scaled<4, 8>(tv, e.act_m, binary) ^ 0x88888888u;
                    const uint32_t* trow = reinterpret_cast<const uint32_t*>(smem + HTAB) + lane * 2 + b;
#pragma unroll
                    for (int u = 0; u < 16; ++u) dacc[u] = __builtin_amdgcn_sdot8((int)Q, (int)trow[u * 128], dacc[u], false);
                } else if constexpr (OUT == QNN_STORE_I4) {
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
        if constexpr (HEAD) {
            int u;
            const int tot = head_reduce16(dacc, lane, u);
            if (lane < 16 && u < hd.units) {
                // BinaryDense / QuantizedDense .call + BN, the reference's op order (as k_dense_packed)
                float v = __fmul_rn((float)tot, hd.scale);
                if (hd.bias) v = __fadd_rn(v, hbias);
                if (hd.bn_inv) v = __fadd_rn(__fmul_rn(v, hinv), hshift);
                hd.y[(long)tile * hd.units + u] = v;         // tile == image: a 4 x 4 pooled map is one 64-row tile
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

// ---------------------------------------------------------------------------------------------------------
// k_conv_mfma_halo (round 3): the pooled int4 layers of the CIFAR VGG (int4 in, Cin 64, 3x3 stride 1 SAME, 2x2 pool,
// int4 out -- BinaryConv2D / QuantizedConv2D.call + BN + clip + MaxPooling2D, models/vgg.py:21-36) with the PIXEL
// operand staged through LDS instead of fetched per tap.
//
// k_conv_mfma_areg above fetches and widens every input pixel once per tap: 9 x (2 loads, 2 address selects, 24
// widening instructions) per 64-row tile, 40 % of the kernel's vector instructions, and the headline pipeline is bound
// by the sum of its instruction streams.  Here a wave's 64 rows are a RECTANGLE of one image (TWP x
// 16/TWP pooled pixels = 2 TWP x 32/TWP conv positions), so the tile's receptive field is a (2 TWP + 2) x (32/TWP + 2)
// pixel region: it is fetched once (4 x 16-byte loads per lane, requested one tile ahead), widened once (code * 16, the
// same byte order as the areg kernel, so the filter image is shared) and written to a WAVE-PRIVATE LDS region; the A
// fragments of all nine taps are then ds_read_b128 with immediate offsets off four per-lane base addresses -- no
// per-tap address arithmetic, no masks: pixels outside the image are zeros in the region (rows above / below fall out
// of the image's buffer descriptor, the left / right halo lanes get an out-of-range offset).
//
// Region layout: two planes (k-block kk = channels [32 kk, 32 kk + 32) permuted as the areg operand), each
// [row][pixel, rows padded to a multiple of 4 pixels = 128 B][2 sixteen-byte slots]; slot = lane half lh XOR (row & 1).
// ds_read_b128 is served in groups of 16 lanes of one half (MI355X_MICROARCH.md, LDS): the 16 pixels of a group are
// 8 distinct columns modulo 8 on two rows of different parity (TWP 8), or 4 + 4 columns covering all residues on two
// row pairs (TWP 4), for every tap; a row pitch of 128 (mod 256) bytes shifts the columns of every other row by four,
// which permutes the residues -- 16 distinct slots of the 256-byte bank row either way, conflict-free.
// Filters: LDS, shared by the workgroup's NW waves (as areg).  No barrier in the main loop: a wave's LDS operations
// execute in order, and the region is its own.
// FOLD (round 4): the epilogue as qnn_fold.h's "bits" form -- the accumulators start from the channel's offset (which carries
// the float's bit pattern), the 2x2 window is pooled on the raw integers as before, and a pooled value costs one v_fma_f32 +
// half a v_cvt_pknorm_i16_f32 + half a packing instruction instead of cvt, add, mul, add, add, v_med3 and a shift-add.
template <int TWP, int NW, bool HEAD, bool FOLD = false>
__global__ __launch_bounds__(NW * 64, 1) void k_conv_mfma_halo(MfmaGeom mg, EpiArgs e, const uint8_t* __restrict__ x,
                                                               const uint8_t* __restrict__ wq8, void* __restrict__ y,
                                                               int ntiles, FastDiv fd_tpi, int txn, FastDiv fd_txn,
                                                               uint32_t img_bytes, HeadArgs hd) {
    constexpr int THP = 16 / TWP;
    constexpr int RW = 2 * TWP + 2, RH = 2 * THP + 2;     // region, pixels
    constexpr int NCH = 2 * RW * RH;                       // 16-byte chunks of packed input (32 B per pixel)
    constexpr int RWP = (RW + 3) & ~3;
    constexpr int PITCH = RWP * 32, PLANE = RH * PITCH, REGION = 2 * PLANE;
    constexpr int B_STEP = 64 * 64, FILT = 9 * B_STEP;
    constexpr int HTAB = FILT + NW * REGION;
    static_assert(NCH <= 256, "four load rounds per lane");
    const ConvGeom& g = mg.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbase = blockIdx.y * 64;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);

    // ---- the slice's filters -> LDS (first four waves; the areg kernel's image and swizzle) ----
    if (tid < 256) {
        const int srow = tid >> 2, sch = tid & 3;
        const int wv = (nbase + srow) * (9 * 64) + sch * 16;
        uint4 wreg[9];
#pragma unroll
        for (int st = 0; st < 9; ++st)
            wreg[st] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wv, st * 64, 0));
#pragma unroll
        for (int st = 0; st < 9; ++st)
            *reinterpret_cast<uint4*>(smem + st * B_STEP + srow * 64 + ((sch ^ ((srow >> 2) & 3)) << 4)) = wreg[st];
    }

    // ---- epilogue constants (as k_conv_mfma_areg) ----
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float mfold = binary ? 1.0f : e.act_m;
    LaneEpi ke[2];
    FoldEpi fe[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        lane_epi_init<QNN_STORE_I4>(ke[b], e, nbase + b * 32 + li, li);
        fe[b].nb = __fdiv_rn(ke[b].bias, e.scale);
        fe[b].ninv = __fmul_rn(__fmul_rn(ke[b].inv, e.scale), mfold);
        fe[b].nshift = __fmul_rn(ke[b].shift, mfold);
    }
    const bool all_pos = !__any((int)(ke[0].neg || ke[1].neg));
    float fda[2] = {0.0f, 0.0f}, fdc[2] = {0.0f, 0.0f};
    int fdb[2] = {0, 0};
    if constexpr (FOLD) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = nbase + b * 32 + li;
            fda[b] = e.fold_a[c]; fdc[b] = e.fold_c[c]; fdb[b] = e.fold_b[c];
        }
    }
    int hu = 0;
    float hbias = 0.0f, hinv = 1.0f, hshift = 0.0f;
    if constexpr (HEAD) {
        for (int i = tid; i < 16 * 64 * 2; i += NW * 64) reinterpret_cast<uint32_t*>(smem + HTAB)[i] = hd.tab[i];
        hu = ((lane & 1) ? 8 : 0) + ((lane & 2) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 8) ? 1 : 0);
        if (hu < hd.units) {
            hbias = hd.bias ? hd.bias[hu] : 0.0f;
            hinv = hd.bn_inv ? hd.bn_inv[hu] : 1.0f;
            hshift = hd.bn_inv ? hd.bn_shift[hu] : 0.0f;
        }
    }
    // this lane's output word after the nibble transpose: local pooled pixel lane_row of the tile's TWP-wide rectangle
    const int jl = li & 7;
    const int lane_row = 2 * (jl & 3) + lh + 8 * (jl >> 2);
    const int lane_off = ((lane_row / TWP) * g.Wp + (lane_row % TWP)) * e.ocw + ((nbase + li) >> 3);

    // ---- staging lanes: chunk j = lane + 64 i is half (j & 1) of region pixel j >> 1 ----
    const int regbase = FILT + wave * REGION;
    int rel[4], wr_addr[4];
    unsigned long long edgeL[4], edgeR[4];
    bool wr_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        const int p = j >> 1, half = j & 1;
        const int ry = p / RW, rx = p - ry * RW;
        wr_ok[i] = j < NCH;
        rel[i] = wr_ok[i] ? (ry * g.W + rx) * 32 + half * 16 : (int)0x80000000;
        wr_addr[i] = regbase + ry * PITCH + rx * 32 + ((half ^ (ry & 1)) << 4);
        edgeL[i] = __ballot(rx == 0);
        edgeR[i] = __ballot(rx == RW - 1);
    }
    // ---- A fragments: rows mt * 32 + li = 4 * (pooled pixel) + (position in its 2x2 window) ----
    int a_even[2], a_odd[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int q = 8 * mt + (li >> 2), sub = li & 3;
        const int ry0 = 2 * (q / TWP) + (sub >> 1), rx0 = 2 * (q % TWP) + (sub & 1);
        a_even[mt] = regbase + ry0 * PITCH + rx0 * 32 + ((lh ^ (ry0 & 1)) << 4);
        a_odd[mt] = a_even[mt] ^ 16;
    }
    int fb_addr[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb_addr[kk] = li * 64 + (((lh * 2 + kk) ^ ((li >> 2) & 3)) << 4);

    const int t_stride = gridDim.x * NW;
    int t = blockIdx.x * NW + wave;

    uint4 L[4];
    long pq_next = 0;
    int n_next = 0;
    auto fetch = [&](int tile) {                       // tile -> (image, tile row, tile column): wave-uniform
        if (tile >= ntiles) return;
        const int n = (int)qnn_div((uint32_t)tile, fd_tpi);
        const int rest = tile - n * (int)fd_tpi.d;
        const int ty = (int)qnn_div((uint32_t)rest, fd_txn);
        const int tx = rest - ty * txn;
        n_next = n;
        pq_next = ((long)n * g.Hp + ty * THP) * g.Wp + tx * TWP;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(x) + (size_t)n * img_bytes, 0, (int)img_bytes, 0x00020000);
        const int origin = ((2 * ty * THP - 1) * g.W + 2 * tx * TWP - 1) * 32;
        const bool tl = tx == 0, tr = tx == txn - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long m = (tl ? edgeL[i] : 0ull) | (tr ? edgeR[i] : 0ull);
            const bool out = __builtin_amdgcn_inverse_ballot_w64(m);
            const int voff = out ? (int)0x80000000 : rel[i] + origin;
            L[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0));
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = L[i];
            const uint4 k0 = make_uint4((q.x << 4) & 0xF0F0F0F0u, q.x & 0xF0F0F0F0u, (q.y << 4) & 0xF0F0F0F0u, q.y & 0xF0F0F0F0u);
            const uint4 k1 = make_uint4((q.z << 4) & 0xF0F0F0F0u, q.z & 0xF0F0F0F0u, (q.w << 4) & 0xF0F0F0F0u, q.w & 0xF0F0F0F0u);
            if (i < 3 || wr_ok[i]) {
                *reinterpret_cast<uint4*>(smem + wr_addr[i]) = k0;
                *reinterpret_cast<uint4*>(smem + wr_addr[i] + PLANE) = k1;
            }
        }
    };

    v16i acc[2][2];
    auto bn = [&](int v, const FoldEpi& f) {
        return __fadd_rn(__fmul_rn(__fadd_rn((float)v, f.nb), f.ninv), f.nshift);
    };
    auto epilogue = [&](long pq0, int image) {
        uint32_t* ytile = reinterpret_cast<uint32_t*>(y) + pq0 * e.ocw;
        int dacc[16];
        if constexpr (HEAD) {
#pragma unroll
            for (int u = 0; u < 16; ++u) dacc[u] = 0;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float tv[8];
            int pv[8];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int i0 = acc[a][b][4 * g4], i1 = acc[a][b][4 * g4 + 1];
                    const int i2 = acc[a][b][4 * g4 + 2], i3 = acc[a][b][4 * g4 + 3];
                    const int mx = max(max(i0, i1), max(i2, i3));
                    int pooled = mx;
                    if (!all_pos) {
                        const int mn = min(min(i0, i1), min(i2, i3));
                        pooled = ke[b].neg ? mn : mx;
                    }
                    if constexpr (FOLD) pv[a * 4 + g4] = pooled;
                    else tv[a * 4 + g4] = bn(pooled, fe[b]);
                }
            uint32_t Q;                                  // eight two's-complement codes, nibble j = value j
            if constexpr (FOLD) {
                uint32_t tp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tp[j] = qnn_fold_pair_bits(pv[j], pv[j + 4], fda[b], fda[b], fdc[b], fdc[b]);
                const uint32_t uo = __builtin_amdgcn_perm(tp[3], tp[1], 0x07030501u);
                const uint32_t ue = __builtin_amdgcn_perm(tp[2], tp[0], 0x07030501u);
                Q = (uo & 0xF0F0F0F0u) | ((ue >> 4) & 0x0F0F0F0Fu);
            } else {
                Q = pack_scaled<4, 8>(tv, e.act_m, binary) ^ 0x88888888u;
            }
            if constexpr (HEAD) {
                const uint32_t* trow = reinterpret_cast<const uint32_t*>(smem + HTAB) + lane * 2 + b;
#pragma unroll
                for (int u = 0; u < 16; ++u) dacc[u] = __builtin_amdgcn_sdot8((int)Q, (int)trow[u * 128], dacc[u], false);
            } else {
                ytile[lane_off + b * 4] = transpose_nib8(Q, ke[0]);       // (a permutation of nibbles: the codes pass through)
            }
        }
        if constexpr (HEAD) {
            int u;
            const int tot = head_reduce16(dacc, lane, u);
            if (lane < 16 && u < hd.units) {
                float v = __fmul_rn((float)tot, hd.scale);
                if (hd.bias) v = __fadd_rn(v, hbias);
                if (hd.bn_inv) v = __fadd_rn(__fmul_rn(v, hinv), hshift);
                hd.y[(long)image * hd.units + u] = v;
            }
        }
    };

    fetch(t);
    __syncthreads();                                   // filters (and the classifier table) are in LDS
    for (; t < ntiles; t += t_stride) {
        const long pq0 = pq_next;
        const int image = n_next;
        stage();                                       // this tile's region (waits for its loads)
        fetch(t + t_stride);                           // the next tile's loads fly under this tile's MFMAs
        // 18 K-steps (tap, kk); the fragments of step s + 1 are requested before the MFMAs of step s are issued (two register
        // sets), so a step's LDS latency lies under the previous step's 128 matrix-pipe cycles
        v4i fa[2][2], fb[2][2];
        auto frags = [&](int st, v4i (&A)[2], v4i (&B)[2]) {
            const int tap = st >> 1, kk = st & 1;
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                A[mt] = *reinterpret_cast<const v4i*>(smem + (dy == 1 ? a_odd[mt] : a_even[mt]) + kk * PLANE +
                                                      dy * PITCH + dx * 32);
#pragma unroll
            for (int b = 0; b < 2; ++b)
                B[b] = *reinterpret_cast<const v4i*>(smem + fb_addr[kk] + tap * B_STEP + b * 2048);
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) frags(st + 1, fa[(st + 1) & 1], fb[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (st == 0) {
                        const int z0 = FOLD ? fdb[b] : 0;      // the fold's offset (with the magic constant): this lane's channel
                        const v16i z = {z0, z0, z0, z0, z0, z0, z0, z0, z0, z0, z0, z0, z0, z0, z0, z0};
                        acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[st & 1][a], fb[st & 1][b], z, 0, 0, 0);
                    } else {
                        acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[st & 1][a], fb[st & 1][b], acc[a][b], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        epilogue(pq0, image);
    }
}

template <int TWP, int NW, bool HEAD, bool FOLD = false>
void launch_halo_one_f(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, const HeadArgs& hd,
                       hipStream_t s) {
    const ConvGeom& g = mg.g;
    constexpr int THP = 16 / TWP;
    const int txn = g.Wp / TWP, tyn = g.Hp / THP;
    const int tpi = txn * tyn;
    const int ntiles = g.N * tpi;
    constexpr int RW = 2 * TWP + 2, RH = 2 * THP + 2, RWP = (RW + 3) & ~3;
    const size_t lds = (size_t)9 * 4096 + (size_t)NW * (2 * RH * RWP * 32) + (HEAD ? 16 * 64 * 2 * 4 : 0);
    const int ny = g.cout / 64;
    int gx = (ntiles + NW - 1) / NW;
    const int cap = 256 / ny > 0 ? 256 / ny : 1;             // one resident workgroup per CU
    if (gx > cap) gx = cap;
    static const bool lds_ok = [] {
        (void)hipFuncSetAttribute((const void*)k_conv_mfma_halo<TWP, NW, HEAD, FOLD>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)lds_ok;
    hipLaunchKernelGGL((k_conv_mfma_halo<TWP, NW, HEAD, FOLD>), dim3((unsigned)gx, (unsigned)ny), dim3(NW * 64), lds, s, mg, e,
                       (const uint8_t*)x, w, y, ntiles, qnn_fastdiv((uint32_t)tpi), txn, qnn_fastdiv((uint32_t)txn),
                       (uint32_t)(g.H * g.W * 32), hd);
}

// a usable fold in the "bits" form (qnn_fold.h mode 2: e.fold_c set) takes the folded epilogue
template <int TWP, int NW, bool HEAD>
void launch_halo_one(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, const HeadArgs& hd,
                     hipStream_t s) {
    if (e.fold_a && e.fold_c && e.fn == QNN_FN_QUANTIZED_TANH) launch_halo_one_f<TWP, NW, HEAD, true>(mg, e, x, w, y, hd, s);
    else launch_halo_one_f<TWP, NW, HEAD, false>(mg, e, x, w, y, hd, s);
}

// 0 = launched.  The pooled map must tile into 8 x 2 or 4 x 4 rectangles; everything else stays on k_conv_mfma_areg.
int halo_width(const MfmaGeom& mg, const EpiArgs& e) {
    const ConvGeom& g = mg.g;
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.pt != 1 || g.pl != 1 || g.pool != 2 || mg.kc != 1) return 0;
    if (g.cout % 64 != 0 || e.out_store != QNN_STORE_I4 || e.res || (g.H & 1) || (g.W & 1)) return 0;
    if ((long)g.N * g.Hp * g.Wp / 16 >= 2000000000L) return 0;
    if (g.Wp % 8 == 0 && g.Hp % 2 == 0) return 8;
    if (g.Wp % 4 == 0 && g.Hp % 4 == 0) return 4;
    return 0;
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
        hipLaunchKernelGGL((k_conv_mfma_areg<XS, OUT, 2, KC>), grid, block, lds, s, mg, e, (const uint8_t*)x, w, y, ntiles, HeadArgs{});
    else
        hipLaunchKernelGGL((k_conv_mfma_areg<XS, OUT, 1, KC>), grid, block, lds, s, mg, e, (const uint8_t*)x, w, y, ntiles, HeadArgs{});
}

template <int KC>
void launch_areg_head(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, const HeadArgs& hd, hipStream_t s) {
    const int ntiles = (int)((mg.total_q * 4 + 63) / 64);
    int gx = (((ntiles + 3) / 4 + 7) / 8) * 8;
    if (gx > 768) gx = 768;
    const size_t lds = (size_t)9 * KC * 64 * 64 + 16 * 64 * 2 * 4;
    static const bool lds_ok = [] {
        (void)hipFuncSetAttribute((const void*)k_conv_mfma_areg<QNN_STORE_I4, QNN_STORE_I4, 2, KC, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        return true;
    }();
    (void)lds_ok;
    hipLaunchKernelGGL((k_conv_mfma_areg<QNN_STORE_I4, QNN_STORE_I4, 2, KC, true>), dim3((unsigned)gx, 1), dim3(256), lds, s, mg, e,
                       (const uint8_t*)x, w, (void*)nullptr, ntiles, hd);
}

// dense int4 codes [units][1024 / 8 words] -> the per-lane table of the fused classifier (see HeadArgs)
__global__ __launch_bounds__(256) void k_head_table(const uint32_t* __restrict__ dp, uint32_t* __restrict__ tab, int units) {
    const int i = blockIdx.x * 256 + threadIdx.x;            // (u * 64 + lane) * 2 + b
    if (i >= 16 * 64 * 2) return;
    const int b = i & 1, lane = (i >> 1) & 63, u = i >> 7;
    const int li = lane & 31, lh = lane >> 5;
    uint32_t word = 0;
    if (u < units)
        for (int j = 0; j < 8; ++j) {
            const int f = (8 * (j >> 2) + 2 * (j & 3) + lh) * 64 + b * 32 + li;     // Flatten index of (pooled pixel, channel)
            word |= ((dp[u * 128 + (f >> 3)] >> (4 * (f & 7))) & 0xFu) << (4 * j);
        }
    tab[i] = word;
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

}  // namespace

int qnn_launch_areg(int x_store, int kc, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                    void* y, hipStream_t s) {
    if (x_store == QNN_STORE_I8)
        return kc == 1 ? launch_areg<QNN_STORE_I8, 1>(mg, e, x, w, y, s) : launch_areg<QNN_STORE_I8, 2>(mg, e, x, w, y, s);
    return kc == 1 ? launch_areg<QNN_STORE_I4, 1>(mg, e, x, w, y, s) : launch_areg<QNN_STORE_I4, 2>(mg, e, x, w, y, s);
}

// Waves per workgroup (they share one LDS copy of the filters): 8 = two per SIMD once there are tiles for every CU's eight
// (measured on 4096 x 16^2 -> 8^2: one / two / three / four waves per SIMD = 42.9 / 32.3 / 33.8 / 35.3 us),
// 4 below that so that small batches still reach every CU.
static int halo_waves(const MfmaGeom& mg, int tw) {
    const long ntiles = (long)mg.g.N * (mg.g.Wp / tw) * (mg.g.Hp / (16 / tw));
    static const int forced = QNN_ENV_INT("QNN_HALO_NW", 0);
    if (forced == 4 || forced == 8) return forced;
    return ntiles >= 256 * 8 ? 8 : 4;
}

int qnn_launch_halo(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s) {
    const int tw = halo_width(mg, e);
    if (tw == 0) return 1;
    const int nw = halo_waves(mg, tw);
    const HeadArgs none{};
    if (tw == 8) {
        if (nw == 8) launch_halo_one<8, 8, false>(mg, e, x, w, y, none, s);
        else launch_halo_one<8, 4, false>(mg, e, x, w, y, none, s);
    } else {
        if (nw == 8) launch_halo_one<4, 8, false>(mg, e, x, w, y, none, s);
        else launch_halo_one<4, 4, false>(mg, e, x, w, y, none, s);
    }
    return 0;
}

int qnn_launch_wres(int x_store, const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w,
                    void* y, hipStream_t s) {
    return x_store == QNN_STORE_I8 ? launch_wres<QNN_STORE_I8>(mg, e, x, w, y, s)
                                   : launch_wres<QNN_STORE_I4>(mg, e, x, w, y, s);
}


// ---- fused conv + classifier (qnn_conv2d_dense_forward): table at prepack time, launch ----
int qnn_head_prepare(qnn_weights* w, hipStream_t s) {
    w->d_head = nullptr;
    if (w->kh != 1 || w->kw != 1 || w->cin != 1024 || w->cout > 16 || w->store != QNN_STORE_I4 || !w->d_packed) return QNN_OK;
    QNN_HIP(hipMalloc((void**)&w->d_head, 16 * 64 * 2 * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_head_table, dim3(8), dim3(256), 0, s, w->d_packed, w->d_head, w->cout);
    QNN_HIP(hipGetLastError());
    return QNN_OK;
}

// 0 = launched.  Conv: int4 in, 3x3 stride 1 SAME, Cin 64 / 128, 64 filters, 2x2 pool, int4 codes out, 4 x 4 pooled map;
// dense: the matching 1024 -> <= 16 head prepacked for int4.  `ed` is the dense layer's epilogue (float32 out, no fn).
int qnn_launch_areg_head(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, const qnn_weights* wd,
                         const EpiArgs& ed, float* y, hipStream_t s, const char** kname) {
    const ConvGeom& g = mg.g;
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.cout != 64 || g.pool != 2 || (mg.kc != 1 && mg.kc != 2)) return 1;
    if (g.Hp * g.Wp != 16 || e.out_store != QNN_STORE_I4 || e.res || !wd->d_head || wd->cin != 1024) return 1;
    HeadArgs hd;
    hd.tab = wd->d_head; hd.bias = ed.bias; hd.bn_inv = ed.bn_inv; hd.bn_shift = ed.bn_shift;
    hd.scale = ed.scale; hd.units = wd->cout; hd.y = y;
    if (!(e.flags & QNN_EPI_NO_HALO) && halo_width(mg, e) == 4 && g.Wp == 4) {   // tile == image
        if (halo_waves(mg, 4) == 8) launch_halo_one<4, 8, true>(mg, e, x, w, nullptr, hd, s);
        else launch_halo_one<4, 4, true>(mg, e, x, w, nullptr, hd, s);
        *kname = "mfma_i4_halo64x64+dense";
        return 0;
    }
    *kname = "mfma_i4_areg64x64+dense";
    if (mg.kc == 1) launch_areg_head<1>(mg, e, x, w, hd, s);
    else launch_areg_head<2>(mg, e, x, w, hd, s);
    return 0;
}
