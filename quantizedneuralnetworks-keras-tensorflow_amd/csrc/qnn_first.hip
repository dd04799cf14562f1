// Float-input first layer on the f32 matrix pipe (v_mfma_f32_32x32x2_f32): exact k-ordered FMA chain.
// Dispatch: qnn_try_launch_mfma (qnn_mfma.hip).
#include "qnn_mfma_common.h"

namespace {

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
#ifndef QNN_FIRST_PRIO
// 1 = a wave's MFMA phase runs at raised priority (s_setprio 1): the other waves' epilogue VALU then only takes the
// issue slots the matrix chain leaves (measured 141 -> 133 us; raising the epilogue instead: 135 us; 0 = off)
#define QNN_FIRST_PRIO 1
#endif

constexpr int first_lds_row_stride(int roww) {          // smallest stride >= roww with stride % 32 == 16
    int rs = (roww / 32) * 32 + 16;
    return rs >= roww ? rs : rs + 32;
}

template <int CIN, int NT, int OUT, int POOL>
__global__ __launch_bounds__(256, QNN_FIRST_WPS) void k_conv_first_lds(ConvGeom g, EpiArgs e,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ wq,
                                                           void* __restrict__ y, long total_q,
                                                           int tiles, int tiles_per_row,
                                                           FastDiv fd_tpr, uint32_t x_bytes, int abl) {
    constexpr int K = 9 * CIN;
    constexpr int KS = (K + 1) / 2;
    constexpr int TROWS = (POOL == 2) ? 4 : 3;        // conv rows + halo
    constexpr int TCOLS = (POOL == 2) ? 18 : 34;      // conv cols + halo
    constexpr int ROWW = TCOLS * CIN;                 // floats per tile row
    // LDS row stride.  The 32 lanes of an operand read sit on two tile rows (POOL == 2: lrow in {0,1},
    // lcol in 0..15): word = lrow*RS + lcol*CIN + const.  lcol*CIN covers 16 distinct banks and so does the
    // second row iff RS == 16 (mod 32) (3*16 == 16, 1*16 == 16): conflict-free ds_read_b32.  With the natural
    // stride 54 (CIN 3) two banks collided in every read (42 % of the LDS cycles were conflict cycles).
    constexpr int RS = (POOL == 2) ? first_lds_row_stride(ROWW) : ROWW;
    constexpr int TE = TROWS * ROWW;                  // floats staged per tile
    constexpr int TL = TROWS * RS;                    // LDS words per tile
    constexpr int NJ = (TE + 63) / 64;                // staging loads per lane
    constexpr bool PACKED = OUT == QNN_STORE_I4 || OUT == QNN_STORE_I8;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* lds = reinterpret_cast<float*>(smem_f) + wv * TL;          // wave-private tile
    const int wave_id = blockIdx.x * 4 + wv;
    const int nwaves = gridDim.x * 4;
    const int cbase = blockIdx.y * (NT * 32);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
    const bool binary = e.fn == QNN_FN_BINARY_TANH;
    const float mfold = (PACKED && !binary) ? e.act_m : 1.0f;

    // ---- per-lane constants ----
    // The block's NT*32 filters are one contiguous run of wq: copy it to LDS with coalesced loads and let every
    // lane pick its operands from there (lane stride K = 27 or 9 words: odd, so conflict-free).  Fetched straight
    // from global memory the 28 operand loads of a wave touch 64 different cache lines each -- 1 792 address
    // cycles per wave, 9 us of a 141 us launch for the 12 waves of a CU (measured: the launch time extrapolates
    // to 11-14 us at zero images).
    float* fw = reinterpret_cast<float*>(smem_f) + 4 * TL;
    for (int i = threadIdx.x; i < NT * 32 * K; i += 256) fw[i] = wq[(long)cbase * K + i];
    __syncthreads();
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
            float w = k < K ? fw[(nt * 32 + li) * K + k] : 0.0f;
            wb[nt][s] = flip ? -w : w;
        }
    }
    // staging: element ej = lane + 64*j of the [TROWS][TCOLS][CIN] tile
    int st_goff[NJ], st_lidx[NJ];
    unsigned long long mX[NJ], mT[NJ], mB[NJ], mL[NJ], mR[NJ];   // lanes outside the tile / on each halo edge
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ej = lane + 64 * j;
        const int r = ej / ROWW, rem = ej - r * ROWW;
        const int col = rem / CIN, ch = rem - col * CIN;
        st_goff[j] = ((r * g.W + col) * CIN + ch) * 4;
        st_lidx[j] = r * RS + rem;
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
        op_idx[s] = (lrow + tap / 3) * RS + (lcol + tap % 3) * CIN + ch;
    }
    // (lane half 1 of the last k-step is padding when K is odd: its filter operand is 0.0f and its A operand is
    // tap 0 / channel 0 of the lane's own receptive field, a value the true sum contains anyway)
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
            if (lane + 64 * j < TE) lds[st_lidx[j]] = stg[j];
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
    // the k-th tile of this wave sits k*x_step bytes further: a SCALAR offset on the buffer load (no VALU);
    // clamped to the wave's last tile, because the two loads issued past the end must stay inside the tensor
    const int my_tiles = (tiles - wave_id + nwaves - 1) / nwaves;
    int kload = 0;
    auto stage_load_next = [&]() {                                  // periodic mode: the next tile of this wave
        const int soff = min(kload, my_tiles - 1) * x_step;
        ++kload;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            stg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, pvoff[j], soff, 0));
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
#pragma unroll
        for (int nc = 0; nc < NT; nc += 2) {
            v16f acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
#if QNN_FIRST_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#elif QNN_FIRST_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wb[nc + u][s], acc[u], 0, 0, 0);
#if QNN_FIRST_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#elif QNN_FIRST_PRIO == 2
            __builtin_amdgcn_s_setprio(1);
#endif
            if (nc + 2 >= NT && !QNN_ABLATE(abl, 2)) {
                __builtin_amdgcn_sched_barrier(0);
                stage_write(0);
                fetch_operands();
                if (periodic) stage_load_next(); else stage_load(min(t + 2 * nwaves, tiles - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (QNN_ABLATE(abl, 1)) {                       // timing experiment: no epilogue arithmetic, one raw store
                if (acc[0][0] + acc[1][0] == 123.456f) ytile[lane_off + nc * 4] = 1u;
                continue;
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
    static const int no_lds = QNN_ENV_INT("QNN_FIRST_GATHER", 0);
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
            const size_t lds_bytes = (size_t)4 * (g.pool == 2 ? 4 * first_lds_row_stride(18 * CIN) : 3 * 34 * CIN) * 4   // one tile per wave
                                     + (size_t)NT * 32 * 9 * CIN * 4;                                                  // + the block's filters
            static const int abl = QNN_ENV_INT("QNN_FIRST_ABL", 0);   // timing ablations: experiment builds only
#define FIRST_LDS_CASE(OUT)                                                                      \
            if (e.out_store == OUT) {                                                            \
                if (g.pool == 2)                                                                 \
                    hipLaunchKernelGGL((k_conv_first_lds<CIN, NT, OUT, 2>), lgrid, block, lds_bytes, s, g, e, xf, wq, y, total_q, (int)ntiles, tpr, fd_tpr, x_bytes, abl); \
                else                                                                             \
                    hipLaunchKernelGGL((k_conv_first_lds<CIN, NT, OUT, 1>), lgrid, block, lds_bytes, s, g, e, xf, wq, y, total_q, (int)ntiles, tpr, fd_tpr, x_bytes, abl); \
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

}  // namespace

int qnn_launch_first(int cin, int nt, const ConvGeom& g, const EpiArgs& e, const void* x, const float* wq,
                     void* y, hipStream_t s) {
    if (nt == 2) return cin == 3 ? launch_first<3, 2>(g, e, x, wq, y, s) : launch_first<1, 2>(g, e, x, wq, y, s);
    return cin == 3 ? launch_first<3, 4>(g, e, x, wq, y, s) : launch_first<1, 4>(g, e, x, wq, y, s);
}
