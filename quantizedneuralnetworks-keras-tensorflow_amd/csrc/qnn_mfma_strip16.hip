// The 16 -> 16 channel 3x3 stride-1 int4 layers (the 224 x 224 stage of the ImageNet ResNet, models/resnet.py:104-129:
// 20 of its 63 convolutions and 40 % of its kernel time) with the folded epilogue (qnn_fold.h), restructured around what
// round 4 measured on k_conv_strip<16, 1, ., ., FOLD> (profiles/r04): once the epilogue is 13 instead of 24-36 vector
// instructions per row the kernel is bound by its VECTOR-MEMORY INSTRUCTIONS, not by arithmetic -- per 16-pixel row one
// 8-byte load, one 2-byte store and (with a merge) one 2-byte shortcut load, each a full pass of 64 lanes through the
// texture-address unit for 128 useful bytes.  Ablations: 20.1 us as is, 16.1 us with one store per eight rows, 14.3 us
// without the input loads, 12.0 us with neither.  So the memory instructions are made few and wide:
//
//   * INPUT: staged through LDS.  One 16-byte load per lane brings SIX rows of the strip (20 pixels = 160 bytes per row
//     incl. the halo, ten lanes per row) into four registers; six rows later they are widened to code * 16 bytes (once
//     per pixel instead of once per tap and row) and two ds_write_b128 put them into a wave-private ring of two 2-KiB
//     slots; a row's MFMA operand is one ds_read_b128 per lane at an immediate offset, no arithmetic: 1/6 instead of 1
//     vector-memory instruction and 2 instead of 6 widening instructions per row.  (LDS-DMA, buffer_load ... lds, would save the registers,
//     but the compiler guards every LDS read behind a pending DMA with s_waitcnt vmcnt(0) and hoists reads over hand
//     placed counts -- measured in the ISA; ordinary loads are counted exactly by the compiler itself.)
//   * SHORTCUT (residual merge): the same, eight lanes per row, six rows per load; a lane's four codes are one
//     ds_read_u16.
//   * OUTPUT: a lane's 16-bit field (four channels of one pixel) of FOUR consecutive rows is transposed across the four
//     16-lane groups of the wave with v_permlane32_swap / v_permlane16_swap (two each, new in gfx950), after which lane
//     (pixel r, group j) holds all 16 channels of pixel r in row y + j: one 8-byte store per lane and four rows.
//   * everything periodic is unrolled twelve rows deep (lcm of the six-row DMA groups, the four-row store groups and
//     the three rotating operand sets) so that every LDS offset, ring slot and s_waitcnt count is an immediate.
//
// Schedule of one period (12 rows; slot e / o = ring slot 0 / 1): after row 3 the input group loaded six rows earlier
// goes into slot e (its rows were last read by row 3) and the next group's load is issued; after row 9 the same for
// slot o; the shortcut ring does it after rows 5 and 11.  A load has six rows (~3 us) to land before its ds_write.
// The arithmetic is k_conv_strip's: A = filters, B = pixels, K block kq = tap dx (kq = 3: zero filter block), the
// fold's offset as the initial accumulator.  Results are bit-identical to k_conv_strip and to the float32 chain
// (tests/test_gpu_fold.py).  Needs an even image width (the DMA moves pixel PAIRS) and a usable fold; everything else
// stays on k_conv_strip.
#include "qnn_mfma_common.h"
#include "qnn_fold.h"

namespace {

constexpr int kXSlot = 2048, kSSlot = 1024; // bytes of a six-row group in LDS: 64 lanes x 32 bytes (widened) / x 16 bytes
constexpr int kXRow = 320, kSRow = 128;     // bytes per staged input row (20 widened pixels) / shortcut row (16 pixels)

// Occupancy (round 4, 64 x 224 x 224: 21.9 us at 6 waves per SIMD, 19.5 / 18.4 / 16.6 / 18.0 / 26.6 us at 5 / 4 / 3 / 2 / 1): the
// persistent grid runs TWO to THREE waves per SIMD -- each wave then owns 76-112 rows of a strip instead of ~38 (fewer ring
// fills) and three waves already cover one another's LDS and matrix-pipe latencies; the register bound is left at four.
#ifndef QNN_S16_BOUNDS
#define QNN_S16_BOUNDS 4
#endif
#ifndef QNN_S16_WPS
#define QNN_S16_WPS 2            // (3 is the fastest for the layer alone, 16.6 against 18.0 us; with two batches in flight 2 leaves
#endif                           // the other batch's kernels more room: ResNet-224 end to end 77.4 -> 78.9 K img/s)
template <bool RES, int FOLD>
__global__ __launch_bounds__(256, QNN_S16_BOUNDS) void k_conv_strip16_lds(MfmaGeom mg, EpiArgs e, const uint8_t* __restrict__ x,
                                                             const uint8_t* __restrict__ wq8, void* __restrict__ y,
                                                             int ntasks, int spr, FastDiv fd_spr, int nch, FastDiv fd_nch,
                                                             int rc, uint32_t img_x, uint32_t img_y) {
    constexpr int WB = 2 * kXSlot + (RES ? 2 * kSSlot : 0);      // LDS bytes per wave
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * WB];
    const ConvGeom& g = mg.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    uint8_t* const xs_lds = smem + wave * WB;                    // two input slots, then two shortcut slots
    uint8_t* const ss_lds = xs_lds + 2 * kXSlot;

    // ---- filters: A operand, row = output channel (as k_conv_strip<16, 1>) ----
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wq8), 0, (int)mg.w_bytes, 0x00020000);
    v4i bw[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int woff = kq < 3 ? (r * 9 + dy * 3 + kq) * 16 : (int)0x80000000;
        bw[dy] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, 0, 0));
    }
    float fa[4], fc[4];
    v4i binit;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = e.fold_a[4 * kq + i]; binit[i] = e.fold_b[4 * kq + i];
        fc[i] = FOLD == 2 ? e.fold_c[4 * kq + i] : 0.0f;
    }
    const int rowb = g.W * 8;                                     // bytes per row: input, output and shortcut alike
    // LDS read addresses of this lane: pixel xs + r + dx - 1 is entry r + dx + 1 of a staged row (it starts at xs - 2)
    const uint8_t* const xrd = xs_lds + (r + (kq < 3 ? kq : 0) + 1) * 16;
    const uint8_t* const srd = ss_lds + r * 8 + kq * 2;
    // DMA lanes: input row = lane / 10 (lanes 60..63 idle), pixel pair lane % 10; shortcut row = lane / 8 (48..63 idle)
    const int xl_row = lane / 10, xl_c = lane - 10 * xl_row;
    const int sl_row = lane >> 3, sl_c = lane & 7;

    auto widen = [&](const uint2& q) -> v4i {
        const uint4 v = make_uint4((q.x << 4) & 0xF0F0F0F0u, q.x & 0xF0F0F0F0u, (q.y << 4) & 0xF0F0F0F0u, q.y & 0xF0F0F0F0u);
        return __builtin_bit_cast(v4i, v);
    };

    for (int task = wid; task < ntasks; task += nw) {
        const uint32_t rest = qnn_div((uint32_t)task, fd_nch);
        const int chunk = task - (int)rest * nch;
        const int n = (int)qnn_div(rest, fd_spr);
        const int xs = ((int)rest - n * spr) * 16;
        const int y0 = chunk * rc;
        const int y1 = min(y0 + rc, g.H);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(x) + (size_t)n * img_x, 0, (int)img_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((uint8_t*)y + (size_t)n * img_y, 0, (int)img_y, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            RES ? (uint8_t*)const_cast<void*>(e.res) + (size_t)n * img_y : (uint8_t*)y, 0, RES ? (int)img_y : 0, 0x00020000);
        // byte offsets (inside the image) of this lane's 16 bytes in the first six-row group; rows above / below the image
        // and pixel pairs left / right of it are out of range = zeros (the width is even: a pair never straddles the edge)
        const int xpx = xs - 2 + 2 * xl_c;
        int xv = (lane < 60 && xpx >= 0 && xpx < g.W) ? ((y0 - 1 + xl_row) * g.W + xpx) * 8 : (int)0x80000000;
        const int spx = xs + 2 * sl_c;
        int sv = (lane < 48 && spx < g.W) ? ((y0 + sl_row) * g.W + spx) * 8 : (int)0x80000000;
        // staging registers: the group a ring slot receives next
        uint4 xg, sg = make_uint4(0, 0, 0, 0);
        auto xload = [&]() {
            xg = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, xv, 0, 0));
            xv += 6 * rowb;
        };
        auto sload = [&]() {
            if constexpr (RES) {
                sg = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rr, sv, 0, 0));
                sv += 6 * rowb;
            }
        };
        // the int4 codes are widened to code * 16 bytes ONCE, on their way into LDS (12 instructions per six rows instead
        // of 6 per row): a row's operand is then one ds_read_b128 and no arithmetic
        auto xput_v = [&](int slot, const uint4& v) {
            v4i* dst = reinterpret_cast<v4i*>(xs_lds + slot * kXSlot + lane * 32);
            dst[0] = widen(make_uint2(v.x, v.y));
            dst[1] = widen(make_uint2(v.z, v.w));
        };
        auto xput = [&](int slot) { xput_v(slot, xg); };
        auto sput = [&](int slot) { if constexpr (RES) *reinterpret_cast<uint4*>(ss_lds + slot * kSSlot + lane * 16) = sg; };
        // output: lane (r, kq) stores pixel xs + r of row (first row of the four-row group) + kq
        const bool pvalid = xs + r < g.W;
        int orow = y0 + kq;
        int ov = (orow * g.W + xs + r) * 8;

        // ---- preamble: groups 0 and 1 of each ring into LDS, group 2 in flight; rows y0 - 1 and y0 into the operand sets ----
        {
            xload(); sload();
            const uint4 x0 = xg, s0 = sg;
            xload(); sload();
            const uint4 x1 = xg, s1 = sg;
            xload(); sload();                                    // (three loads of each ring in flight before the first wait)
            xput_v(0, x0);
            xput_v(1, x1);
            if constexpr (RES) {
                *reinterpret_cast<uint4*>(ss_lds + lane * 16) = s0;
                *reinterpret_cast<uint4*>(ss_lds + kSSlot + lane * 16) = s1;
            }
        }
        // operand sets: input row k (relative to y0) lives in X[k & 3] and at ring position (k + 1) % 12
        v4i X[4];
        X[3] = *reinterpret_cast<const v4i*>(xrd + 0 * kXRow);
        X[0] = *reinterpret_cast<const v4i*>(xrd + 1 * kXRow);
        X[1] = *reinterpret_cast<const v4i*>(xrd + 2 * kXRow);
        X[2] = *reinterpret_cast<const v4i*>(xrd + 3 * kXRow);
        uint32_t R[4] = {0, 0, 0, 0};                            // a four-row group's fields (bytes 1 and 3 of each)

        auto store_group = [&](bool all) {
            // 4 x 4 transpose between the rows of the group (registers) and the wave's four 16-lane groups
            auto a = __builtin_amdgcn_permlane32_swap(R[0], R[2], false, false);
            auto b = __builtin_amdgcn_permlane32_swap(R[1], R[3], false, false);
            auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
            auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
            // lane (r, j) now holds the four fields of pixel r in row j: c[0], c[1], d[0], d[1], each in bytes 1 and 3
            const uint32_t lo = __builtin_amdgcn_perm(c[1], c[0], 0x07050301u);
            const uint32_t hi = __builtin_amdgcn_perm(d[1], d[0], 0x07050301u);
            const int off = (pvalid && (all || orow < y1)) ? ov : (int)0x80000000;
            typedef unsigned int u2v __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u2v{lo, hi}, yr, off, 0, 0);
            ov += 4 * rowb;
            orow += 4;
        };

        // Software pipeline (QNN_S16_PIPE): an iteration issues the three MFMAs of row J + 1 BETWEEN the pieces of row J's
        // epilogue.  A v_mfma_i32_16x16x64_i8 occupies the matrix pipe for 16 cycles but holds the SIMD's vector issue for 8
        // (MI355X_MICROARCH.md, cycle constants): back to back the three dependent MFMAs of a row cost 48 cycles with nothing
        // else issued, with four independent vector instructions behind each of them 3 x (8 + 16).  Left to itself the
        // compiler hoists the MFMAs of several rows into clusters (with s_nop between dependent ones) and leaves the vector
        // work as one long tail; the scheduling barriers below pin the interleaving (vector and matrix instructions may not
        // cross them; memory instructions and scalar code may).
        constexpr int kPin = 0x4 | 0x10 | 0x80;                  // sched_barrier mask: SALU, VMEM, DS may cross
        v4i accp;                                                // accumulators of the row whose epilogue comes next
        {
            accp = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[0], X[3], binit, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[1], X[0], accp, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[2], X[1], accp, 0, 0, 0);
        }
        auto body = [&](auto jc, bool full) {
            constexpr int J = decltype(jc)::value;               // epilogue of row J, MFMAs of row J + 1 (period 12)
            // the operand of the row after next is requested now (input row J + 3 -> X[(J + 3) & 3]): a whole iteration
            // of LDS latency before iteration J + 1 reads it
            constexpr int I = (J + 4) % 12;
            X[(J + 3) % 4] = *reinterpret_cast<const v4i*>(xrd + (I / 6) * kXSlot + (I % 6) * kXRow);
            uint32_t scf = 0;
            if constexpr (RES) {
                scf = *reinterpret_cast<const unsigned short*>(srd + (J / 6) * kSSlot + (J % 6) * kSRow);
            }
            v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[0], X[J % 4], binit, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(kPin);
            // folded epilogue of row J (qnn_fold.h), pairs (c0, c2), (c1, c3)
            uint32_t pe = qnn_fold_pair_m<FOLD>(accp[0], accp[2], fa[0], fa[2], fc[0], fc[2]);
            uint32_t y2 = 0;
            if constexpr (RES) {
                const uint32_t w = scf ^ 0x8888u;
                y2 = __builtin_amdgcn_perm(0u, w, 0x0C010C00u);
            }
            __builtin_amdgcn_sched_barrier(kPin);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[1], X[(J + 1) % 4], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(kPin);
            uint32_t po = qnn_fold_pair_m<FOLD>(accp[1], accp[3], fa[1], fa[3], fc[1], fc[3]);
            if constexpr (RES) pe = qnn_fold_merge(pe, (y2 & 0x000F000Fu) << 10);
            __builtin_amdgcn_sched_barrier(kPin);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(bw[2], X[(J + 2) % 4], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(kPin);
            if constexpr (RES) po = qnn_fold_merge(po, (y2 & 0x00F000F0u) << 6);
            R[J % 4] = (po & 0xF000F000u) | ((pe >> 4) & ~0xF000F000u);       // bytes 1, 3 = (c1:c0), (c3:c2)
            if constexpr (J % 4 == 3) store_group(full);
            if constexpr (J == 3) { xput(0); xload(); }          // slot 0's rows were last read one row ago
            if constexpr (J == 5) { sput(0); sload(); }
            if constexpr (J == 9) { xput(1); xload(); }
            if constexpr (J == 11) { sput(1); sload(); }
            accp = acc;
        };
#define B16(J, F) body(std::integral_constant<int, J>{}, F)
        int yy = y0;
        for (; yy + 12 <= y1; yy += 12) {
            B16(0, true); B16(1, true); B16(2, true); B16(3, true); B16(4, true); B16(5, true);
            B16(6, true); B16(7, true); B16(8, true); B16(9, true); B16(10, true); B16(11, true);
        }
        const int rem = y1 - yy;
        if (rem > 0) B16(0, false);
        if (rem > 1) B16(1, false);
        if (rem > 2) B16(2, false);
        if (rem > 3) B16(3, false);
        if (rem > 4) B16(4, false);
        if (rem > 5) B16(5, false);
        if (rem > 6) B16(6, false);
        if (rem > 7) B16(7, false);
        if (rem > 8) B16(8, false);
        if (rem > 9) B16(9, false);
        if (rem > 10) B16(10, false);
#undef B16
        if (rem & 3) store_group(false);                         // the last, partial group (rows past y1 are masked)
    }
}

}  // namespace

// 0 = launched; 1 = not eligible (the caller falls back to k_conv_strip)
int qnn_launch_strip16_lds(const MfmaGeom& mg, const EpiArgs& e, const void* x, const uint8_t* w, void* y, hipStream_t s) {
    const ConvGeom& g = mg.g;
    const int res = !e.res ? 0 : e.res_store == QNN_STORE_I4 ? 1 : 2;
    if (!e.fold_a || res == 2 || g.cin != 16 || g.cout != 16 || g.stride != 1 || (g.W & 1) || e.ocw != 2) return 1;
    const int spr = (g.W + 15) / 16;
    const double img = (double)g.H * g.W * 8.0;
    if (img >= 1.0e9) return 1;
    static const int wps = QNN_ENV_INT("QNN_S16_WPS", QNN_S16_WPS);       // (A/B switch, experiment builds only)
    const int blocks_cap = 256 * wps;
    const long nwaves = (long)blocks_cap * 4;
    // rows per task: whole rounds of the persistent grid; a round costs rc rows + ~3 rows of pipeline fill.  Multiples of
    // four keep every store group full except the image's last.
    int best_rc = g.H, best_nch = 1;
    double best_cost = 1e300;
    for (int rc = 4; rc <= g.H + 3; rc += 4) {
        const int nch = (g.H + rc - 1) / rc;
        const long tasks = (long)g.N * spr * nch;
        const long rounds = (tasks + nwaves - 1) / nwaves;
        const double cost = (double)rounds * (rc + 3);
        if (cost < best_cost) { best_cost = cost; best_rc = rc; best_nch = nch; }
    }
    const long ntasks_l = (long)g.N * spr * best_nch;
    if (ntasks_l >= 2000000000L) return 1;
    long blocks = (ntasks_l + 3) / 4;
    if (blocks > blocks_cap) blocks = blocks_cap;
    const dim3 grid((unsigned)blocks), block(256);
#define S16_CASE(RES_, FOLD_)                                                                                            \
    if ((res != 0) == RES_ && fold == FOLD_)                                                                             \
        hipLaunchKernelGGL((k_conv_strip16_lds<RES_, FOLD_>), grid, block, 0, s, mg, e, (const uint8_t*)x, w, y, (int)ntasks_l, \
                           spr, qnn_fastdiv((uint32_t)spr), best_nch, qnn_fastdiv((uint32_t)best_nch), best_rc, (uint32_t)img,  \
                           (uint32_t)img);
    const int fold = e.fold_c ? 2 : 1;
    S16_CASE(false, 1) S16_CASE(false, 2) S16_CASE(true, 1) S16_CASE(true, 2)
#undef S16_CASE
    return 0;
}
