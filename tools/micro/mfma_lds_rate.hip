// What the inner loop of k_conv_mfma_halo / k_conv_mfma_areg can reach: v_mfma_i32_32x32x32_i8 with BOTH operands re-read from
// LDS (one ds_read_b128 per MFMA: 2 x 2 register blocking, fragments requested one K-step ahead), random bytes, no other work.
// Prints ns per MFMA per SIMD, the implied chip rate and the clock the chip held (s_memtime / s_memrealtime) at 1 / 2 / 3 waves
// per SIMD, and the same loop with the operands kept in registers.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds_rate.hip -o /tmp/mfma_lds_rate && /tmp/mfma_lds_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NW, bool LDS>
__global__ __launch_bounds__(NW * 64, 1) void k(const uint4* __restrict__ rnd, int iters, int* out, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BYTES = 36864 + NW * 9216;
    for (int i = threadIdx.x; i < BYTES / 16; i += NW * 64) reinterpret_cast<uint4*>(smem)[i] = rnd[(i + blockIdx.x * 7) & 16383];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int a0 = 36864 + wave * 9216 + (li & 15) * 32 + ((li >> 4) & 1) * 768 + (lh << 4);
    const int b0 = li * 64 + (((lh * 2) ^ ((li >> 2) & 3)) << 4);
    v16i acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int j = 0; j < 16; ++j) acc[a][b][j] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    v4i fa[2][2], fb[2][2];
    auto frags = [&](int st, v4i (&A)[2], v4i (&B)[2]) {
        if (LDS) {
            A[0] = *reinterpret_cast<const v4i*>(smem + a0 + (st % 9) * 32 + (st & 1) * 4608);
            A[1] = *reinterpret_cast<const v4i*>(smem + a0 + (st % 9) * 32 + (st & 1) * 4608 + 1536);
            B[0] = *reinterpret_cast<const v4i*>(smem + b0 + (st >> 1) * 4096);
            B[1] = *reinterpret_cast<const v4i*>(smem + b0 + (st >> 1) * 4096 + 2048);
        }
    };
    frags(0, fa[0], fb[0]);
    if (!LDS) { frags(1, fa[1], fb[1]); for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { fa[i][j] = *reinterpret_cast<const v4i*>(smem + a0 + i * 32 + j * 1536); fb[i][j] = *reinterpret_cast<const v4i*>(smem + b0 + i * 4096 + j * 2048); } }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (LDS) frags((st + 1) % 18, fa[(st + 1) & 1], fb[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[st & 1][a], fb[st & 1][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int j = 0; j < 16; ++j) s ^= acc[a][b][j];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (lane == 0) { stamps[(blockIdx.x * NW + wave) * 2] = t1 - t0; stamps[(blockIdx.x * NW + wave) * 2 + 1] = r1 - r0; }
}

template <int NW, bool LDS>
static void run(const uint4* rnd) {
    const int iters = 400, blocks = 256;
    int* out; unsigned long long* st;
    hipMalloc(&out, blocks * NW * 64 * 4); hipMalloc(&st, blocks * NW * 16);
    const size_t lds = 36864 + NW * 9216;
    hipFuncSetAttribute((const void*)k<NW, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {                      // ~tens of ms of back-to-back launches before the timed one
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<NW, LDS>), dim3(blocks), dim3(NW * 64), lds, 0, rnd, iters, out, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    ms /= 20;
    unsigned long long* h = (unsigned long long*)malloc(blocks * NW * 16);
    hipMemcpy(h, st, blocks * NW * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks * NW; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double ghz = cyc / real * 0.1;                     // s_memrealtime ticks at 100 MHz
    const double n_per_simd = (double)iters * 72 * (NW / 4);
    const double ns = ms * 1e6 / n_per_simd;
    printf("%s  %d waves/SIMD: %.2f ns per MFMA per SIMD (%.1f cycles at the %.2f GHz held) -> %.0f T MAC/s chip\n",
           LDS ? "operands from LDS " : "operands in regs  ", NW / 4, ns, ns * ghz, ghz, 32768.0 / ns * 1024 / 1e3);
    free(h); hipFree(out); hipFree(st);
}

int main() {
    uint4* rnd; const size_t n = 16384;
    uint4* h = (uint4*)malloc(n * 16);
    srand(7);
    for (size_t i = 0; i < n * 4; ++i) ((uint32_t*)h)[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    hipMalloc(&rnd, n * 16); hipMemcpy(rnd, h, n * 16, hipMemcpyHostToDevice);
    run<4, false>(rnd); run<8, false>(rnd); run<12, false>(rnd);
    run<4, true>(rnd); run<8, true>(rnd); run<12, true>(rnd);
    return 0;
}
