// Instruction-throughput microbenchmark for the ops the low-bit kernels are built on.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int NACC = 8;      // independent accumulators per lane
constexpr int NW = 16;       // scalar words cycled through

template <int OP>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ sw, uint32_t* out, int iters) {
    uint32_t a[NACC];
    int acc[NACC];
    float facc[NACC];
    for (int i = 0; i < NACC; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u; acc[i] = 0; facc[i] = 0.f; }
    uint32_t w[NW];
    for (int j = 0; j < NW; ++j) w[j] = sw[j];   // uniform -> SGPRs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (OP == 0) acc[i] = __builtin_amdgcn_sdot8((int)a[i], (int)w[j], acc[i], false);
                else if constexpr (OP == 1) acc[i] += __popc(a[i] ^ w[j]);
                else if constexpr (OP == 2) facc[i] = fmaf(__uint_as_float(a[i]), __uint_as_float(w[j]), facc[i]);
                else if constexpr (OP == 3) acc[i] = __builtin_amdgcn_sdot4((int)a[i], (int)w[j], acc[i], false);
                else if constexpr (OP == 4) acc[i] += __popc(a[i] & w[j]);          // and+bcnt
                else if constexpr (OP == 5) acc[i] = __builtin_amdgcn_udot8(a[i], w[j], (uint32_t)acc[i], false);
            }
        }
        // perturb so the compiler cannot hoist
        for (int j = 0; j < NW; ++j) w[j] = __builtin_amdgcn_readfirstlane(w[j] + it);
    }
    uint32_t r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i] + __float_as_uint(facc[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int ops_per_elem, double macs_per_op, const uint32_t* sw, uint32_t* out) {
    for (int wps : {1, 2, 4, 8}) {     // waves per SIMD
        const int blocks = 256 * wps;  // 256 CUs x (wps blocks of 4 waves)
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, sw, out, 64);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, sw, out, ITERS);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double winstr = (double)ITERS * NW * NACC * ops_per_elem;      // per wave
        const double total_w = (double)blocks * 4;                            // waves
        const double per_simd = winstr * total_w / 1024.0;                    // wave-instr per SIMD
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-12s waves/SIMD=%d  %.3f ms  %.2f cyc/wave-instr/SIMD (@2.4GHz)  %.1f T elem-op/s\n", name, wps, ms,
               cyc / per_simd, winstr / ops_per_elem * total_w * 64 * macs_per_op / (ms * 1e-3) / 1e12);
    }
    return 0;
}

int main() {
    uint32_t* sw; uint32_t* out;
    CHECK(hipMalloc(&sw, 4096)); CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    std::vector<uint32_t> h(1024);
    for (auto& v : h) v = rand();
    CHECK(hipMemcpy(sw, h.data(), 4096, hipMemcpyHostToDevice));
    run<0>("sdot8_i4", 1, 8, sw, out);
    run<5>("udot8_u4", 1, 8, sw, out);
    run<3>("sdot4_i8", 1, 4, sw, out);
    run<1>("xor+bcnt", 2, 32, sw, out);
    run<4>("and+bcnt", 2, 32, sw, out);
    run<2>("fma_f32", 1, 1, sw, out);
    return 0;
}
