// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_shape tools/ubench_shape.hip   (run on the GPU box; the executable is not committed)
// i8 MFMA shape vs sustained rate on random operands (DVFS): 32x32x32 vs 16x16x64, gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const v4i* __restrict__ in, int* out, int iters) {
    // 64x64 per wave: SHAPE 0: 2x2 tiles of 32x32x32 (x2 k-blocks); SHAPE 1: 4x4 tiles of 16x16x64
    v4i fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = in[threadIdx.x * 8 + i]; fb[i] = in[threadIdx.x * 8 + 4 + i]; }
    v16i acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    v4i c16[16];
    for (int a = 0; a < 16; ++a) c16[a] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a * 2 + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kk * 2 + a], fb[kk * 2 + b], acc[a * 2 + b], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    c16[a * 4 + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[b], c16[a * 4 + b], 0, 0, 0);
        }
    }
    int r = 0;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) r += acc[a][i];
    for (int a = 0; a < 16; ++a) for (int i = 0; i < 4; ++i) r += c16[a][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int SHAPE> int run(const v4i* in, int* out, int wps, const char* tag) {
    const int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<SHAPE>), dim3(256 * wps), dim3(256), 0, 0, in, out, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<SHAPE>), dim3(256 * wps), dim3(256), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double macs = (double)iters * 64 * 64 * 64 * 4.0 * 256 * wps;   // per wave-iteration 64x64x64
    printf("%-10s %s waves/SIMD=%d: %.3f ms  %.0f T MAC/s (%.1f %% of 2516)\n", SHAPE ? "16x16x64" : "32x32x32", tag, wps, ms,
           macs / (ms * 1e-3) / 1e12, macs / (ms * 1e-3) / 1e12 / 25.16);
    return 0;
}
int main() {
    const size_t n = 256 * 8;
    v4i* h = (v4i*)malloc(n * sizeof(v4i));
    v4i *drand, *dzero; int* out;
    CHECK(hipMalloc(&drand, n * sizeof(v4i))); CHECK(hipMalloc(&dzero, n * sizeof(v4i))); CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (v4i){rand() ^ (rand() << 16), rand() ^ (rand() << 16), rand() ^ (rand() << 16), rand() ^ (rand() << 16)};
    CHECK(hipMemcpy(drand, h, n * sizeof(v4i), hipMemcpyHostToDevice));
    CHECK(hipMemset(dzero, 0, n * sizeof(v4i)));
    for (int wps : {1, 2, 4}) {
        run<0>(dzero, out, wps, "zeros "); run<1>(dzero, out, wps, "zeros ");
        run<0>(drand, out, wps, "random"); run<1>(drand, out, wps, "random");
    }
    return 0;
}
