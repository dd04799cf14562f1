// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f32 tools/ubench_f32.hip   (run on the GPU box; the executable is not committed)
// f32 MFMA 32x32x2: sustained cycles per MFMA vs number of accumulator chains and waves/SIMD (random operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int NC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* out, int iters) {
    float a[14], b[2][14];
    for (int i = 0; i < 14; ++i) { a[i] = in[threadIdx.x * 64 + i]; b[0][i] = in[threadIdx.x * 64 + 14 + i]; b[1][i] = in[threadIdx.x * 64 + 28 + i]; }
    v16f acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 14; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = (NC == 2) ? u : (u + 2 * (s & 1));      // 2 chains (as the kernel) or 4
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[u][s], acc[c], 0, 0, 0);
            }
    }
    float r = 0;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NC> int run(const float* in, float* out, int wps) {
    const int iters = 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NC>), dim3(256 * wps), dim3(256), 0, 0, in, out, 16);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NC>), dim3(256 * wps), dim3(256), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double n_mfma = (double)iters * 28 * wps;    // per SIMD
    printf("chains=%d waves/SIMD=%d: %.3f ms -> %.1f nominal-2.4GHz cycles per MFMA per SIMD; %.1f TFLOP/s\n", NC, wps, ms,
           ms * 1e-3 * 2.4e9 / n_mfma, n_mfma * 1024 * 4096.0 / (ms * 1e-3) / 1e12);
    return 0;
}
int main() {
    const size_t n = 256 * 64;
    float* h = (float*)malloc(n * 4); srand(2);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *din, *out; CHECK(hipMalloc(&din, n * 4)); CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice));
    for (int wps : {1, 2, 3, 4}) { run<2>(din, out, wps); run<4>(din, out, wps); }
    return 0;
}
