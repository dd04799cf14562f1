// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma tools/ubench_mfma.hip   (run on the GPU box; the executable is not committed)
// How much VALU work co-executes with MFMA of different types on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// KIND 0: f32 32x32x2, 1: i8 32x32x32 ; NV = independent VALU fma per MFMA, issued by the SAME wave
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    v16f accf = {0}; v16i acci = {0};
    float a = threadIdx.x * 0.001f + seed, b = 1.0001f;
    v4i ia = {(int)threadIdx.x, 2, 3, 4}, ib = {5, 6, 7, (int)threadIdx.x};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (KIND == 0) accf = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accf, 0, 0, 0);
            else acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, acci, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += accf[i] + (float)acci[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND, int NV>
int run(const char* name, float* out, int wps) {
    const int iters = 4096;
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, out, 16, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)iters * 8 * wps;     // wave-MFMAs per SIMD
    printf("%-10s NV=%2d waves/SIMD=%d : %.3f ms  -> %.1f cycles@2.4GHz per MFMA slot (per SIMD)\n", name, NV, wps, ms,
           ms * 1e-3 * 2.4e9 / mfma_per_simd);
    return 0;
}
int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int wps : {1, 2, 4}) {
        run<0, 0>("f32_32x32x2", out, wps); run<0, 4>("f32_32x32x2", out, wps); run<0, 8>("f32_32x32x2", out, wps);
        run<0, 16>("f32_32x32x2", out, wps); run<0, 32>("f32_32x32x2", out, wps);
    }
    for (int wps : {1, 2, 4}) {
        run<1, 0>("i8_32x32x32", out, wps); run<1, 4>("i8_32x32x32", out, wps); run<1, 8>("i8_32x32x32", out, wps);
        run<1, 16>("i8_32x32x32", out, wps);
    }
    return 0;
}
