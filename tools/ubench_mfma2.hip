// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma2 tools/ubench_mfma2.hip   (run on the GPU box; the executable is not committed)
// Do a MFMA-only wave and a VALU-only wave on the same SIMD overlap (gfx950)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// MODE 0: all waves MFMA only; 1: all waves VALU only; 2: waves 0-3 MFMA, waves 4-7 VALU (512-thread WG: 2 waves/SIMD)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
    v16f accf = {0}; v16i acci = {0};
    float a = threadIdx.x * 0.001f + seed, b = 1.0001f;
    v4i ia = {(int)threadIdx.x, 2, 3, 4}, ib = {5, 6, 7, (int)threadIdx.x};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    if (do_mfma)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (KIND == 0) accf = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accf, 0, 0, 0);
                else acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, acci, 0, 0, 0);
            }
        }
    if (do_valu)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 128; ++u) v[u & 7] = fmaf(v[u & 7], b, 0.5f);
        }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += accf[i] + (float)acci[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int KIND> int run(const char* name, float* out) {
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, 16, mode, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, 4096, mode, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s mode=%d (%s): %.3f ms\n", name, mode, mode == 0 ? "2 MFMA waves/SIMD" : mode == 1 ? "2 VALU waves/SIMD" : "1 MFMA + 1 VALU wave/SIMD", ms);
    }
    return 0;
}
int main() { float* out; CHECK(hipMalloc(&out, 256 * 512 * 4)); run<0>("f32_32x32x2", out); run<1>("i8_32x32x32", out); return 0; }
