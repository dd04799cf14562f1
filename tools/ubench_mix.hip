// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mix tools/ubench_mix.hip   (run on the GPU box; the executable is not committed)
// Does int-VALU work of one wave hide under the i8 MFMAs of the other waves on a SIMD when
// every wave alternates bursts of both (the shape of the conv main loops)?  gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NM, int NV, int DEP>   // DEP=1: the MFMA A operands are produced by the VALU burst
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
    v16i acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    v4i ia = {(int)threadIdx.x, seed, 3, 4}, ib = {5, 6, seed, (int)threadIdx.x};
    uint32_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * 77 + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[j & 7]) : "v"(0xF0F0F0F1u + (uint32_t)seed));
        if (DEP) { ia[0] = (int)v[0]; ia[1] = (int)v[1]; ia[2] = (int)v[2]; ia[3] = (int)v[3]; }
#pragma unroll
        for (int u = 0; u < NM; ++u)
            acc[u & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, acc[u & 3], 0, 0, 0);
    }
    int r = 0;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) r += acc[a][i];
    for (int i = 0; i < 8; ++i) r += (int)v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NM, int NV, int DEP>
int run(int* out, int wps) {
    const int iters = 2048;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NM, NV, DEP>), dim3(256 * wps), dim3(256), 0, 0, out, 8, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NM, NV, DEP>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 1);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);   // cycles per (wave-iteration) per SIMD
    printf("NM=%2d NV=%3d dep=%d waves/SIMD=%d : %.3f ms -> %.0f cyc per wave-iteration per SIMD (MFMA alone %d, VALU alone ~%d)\n",
           NM, NV, DEP, wps, ms, cyc, NM * 32, NV * 4);
    return 0;
}
int main() {
    int* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int wps : {1, 2, 3, 4}) {
        run<8, 0, 0>(out, wps); run<0, 40, 0>(out, wps); run<8, 40, 0>(out, wps); run<8, 40, 1>(out, wps);
        run<8, 80, 0>(out, wps);
    }
    return 0;
}
