// Micro-benchmark: issue rate of the int8 / bf16 MFMA shapes on gfx950, MFMA only, 8 (16x16) or 4 (32x32) independent
// accumulators per wave, 1 / 2 / 4 waves per SIMD.  Prints ns per MFMA per SIMD and the implied chip rate.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256, 2) void k(int iters, int* out) {
    v4i a = {(int)threadIdx.x * 0x01010101, 0x11223344, 0x55667788, 0x7f017f01}, b = {0x04050607, 0x01020304, 0x7f7f7f7f, (int)blockIdx.x};
    v4i acc4[8]; v16i acc16[4]; v4f accf4[8]; v16f accf16[4];
    for (int i = 0; i < 8; ++i) { acc4[i] = v4i{0, 0, 0, 0}; accf4[i] = v4f{0, 0, 0, 0}; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) { acc16[i][j] = 0; accf16[i][j] = 0; }
    v8bf ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(float)(threadIdx.x + i); hb[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 0) acc4[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[m], 0, 0, 0);
            if (KIND == 1) acc16[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc16[m & 3], 0, 0, 0);
            if (KIND == 2) accf4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, accf4[m], 0, 0, 0);
            if (KIND == 3) accf16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, accf16[m & 3], 0, 0, 0);
            if (KIND == 4) acc4[m] = __builtin_amdgcn_mfma_i32_16x16x32_i8(((long)a.x << 32) | (unsigned)a.y, ((long)b.x << 32) | (unsigned)b.y, acc4[m], 0, 0, 0);   // the CDNA3 shape (K = 32)
        }
    }
    int s = 0;
    // every accumulator element is used: a partial use lets the compiler split the tuples and shuffle AGPRs in the loop
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s ^= acc4[i][j] ^ __float_as_int(accf4[i][j]);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s ^= acc16[i][j] ^ __float_as_int(accf16[i][j]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, double flops, int wps) {
    const int iters = 4000, blocks = 256 * wps;
    int* out;
    if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess) return;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_per_simd = (double)iters * 8 * wps;
    const double ns = ms * 1e6 / n_per_simd;
    printf("%-16s waves/SIMD=%d: %.2f ns per MFMA per SIMD -> %.0f T(FL)OP/s chip\n", name, wps, ns, flops / ns * 1024 / 1e3);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("i8 16x16x64", 2.0 * 16 * 16 * 64, w);
        run<1>("i8 32x32x32", 2.0 * 32 * 32 * 32, w);
        run<2>("bf16 16x16x32", 2.0 * 16 * 16 * 32, w);
        run<3>("bf16 32x32x16", 2.0 * 32 * 32 * 16, w);
        run<4>("i8 16x16x32", 2.0 * 16 * 16 * 32, w);
    }
    return 0;
}
