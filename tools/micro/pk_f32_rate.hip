// Issue rate of the packed float32 VALU operations (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE float32 results
// per lane and instruction) against their scalar forms, one wave per SIMD and four per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pk_f32_rate.hip -o /tmp/pk_f32_rate && /tmp/pk_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    v2f x[8];
    for (int i = 0; i < 8; ++i) x[i] = v2f{(float)threadIdx.x + i, (float)threadIdx.x - i};
    const v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {            // packed: one mul + one add per pair
                x[i] = x[i] * va;
                x[i] = x[i] + vb;
            } else {                    // scalar: two muls + two adds per pair
                float p = x[i].x, q = x[i].y;
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(p) : "v"(a));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q) : "v"(a));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(p) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(q) : "v"(b));
                x[i] = v2f{p, q};
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 1024 * 8);
    const int iters = 4096;
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = 256 * wps;      // 4 waves per block = one per SIMD of a CU
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 1e-9f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 1e-9f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double pair_ops = (double)iters * 8 * 2;          // (mul, add) on 8 pairs per iteration per wave
            printf("%s  %d wave(s)/SIMD: %.3f ms, %.2f ns per (mul+add on a float pair) per wave\n",
                   mode == 0 ? "v_pk_*_f32" : "scalar    ", wps, ms, ms * 1e6 / pair_ops / wps * wps);
        }
    return 0;
}
