// Micro-benchmark: how do v_mfma_i32_16x16x64_i8 and the VALU instructions that consume its results share a SIMD of
// gfx950?  One "round" = 8 groups of (3 MFMAs with C = 0, then 8 v_lshl_add_u32 + 3 max on the results), the inner
// pattern of csrc/qnn_first_fixed.hip.  Modes: 0 MFMA only, 1 VALU only, 2 dependent in program order, 3 dependent,
// software-pipelined (group g+1's MFMAs issued before group g's combine).  1 / 2 / 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int combine(const v4i& a0, const v4i& a1, const v4i& a2) {
    int raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int hi = (int)(((uint32_t)a2[i] << 8) + (uint32_t)a1[i]);
        asm("" : "+v"(hi));
        raw[i] = (int)(((uint32_t)hi << 8) + (uint32_t)a0[i]);
    }
    return max(max(raw[0], raw[1]), max(raw[2], raw[3]));
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(int iters, const v4i* __restrict__ in, int* out) {
    v4i A[2][3], B[4];
    for (int i = 0; i < 6; ++i) A[i / 3][i % 3] = in[threadIdx.x * 16 + i];
    for (int i = 0; i < 4; ++i) B[i] = in[threadIdx.x * 16 + 6 + i];
    const v4i z = {0, 0, 0, 0};
    int T = 0;
    v4i f0 = in[threadIdx.x], f1 = in[threadIdx.x + 1], f2 = in[threadIdx.x + 2];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            v4i s = z;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][0], B[g & 3], z, 0, 0, 0);
                v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][1], B[g & 3], z, 0, 0, 0);
                v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][2], B[g & 3], z, 0, 0, 0);
                asm volatile("" ::"v"(a0), "v"(a1), "v"(a2));
            }
            A[0][0][0] += 1;
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                T += combine(f0, f1, f2);
                f0[g & 3] += T; f1[g & 3] ^= T; f2[(g + 1) & 3] += T;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 4) {                  // MFMAs plus the same VALU work on registers the MFMAs do not write
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][0], B[g & 3], z, 0, 0, 0);
                v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][1], B[g & 3], z, 0, 0, 0);
                v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][2], B[g & 3], z, 0, 0, 0);
                asm volatile("" ::"v"(a0), "v"(a1), "v"(a2));
                T += combine(f0, f1, f2);
                f0[g & 3] += T; f1[g & 3] ^= T; f2[(g + 1) & 3] += T;
                __builtin_amdgcn_sched_barrier(0);
            }
            A[0][0][0] += 1;
        } else if (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][0], B[g & 3], z, 0, 0, 0);
                v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][1], B[g & 3], z, 0, 0, 0);
                v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[g >> 2][2], B[g & 3], z, 0, 0, 0);
                T += combine(a0, a1, a2);
                __builtin_amdgcn_sched_barrier(0);
            }
            A[0][0][0] += 1;
        } else {
            v4i a[2][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) a[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0][j], B[0], z, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) a[(g + 1) & 1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[(g + 1) >> 2][j], B[(g + 1) & 3], z, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                T += combine(a[g & 1][0], a[g & 1][1], a[g & 1][2]);
                __builtin_amdgcn_sched_barrier(0);
            }
            A[0][0][0] += 1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = T + f1[0] + f1[3] + f2[1] + f2[2] + f0[0] + f0[1] + f0[2] + f0[3] + A[0][0][0];
}

template <int MODE>
static void run(const v4i* in, int* out, int wps) {
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, 100, in, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, iters, in, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static const char* names[] = {"MFMA only (24)", "VALU only (88)", "dependent, in order", "dependent, pipelined", "independent VALU"};
    printf("%-22s waves/SIMD=%d: %.1f ns per wave-round\n", names[MODE], wps, ms * 1e6 / iters / wps);
}

int main() {
    const size_t n = 256 * 16 + 8;
    v4i* h = (v4i*)malloc(n * sizeof(v4i));
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (v4i){rand() ^ (rand() << 16), rand() ^ (rand() << 16), rand() ^ (rand() << 16), rand() ^ (rand() << 16)};
    v4i* in; int* out;
    if (hipMalloc(&in, n * sizeof(v4i)) != hipSuccess || hipMalloc(&out, 256 * 4 * 256 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(in, h, n * sizeof(v4i), hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) { run<0>(in, out, w); run<1>(in, out, w); run<2>(in, out, w); run<3>(in, out, w); run<4>(in, out, w); }
    return 0;
}
