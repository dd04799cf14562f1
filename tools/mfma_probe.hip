// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip   (run on the GPU box; the executable is not committed)
// Probe the A/B operand lane maps of v_mfma_i32_32x32x32_i8 and _16x16x64_i8 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k32(const int8_t* A, const int8_t* B, int* D) {   // A[32][32] (m,k), B[32][32] (k,n) , D[32][32]
    int l = threadIdx.x, i = l & 31, h = l >> 5;
    v4i a, b;
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; ++j) { ab[j] = A[i * 32 + 16 * h + j]; bb[j] = B[(16 * h + j) * 32 + i]; }
    a = *(v4i*)ab; b = *(v4i*)bb;
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * h; D[row * 32 + i] = c[r]; }
}
__global__ void k16(const int8_t* A, const int8_t* B, int* D) {   // A[16][64], B[64][16], D[16][16]
    int l = threadIdx.x, i = l & 15, q = l >> 4;
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; ++j) { ab[j] = A[i * 64 + 16 * q + j]; bb[j] = B[(16 * q + j) * 16 + i]; }
    v4i a = *(v4i*)ab, b = *(v4i*)bb;
    v4i c = {0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { int row = q * 4 + r; D[row * 16 + i] = c[r]; }
}
int main() {
    int8_t hA[32 * 64], hB[64 * 32]; int hD[32 * 32], ref[32 * 32];
    srand(1);
    for (auto& v : hA) v = rand() % 255 - 127;
    for (auto& v : hB) v = rand() % 255 - 127;
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int k = 0; k < 32; ++k) s += hA[m * 32 + k] * hB[k * 32 + n]; if (s != hD[m * 32 + n]) ++bad; }
    printf("32x32x32_i8 hypothesis k=16*(l>>5)+j : %d mismatches of 1024\n", bad);
    k16<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 16 * 16 * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { int s = 0; for (int k = 0; k < 64; ++k) s += hA[m * 64 + k] * hB[k * 16 + n]; if (s != hD[m * 16 + n]) ++bad; }
    printf("16x16x64_i8 hypothesis k=16*(l>>4)+j : %d mismatches of 256\n", bad);
    return 0;
}
