// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_bf16_probe tools/mfma_bf16_probe.hip   (run on the GPU box; the executable is not committed)
// What does v_mfma_f32_32x32x16_bf16 compute, bit for bit?  Compare with
//  H1: round_to_f32( C + sum_k a_k*b_k )   (exact sum, one rounding)
//  H2: sequential fp32 FMA chain k = 0..15 starting from C
//  H3: exact sum of the 16 products rounded to f32, then + C rounded
//  H4: two exact half sums (k 0..7, 8..15): C' = rnd(C + s0); D = rnd(C' + s1)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const uint16_t* A, const uint16_t* B, const float* C, float* D) {  // A[32][16], B[16][32], C/D[32][32]
    int l = threadIdx.x, i = l & 31, h = l >> 5;
    v8s a, b; v16f c;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[i * 16 + 8 * h + j]; b[j] = (short)B[(8 * h + j) * 32 + i]; }
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}
static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
// exact: value = num * 2^-SH as __int128
typedef __int128 i128;
static const int SH = 60;
static i128 to_fixed(double v) { return (i128)ldexp(v, SH); }   // exact for our restricted exponents
static float round_fixed(i128 x) {   // RNE of x*2^-SH to float
    if (x == 0) return 0.f;
    int neg = x < 0; unsigned __int128 m = neg ? (unsigned __int128)(-x) : (unsigned __int128)x;
    int hb = 127; while (!((m >> hb) & 1)) --hb;
    int drop = hb - 23;
    uint64_t mant; 
    if (drop <= 0) mant = (uint64_t)(m << (-drop));
    else {
        unsigned __int128 q = m >> drop, rem = m & ((((unsigned __int128)1) << drop) - 1), half = ((unsigned __int128)1) << (drop - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;
        mant = (uint64_t)q;
    }
    double v = ldexp((double)mant, (drop > 0 ? drop : drop) - SH);
    if (drop <= 0) v = ldexp((double)mant, drop - SH);
    return (float)(neg ? -v : v);
}
int main() {
    const int T = 400;
    uint16_t hA[32 * 16], hB[16 * 32]; float hC[1024], hD[1024];
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    long n = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0, b9 = 0;
    srand(7);
    for (int t = 0; t < T; ++t) {
        for (auto& v : hA) { int e = 127 - 3 + rand() % 7; v = (uint16_t)(((rand() & 1) << 15) | (e << 7) | (rand() & 127)); }
        for (auto& v : hB) { int e = 127 - 3 + rand() % 7; v = (uint16_t)(((rand() & 1) << 15) | (e << 7) | (rand() & 127)); }
        for (auto& v : hC) { uint32_t u = ((uint32_t)(rand() & 1) << 31) | ((uint32_t)(127 - 4 + rand() % 10) << 23) | ((uint32_t)rand() & 0x7FFFFF); memcpy(&v, &u, 4); if (t % 3 == 0) v = 0.f; }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double p[16]; i128 s = 0, s0 = 0, s1 = 0;
            float chain = hC[i * 32 + j];
            for (int kk = 0; kk < 16; ++kk) {
                p[kk] = (double)bf2f(hA[i * 16 + kk]) * (double)bf2f(hB[kk * 32 + j]);
                s += to_fixed(p[kk]); (kk < 8 ? s0 : s1) += to_fixed(p[kk]);
                chain = fmaf(bf2f(hA[i * 16 + kk]), bf2f(hB[kk * 32 + j]), chain);
            }
            i128 c = to_fixed((double)hC[i * 32 + j]);
            float h1 = round_fixed(c + s);
            float h3 = round_fixed(to_fixed((double)round_fixed(s)) + c);
            float h4 = round_fixed(to_fixed((double)round_fixed(c + s0)) + s1);
            // H5: four groups of 4 (sequential, exact within a group)
            float acc5 = hC[i * 32 + j], acc6 = hC[i * 32 + j];
            for (int g = 0; g < 4; ++g) { i128 sg = 0; for (int kk = 4 * g; kk < 4 * g + 4; ++kk) sg += to_fixed(p[kk]); acc5 = round_fixed(to_fixed((double)acc5) + sg); }
            // H6: eight groups of 2
            for (int g = 0; g < 8; ++g) { i128 sg = to_fixed(p[2 * g]) + to_fixed(p[2 * g + 1]); acc6 = round_fixed(to_fixed((double)acc6) + sg); }
            // H7: halves in the other order (k 8..15 first)
            float h7 = round_fixed(to_fixed((double)round_fixed(c + s1)) + s0);
            // H8: half sums rounded to f32 first, then added to C sequentially
            float h8 = round_fixed(to_fixed((double)round_fixed(c + to_fixed((double)round_fixed(s0)))) + to_fixed((double)round_fixed(s1)));
            // H9: C + (rnd(s0) + rnd(s1)) exact then round
            float h9 = round_fixed(c + to_fixed((double)round_fixed(s0)) + to_fixed((double)round_fixed(s1)));
            b5 += (hD[i * 32 + j] != acc5); b6 += (hD[i * 32 + j] != acc6); b7 += (hD[i * 32 + j] != h7); b8 += (hD[i * 32 + j] != h8); b9 += (hD[i * 32 + j] != h9);
            float d = hD[i * 32 + j];
            ++n; b1 += (d != h1); b2 += (d != chain); b3 += (d != h3); b4 += (d != h4);
            if (t == 0 && i == 0 && j < 3) printf("d=%.9g h1=%.9g chain=%.9g h3=%.9g h4=%.9g\n", d, h1, chain, h3, h4);
        }
    }
    printf("H5(4x4 seq)=%ld H6(8x2 seq)=%ld H7(halves swapped)=%ld H8(rounded halves seq)=%ld H9(C+rnd halves)=%ld\n", b5, b6, b7, b8, b9);
    printf("n=%ld mismatches: H1(exact,1 rounding)=%ld  H2(fma chain)=%ld  H3(sum rounded, +C)=%ld  H4(two halves)=%ld\n", n, b1, b2, b3, b4);
    return 0;
}
