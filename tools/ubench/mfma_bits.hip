// Tuning aid: is C + A.B over k = 0..31 by ONE v_mfma_f32_16x16x32_bf16 bit-identical to TWO chained
// v_mfma_f32_32x32x16_bf16 (k 0..15 then 16..31)?  Decides whether a tile may switch instruction shape without
// changing result bits.  Build: hipcc --offload-arch=gfx950 -O2 mfma_bits.hip -o mfma_bits.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// A: 32 x 32 (row, k) bf16 bits, B: 32 x 32 (k, col), C0: 32 x 32 fp32; out16 / out32: 32 x 32 fp32 (only [0,16)^2 compared for out16)
__global__ void k(const uint16_t* A, const uint16_t* B, const float* C0, float* out16, float* out32, float* out16x2) {
    const int l = threadIdx.x;
    {   // 16x16x32: rows 0..15, cols 0..15, k 0..31
        bf16x8 a, b;
        uint16_t ta[8], tb[8];
        for (int e = 0; e < 8; ++e) { ta[e] = A[(l % 16) * 32 + (l / 16) * 8 + e]; tb[e] = B[((l / 16) * 8 + e) * 32 + (l % 16)]; }
        memcpy(&a, ta, 16); memcpy(&b, tb, 16);
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = C0[((l / 16) * 4 + r) * 32 + (l % 16)];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out16[((l / 16) * 4 + r) * 32 + (l % 16)] = c[r];
    }
    {   // 32x32x16 twice
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = C0[((r / 4) * 8 + (l / 32) * 4 + (r % 4)) * 32 + (l % 32)];
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a, b;
            uint16_t ta[8], tb[8];
            for (int e = 0; e < 8; ++e) { ta[e] = A[(l % 32) * 32 + ks * 16 + (l / 32) * 8 + e]; tb[e] = B[(ks * 16 + (l / 32) * 8 + e) * 32 + (l % 32)]; }
            memcpy(&a, ta, 16); memcpy(&b, tb, 16);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) out32[((r / 4) * 8 + (l / 32) * 4 + (r % 4)) * 32 + (l % 32)] = c[r];
    }
    {   // 16x16x32 fed with a different k -> lane-group assignment (k = 16*(g&1) + 8*(g>>1) + e): does the order inside matter?
        bf16x8 a, b;
        uint16_t ta[8], tb[8];
        const int g = l / 16, kb = 16 * (g & 1) + 8 * (g >> 1);
        for (int e = 0; e < 8; ++e) { ta[e] = A[(l % 16) * 32 + kb + e]; tb[e] = B[(kb + e) * 32 + (l % 16)]; }
        memcpy(&a, ta, 16); memcpy(&b, tb, 16);
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = C0[((l / 16) * 4 + r) * 32 + (l % 16)];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out16x2[((l / 16) * 4 + r) * 32 + (l % 16)] = c[r];
    }
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    uint16_t hA[1024], hB[1024]; float hC[1024], o16[1024], o32[1024], o16b[1024];
    uint16_t *dA, *dB; float *dC, *d16, *d32, *d16b;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096); hipMalloc(&d16, 4096); hipMalloc(&d32, 4096); hipMalloc(&d16b, 4096);
    long diff = 0, diffb = 0, total = 0; double maxrel = 0, err16 = 0, err32 = 0;
    srand(1);
    for (int trial = 0; trial < 2000; ++trial) {
        const float scale = (trial % 4 == 0) ? 1.f : (trial % 4 == 1) ? 100.f : (trial % 4 == 2) ? 1e-3f : 7.3f;
        for (int i = 0; i < 1024; ++i) {
            hA[i] = f2bf(((rand() / (float)RAND_MAX) * 2 - 1) * scale);
            hB[i] = f2bf(((rand() / (float)RAND_MAX) * 2 - 1));
            hC[i] = ((rand() / (float)RAND_MAX) * 2 - 1) * scale * (trial % 3);
        }
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, d16, d32, d16b);
        hipMemcpy(o16, d16, 4096, hipMemcpyDeviceToHost); hipMemcpy(o32, d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(o16b, d16b, 4096, hipMemcpyDeviceToHost);
        for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            double ref = hC[r * 32 + c];
            for (int kk = 0; kk < 32; ++kk) ref += (double)bf2f(hA[r * 32 + kk]) * (double)bf2f(hB[kk * 32 + c]);
            const float x = o16[r * 32 + c], y = o32[r * 32 + c], z = o16b[r * 32 + c];
            ++total;
            if (memcmp(&x, &y, 4)) { ++diff; const double rel = fabs((double)x - y) / (fabs(ref) + 1e-30); if (rel > maxrel) maxrel = rel; }
            if (memcmp(&x, &z, 4)) ++diffb;
            err16 += fabs(x - ref); err32 += fabs(y - ref);
        }
    }
    printf("elements %ld: 16x16x32 vs 2 x 32x32x16 differ in %ld (max rel %.3g); 16x16x32 with permuted k groups differs in %ld\n", total, diff, maxrel, diffb);
    printf("mean |err| vs fp64: 16x16x32 %.4g   2 x 32x32x16 %.4g\n", err16 / total, err32 / total);
    return 0;
}
