// Tuning aid: ONE wave per SIMD (256 threads per CU) issuing [1 MFMA 16x16x32 bf16 + N VALU] groups: how many VALU instructions of
// which kind hide behind a 16-cycle MFMA inside one wave's own instruction stream?  (MI355X; prints shader cycles per group.)
// Build + run: hipcc --offload-arch=gfx950 -O3 mfma_valu_mix.hip -o /tmp/mvm && /tmp/mvm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GROUPS 32
template <int KIND, int N>          // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_med3_f32, 3: dependent v_fma_f32 chain per slot pair,
                                    // 4: v_fma_f32 beside v_mfma_f32_32x32x16_bf16 (32 cycles) instead of 16x16x32, 5: v_pk_fma_f32 beside it
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    float s[8];
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) { s[i] = threadIdx.x * 0.5f + i; a[i] = f32x2{(float)threadIdx.x + i, 1.0f}; }
    f32x4 acc[4] = {};
    f32x16 big[2] = {};
    bf16x8 fa = {}, fb = {};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (KIND >= 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[g & 1]) : "v"(fa), "v"(fb));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int c = (g * N + i) & 7;
                if (KIND == 0 || KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[c]) : "v"(s[(c + 3) & 7]));
                else if (KIND == 1 || KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(a[(c + 3) & 7]));
                else if (KIND == 2) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(s[c]) : "v"(s[(c + 3) & 7]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(s[c & 1]));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s[i] + a[i].x + a[i].y;
    for (int i = 0; i < 4; ++i) r += acc[i].x;
    r += big[0][0] + big[1][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND, int N> void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 200;
    k<KIND, N><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256 * 4; ++i) s += h[i];
    printf("1 mfma%s + %d %-22s : %6.2f cycles per group\n", KIND >= 4 ? " 32x32x16" : "", N, name, s / (256 * 4) / (iters * GROUPS));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>("(nothing)");
    run<0, 1>("v_fma_f32"); run<0, 2>("v_fma_f32"); run<0, 3>("v_fma_f32"); run<0, 4>("v_fma_f32"); run<0, 5>("v_fma_f32"); run<0, 6>("v_fma_f32"); run<0, 8>("v_fma_f32");
    run<1, 1>("v_pk_fma_f32"); run<1, 2>("v_pk_fma_f32"); run<1, 3>("v_pk_fma_f32"); run<1, 4>("v_pk_fma_f32");
    run<2, 2>("v_med3_f32"); run<2, 4>("v_med3_f32");
    run<3, 2>("dependent v_fma_f32"); run<3, 4>("dependent v_fma_f32");
    run<4, 0>("(nothing)"); run<4, 2>("v_fma_f32"); run<4, 4>("v_fma_f32"); run<4, 5>("v_fma_f32"); run<4, 6>("v_fma_f32"); run<4, 8>("v_fma_f32"); run<4, 10>("v_fma_f32");
    run<5, 1>("v_pk_fma_f32"); run<5, 2>("v_pk_fma_f32"); run<5, 4>("v_pk_fma_f32");
    return 0;
}
