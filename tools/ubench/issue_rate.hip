// Tuning aid: instruction issue cost per wave on gfx950 with 1 or 2 waves per SIMD (256 / 512 threads, one
// workgroup per CU).  Build + run: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP 64
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ char smem[];
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, 1.0f};
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = threadIdx.x * 0.5f + i;
    f32x4 acc[4] = {};
    f32x4 acc16[16] = {};
    f32x16 big[4] = {};
    bf16x8 fa = {}, fb = {};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {   // packed fp32 fma, 8 independent chains
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            } else if (MODE == 1) {   // scalar fp32 fma
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(s[(i + 1) & 7]));
            } else if (MODE == 2) {   // v_rcp
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
            } else if (MODE == 3) {   // salu
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
            } else if (MODE == 4) {   // mfma 16x16x32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
            } else if (MODE == 5) {   // 1 mfma + 3 pk_fma interleaved
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[3 * i]) : "v"(a[7]));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[3 * i + 1]) : "v"(a[7]));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[3 * i + 2]) : "v"(a[7]));
                }
            } else if (MODE == 6) {   // role split: waves 0-3 mfma only, waves 4-7 pk_fma only
                if ((threadIdx.x >> 8) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                }
            } else if (MODE == 8) {   // mfma 16x16x32, 16 independent accumulators (the p8 phase)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(r & 1) * 8 + i]) : "v"(fa), "v"(fb));
            } else if (MODE == 9) {   // mfma 32x32x16, 4 independent accumulators
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(fa), "v"(fb));
            } else if (MODE == 10 || MODE == 11) {   // ping-pong mimic: waves 0-3 mfma, waves 4-7 ds_read_b128
                if ((threadIdx.x >> 8) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (MODE == 10) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(r & 1) * 8 + i]) : "v"(fa), "v"(fb));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(fa), "v"(fb));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(0)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); asm volatile("" :: "v"(v)); }
                }
            } else if (MODE == 12 || MODE == 13 || MODE == 14) {   // as 10, accumulators in AGPRs (12), reads into AGPRs (13), both (14)
                if ((threadIdx.x >> 8) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (MODE == 13) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(r & 1) * 8 + i]) : "v"(fa), "v"(fb));
                        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[(r & 1) * 8 + i]) : "v"(fa), "v"(fb));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f32x4 v;
                        if (MODE == 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(0));
                        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(0));
                        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                        if (MODE == 12) asm volatile("" :: "v"(v)); else asm volatile("" :: "a"(v));
                    }
                }
            } else if (MODE == 15) {   // as 10 with ds_read_b64 (half the bytes per instruction)
                if ((threadIdx.x >> 8) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(r & 1) * 8 + i]) : "v"(fa), "v"(fb));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { f32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 8u), "n"(0)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); asm volatile("" :: "v"(v)); }
                }
            } else if (MODE >= 16 && MODE <= 20) {   // MFMA wave | other wave: 16 setprio+ds_read, 17 reads without waits, 18 s_nop, 19 salu, 20 s_sleep
                if ((threadIdx.x >> 8) == 0) {
                    if (MODE == 16) asm volatile("s_setprio 3");
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(r & 1) * 8 + i]) : "v"(fa), "v"(fb));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (MODE == 16) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(0)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); asm volatile("" :: "v"(v)); }
                        if (MODE == 17) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(0)); asm volatile("" :: "v"(v)); }
                        if (MODE == 18) asm volatile("s_nop 7");
                        if (MODE == 19) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                        if (MODE == 20) asm volatile("s_sleep 2");
                    }
                    if (MODE == 17) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else if (MODE == 21 || MODE == 22 || MODE == 23) {   // 32x32x16 with 1 / 2 / 3 accumulators in rotation (dependent chains)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[MODE == 21 ? 0 : MODE == 22 ? (i & 1) : (i % 3)]) : "v"(fa), "v"(fb));
            } else if (MODE == 24 || MODE == 25) {   // 16x16x32 with 1 / 2 accumulators
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[MODE == 24 ? 0 : (i & 1)]) : "v"(fa), "v"(fb));
            } else if (MODE >= 30 && MODE <= 36) {   // ds_read_b128 address patterns of the p8 tile (128-byte rows, chunk ^ (row & 7))
                const unsigned l = threadIdx.x & 63;
                unsigned addr;
                if (MODE == 30) { const unsigned row = l & 15; addr = row * 128 + (((l >> 4) ^ (row & 7)) << 4); }          // 16x16x32 operand
                else if (MODE == 31) { const unsigned row = l & 31; addr = row * 128 + (((l >> 5) ^ (row & 7)) << 4); }     // 32x32x16 operand
                else if (MODE == 32) { const unsigned row = l & 31; addr = row * 128 + ((((l >> 5) ^ (row & 7)) ^ ((row >> 3) & 1) * 2) << 4); }   // + row bit 3 into chunk bit 1
                else if (MODE == 33) { const unsigned row = l & 31; addr = row * 128 + ((((l >> 5) ^ (row & 7)) ^ ((row >> 3) & 3) * 2) << 4); }                 // + row bits 3,4 into chunk bits 1,2
                else if (MODE == 34) { const unsigned row = l & 31; addr = row * 128 + ((((l >> 5) ^ (row & 7)) ^ ((row >> 4) & 1)) << 4); }     // + row bit 4 into chunk bit 0
                else if (MODE == 35) { const unsigned row = l & 31; addr = row * 128 + ((((l >> 5) ^ (row & 7)) ^ ((row >> 3) & 1)) << 4); }     // + row bit 3 into chunk bit 0
                else { const unsigned row = l & 31; addr = row * 128 + ((((l >> 5) ^ (row & 7)) ^ ((row >> 3) & 1) ^ (((row >> 4) & 1) << 1)) << 4); }   // bit3->c0, bit4->c1
                addr += (threadIdx.x >> 6) * 4096;
#pragma unroll
                for (int i = 0; i < 8; ++i) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0)); asm volatile("" :: "v"(v)); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE >= 40 && MODE <= 45) {   // ds_read_b128 on 64-byte rows (16x16x32 operand: row = lane & 15, chunk = lane >> 4), swizzles
                const unsigned l = threadIdx.x & 63, row = l & 15, ch = l >> 4;
                unsigned sw;
                if (MODE == 40) sw = 0;                                   // no swizzle
                else if (MODE == 41) sw = (row & 8) >> 2;                  // token_mlp today: row bit 3 -> chunk bit 1
                else if (MODE == 42) sw = (row >> 2) & 3;                  // row bits 2,3 -> chunk bits 0,1
                else if (MODE == 43) sw = (row >> 1) & 3;                  // row bits 1,2
                else if (MODE == 44) sw = ((row >> 2) & 1) | ((row >> 3) << 1);   // same as 42
                else sw = ((row >> 2) & 1) ^ (((row >> 3) & 1) << 1) ^ ((row >> 2) & 2);
                unsigned addr = row * 64 + ((ch ^ sw) << 4) + (threadIdx.x >> 6) * 4096;
#pragma unroll
                for (int i = 0; i < 8; ++i) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0)); asm volatile("" :: "v"(v)); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 7) {   // ds_read_b128, conflict-free lane-linear
#pragma unroll
                for (int i = 0; i < 8; ++i) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(0)); acc[i & 3] += v; }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
    for (int i = 0; i < 4; ++i) r += acc[i].x + big[i][0];
    for (int i = 0; i < 16; ++i) r += acc16[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    const int iters = 200;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<MODE><<<256, threads, 100 * 1024>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0; int n0 = 0, n1 = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { if (w < 4) { s0 += h[b * 8 + w]; ++n0; } else { s1 += h[b * 8 + w]; ++n1; } }
    printf("%-34s %3d thr: waves0-3 %6.2f clk/instr", name, threads, s0 / n0 / (iters * REP));
    if (n1) printf("   waves4-7 %6.2f clk/instr", s1 / n1 / (iters * REP));
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int thr : {256, 512}) {
        run<0>("v_pk_fma_f32 (8 chains)", thr);
        run<1>("v_fma_f32 (8 chains)", thr);
        run<2>("v_rcp_f32", thr);
        run<3>("s_add_u32", thr);
        run<4>("v_mfma_16x16x32_bf16", thr);
        run<5>("1 mfma + 3 pk_fma", thr);
        run<7>("ds_read_b128 + 4 v_add", thr);
        run<30>("ds_read_b128 p8 16x16 operand", thr);
        run<31>("ds_read_b128 p8 32x32 operand", thr);
        run<32>("ds_read_b128 32x32 + row bit3", thr);
        run<33>("ds_read_b128 32x32 + row bits34", thr);
        run<34>("ds_read_b128 32x32 row bit4->c0", thr);
        run<35>("ds_read_b128 32x32 row bit3->c0", thr);
        run<36>("ds_read_b128 32x32 b3->c0 b4->c1", thr);
        run<40>("ds_read_b128 64B rows, no swizzle", thr);
        run<41>("ds_read_b128 64B rows, bit3->c1 (token)", thr);
        run<42>("ds_read_b128 64B rows, bits23->c01", thr);
        run<43>("ds_read_b128 64B rows, bits12->c01", thr);
        run<8>("v_mfma_16x16x32_bf16 (16 accs)", thr);
        run<9>("v_mfma_32x32x16_bf16 (4 accs)", thr);
        run<21>("v_mfma_32x32x16_bf16 (1 acc)", thr);
        run<22>("v_mfma_32x32x16_bf16 (2 accs)", thr);
        run<23>("v_mfma_32x32x16_bf16 (3 accs)", thr);
        run<24>("v_mfma_16x16x32_bf16 (1 acc)", thr);
        run<25>("v_mfma_16x16x32_bf16 (2 accs)", thr);
    }
    run<10>("split: w0-3 mfma16 | w4-7 ds_read", 512);
    run<11>("split: w0-3 mfma32 | w4-7 ds_read", 512);
    run<12>("split: mfma16 acc=AGPR | ds_read", 512);
    run<13>("split: mfma16 | ds_read -> AGPR", 512);
    run<14>("split: mfma16 acc=AGPR | ds_read->AGPR", 512);
    run<15>("split: mfma16 | ds_read_b64", 512);
    run<16>("split: mfma16 prio3 | ds_read", 512);
    run<17>("split: mfma16 | ds_read x8, 1 wait", 512);
    run<18>("split: mfma16 | s_nop 7", 512);
    run<19>("split: mfma16 | s_add_u32", 512);
    run<20>("split: mfma16 | s_sleep 2", 512);
    run<6>("split: w0-3 mfma | w4-7 pk_fma", 512);
    return 0;
}
