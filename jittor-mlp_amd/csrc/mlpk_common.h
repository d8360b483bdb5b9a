// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mlpk.h"

namespace mlpk {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = MLPK_F32; };
template <> struct dtype_of<f16_t> { static constexpr int value = MLPK_F16; };
template <> struct dtype_of<bf16_t> { static constexpr int value = MLPK_BF16; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// Exact-form GELU, 0.5 x (1 + erf(x / sqrt 2)), with erf by Abramowitz & Stegun 7.1.26
// (|abs err| <= 1.5e-7): 1 rcp + 1 exp + 6 fma instead of libm's branchy erff, so the
// GEMM epilogue stays a small fraction of the MFMA time.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __expf(-ax * ax);
    const float r = __builtin_fmaf(-p, e, 1.0f);
    return __builtin_copysignf(r, x);
}

__device__ __forceinline__ float gelu_f(float x) {
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

// GELU for 16-bit outputs, division-free:
//   gelu(x) = x * Phi(x),  Phi(x) ~= 0.5 + t * Q(t^2 - 1),  t = clamp(x * sqrt2 / 4.5, -sqrt2, sqrt2)
// Q = degree-10 polynomial in u = t^2 - 1 in [-1, 1] (weighted minimax fit of the error of Phi, tools/fit_gelu_poly.py; sum |c| =
// 1.27, so fp32 Horner is well conditioned).  16 VALU instructions per PAIR of elements (everything but the clamp pairs into
// v_pk_* ops) and no transcendental; the form it replaces -- erf by Abramowitz & Stegun 7.1.28, 1 - (1 + a1 z + .. + a6 z^6)^-16 --
// took 19 + 2 v_rcp_f32 per pair, and the kernels that evaluate GELU are bound by VALU instruction count (the fused
// token-mixing kernel: profiles/r02_token_mlp_ablation.txt; the fc1 epilogue of the channel MLP).
// |Phi error| <= 2.7e-6 everywhere (|x| > 4.5 clamps to Phi(4.5) = 1 - 3.4e-6), |gelu error| <= 3.7e-6 on |x| <= 4.5: 60x below
// half an ulp of f16 at that magnitude; checked in fp32 emulation by tests/test_host_cpu.py against the coefficients HERE.
// float outputs keep gelu_f.
// Two grades (tools/fit_gelu_poly.py A K), by storage type of the result:
//   f16 : A = 4.5, 11 coefficients -- |Phi error| <= 2.7e-6, |gelu error| <= 3.7e-6 on |x| <= 4.5 (60x below half an ulp of f16)
//   bf16: A = 4.0,  8 coefficients -- |Phi error| <= 5.3e-5, |gelu error| <= 9e-5 on |x| <= 4 (half an ulp of bf16 is 2^-9 relative:
//         the polynomial error stays below the rounding of every result above 0.04 in magnitude); three fewer fma per element in the
//         kernels whose epilogues are bound by VALU issue (round 3: the q4 GEMM's fillers, the token-mixing kernel).  At this grade the
//         SAME fit is evaluated in the variable the kernels have at hand, without the scaling multiply in front of the clamp:
//             Phi(x) ~= 0.5 + t * R(t * t),  t = clamp(x, -4, 4)          (tools/fit_gelu_poly.py 4.0 8 raw)
//         -- R's coefficients sum to 81 in the variable (t/4)^2, i.e. ~5e-6 of fp32 cancellation, nothing against 9e-5; the f16
//         grade (sum 500+) keeps the centred variable.  11 operations per element instead of 12.
#define MLPK_GELUP_SCALE 0.314269681f
#define MLPK_GELUP_COEFS {0.00260713836f, -0.00718860654f, 0.00979797821f, -0.0172248576f, 0.0355015062f, -0.0601866171f, 0.090279378f, -0.127707109f, 0.174028099f, -0.245624334f, 0.499268919f}
#define MLPK_GELUP_CLAMP_BF16 4.0f
#define MLPK_GELUP_COEFS_BF16 {-1.58078628e-09f, 1.21711111e-07f, -4.10086659e-06f, 8.06673925e-05f, -0.00104820437f, 0.00966487452f, -0.0661753789f, 0.39884752f}
// Round 4 -- the logistic bf16 grade (round 5: behind -DMLPK_GELU_BF16_SIG; the polynomial above behind -DMLPK_GELU_BF16_POLY; default: "h2b" below):
//         gelu(x) = x * Phi(x),   Phi(x) ~= 1 / (1 + 2^(x * (K0 + K1 |x| + K2 x^2)))          (tools/fit_gelu_sig.py)
// a logistic with a cubic exponent, odd in x.  SEVEN instructions per element -- fma, fma (|x| is a source modifier), mul, v_exp_f32,
// add, v_rcp_f32, mul -- against eleven: the epilogues that carry a GELU are bound by the number of instructions one wave can issue
// behind its MFMAs (the q4 GEMM's fillers, the fused token-mixing kernel: 6.6 VALU operations per MFMA where 5 are free), and a
// transcendental costs about two cycles more than a plain operation there (tools/ubench/q4_slots.py exp* / alt_* / target* rows).
// |gelu error| <= 1.4e-4 for EVERY finite x (|Phi error| <= 3.7e-4 near 0 where gelu itself is small, <= 7e-5 beyond |x| = 2); the
// exponent's leading coefficient has the sign of K0, so Phi -> 0 / 1 and gelu -> -0 / x in the tails without a clamp -- the clamped
// polynomial's error grew like 5e-5 |x| beyond its interval (gelu(-1000) = -0.05).  Checked in emulated fp32 by tests/test_host_cpu.py.
#define MLPK_GELUS_K0 -2.28684449f
#define MLPK_GELUS_K1 -0.0305621661f
#define MLPK_GELUS_K2 -0.0905431807f
__device__ __forceinline__ float gelu_sig_f(float x) {
    const float a = __builtin_fabsf(x);
    float q = __builtin_fmaf(a, MLPK_GELUS_K2, MLPK_GELUS_K1);
    q = __builtin_fmaf(a, q, MLPK_GELUS_K0);
    const float e = __builtin_amdgcn_exp2f(x * q);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Round 5 -- the bf16 grade every kernel evaluates now ("h2b"; -DMLPK_GELU_BF16_SIG keeps the logistic form above, -DMLPK_GELU_BF16_POLY the
// polynomial, for A/B builds):  gelu(x) = x * Phi(h),  h = f16(x) (nearest-even),  Phi in PACKED f16 -- two elements per instruction:
//         t = h * S;  u = t * t - 1;  Phi = clamp01(0.5 + t * Q(u)),  Q = 7 coefficients by Horner            (tools/fit_gelu_h2.py)
// -- and the product in fp32 on the unrounded x (v_fma_mix_f32 reads the f16 half as an fp32 source).  Per PAIR: 1 convert + 9 packed + 2
// mixed = 12 plain instructions, none transcendental, against 14 of which four are quarter-rate (v_exp_f32 / v_rcp_f32): in the epilogues that
// are bound by VALU time (gMLP's channel_proj1, the narrow channel MLPs) a transcendental costs four plain issue cycles, in the q4 GEMM's
// filler stream two (profiles/r05_issue_slots_packed_gelu.txt: the fc1 loop 39.45 -> 37.71 cycles per MFMA).  No operand clamp: the fit's
// leading coefficient is positive, so beyond |x| = 4 the polynomial runs off in the direction the result clamp wants (h = +-inf included:
// every intermediate is +-inf of the right sign, never inf - inf), and x itself stays fp32, so gelu(-1e6) = -1e6 * 0 = -0.
// |gelu error| <= 9e-4 |x| on |x| >= 0.25 (half an ulp of bf16 is 2e-3 |x|), <= 1.7e-4 below; rms 1.2e-4 .. 3.5e-4 for x ~ N(0, 0.5 .. 2): +1 .. 3 %
// on the rms error of the bf16-rounded result (tests/test_host_cpu.py evaluates it over every f16 input, tools/fit_gelu_h2.py prints the table).
#define MLPK_GELUH_SCALE 0.353515625f
#define MLPK_GELUH_COEFS {0.01392364501953125f, -0.046875f, 0.07305908203125f, -0.100830078125f, 0.155029296875f, -0.2386474609375f, 0.497802734375f}
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2_t gelu_h2_phi(const h2_t h) {
    constexpr float c[7] = MLPK_GELUH_COEFS;
    const h2_t t = h * h2_t{(_Float16)MLPK_GELUH_SCALE, (_Float16)MLPK_GELUH_SCALE};
    const h2_t u = __builtin_elementwise_fma(t, t, h2_t{(_Float16)-1.0f, (_Float16)-1.0f});
    h2_t q = __builtin_elementwise_fma(h2_t{(_Float16)c[0], (_Float16)c[0]}, u, h2_t{(_Float16)c[1], (_Float16)c[1]});
#pragma unroll
    for (int i = 2; i < 7; ++i) q = __builtin_elementwise_fma(q, u, h2_t{(_Float16)c[i], (_Float16)c[i]});
    const h2_t p = __builtin_elementwise_fma(t, q, h2_t{(_Float16)0.5f, (_Float16)0.5f});
    return __builtin_elementwise_min(__builtin_elementwise_max(p, h2_t{(_Float16)0.0f, (_Float16)0.0f}), h2_t{(_Float16)1.0f, (_Float16)1.0f});
}
__device__ __forceinline__ float gelu_h2b_f(float x) {         // the scalar form: the same f16 operations on one lane, hence the same bits
    const h2_t p = gelu_h2_phi(h2_t{(_Float16)x, (_Float16)x});
    return x * (float)p.x;
}
// The same Phi for N >= 2 pairs with the instruction ORDER fixed: one step of all N chains, then the next step.  A packed-f16 (VOP3P) result
// needs one wait state before the VALU instruction that reads it (gfx940+ destination forwarding); hipcc's own schedule of gelu_h2_phi ran one
// chain at a time and padded EVERY step with s_nop 0 -- 21 issue slots per pair, slower than the logistic form it replaces (AS-MLP-T -3 %
// same-box, profiles/r05_gelu_h2b_ab.txt).  Step-major order puts N - 1 independent instructions between dependent ones: no padding at all.
// asm volatile statements keep their order; everything else (the MFMAs and loads around an epilogue) still schedules freely between them.
template <int N> __device__ __forceinline__ void gelu_h2_phi_n(unsigned (&h)[N]) {          // h: packed f16 pairs in, Phi out
    static_assert(N >= 2, "two chains at least: the interleave IS the hazard distance");
    constexpr float c[7] = MLPK_GELUH_COEFS;
    auto pk = [](float v) { const unsigned short b = __builtin_bit_cast(unsigned short, (_Float16)v); return (unsigned)b | ((unsigned)b << 16); };
    const unsigned kS = pk(MLPK_GELUH_SCALE), c0 = pk(c[0]), k1 = pk(c[1]), k2 = pk(c[2]), k3 = pk(c[3]), k4 = pk(c[4]), k5 = pk(c[5]), k6 = pk(c[6]);
    unsigned t[N], u[N], q[N];
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(t[k]) : "v"(h[k]), "s"(kS));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %1, %1, -1.0 op_sel_hi:[1,1,0]" : "=v"(u[k]) : "v"(t[k]));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(q[k]) : "v"(c0), "v"(u[k]), "s"(k1));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(q[k]) : "v"(u[k]), "s"(k2));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(q[k]) : "v"(u[k]), "s"(k3));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(q[k]) : "v"(u[k]), "s"(k4));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(q[k]) : "v"(u[k]), "s"(k5));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(q[k]) : "v"(u[k]), "s"(k6));
#pragma unroll
    for (int k = 0; k < N - 1; ++k) asm volatile("v_pk_fma_f16 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(h[k]) : "v"(t[k]), "v"(q[k]));
    // (the compiler does not know that the last statement writes a packed result: the wait state in front of whatever reads it is spelled out)
    asm volatile("v_pk_fma_f16 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0" : "=v"(h[N - 1]) : "v"(t[N - 1]), "v"(q[N - 1]));
}

// gelu on N independent pairs with the N dependency chains interleaved step by step: a single wave running ONE
// chain is latency-bound (each v_pk op waits for its predecessor); N = 4 keeps the VALU issuing back to back.
// RAW: the bf16 grade's form (t = clamp(x, -scale, scale), u = t * t); else t = clamp(x * scale, -sqrt2, sqrt2), u = t * t - 1
template <int N, int K, bool RAW> __device__ __forceinline__ void gelu_pk_impl(f32x2 (&x)[N], const float (&c)[K], const float scale) {
    constexpr float r2 = 1.41421356237f;
    f32x2 t[N], u[N], q[N];
    if constexpr (RAW) {
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = f32x2{__builtin_amdgcn_fmed3f(x[k].x, -scale, scale), __builtin_amdgcn_fmed3f(x[k].y, -scale, scale)};
#pragma unroll
        for (int k = 0; k < N; ++k) u[k] = t[k] * t[k];
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = x[k] * f32x2{scale, scale};
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = f32x2{__builtin_amdgcn_fmed3f(t[k].x, -r2, r2), __builtin_amdgcn_fmed3f(t[k].y, -r2, r2)};
#pragma unroll
        for (int k = 0; k < N; ++k) u[k] = __builtin_elementwise_fma(t[k], t[k], f32x2{-1.0f, -1.0f});
    }
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = __builtin_elementwise_fma(u[k], f32x2{c[0], c[0]}, f32x2{c[1], c[1]});
#pragma unroll
    for (int i = 2; i < K; ++i)
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = __builtin_elementwise_fma(q[k], u[k], f32x2{c[i], c[i]});
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = x[k] * __builtin_elementwise_fma(t[k], q[k], f32x2{0.5f, 0.5f});
}

template <typename T, int N> __device__ __forceinline__ void gelu_pk_n(f32x2 (&x)[N]) {
#if !defined(MLPK_GELU_BF16_POLY) && !defined(MLPK_GELU_BF16_SIG)
    if constexpr (dtype_of<T>::value == MLPK_BF16) {
        if constexpr (N >= 2) {
            unsigned h[N];
#pragma unroll
            for (int k = 0; k < N; ++k) h[k] = __builtin_bit_cast(unsigned, h2_t{(_Float16)x[k].x, (_Float16)x[k].y});
            gelu_h2_phi_n<N>(h);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const h2_t p = __builtin_bit_cast(h2_t, h[k]);
                x[k] = f32x2{x[k].x * (float)p.x, x[k].y * (float)p.y};
            }
        } else {
            const h2_t p = gelu_h2_phi(h2_t{(_Float16)x[0].x, (_Float16)x[0].y});
            x[0] = f32x2{x[0].x * (float)p.x, x[0].y * (float)p.y};
        }
        return;
    }
#endif
#ifndef MLPK_GELU_BF16_POLY
    if constexpr (dtype_of<T>::value == MLPK_BF16) {
        // the N pairs step by step (the compiler is free to pair the fma / mul steps into v_pk_*: same bits either way)
        f32x2 a[N], q[N];
#pragma unroll
        for (int k = 0; k < N; ++k) a[k] = f32x2{__builtin_fabsf(x[k].x), __builtin_fabsf(x[k].y)};
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = __builtin_elementwise_fma(a[k], f32x2{MLPK_GELUS_K2, MLPK_GELUS_K2}, f32x2{MLPK_GELUS_K1, MLPK_GELUS_K1});
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = __builtin_elementwise_fma(a[k], q[k], f32x2{MLPK_GELUS_K0, MLPK_GELUS_K0});
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = x[k] * q[k];
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = f32x2{__builtin_amdgcn_exp2f(q[k].x), __builtin_amdgcn_exp2f(q[k].y)};
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = q[k] + f32x2{1.0f, 1.0f};
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = f32x2{__builtin_amdgcn_rcpf(q[k].x), __builtin_amdgcn_rcpf(q[k].y)};
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = x[k] * q[k];
        return;
    }
#endif
    if constexpr (dtype_of<T>::value == MLPK_BF16) {
        constexpr float c[8] = MLPK_GELUP_COEFS_BF16;
        gelu_pk_impl<N, 8, true>(x, c, MLPK_GELUP_CLAMP_BF16);
    } else {
        constexpr float c[11] = MLPK_GELUP_COEFS;
        gelu_pk_impl<N, 11, false>(x, c, MLPK_GELUP_SCALE);
    }
}

template <typename T> __device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {
    f32x2 v[1] = {x};
    gelu_pk_n<T, 1>(v);
    return v[0];
}

// scalar form of gelu_pk (the same operation sequence, hence the same results)
template <int K, bool RAW> __device__ __forceinline__ float gelu16_impl(float x, const float (&c)[K], const float scale) {
    constexpr float r2 = 1.41421356237f;
    const float t = RAW ? __builtin_amdgcn_fmed3f(x, -scale, scale) : __builtin_amdgcn_fmed3f(x * scale, -r2, r2);
    const float u = RAW ? t * t : __builtin_fmaf(t, t, -1.0f);
    float q = __builtin_fmaf(u, c[0], c[1]);
#pragma unroll
    for (int i = 2; i < K; ++i) q = __builtin_fmaf(q, u, c[i]);
    return x * __builtin_fmaf(t, q, 0.5f);
}

template <typename T> __device__ __forceinline__ float gelu16_f(float x) {
#if !defined(MLPK_GELU_BF16_POLY) && !defined(MLPK_GELU_BF16_SIG)
    if constexpr (dtype_of<T>::value == MLPK_BF16) return gelu_h2b_f(x);
#endif
#ifndef MLPK_GELU_BF16_POLY
    if constexpr (dtype_of<T>::value == MLPK_BF16) return gelu_sig_f(x);
#endif
    if constexpr (dtype_of<T>::value == MLPK_BF16) {
        constexpr float c[8] = MLPK_GELUP_COEFS_BF16;
        return gelu16_impl<8, true>(x, c, MLPK_GELUP_CLAMP_BF16);
    } else {
        constexpr float c[11] = MLPK_GELUP_COEFS;
        return gelu16_impl<11, false>(x, c, MLPK_GELUP_SCALE);
    }
}

template <typename T> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (sizeof(T) == 2) return gelu16_f<T>(x);
    else return gelu_f(x);
}

// Sum over the 64 lanes of a wave, result in every lane.  DPP row operations (6 VALU instructions + one readlane)
// instead of six __shfl_xor, which compile to ds_bpermute_b32 round trips through the LDS crossbar.
__device__ __forceinline__ float wave_sum(float v) {
    // within each row of 16 lanes: xor 1, xor 2 (quad_perm), then half-mirror / mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    // every lane of a row now holds the row's sum: add rows 0+1 and 2+3, then the two halves
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));   // row_bcast15 -> rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));   // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum over each row of 16 lanes (the first four steps of wave_sum), result in every lane of the row.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Sum over each group of 4 consecutive lanes (the first two steps of row16_sum), result in every lane of the group: with one 16-byte
// chunk (8 columns) per lane this is the 32-column partial of the by-product row statistics, (c0 + c1) + (c2 + c3) -- the ONE
// reduction order every tile of the library uses (mlpk.h row_part), so a row's statistics do not depend on the tile that stored it.
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}

// (sum, sum of squares) of the eight 16-bit values of one 16-byte chunk, accumulated in fp32 by the packed dot products
// (v_dot2c_f32_bf16 / v_dot2c_f32_f16: two products + the accumulator per instruction, 8 instructions per chunk where
// convert + add + fma take 24)
template <typename T> __device__ __forceinline__ void chunk_sums(const u32x4 v, float& s1, float& s2) {
    static_assert(sizeof(T) == 2, "16-bit storage types");
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (dtype_of<T>::value == MLPK_BF16) {
            typedef __attribute__((ext_vector_type(2))) __bf16 pair_t;
            const pair_t a = __builtin_bit_cast(pair_t, w[k]);
            const pair_t one = {(__bf16)1.0f, (__bf16)1.0f};
            s1 = __builtin_amdgcn_fdot2_f32_bf16(a, one, s1, false);
            s2 = __builtin_amdgcn_fdot2_f32_bf16(a, a, s2, false);
        } else {
            typedef __attribute__((ext_vector_type(2))) _Float16 pair_t;
            const pair_t a = __builtin_bit_cast(pair_t, w[k]);
            const pair_t one = {(_Float16)1.0f, (_Float16)1.0f};
            s1 = __builtin_amdgcn_fdot2(a, one, s1, false);
            s2 = __builtin_amdgcn_fdot2(a, a, s2, false);
        }
    }
}

// Bijective XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): gives every XCD a
// contiguous range of logical tile ids so that neighbouring tiles share operand panels in
// that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

}  // namespace mlpk

int mlpk_dwconv_direct(int dtype, const void* x, void* out, int B, int H, int W, int C, int k, const float* w,
                       const float* bias, const float* bn_scale, const float* bn_shift, void* stream);

#define MLPK_LAUNCH_CHECK()                            \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)
