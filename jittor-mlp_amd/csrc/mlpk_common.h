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

// GELU for 16-bit outputs: erf by Abramowitz & Stegun 7.1.28, erf(z) = 1 - (1 + a1 z + .. + a6 z^6)^-16
// (|err| <= 3e-7), with the 1/sqrt(2) folded into the coefficients: ONE quarter-rate instruction (rcp) per
// element instead of two (rcp + exp), everything else pairs into v_pk_* ops.  The polynomial is additionally scaled
// by 2^(1/16), so that its 16th power carries the factor 2 and the reciprocal comes out already halved:
//   gelu(x) = x/2 + |x|/2 * erf(|x|/sqrt 2) = max(x, 0) - |x| * D'(|x|)^-16,    D' = 2^(1/16) * D
// 17 VALU instructions per PAIR of elements.  |GELU error| <= 7.1e-7 (measured over [-12, 12]), far below half an
// ulp of f16/bf16; float outputs keep gelu_f.
#define MLPK_GELU_C0 1.0442737340927124f
#define MLPK_GELU_C1 0.052075162529945374f
#define MLPK_GELU_C2 0.02207699790596962f
#define MLPK_GELU_C3 0.003422739217057824f
#define MLPK_GELU_C4 3.9686136005911976e-05f
#define MLPK_GELU_C5 5.105520904180594e-05f
#define MLPK_GELU_C6 5.62129980608006e-06f

// gelu on N independent pairs with the N dependency chains interleaved step by step: a single wave running ONE
// chain is latency-bound (each v_pk op waits for its predecessor); N = 4 keeps the VALU issuing back to back.
template <int N> __device__ __forceinline__ void gelu_pk_n(f32x2 (&x)[N]) {
    f32x2 ax[N], d[N];
#pragma unroll
    for (int c = 0; c < N; ++c) ax[c] = __builtin_elementwise_abs(x[c]);
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = __builtin_elementwise_fma(ax[c], f32x2{MLPK_GELU_C6, MLPK_GELU_C6}, f32x2{MLPK_GELU_C5, MLPK_GELU_C5});
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = __builtin_elementwise_fma(d[c], ax[c], f32x2{MLPK_GELU_C4, MLPK_GELU_C4});
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = __builtin_elementwise_fma(d[c], ax[c], f32x2{MLPK_GELU_C3, MLPK_GELU_C3});
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = __builtin_elementwise_fma(d[c], ax[c], f32x2{MLPK_GELU_C2, MLPK_GELU_C2});
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = __builtin_elementwise_fma(d[c], ax[c], f32x2{MLPK_GELU_C1, MLPK_GELU_C1});
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = __builtin_elementwise_fma(d[c], ax[c], f32x2{MLPK_GELU_C0, MLPK_GELU_C0});
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < N; ++c) d[c] = d[c] * d[c];
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = f32x2{__builtin_amdgcn_rcpf(d[c].x), __builtin_amdgcn_rcpf(d[c].y)};
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const f32x2 relu = __builtin_elementwise_fma(x[c], f32x2{0.5f, 0.5f}, ax[c] * 0.5f);     // max(x, 0) = x/2 + |x|/2
        x[c] = __builtin_elementwise_fma(-ax[c], d[c], relu);
    }
}

// Division-free form for the kernels whose time is VALU instructions (the fused token-mixing kernel: its two waves per SIMD
// serialise on the VALU port, a v_rcp_f32 costs four plain slots there):
//   gelu(x) = x * Phi(x),  Phi(x) ~= 0.5 + t * Q(t^2 - 1),  t = clamp(x * sqrt2 / 4.5, -sqrt2, sqrt2)
// Q = degree-10 polynomial in u = t^2 - 1 in [-1, 1] (weighted minimax fit of the error of Phi, tools/fit_gelu_poly.py; sum |c| =
// 1.27, so fp32 Horner is well conditioned).  16 VALU per PAIR and no transcendental, vs 19 + 2 v_rcp_f32 above.
// |Phi error| <= 2.7e-6 everywhere (|x| > 4.5 clamps to Phi(4.5) = 1 - 3.4e-6), |gelu error| <= 3.7e-6 on |x| <= 4.5: 60x below
// half an ulp of f16 at that magnitude; checked in fp32 emulation by tests/test_host_cpu.py against the coefficients HERE.
#define MLPK_GELUP_SCALE 0.314269681f
#define MLPK_GELUP_COEFS {0.00260713836f, -0.00718860654f, 0.00979797821f, -0.0172248576f, 0.0355015062f, -0.0601866171f, 0.090279378f, -0.127707109f, 0.174028099f, -0.245624334f, 0.499268919f}
template <int N> __device__ __forceinline__ void gelu_poly_pk_n(f32x2 (&x)[N]) {
    constexpr float c[11] = MLPK_GELUP_COEFS;
    constexpr float r2 = 1.41421356237f;
    f32x2 t[N], u[N], q[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = x[k] * f32x2{MLPK_GELUP_SCALE, MLPK_GELUP_SCALE};
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = f32x2{__builtin_amdgcn_fmed3f(t[k].x, -r2, r2), __builtin_amdgcn_fmed3f(t[k].y, -r2, r2)};
#pragma unroll
    for (int k = 0; k < N; ++k) u[k] = __builtin_elementwise_fma(t[k], t[k], f32x2{-1.0f, -1.0f});
#pragma unroll
    for (int k = 0; k < N; ++k) q[k] = __builtin_elementwise_fma(u[k], f32x2{c[0], c[0]}, f32x2{c[1], c[1]});
#pragma unroll
    for (int i = 2; i < 11; ++i)
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = __builtin_elementwise_fma(q[k], u[k], f32x2{c[i], c[i]});
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = x[k] * __builtin_elementwise_fma(t[k], q[k], f32x2{0.5f, 0.5f});
}

__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {
    f32x2 v[1] = {x};
    gelu_pk_n<1>(v);
    return v[0];
}

// scalar form of gelu_pk (the same operation sequence, hence the same results)
__device__ __forceinline__ float gelu16_f(float x) {
    const float ax = __builtin_fabsf(x);
    float d = __builtin_fmaf(ax, MLPK_GELU_C6, MLPK_GELU_C5);
    d = __builtin_fmaf(d, ax, MLPK_GELU_C4);
    d = __builtin_fmaf(d, ax, MLPK_GELU_C3);
    d = __builtin_fmaf(d, ax, MLPK_GELU_C2);
    d = __builtin_fmaf(d, ax, MLPK_GELU_C1);
    d = __builtin_fmaf(d, ax, MLPK_GELU_C0);
    d = d * d;
    d = d * d;
    d = d * d;
    d = d * d;
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(-ax, r, __builtin_fmaf(x, 0.5f, ax * 0.5f));
}

template <typename T> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (sizeof(T) == 2) return gelu16_f(x);
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
