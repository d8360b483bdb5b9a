// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mlpk.h"

namespace mlpk {

typedef __attribute__((ext_vector_type(4))) float f32x4;
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
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
