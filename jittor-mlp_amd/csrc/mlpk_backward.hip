// Backward of the hot path (round 5): the element-wise, normalisation and layout kernels that autograd through the
// GEMM epilogues needs beside mlpk_gemm_nt itself (the two products of a Linear's backward, dX = dY W and dW = dY^T X, ARE mlpk_gemm_nt
// calls on transposed operands).  What they follow in the reference: nn.GELU (mlp_mixer.py:21), nn.LayerNorm (mlp_mixer.py:9),
// nn.Linear / Conv1d(k=1) bias gradients (mlp_mixer.py:19-25), the token <-> channel transposes of the token-mixing FeedForward
// (mlp_mixer.py:34: dense = Conv1d over the patch axis), Reduce('b n c -> b c', 'mean') (mlp_mixer.py:63), and BatchNorm2d's batch
// statistics (conv_mixer.py:20,28,31).  All HBM-bound; none of them is on the inference path.  fp32 math, one rounding per stored value.
#include "mlpk_common.h"

namespace mlpk {

// exact-form GELU and its derivative: gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
    const float Phi = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    return __builtin_fmaf(x, phi, Phi);
}

// mode 0: out = gelu(a);  mode 1: out = b * gelu'(a)   (a = the pre-activation, b = the incoming gradient)
template <typename T, int MODE> __global__ void __launch_bounds__(256) gelu_ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                                                     int64_t rows, int cols, int64_t ld) {
    const int64_t total = rows * (int64_t)cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int64_t o = r * ld + (i - r * cols);
        const float x = to_f32<T>(a[o]);
        out[o] = from_f32<T>(MODE == 0 ? gelu_f(x) : to_f32<T>(b[o]) * gelu_grad_f(x));
    }
}

// LayerNorm backward over the last axis (nn.LayerNorm, biased variance): one wave per row, x^ = (x - mean) rstd,
//   dx = rstd (g - mean_c(g) - x^ mean_c(g x^)),  g = dy gamma;   partial dgamma / dbeta per workgroup (4 rows at a time, deterministic order)
template <typename T> __global__ void __launch_bounds__(256) layernorm_bwd_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
                                                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                                  const T* __restrict__ dy, int64_t lddy, T* __restrict__ dx, int64_t lddx,
                                                                                  float* __restrict__ part, int64_t rows, int C, int rows_per_wg) {
    extern __shared__ float sm[];          // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* dg = sm + (size_t)wave * 2 * C;
    float* db = dg + C;
    for (int c = lane; c < C; c += 64) { dg[c] = 0.f; db[c] = 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    for (int64_t r = r0 + wave; r < r1; r += 4) {
        const float mu = mean[r], rs = rstd[r];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xh = (to_f32<T>(x[r * ldx + c]) - mu) * rs;
            const float g = to_f32<T>(dy[r * lddy + c]) * gamma[c];
            s1 += g;
            s2 += g * xh;
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
        for (int c = lane; c < C; c += 64) {
            const float xh = (to_f32<T>(x[r * ldx + c]) - mu) * rs;
            const float d = to_f32<T>(dy[r * lddy + c]);
            dx[r * lddx + c] = from_f32<T>(rs * (d * gamma[c] - s1 - xh * s2));
            dg[c] += d * xh;               // (lane-private columns: no race inside the wave)
            db[c] += d;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c / C, cc = c - which * C;
        float v = 0.f;
        for (int w = 0; w < 4; ++w) v += sm[(size_t)w * 2 * C + (size_t)which * C + cc];
        part[((size_t)blockIdx.x * 2 + which) * C + cc] = v;
    }
}

// column sums over rows, fp32 out: out[c] = sum_r f(x[r, c]) (SQ: f = square).  One workgroup per 64 columns, 4 row lanes, a fixed
// summation order (no atomics): the bias gradients of a Linear, the second stage of the LayerNorm parameter gradients, BatchNorm's batch sums.
template <typename T, bool SQ> __global__ void __launch_bounds__(256) col_sum_kernel(const T* __restrict__ x, const T* __restrict__ sub, int64_t rows, int cols, int64_t ld,
                                                                                     float* __restrict__ out) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float s = 0.f, comp = 0.f;             // Kahan: 50 000 rows of one sign must not lose the small terms
    if (c < cols)
        for (int64_t r = rl; r < rows; r += 4) {
            float v = to_f32<T>(x[r * ld + c]);
            if (sub) v -= to_f32<T>(sub[r * ld + c]);      // statistics of a difference: what a kernel with a built-in residual added
            if (SQ) v *= v;
            const float y = v - comp;
            const float t = s + y;
            comp = (t - s) - y;
            s = t;
        }
    sm[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) out[c] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// batched 2-D transpose with an optional addend: out[b, c, r] = in[b, r, c] (+ res[b, c, r]);  in: (batch, R, ld_in), out: (batch, Cc, ld_out)
template <typename T> __global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, int64_t ld_in, T* __restrict__ out, int64_t ld_out,
                                                                              const T* __restrict__ res, int64_t ld_res, int R, int Cc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows of 32 per pass
    const T* ib = in + (size_t)b * R * ld_in;
    T* ob = out + (size_t)b * Cc * ld_out;
    const T* rb = res ? res + (size_t)b * Cc * ld_res : nullptr;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < Cc) tile[k][tx] = to_f32<T>(ib[(size_t)(r0 + k) * ld_in + c0 + tx]);
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < Cc && r0 + tx < R) {
            float v = tile[tx][k];
            if (rb) v += to_f32<T>(rb[(size_t)(c0 + k) * ld_res + r0 + tx]);
            ob[(size_t)(c0 + k) * ld_out + r0 + tx] = from_f32<T>(v);
        }
}

// out[b, s, c] = scale * in[b, c]: the backward of the token mean (mlp_mixer.py:63)
template <typename T> __global__ void __launch_bounds__(256) broadcast_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int S, int C, float scale) {
    const int64_t total = (int64_t)B * S * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t b = i / ((int64_t)S * C);
        out[i] = from_f32<T>(to_f32<T>(in[b * C + c]) * scale);
    }
}

// x[m, c] += t[m % period, c] (t fp32, one rounding): SwinMLP's absolute position embedding added to every image's tokens (swin_mlp.py:437-438)
template <typename T> __global__ void __launch_bounds__(256) add_periodic_kernel(T* __restrict__ x, int64_t ldx, const float* __restrict__ t, int64_t rows, int C,
                                                                                  int period) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t m = i / C;
        T* px = x + m * ldx + c;
        *px = from_f32<T>(to_f32<T>(*px) + t[(m % period) * C + c]);
    }
}

static unsigned ew_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace mlpk

#define BW_DISPATCH(DT, CALL_F32, CALL_F16, CALL_BF16) \
    switch (DT) {                                      \
        case MLPK_F32: CALL_F32; break;                \
        case MLPK_F16: CALL_F16; break;                \
        case MLPK_BF16: CALL_BF16; break;              \
        default: return MLPK_EDTYPE;                   \
    }

extern "C" int mlpk_gelu_elementwise(int dtype, int mode, const void* a, const void* b, void* out, int64_t rows, int cols, int64_t ld, void* stream) {
    using namespace mlpk;
    if (!a || !out || (mode == 1 && !b)) return MLPK_ENULL;
    if (mode != 0 && mode != 1) return MLPK_EMODE;
    if (rows <= 0 || cols <= 0 || ld < cols) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid(rows * (int64_t)cols);
#define GE(TT)                                                                                                                                       \
    if (mode == 0) hipLaunchKernelGGL((gelu_ew_kernel<TT, 0>), dim3(g), dim3(256), 0, s, (const TT*)a, (const TT*)b, (TT*)out, rows, cols, ld);        \
    else hipLaunchKernelGGL((gelu_ew_kernel<TT, 1>), dim3(g), dim3(256), 0, s, (const TT*)a, (const TT*)b, (TT*)out, rows, cols, ld)
    BW_DISPATCH(dtype, GE(float), GE(f16_t), GE(bf16_t))
#undef GE
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_layernorm_backward_blocks(int64_t rows) { return (int)((rows + 255) / 256); }

extern "C" int mlpk_layernorm_backward(int dtype, const void* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma, const void* dy,
                                       int64_t lddy, void* dx, int64_t lddx, float* part, int64_t rows, int C, void* stream) {
    using namespace mlpk;
    if (!x || !mean || !rstd || !gamma || !dy || !dx || !part) return MLPK_ENULL;
    if (rows <= 0 || C <= 0 || ldx < C || lddy < C || lddx < C) return MLPK_ESHAPE;
    if ((size_t)C * 8 * sizeof(float) > 64 * 1024) return MLPK_ESHAPE;            // [4][2][C] floats of LDS: C <= 2048
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nb = mlpk_layernorm_backward_blocks(rows);
    const size_t lds = (size_t)C * 8 * sizeof(float);
#define LB(TT)                                                                                                                                   \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TT>), dim3(nb), dim3(256), lds, s, (const TT*)x, ldx, mean, rstd, gamma, (const TT*)dy, lddy, (TT*)dx, lddx, \
                       part, rows, C, 256)
    BW_DISPATCH(dtype, LB(float), LB(f16_t), LB(bf16_t))
#undef LB
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_col_sum(int dtype, const void* x, const void* sub, int64_t rows, int cols, int64_t ld, int square, float* out, void* stream) {
    using namespace mlpk;
    if (!x || !out) return MLPK_ENULL;
    if (rows <= 0 || cols <= 0 || ld < cols) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = (unsigned)((cols + 63) / 64);
#define CS(TT)                                                                                                        \
    if (square) hipLaunchKernelGGL((col_sum_kernel<TT, true>), dim3(g), dim3(256), 0, s, (const TT*)x, (const TT*)sub, rows, cols, ld, out); \
    else hipLaunchKernelGGL((col_sum_kernel<TT, false>), dim3(g), dim3(256), 0, s, (const TT*)x, (const TT*)sub, rows, cols, ld, out)
    BW_DISPATCH(dtype, CS(float), CS(f16_t), CS(bf16_t))
#undef CS
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_transpose_batched(int dtype, const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* res, int64_t ld_res, int batch,
                                      int R, int Cc, void* stream) {
    using namespace mlpk;
    if (!in || !out) return MLPK_ENULL;
    if (batch <= 0 || R <= 0 || Cc <= 0 || ld_in < Cc || ld_out < R || (res && ld_res < R) || batch > 65535) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g((Cc + 31) / 32, (R + 31) / 32, batch);
    if (g.y > 65535) return MLPK_ESHAPE;
#define TR(TT) hipLaunchKernelGGL((transpose_kernel<TT>), g, dim3(256), 0, s, (const TT*)in, ld_in, (TT*)out, ld_out, (const TT*)res, ld_res, R, Cc)
    BW_DISPATCH(dtype, TR(float), TR(f16_t), TR(bf16_t))
#undef TR
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_broadcast_rows(int dtype, const void* in, void* out, int B, int S, int C, float scale, void* stream) {
    using namespace mlpk;
    if (!in || !out) return MLPK_ENULL;
    if (B <= 0 || S <= 0 || C <= 0) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid((int64_t)B * S * C);
#define BR(TT) hipLaunchKernelGGL((broadcast_rows_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)in, (TT*)out, B, S, C, scale)
    BW_DISPATCH(dtype, BR(float), BR(f16_t), BR(bf16_t))
#undef BR
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_add_periodic(int dtype, void* x, int64_t ldx, const float* t, int64_t rows, int C, int period, void* stream) {
    using namespace mlpk;
    if (!x || !t) return MLPK_ENULL;
    if (rows <= 0 || C <= 0 || period <= 0 || ldx < C) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid(rows * C);
#define AP(TT) hipLaunchKernelGGL((add_periodic_kernel<TT>), dim3(g), dim3(256), 0, s, (TT*)x, ldx, t, rows, C, period)
    BW_DISPATCH(dtype, AP(float), AP(f16_t), AP(bf16_t))
#undef AP
    MLPK_LAUNCH_CHECK();
    return 0;
}
