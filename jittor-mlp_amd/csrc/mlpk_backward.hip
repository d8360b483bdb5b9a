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

// ---- round 6: the element-wise / normalisation / remap derivatives the other families' train mode needs ----
// per-column / per-row-group combinations of one or two row-major tensors, fp32 math, one rounding:
//   0: out = a * g[c] + h[c]           Aff (res_mlp.py:17-19), GroupNorm / BatchNorm affine; dx of a per-channel scale (h NULL)
//   1: out = a * b                     the SGU gate u * v and its two derivatives (g_mlp.py:21)
//   2: out = a + g[c] * b              x + gamma * f(x) (res_mlp.py:53,55); g NULL: a + b (sum of two gradient paths, x_lr + x_td as_mlp.py:91)
//   3: out = a * g[m / period]         stochastic depth's per-sample scale and its derivative (as_mlp.py:159-160)
//   4: out = a * g[c] + b * h[c] + k[c]   BatchNorm (batch statistics) backward: dx = (gamma / sigma) (dy - mean(dy) - x^ mean(dy x^)) (conv_mixer.py:20,28,31)
//   5: out = a * g[i, c] + h[i, c] + b,  i = m / period   SplitAttention's weighted sum over the branches, one term per call (vip.py:54-56,
//      s2_mlp_v2.py:48-50), and its derivative w.r.t. a branch: dy * bar_a[k] + the broadcast gradient of the pixel sum (h, b optional)
template <typename T, int MODE> __global__ void __launch_bounds__(256) ew_cols_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                                                                                     const float* __restrict__ g, const float* __restrict__ h,
                                                                                     const float* __restrict__ k, T* __restrict__ out, int64_t ldo,
                                                                                     int64_t rows, int cols, int period) {
    const int64_t total = rows * (int64_t)cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float av = to_f32<T>(a[r * lda + c]);
        float v;
        if (MODE == 0) v = __builtin_fmaf(av, g ? g[c] : 1.f, h ? h[c] : 0.f);
        else if (MODE == 1) v = av * to_f32<T>(b[r * ldb + c]);
        else if (MODE == 2) v = __builtin_fmaf(g ? g[c] : 1.f, to_f32<T>(b[r * ldb + c]), av);
        else if (MODE == 3) v = av * g[r / period];
        else if (MODE == 4) v = __builtin_fmaf(av, g[c], __builtin_fmaf(to_f32<T>(b[r * ldb + c]), h[c], k[c]));
        else {
            const int64_t pi = (r / period) * cols + c;
            v = __builtin_fmaf(av, g[pi], (h ? h[pi] : 0.f) + (b ? to_f32<T>(b[r * ldb + c]) : 0.f));
        }
        out[r * ldo + c] = from_f32<T>(v);
    }
}

// out[c] = sum over rows of x[r, c] * y[r, c], fp32 (Kahan, fixed order): the gradient of a per-channel scale (Aff alpha, layer scale gamma,
// GroupNorm / BatchNorm weight: sum of dy * x^)
// (blockIdx.y = segment of `rows` consecutive rows: mlpk_col_dot_seg -- per-image sums, the gradient of SplitAttention's weights)
template <typename T> __global__ void __launch_bounds__(256) col_dot_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ y, int64_t ldy, int64_t rows,
                                                                            int cols, float* __restrict__ out) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float s = 0.f, comp = 0.f;
    x += (int64_t)blockIdx.y * rows * ldx;
    y += (int64_t)blockIdx.y * rows * ldy;
    out += (int64_t)blockIdx.y * cols;
    if (c < cols)
        for (int64_t r = rl; r < rows; r += 4) {
            const float v = to_f32<T>(x[r * ldx + c]) * to_f32<T>(y[r * ldy + c]);
            const float yk = v - comp;
            const float t = s + yk;
            comp = (t - s) - yk;
            s = t;
        }
    sm[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) out[c] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// GroupNorm(1, C) backward on channel-last samples (as_mlp.py:343-344: MyNorm = GroupNorm(1, dim)): a sample is glen = H W C contiguous
// values; xh = the normalised values (before the affine), g = dy * gamma[c] (the caller's mode-0 pass):
//   dx = rstd[b] * (g - mean_b(g) - xh * mean_b(g xh))
// one workgroup per sample, two passes over it (the second hits the L2), fp32 sums in a fixed order
template <typename T> __global__ void __launch_bounds__(1024) group_norm_bwd_kernel(const T* __restrict__ xh, const T* __restrict__ g, const float* __restrict__ rstd,
                                                                                    T* __restrict__ dx, int64_t glen) {
    __shared__ float sm[2][16];
    const int64_t base = (int64_t)blockIdx.x * glen;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = threadIdx.x; i < glen; i += 1024) {
        const float gv = to_f32<T>(g[base + i]);
        s1 += gv;
        s2 = __builtin_fmaf(gv, to_f32<T>(xh[base + i]), s2);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s1; sm[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
    for (int w = 0; w < 16; ++w) { t1 += sm[0][w]; t2 += sm[1][w]; }
    const float m1 = t1 / (float)glen, m2 = t2 / (float)glen, rs = rstd[blockIdx.x];
    for (int64_t i = threadIdx.x; i < glen; i += 1024)
        dx[base + i] = from_f32<T>(rs * (to_f32<T>(g[base + i]) - m1 - to_f32<T>(xh[base + i]) * m2));
}

// adjoint of mlpk_shift_nhwc (as_mlp.py:84-89 through the channel-last layout): grad_in[n,h,w,c] = grad_out[n,h-s,w,c] (dim 2) |
// grad_out[n,h,w-s,c] (dim 3), s = ksz / 2 - c / group, zero outside -- shift_backward_grad_input_kernel (utils/shift_cuda.py:75-103) on (N,H,W,C)
template <typename T> __global__ void __launch_bounds__(256) shift_nhwc_bwd_kernel(const T* __restrict__ go, T* __restrict__ gi, int N, int H, int W, int C, int ksz,
                                                                                   int dim) {
    const int64_t total = (int64_t)N * H * W * C;
    const int group = (C + ksz - 1) / ksz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int w = (int)((idx / C) % W);
        const int h = (int)((idx / ((int64_t)C * W)) % H);
        const int s = ksz / 2 - c / group;
        T v = from_f32<T>(0.f);
        if (dim == 2) {
            if (h - s >= 0 && h - s < H) v = go[idx - (int64_t)s * W * C];
        } else {
            if (w - s >= 0 && w - s < W) v = go[idx - (int64_t)s * C];
        }
        gi[idx] = v;
    }
}

// the im2col half of a kernel == stride convolution on channel-last tensors, and its adjoint (a permutation: every element moves once):
// PatchMerging's gather (as_mlp.py:207-211: cat of x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2] over the channels = order 1) and the
// stage convolutions of S2-MLPv2 read from the previous stage's channel-last rows (s2_mlp_v2.py:118-119 = mlpk_patchify NHWC order 0).
//   dir 0: (B,H,W,C) -> (B,H/ph,W/pw, ph*pw*C), column ((i*pw + j) [order 0] | (j*ph + i) [order 1]) * C + c;   dir 1: back
template <typename T> __global__ void __launch_bounds__(256) patch_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C, int ph, int pw,
                                                                               int order, int dir) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int x = (int)((idx / C) % W);
        const int y = (int)((idx / ((int64_t)C * W)) % H);
        const int64_t b = idx / ((int64_t)C * W * H);
        const int i = y % ph, j = x % pw;
        const int q = order == 0 ? i * pw + j : j * ph + i;
        const int64_t m = ((b * (H / ph) + y / ph) * (W / pw) + x / pw) * ((int64_t)ph * pw * C) + (int64_t)q * C + c;
        if (dir == 0) dst[m] = src[idx];
        else dst[idx] = src[m];
    }
}

// depthwise Conv2d(k, groups = C, padding = "same") on channel-last tensors without an epilogue (train mode keeps the pre-activation), and its
// adjoint (the gradient w.r.t. the input): w fp32 [k*k][C] tap-major; "same" pads (k-1)/2 before and k/2 after (conv_mixer.py:25)
//   fwd:     out[b,y,x,c] = bias[c] + sum_{i,j} w[i,j,c] in[b, y+i-p, x+j-p, c]
//   adjoint: out[b,y,x,c] =           sum_{i,j} w[i,j,c] in[b, y-i+p, x-j+p, c]
template <typename T> __global__ void __launch_bounds__(256) dwconv_plain_kernel(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ w,
                                                                                 const float* __restrict__ bias, int B, int H, int W, int C, int k, int adj) {
    const int64_t total = (int64_t)B * H * W * C;
    const int p = (k - 1) / 2;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int x = (int)((idx / C) % W);
        const int y = (int)((idx / ((int64_t)C * W)) % H);
        const int64_t b = idx / ((int64_t)C * W * H);
        float acc = (bias && !adj) ? bias[c] : 0.f;
        for (int i = 0; i < k; ++i) {
            const int yy = adj ? y - i + p : y + i - p;
            if (yy < 0 || yy >= H) continue;
            for (int j = 0; j < k; ++j) {
                const int xx = adj ? x - j + p : x + j - p;
                if (xx < 0 || xx >= W) continue;
                acc = __builtin_fmaf(w[(size_t)(i * k + j) * C + c], to_f32<T>(in[((b * H + yy) * W + xx) * (int64_t)C + c]), acc);
            }
        }
        out[idx] = from_f32<T>(acc);
    }
}

// gradient of the depthwise taps: dw[i,j,c] = sum_{b,y,x} dy[b,y,x,c] x[b, y+i-p, x+j-p, c].  One workgroup per (64 channels, tap): 4 pixel
// lanes, Kahan sums, fixed order (no atomics)
template <typename T> __global__ void __launch_bounds__(256) dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, int B, int H,
                                                                                 int W, int C, int k) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;
    const int tap = blockIdx.y, i = tap / k, j = tap - i * k, p = (k - 1) / 2;
    float s = 0.f, comp = 0.f;
    if (c < C) {
        const int64_t npx = (int64_t)B * H * W;
        for (int64_t px = pl; px < npx; px += 4) {
            const int xo = (int)(px % W), yo = (int)((px / W) % H);
            const int yy = yo + i - p, xx = xo + j - p;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int64_t src = px + (int64_t)(yy - yo) * W + (xx - xo);
            const float v = to_f32<T>(dy[px * C + c]) * to_f32<T>(x[src * C + c]);
            const float yk = v - comp;
            const float t = s + yk;
            comp = (t - s) - yk;
            s = t;
        }
    }
    sm[pl][threadIdx.x & 63] = s;
    __syncthreads();
    if (pl == 0 && c < C) dw[(size_t)tap * C + c] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// softmax over the k = 3 branches, backward (vip.py:52-53, s2_mlp_v2.py:46-47): bar, dbar, dhat fp32 [B][3][C]
//   dhat[k] = bar[k] * (dbar[k] - sum_j bar[j] dbar[j])
__global__ void __launch_bounds__(256) split_softmax_bwd_kernel(const float* __restrict__ bar, const float* __restrict__ dbar, float* __restrict__ dhat, int B, int C) {
    const int64_t total = (int64_t)B * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / C;
        const int c = (int)(i - b * C);
        const int64_t o = b * 3 * C + c;
        const float a0 = bar[o], a1 = bar[o + C], a2 = bar[o + 2 * C];
        const float d0 = dbar[o], d1 = dbar[o + C], d2 = dbar[o + 2 * C];
        const float sdot = __builtin_fmaf(a0, d0, __builtin_fmaf(a1, d1, a2 * d2));
        dhat[o] = a0 * (d0 - sdot);
        dhat[o + C] = a1 * (d1 - sdot);
        dhat[o + 2 * C] = a2 * (d2 - sdot);
    }
}

// S2-MLPv2's spatial_shift1 / spatial_shift2 (s2_mlp_v2.py:15-29) on (B, D1, D2, C) rows with strides ldi / ldo, out of place, and the
// backward the REFERENCE's autograd performs.  The reference assigns in place on overlapping views: its forward smears the +1 groups
// (mode 1: y[i] = x[0]; mode 0 = the intended shift y[i] = x[i-1]), while its autograd (CopySlices on the pre-assignment values) returns
// the adjoint of the INTENDED shift whatever the forward did -- checked on the reference itself (tests/golden/make_golden.py --only traingrad).
//   which 1: groups (dim1,+1), (dim1,-1), (dim2,+1), (dim2,-1);  which 2: (dim2,+1), (dim2,-1), (dim1,+1), (dim1,-1)
//   adjoint: dx[j] = dy[j+1] (j+1 < n) + (j == 0 ? dy[0] : 0) for +1;   dx[j] = dy[j-1] (j >= 1) + (j == n-1 ? dy[n-1] : 0) for -1
template <typename T> __global__ void __launch_bounds__(256) s2_shift2_kernel(const T* __restrict__ in, int64_t ldi, T* __restrict__ out, int64_t ldo, int B, int D1,
                                                                              int D2, int C, int which, int mode, int adjoint) {
    const int64_t total = (int64_t)B * D1 * D2 * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int64_t px = idx / C;
        const int j = (int)(px % D2);
        const int i = (int)((px / D2) % D1);
        const int grp = c < C / 4 ? 0 : (c < C / 2 ? 1 : (c < C * 3 / 4 ? 2 : 3));
        const bool ax1 = which == 1 ? grp < 2 : grp >= 2;          // the axis this group moves along
        const int dir = (grp & 1) ? -1 : +1;
        const int n = ax1 ? D1 : D2, pos = ax1 ? i : j;
        const int64_t step = ax1 ? (int64_t)D2 : 1;                // rows between neighbours along the axis
        float v;
        if (!adjoint) {
            int src = dir < 0 ? (pos + 1 < n ? pos + 1 : n - 1) : (mode == 1 ? 0 : (pos > 0 ? pos - 1 : 0));
            v = to_f32<T>(in[(px + (int64_t)(src - pos) * step) * ldi + c]);
        } else if (dir > 0) {
            v = (pos + 1 < n ? to_f32<T>(in[(px + step) * ldi + c]) : 0.f) + (pos == 0 ? to_f32<T>(in[px * ldi + c]) : 0.f);
        } else {
            v = (pos >= 1 ? to_f32<T>(in[(px - step) * ldi + c]) : 0.f) + (pos == n - 1 ? to_f32<T>(in[px * ldi + c]) : 0.f);
        }
        out[px * ldo + c] = from_f32<T>(v);
    }
}

// A remap given as an index table, the same for every image of the batch: dst[(b, i), :] = sum over q < kmax of src[(b, idx[i * kmax + q]), :]
// (rows of `width` contiguous elements; idx < 0: nothing).  kmax = 1 is a gather -- zero padding, circular padding, rolls, window partitions,
// region rearranges, overlapping-window im2col, per-channel shifts (width = 1) -- and the table of the INVERSE relation (every source's list of
// readers, kmax = the largest multiplicity) is its adjoint, again as a gather: no atomics, a fixed summation order.  The tables are built once
// per shape by running the reference's own index arithmetic (torch.roll / F.pad / view / permute) on a tensor of positions.
template <typename T> __global__ void __launch_bounds__(256) index_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ idx, int batch,
                                                                                 int64_t n_out, int64_t n_in, int width, int kmax) {
    const int64_t total = (int64_t)batch * n_out * width;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int e = (int)(t % width);
        const int64_t r = t / width;
        const int64_t i = r % n_out, b = r / n_out;
        float acc = 0.f;
        for (int q = 0; q < kmax; ++q) {
            const int j = idx[i * kmax + q];
            if (j >= 0) acc += to_f32<T>(src[(b * n_in + j) * width + e]);
        }
        dst[t] = from_f32<T>(acc);
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

extern "C" int mlpk_ew_cols(int dtype, int mode, const void* a, int64_t lda, const void* b, int64_t ldb, const float* g, const float* h, const float* k,
                            void* out, int64_t ldo, int64_t rows, int cols, int period, void* stream) {
    using namespace mlpk;
    if (!a || !out) return MLPK_ENULL;
    if (mode < 0 || mode > 5) return MLPK_EMODE;
    if (((mode == 1 || mode == 2 || mode == 4) && !b) || ((mode == 3 || mode == 5) && !g) || (mode == 4 && (!g || !h || !k))) return MLPK_ENULL;
    if (rows <= 0 || cols <= 0 || lda < cols || ldo < cols || (b && ldb < cols) || ((mode == 3 || mode == 5) && period <= 0)) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned gr = ew_grid(rows * (int64_t)cols);
#define EW(TT, MM) hipLaunchKernelGGL((ew_cols_kernel<TT, MM>), dim3(gr), dim3(256), 0, s, (const TT*)a, lda, (const TT*)b, ldb, g, h, k, (TT*)out, ldo, rows, cols, period)
#define EWM(TT)                  \
    switch (mode) {              \
        case 0: EW(TT, 0); break; \
        case 1: EW(TT, 1); break; \
        case 2: EW(TT, 2); break; \
        case 3: EW(TT, 3); break; \
        case 4: EW(TT, 4); break; \
        default: EW(TT, 5); break; \
    }
    BW_DISPATCH(dtype, EWM(float), EWM(f16_t), EWM(bf16_t))
#undef EWM
#undef EW
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_col_dot(int dtype, const void* x, int64_t ldx, const void* y, int64_t ldy, int64_t rows, int cols, float* out, void* stream) {
    using namespace mlpk;
    if (!x || !y || !out) return MLPK_ENULL;
    if (rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = (unsigned)((cols + 63) / 64);
#define CD(TT) hipLaunchKernelGGL((col_dot_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)x, ldx, (const TT*)y, ldy, rows, cols, out)
    BW_DISPATCH(dtype, CD(float), CD(f16_t), CD(bf16_t))
#undef CD
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_group_norm_backward(int dtype, const void* xh, const void* g, const float* rstd, void* dx, int groups, int64_t glen, void* stream) {
    using namespace mlpk;
    if (!xh || !g || !rstd || !dx) return MLPK_ENULL;
    if (groups <= 0 || glen <= 0) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define GB(TT) hipLaunchKernelGGL((group_norm_bwd_kernel<TT>), dim3((unsigned)groups), dim3(1024), 0, s, (const TT*)xh, (const TT*)g, rstd, (TT*)dx, glen)
    BW_DISPATCH(dtype, GB(float), GB(f16_t), GB(bf16_t))
#undef GB
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_shift_nhwc_backward(int dtype, const void* grad_out, void* grad_in, int N, int H, int W, int C, int kernel_size, int dim, void* stream) {
    using namespace mlpk;
    if (!grad_out || !grad_in) return MLPK_ENULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return MLPK_ESHAPE;
    if (kernel_size < 3 || !(kernel_size & 1)) return MLPK_ESHAPE;
    if (dim != 2 && dim != 3) return MLPK_EMODE;
    if (grad_out == grad_in) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid((int64_t)N * H * W * C);
#define SB(TT) hipLaunchKernelGGL((shift_nhwc_bwd_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)grad_out, (TT*)grad_in, N, H, W, C, kernel_size, dim)
    BW_DISPATCH(dtype, SB(float), SB(f16_t), SB(bf16_t))
#undef SB
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_patch_rows_nhwc(int dtype, int dir, int order, const void* src, void* dst, int B, int H, int W, int C, int ph, int pw, void* stream) {
    using namespace mlpk;
    if (!src || !dst) return MLPK_ENULL;
    if ((dir != 0 && dir != 1) || (order != 0 && order != 1)) return MLPK_EMODE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || H % ph || W % pw || src == dst) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid((int64_t)B * H * W * C);
#define PR(TT) hipLaunchKernelGGL((patch_rows_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)src, (TT*)dst, B, H, W, C, ph, pw, order, dir)
    BW_DISPATCH(dtype, PR(float), PR(f16_t), PR(bf16_t))
#undef PR
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_merge2x2_nhwc(int dtype, int dir, const void* src, void* dst, int B, int H, int W, int C, void* stream) {
    return mlpk_patch_rows_nhwc(dtype, dir, 1, src, dst, B, H, W, C, 2, 2, stream);
}

extern "C" int mlpk_dwconv_plain_nhwc(int dtype, int adjoint, const void* in, void* out, int B, int H, int W, int C, int k, const float* w, const float* bias,
                                      void* stream) {
    using namespace mlpk;
    if (!in || !out || !w) return MLPK_ENULL;
    if (adjoint != 0 && adjoint != 1) return MLPK_EMODE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k < 1 || k > 13 || in == out) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid((int64_t)B * H * W * C);
#define DP(TT) hipLaunchKernelGGL((dwconv_plain_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)in, (TT*)out, w, bias, B, H, W, C, k, adjoint)
    BW_DISPATCH(dtype, DP(float), DP(f16_t), DP(bf16_t))
#undef DP
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_dwconv_wgrad_nhwc(int dtype, const void* x, const void* dy, float* dw, int B, int H, int W, int C, int k, void* stream) {
    using namespace mlpk;
    if (!x || !dy || !dw) return MLPK_ENULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k < 1 || k > 13) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((C + 63) / 64), (unsigned)(k * k));
#define DW(TT) hipLaunchKernelGGL((dwconv_wgrad_kernel<TT>), g, dim3(256), 0, s, (const TT*)x, (const TT*)dy, dw, B, H, W, C, k)
    BW_DISPATCH(dtype, DW(float), DW(f16_t), DW(bf16_t))
#undef DW
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_col_dot_seg(int dtype, const void* x, int64_t ldx, const void* y, int64_t ldy, int segments, int64_t seg_rows, int cols, float* out,
                                void* stream) {
    using namespace mlpk;
    if (!x || !y || !out) return MLPK_ENULL;
    if (segments <= 0 || segments > 65535 || seg_rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((cols + 63) / 64), (unsigned)segments);
#define CDS(TT) hipLaunchKernelGGL((col_dot_kernel<TT>), g, dim3(256), 0, s, (const TT*)x, ldx, (const TT*)y, ldy, seg_rows, cols, out)
    BW_DISPATCH(dtype, CDS(float), CDS(f16_t), CDS(bf16_t))
#undef CDS
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_split_softmax_backward(const float* bar, const float* dbar, float* dhat, int B, int C, void* stream) {
    using namespace mlpk;
    if (!bar || !dbar || !dhat) return MLPK_ENULL;
    if (B <= 0 || C <= 0) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(split_softmax_bwd_kernel, dim3(ew_grid((int64_t)B * C)), dim3(256), 0, s, bar, dbar, dhat, B, C);
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_s2_shift2(int dtype, int which, int mode, int adjoint, const void* in, int64_t ldi, void* out, int64_t ldo, int B, int D1, int D2, int C,
                              void* stream) {
    using namespace mlpk;
    if (!in || !out) return MLPK_ENULL;
    if ((which != 1 && which != 2) || (mode != 0 && mode != 1) || (adjoint != 0 && adjoint != 1)) return MLPK_EMODE;
    if (B <= 0 || D1 <= 0 || D2 <= 0 || C <= 0 || ldi < C || ldo < C || in == out) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid((int64_t)B * D1 * D2 * C);
#define S2(TT) hipLaunchKernelGGL((s2_shift2_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)in, ldi, (TT*)out, ldo, B, D1, D2, C, which, mode, adjoint)
    BW_DISPATCH(dtype, S2(float), S2(f16_t), S2(bf16_t))
#undef S2
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_index_gather(int dtype, const void* src, void* dst, const int* idx, int batch, int64_t n_out, int64_t n_in, int width, int kmax, void* stream) {
    using namespace mlpk;
    if (!src || !dst || !idx) return MLPK_ENULL;
    if (batch <= 0 || n_out <= 0 || n_in <= 0 || width <= 0 || kmax <= 0 || n_in > 0x7fffffffLL || src == dst) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ew_grid((int64_t)batch * n_out * width);
#define IG(TT) hipLaunchKernelGGL((index_gather_kernel<TT>), dim3(g), dim3(256), 0, s, (const TT*)src, (TT*)dst, idx, batch, n_out, n_in, width, kmax)
    BW_DISPATCH(dtype, IG(float), IG(f16_t), IG(bf16_t))
#undef IG
    MLPK_LAUNCH_CHECK();
    return 0;
}
