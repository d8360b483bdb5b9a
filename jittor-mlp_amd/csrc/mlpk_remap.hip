// Index remaps and the reductions around them: AS-MLP axial shift (the reference's one native
// op), S2-MLP spatial shifts, split attention (ViP / S2-MLPv2) and the ConvMixer depthwise half.
// All channel-last kernels put the 64 lanes of a wave on 64 consecutive channels, so every
// global access is a contiguous 128/256-byte run whatever the spatial gather does; none of the
// shifts is ever materialised as a tensor except where the reference's API returns one.
#include "mlpk_common.h"
#include <cstdlib>

namespace mlpk {

// ================================ AS-MLP axial shift ================================
// shift_cuda.py:44-72: group = ceil(C/k), g = c / group, s = k/2 - g, zero fill.
// sign = +1: the forward gather; sign = -1: its adjoint, shift_backward_grad_input_kernel (shift_cuda.py:75-103: bottom_diff[h] = top_diff[h - s])
template <typename T>
__global__ void __launch_bounds__(256) shift_nchw_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int C,
                                                         int H, int W, int ksz, int dim, int sign) {
    const int64_t total = (int64_t)N * C * H * W;
    const int group = (C + ksz - 1) / ksz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int w = (int)(idx % W);
        const int h = (int)((idx / W) % H);
        const int c = (int)((idx / ((int64_t)W * H)) % C);
        const int s = sign * (ksz / 2 - c / group);
        T v = from_f32<T>(0.f);
        if (dim == 2) {
            if (h + s >= 0 && h + s < H) v = in[idx + (int64_t)s * W];
        } else {
            if (w + s >= 0 && w + s < W) v = in[idx + s];
        }
        out[idx] = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) shift_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H,
                                                         int W, int C, int ksz, int dim) {
    const int64_t total = (int64_t)N * H * W * C;
    const int group = (C + ksz - 1) / ksz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int w = (int)((idx / C) % W);
        const int h = (int)((idx / ((int64_t)C * W)) % H);
        const int s = ksz / 2 - c / group;
        T v = from_f32<T>(0.f);
        if (dim == 2) {
            if (h + s >= 0 && h + s < H) v = in[idx + (int64_t)s * W * C];
        } else {
            if (w + s >= 0 && w + s < W) v = in[idx + (int64_t)s * C];
        }
        out[idx] = v;
    }
}

// 16-bit types, C % 8 == 0: one thread moves 8 channels (16 bytes).  A vector that straddles two shift groups is
// assembled from the two source positions; 32-bit index arithmetic (one division pair per 16 bytes, not per element).
template <typename T>
__global__ void __launch_bounds__(256) shift_nhwc_vec_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H,
                                                             int W, int C, int ksz, int dim) {
    const int CV = C / 8;
    const unsigned total = (unsigned)N * H * W * CV;
    const int group = (C + ksz - 1) / ksz;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned px = idx / CV;                       // pixel (n, h, w)
        const int c = (int)(idx - px * CV) * 8;
        const int w = (int)(px % W);
        const int h = (int)((px / W) % H);
        const int g0 = c / group, g1 = (c + 7) / group;
        const int pos = dim == 2 ? h : w, lim = dim == 2 ? H : W;
        const int step = dim == 2 ? W * C : C;
        const T* src = in + (size_t)px * C + c;
        const int s0 = ksz / 2 - g0;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (pos + s0 >= 0 && pos + s0 < lim) v = *reinterpret_cast<const u32x4*>(src + (ptrdiff_t)s0 * step);
        if (g1 != g0) {
            const int s1 = ksz / 2 - g1;
            u32x4 v1 = {0u, 0u, 0u, 0u};
            if (pos + s1 >= 0 && pos + s1 < lim) v1 = *reinterpret_cast<const u32x4*>(src + (ptrdiff_t)s1 * step);
            T a[8], b[8];
            __builtin_memcpy(a, &v, 16);
            __builtin_memcpy(b, &v1, 16);
            const int first1 = g1 * group - c;              // first element of the vector that belongs to group g1
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = e >= first1 ? b[e] : a[e];
            __builtin_memcpy(&v, a, 16);
        }
        *reinterpret_cast<u32x4*>(out + (size_t)px * C + c) = v;
    }
}

// ================================ CycleFC sampling (CycleMLP) ================================
// cycle_mlp.py:104-131: torchvision's deform_conv2d with a 1 x 1 kernel and the fixed integer offsets of gen_offset --
// input channel c is read one cycle step away, d(c) = (c + k/2) % k - k/2, along W (kernel (1,k), `sfc_h`) or along H
// (kernel (k,1), `sfc_w`), zero outside the map (bilinear sampling at an integer point outside [0, size) is 0).  Both
// gathered copies of the channel-last activation are produced in one pass (the 1 x 1 convolutions that follow are GEMMs).
template <typename T>
__global__ void __launch_bounds__(256) cycle_shift_kernel(const T* __restrict__ in, T* __restrict__ out_h, T* __restrict__ out_w, int B,
                                                          int H, int W, int C, int kw, int kh, int ldi, int ldo) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int64_t px = idx / C;
        const int x = (int)(px % W);
        const int y = (int)((px / W) % H);
        if (out_h) {
            const int d = (c + kw / 2) % kw - kw / 2;
            out_h[px * ldo + c] = (x + d >= 0 && x + d < W) ? in[(px + d) * ldi + c] : from_f32<T>(0.f);
        }
        if (out_w) {
            const int d = (c + kh / 2) % kh - kh / 2;
            out_w[px * ldo + c] = (y + d >= 0 && y + d < H) ? in[(px + (int64_t)d * W) * ldi + c] : from_f32<T>(0.f);
        }
    }
}

// 16-bit types, C % 8 == 0: one thread produces 8 channels (16 bytes) of both outputs from the <= k neighbouring pixels'
// vectors at the same channel offset (all but the centre one are L1 / L2 hits: the neighbours' threads read them too).
// (round 5, mlpk_cycle_shift_ln: `in` is the UN-normalised tensor and the LayerNorm in front of the three branches -- cycle_mlp.py:195, per pixel
//  over C -- is applied to every element on the way, with the statistics of the pixel it comes FROM; what lies outside the map stays zero)
template <typename T, int K>
__global__ void __launch_bounds__(256) cycle_shift_vec_kernel(const T* __restrict__ in, T* __restrict__ out_h, T* __restrict__ out_w, int B,
                                                              int H, int W, int C, int ldi, int ldo, const float* __restrict__ mean = nullptr,
                                                              const float* __restrict__ rstd = nullptr, const float* __restrict__ gamma = nullptr,
                                                              const float* __restrict__ beta = nullptr) {
    const int CV = C / 8;
    const unsigned total = (unsigned)B * H * W * CV;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned px = idx / CV;
        const int c = (int)(idx - px * CV) * 8;
        const int x = (int)(px % W);
        const int y = (int)((px / W) % H);
        const T* src = in + (size_t)px * ldi + c;
        T oh[8], ow[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) oh[e] = ow[e] = from_f32<T>(0.f);
        float gg[8], bb[8];
        if (mean) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { gg[e] = gamma[c + e]; bb[e] = beta[c + e]; }
        }
#pragma unroll
        for (int d = -(K / 2); d <= K / 2; ++d) {
            // elements of this vector whose cycle step is d: (c + e + K/2) % K == d + K/2
            const int first = ((d + K / 2 - (c + K / 2) % K) % K + K) % K;
            if (first < 8) {
                if (out_h && x + d >= 0 && x + d < W) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (ptrdiff_t)d * ldi);
                    T a[8];
                    __builtin_memcpy(a, &v, 16);
                    if (mean) {
                        const float mu = mean[px + d], rs = rstd[px + d];
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = from_f32<T>(__builtin_fmaf((to_f32(a[e]) - mu) * rs, gg[e], bb[e]));   // (norm_apply's form)
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= first && (e - first) % K == 0) oh[e] = a[e];
                }
                if (out_w && y + d >= 0 && y + d < H) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (ptrdiff_t)d * W * ldi);
                    T a[8];
                    __builtin_memcpy(a, &v, 16);
                    if (mean) {
                        const float mu = mean[(ptrdiff_t)px + d * W], rs = rstd[(ptrdiff_t)px + d * W];
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = from_f32<T>(__builtin_fmaf((to_f32(a[e]) - mu) * rs, gg[e], bb[e]));
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= first && (e - first) % K == 0) ow[e] = a[e];
                }
            }
        }
        if (out_h) {
            u32x4 v;
            __builtin_memcpy(&v, oh, 16);
            *reinterpret_cast<u32x4*>(out_h + (size_t)px * ldo + c) = v;
        }
        if (out_w) {
            u32x4 v;
            __builtin_memcpy(&v, ow, 16);
            *reinterpret_cast<u32x4*>(out_w + (size_t)px * ldo + c) = v;
        }
    }
}

// ================================ S2 spatial shifts ================================
// Tensor (B, D1, D2, C).  Channel quarters as the slices of s2_mlp_v2.py:17-20.
// which = 1: spatial_shift1 -> (d1,+1),(d1,-1),(d2,+1),(d2,-1); which = 2: spatial_shift2 ->
// (d2,+1),(d2,-1),(d1,+1),(d1,-1).  "+1": y[i] = x[i-1] (border keeps x[0]); the reference's
// in-place assignment makes that y[i] = x[0] for all i (MLPK_SHIFT_S2_REF).  "-1": y[i] = x[min(i+1,n-1)].
__device__ __forceinline__ void s2_source(int which, int mode, int c, int C, int i1, int i2, int D1, int D2, int& j1,
                                          int& j2) {
    j1 = i1;
    j2 = i2;
    if (mode == MLPK_SHIFT_NONE || which == 0) return;
    int q;
    if (c < C / 4) q = 0;
    else if (c < C / 2) q = 1;
    else if (c < C * 3 / 4) q = 2;
    else q = 3;
    const bool plus = (q & 1) == 0;
    const bool axis1 = which == 1 ? (q < 2) : (q >= 2);
    if (axis1) {
        if (plus) j1 = mode == MLPK_SHIFT_S2_REF ? 0 : (i1 > 0 ? i1 - 1 : 0);
        else j1 = i1 + 1 < D1 ? i1 + 1 : D1 - 1;
    } else {
        if (plus) j2 = mode == MLPK_SHIFT_S2_REF ? 0 : (i2 > 0 ? i2 - 1 : 0);
        else j2 = i2 + 1 < D2 ? i2 + 1 : D2 - 1;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) s2_shift_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int D1,
                                                       int D2, int C, int ldi, int ldo, int mode) {
    const int64_t total = (int64_t)B * D1 * D2 * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int64_t px = idx / C;
        const int i2 = (int)(px % D2);
        const int i1 = (int)((px / D2) % D1);
        const int64_t b = px / ((int64_t)D1 * D2);
        int j1, j2;
        s2_source(1, mode, c, C, i1, i2, D1, D2, j1, j2);
        out[px * ldo + c] = in[((b * D1 + j1) * D2 + j2) * ldi + c];
    }
}

// ================================ split attention ================================
struct SplitArgs {
    const void* x0;
    const void* x1;
    const void* x2;
    int ld0, ld1, ld2;
    int B, D1, D2, C, mode;
};

// a[b,c] = sum over the three branches and all pixels; workgroup = (image, 64 channels),
// 4 pixel phases x 64 channel lanes, fp32 accumulation.
template <typename T>
__global__ void __launch_bounds__(256) split_sum_kernel(const SplitArgs p, float* __restrict__ a, const float scale) {
    __shared__ float red[4][64];
    const int tid = threadIdx.x;
    const int cl = tid & 63;
    const int ph = tid >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int b = blockIdx.x;
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);
    const T* x2 = reinterpret_cast<const T*>(p.x2);
    float s = 0.f;
    if (c < p.C) {
        const int npx = p.D1 * p.D2;
        for (int px = ph; px < npx; px += 4) {
            const int i1 = px / p.D2, i2 = px - i1 * p.D2;
            int j1, j2;
            s2_source(1, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
            s += to_f32(x0[(((int64_t)b * p.D1 + j1) * p.D2 + j2) * p.ld0 + c]);
            s2_source(2, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
            s += to_f32(x1[(((int64_t)b * p.D1 + j1) * p.D2 + j2) * p.ld1 + c]);
            s += to_f32(x2[(((int64_t)b * p.D1 + i1) * p.D2 + i2) * p.ld2 + c]);
        }
    }
    red[ph][cl] = s;
    __syncthreads();
    if (tid < 64 && c < p.C) a[(int64_t)b * p.C + c] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) * scale;
}

__global__ void __launch_bounds__(256) split_softmax_kernel(const float* __restrict__ hat, float* __restrict__ bar,
                                                            int B, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    const float* h = hat + (int64_t)b * 3 * C;
    const float h0 = h[c], h1 = h[C + c], h2 = h[2 * C + c];
    const float m = fmaxf(h0, fmaxf(h1, h2));
    const float e0 = __expf(h0 - m), e1 = __expf(h1 - m), e2 = __expf(h2 - m);
    const float inv = 1.0f / (e0 + e1 + e2);
    float* o = bar + (int64_t)b * 3 * C;
    o[c] = e0 * inv;
    o[C + c] = e1 * inv;
    o[2 * C + c] = e2 * inv;
}

template <typename T>
__global__ void __launch_bounds__(256) split_apply_kernel(const SplitArgs p, const float* __restrict__ bar,
                                                          T* __restrict__ out, int ldo) {
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);
    const T* x2 = reinterpret_cast<const T*>(p.x2);
    const int64_t total = (int64_t)p.B * p.D1 * p.D2 * p.C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % p.C);
        const int64_t px = idx / p.C;
        const int i2 = (int)(px % p.D2);
        const int i1 = (int)((px / p.D2) % p.D1);
        const int64_t b = px / ((int64_t)p.D1 * p.D2);
        const float* w = bar + b * 3 * p.C;
        int j1, j2;
        s2_source(1, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
        float v = w[c] * to_f32(x0[((b * p.D1 + j1) * p.D2 + j2) * p.ld0 + c]);
        s2_source(2, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
        v += w[p.C + c] * to_f32(x1[((b * p.D1 + j1) * p.D2 + j2) * p.ld1 + c]);
        v += w[2 * p.C + c] * to_f32(x2[px * p.ld2 + c]);
        out[px * ldo + c] = from_f32<T>(v);
    }
}


// ---- 16-byte (8-channel) versions, used when every channel quarter is a multiple of 8 so that one
// vector never straddles two shift groups: thread = 8 consecutive channels of one pixel ----
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
    T e[8];
    __builtin_memcpy(e, &t, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = to_f32(e[i]);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]) {
    T e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = from_f32<T>(v[i]);
    u32x4 t;
    __builtin_memcpy(&t, e, 16);
    *reinterpret_cast<u32x4*>(p) = t;
}

template <typename T>
__global__ void __launch_bounds__(256) split_sum_vec_kernel(const SplitArgs p, float* __restrict__ a, const float scale) {
    __shared__ float red[32][64 + 1];
    const int tid = threadIdx.x;
    const int cl = (tid & 7) * 8;
    const int ph = tid >> 3;                       // 32 pixel phases
    const int c = blockIdx.y * 64 + cl;
    const int b = blockIdx.x;
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);
    const T* x2 = reinterpret_cast<const T*>(p.x2);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (c < p.C) {
        const int npx = p.D1 * p.D2;
        for (int px = ph; px < npx; px += 32) {
            const int i1 = px / p.D2, i2 = px - i1 * p.D2;
            int j1, j2;
            float v[8];
            s2_source(1, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
            ld8<T>(x0 + (((int64_t)b * p.D1 + j1) * p.D2 + j2) * p.ld0 + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
            s2_source(2, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
            ld8<T>(x1 + (((int64_t)b * p.D1 + j1) * p.D2 + j2) * p.ld1 + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
            ld8<T>(x2 + (((int64_t)b * p.D1 + i1) * p.D2 + i2) * p.ld2 + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ph][cl + e] = s[e];
    __syncthreads();
    if (tid < 64) {
        const int cc = blockIdx.y * 64 + tid;
        if (cc < p.C) {
            float t = 0.f;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) t += red[i][tid];
            a[(int64_t)b * p.C + cc] = t * scale;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) split_apply_vec_kernel(const SplitArgs p, const float* __restrict__ bar,
                                                              T* __restrict__ out, int ldo) {
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);
    const T* x2 = reinterpret_cast<const T*>(p.x2);
    const int cv = p.C / 8;
    const int64_t total = (int64_t)p.B * p.D1 * p.D2 * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * 8;
        const int64_t px = idx / cv;
        const int i2 = (int)(px % p.D2);
        const int i1 = (int)((px / p.D2) % p.D1);
        const int64_t b = px / ((int64_t)p.D1 * p.D2);
        const float* w = bar + b * 3 * p.C + c;
        float w0[8], w1[8], w2[8], v0[8], v1[8], v2[8], o[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(w + e);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(w + p.C + e);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(w + 2 * p.C + e);
            w0[e] = t0.x; w0[e + 1] = t0.y; w0[e + 2] = t0.z; w0[e + 3] = t0.w;
            w1[e] = t1.x; w1[e + 1] = t1.y; w1[e + 2] = t1.z; w1[e + 3] = t1.w;
            w2[e] = t2.x; w2[e + 1] = t2.y; w2[e + 2] = t2.z; w2[e + 3] = t2.w;
        }
        int j1, j2;
        s2_source(1, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
        ld8<T>(x0 + ((b * p.D1 + j1) * p.D2 + j2) * p.ld0 + c, v0);
        s2_source(2, p.mode, c, p.C, i1, i2, p.D1, p.D2, j1, j2);
        ld8<T>(x1 + ((b * p.D1 + j1) * p.D2 + j2) * p.ld1 + c, v1);
        ld8<T>(x2 + px * p.ld2 + c, v2);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = w0[e] * v0[e] + w1[e] * v1[e] + w2[e] * v2[e];
        st8<T>(out + px * ldo + c, o);
    }
}

// ================================ ViP: split attention straight on the permuted branch outputs ================================
// vip.py:69-76: the H- and W-branch Linears run on rearranged rows, so their outputs come out as
//   zh[((b*W + w)*G + g)*ldh + h*seg + q] = xH[b,h,w,g*seg + q]      zw[((b*H + h)*G + g)*ldw + w*seg + q] = xW[b,h,w,g*seg + q]
// and the reference rearranges them back ('b w c (h s) -> b h w (c s)') before SplitAttention.  Here the inverse rearrange is the
// ADDRESS of the load: the reduction and the weighted sum of SplitAttention read both tensors where they lie (with seg % 4 == 0
// a lane's 8 channels are two 8-byte pieces of a permuted row), so neither xH nor xW is ever written in (B,H,W,C) order.
template <typename T>
__device__ __forceinline__ void vip_ld8(const T* __restrict__ z, const int ldz, const int64_t row0, const int G, const int seg, const int pos,
                                        const int c, float (&v)[8]) {
    // row0 = (b*W + w) * G (H branch, pos = h) or (b*H + h) * G (W branch, pos = w); channels c .. c+7, c % 8 == 0
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int cc = c + 4 * part;
        const int g = cc / seg;
        const int q = cc - g * seg;                                   // multiple of 4: the 4 channels stay inside group g
        const u32x2 t = *reinterpret_cast<const u32x2*>(z + (row0 + g) * ldz + pos * seg + q);
        T e[4];
        __builtin_memcpy(e, &t, 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * part + i] = to_f32(e[i]);
    }
}

struct VipSplitArgs {
    const void* zh;
    const void* zw;
    const void* xc;
    int ldh, ldw, ldc;
    int B, H, W, C, seg;
};

// out[b,h,w,c] = bar[b,0,c] xH + bar[b,1,c] xW + bar[b,2,c] xC
template <typename T>
__global__ void __launch_bounds__(256) vip_split_apply_kernel(const VipSplitArgs p, const float* __restrict__ bar, T* __restrict__ out, int ldo) {
    const T* zh = reinterpret_cast<const T*>(p.zh);
    const T* zw = reinterpret_cast<const T*>(p.zw);
    const T* xc = reinterpret_cast<const T*>(p.xc);
    const int cv = p.C / 8;
    const int G = p.C / p.seg;
    const int64_t total = (int64_t)p.B * p.H * p.W * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * 8;
        const int64_t px = idx / cv;
        const int w = (int)(px % p.W);
        const int h = (int)((px / p.W) % p.H);
        const int64_t b = px / ((int64_t)p.H * p.W);
        const float* wt = bar + b * 3 * p.C + c;
        float w0[8], w1[8], w2[8], v0[8], v1[8], v2[8], o[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(wt + e);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(wt + p.C + e);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(wt + 2 * p.C + e);
            w0[e] = t0.x; w0[e + 1] = t0.y; w0[e + 2] = t0.z; w0[e + 3] = t0.w;
            w1[e] = t1.x; w1[e + 1] = t1.y; w1[e + 2] = t1.z; w1[e + 3] = t1.w;
            w2[e] = t2.x; w2[e + 1] = t2.y; w2[e + 2] = t2.z; w2[e + 3] = t2.w;
        }
        vip_ld8<T>(zh, p.ldh, (b * p.W + w) * G, G, p.seg, h, c, v0);
        vip_ld8<T>(zw, p.ldw, (b * p.H + h) * G, G, p.seg, w, c, v1);
        ld8<T>(xc + px * p.ldc + c, v2);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = w0[e] * v0[e] + w1[e] * v1[e] + w2[e] * v2[e];
        st8<T>(out + px * ldo + c, o);
    }
}

// The same weighted sum on 8 x 8 pixel tiles: read pixel by pixel, a lane's piece of zh is 8-16 bytes of a row that the next
// lane does not touch (row stride ldh), i.e. a quarter of every 64-byte sector (measured 2.7 TB/s).  A tile of 8 h x 8 w pixels
// needs, of the h branch, the columns (h0..h0+7, all q) of the rows (b, w, g) for 8 w: runs of 8 * seg contiguous elements (192
// bytes at seg = 12) -- and of the w branch the mirror image.  Both sets of runs are staged in LDS as they lie (16-byte copies),
// the (pixel, 8 channels) items then pick their 8-byte pieces out of LDS; xc, the weights and the output are contiguous anyway.
template <typename T>
__global__ void __launch_bounds__(512) vip_split_apply_tile_kernel(const VipSplitArgs p, const float* __restrict__ bar, T* __restrict__ out, int ldo, int parts) {
    // round 6: `parts` workgroups share a pixel tile, each taking G / parts channel groups -- the staged tiles of one workgroup then fill a
    // third of the LDS instead of two thirds, so that three workgroups (not one) live on a CU and one's load phase runs under another's
    // weighted-sum phase (the kernel has no pipeline of its own); the runs read from zh / zw stay 8 x seg elements long.  Same arithmetic.
    static_assert(sizeof(T) == 2, "16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) char vsmem[];
    const int tid = threadIdx.x;
    const int G = p.C / p.seg;
    const int Gs = G / parts;                                // channel groups of this workgroup
    const int Cs = Gs * p.seg;
    const int run = 8 * p.seg;                              // elements per run (one (w, g) row x 8 h, or one (h, g) row x 8 w)
    T* const zh_t = reinterpret_cast<T*>(vsmem);            // [8 w][Gs][8 h * seg]
    T* const zw_t = zh_t + 8 * Gs * run;                    // [8 h][Gs][8 w * seg]
    float* const wt = reinterpret_cast<float*>(zw_t + 8 * Gs * run);   // bar[b] of this workgroup's channels: 3 x Cs
    const int tw = p.W >> 3, th = p.H >> 3;
    const int part = blockIdx.x % parts;
    const int bt = blockIdx.x / parts;
    const int b = bt / (th * tw);
    const int t = bt - b * (th * tw);
    const int h0 = (t / tw) * 8, w0 = (t % tw) * 8;
    const int g0 = part * Gs, c0 = part * Cs;
    const T* zh = reinterpret_cast<const T*>(p.zh);
    const T* zw = reinterpret_cast<const T*>(p.zw);
    const T* xc = reinterpret_cast<const T*>(p.xc);
    const int ppr = p.seg;                                  // 16-byte pieces per run: 8 * seg * 2 / 16
    for (int idx = tid; idx < 8 * Gs * ppr; idx += 512) {
        const int r = idx / ppr, pc = idx - r * ppr;        // r = l * Gs + g
        const int l = r / Gs, g = r - l * Gs;
        *reinterpret_cast<u32x4*>(zh_t + (size_t)r * run + pc * 8) =
            *reinterpret_cast<const u32x4*>(zh + (((int64_t)b * p.W + w0 + l) * G + g0 + g) * p.ldh + h0 * p.seg + pc * 8);
        *reinterpret_cast<u32x4*>(zw_t + (size_t)r * run + pc * 8) =
            *reinterpret_cast<const u32x4*>(zw + (((int64_t)b * p.H + h0 + l) * G + g0 + g) * p.ldw + w0 * p.seg + pc * 8);
    }
    for (int i = tid; i < 3 * Cs; i += 512) {
        const int k = i / Cs;
        wt[i] = bar[(int64_t)b * 3 * p.C + k * p.C + c0 + (i - k * Cs)];
    }
    __syncthreads();
    const int cv = Cs / 8;
    for (int idx = tid; idx < 64 * cv; idx += 512) {
        const int pl = idx / cv;
        const int c = (idx - pl * cv) * 8;                  // channel inside this workgroup's range
        const int hl = pl >> 3, wl = pl & 7;
        const int64_t px = ((int64_t)b * p.H + h0 + hl) * p.W + w0 + wl;
        float v0[8], v1[8], v2[8], o[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cc = c + 4 * half;
            const int g = cc / p.seg;
            const int q = cc - g * p.seg;                   // multiple of 4: the 4 channels stay inside group g
            const u32x2 a0 = *reinterpret_cast<const u32x2*>(zh_t + (size_t)(wl * Gs + g) * run + hl * p.seg + q);
            const u32x2 a1 = *reinterpret_cast<const u32x2*>(zw_t + (size_t)(hl * Gs + g) * run + wl * p.seg + q);
            T e0[4], e1[4];
            __builtin_memcpy(e0, &a0, 8);
            __builtin_memcpy(e1, &a1, 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v0[4 * half + i] = to_f32(e0[i]); v1[4 * half + i] = to_f32(e1[i]); }
        }
        ld8<T>(xc + px * p.ldc + c0 + c, v2);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = wt[c + e] * v0[e] + wt[Cs + c + e] * v1[e] + wt[2 * Cs + c + e] * v2[e];
        st8<T>(out + px * ldo + c0 + c, o);
    }
}

// ================================ AS-MLP: GroupNorm + GELU + both axial shifts in one pass ================================
// as_mlp.py:64-66,84-95: t = gelu(GroupNorm(1,C)(conv1(x))) is only ever read through the two axial shifts (conv2_1 takes the
// W-shifted, conv2_2 the H-shifted copy), so t itself is never stored: this kernel reads the conv1 output u once per output
// and writes  out_w[n,h,w,c] = t[n,h,w+s,c],  out_h[n,h,w,c] = t[n,h+s,w,c]  (s = k/2 - c / ceil(C/k), zero outside the map,
// utils/shift_cuda.py:49-69) with  t = act((u - mean[n]) * rstd[n] * gamma[c] + beta[c])  computed on the fly.  One thread
// = 8 channels (16 bytes) of both outputs; a vector that straddles two shift groups is assembled from two source pixels.
template <typename T>
__global__ void __launch_bounds__(256) norm_shift_vec_kernel(const T* __restrict__ in, T* __restrict__ out_w, T* __restrict__ out_h, int N, int H,
                                                             int W, int C, int ksz, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act) {
    const int CV = C / 8;
    const unsigned total = (unsigned)N * H * W * CV;
    const int group = (C + ksz - 1) / ksz;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned px = idx / CV;                       // pixel (n, h, w)
        const int c = (int)(idx - px * CV) * 8;
        const int w = (int)(px % W);
        const int h = (int)((px / W) % H);
        const int n = (int)(px / ((unsigned)W * H));
        const float mu = mean[n], rs = rstd[n];
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = gamma ? gamma[c + e] : 1.f;
            sc[e] = rs * g;
            sh[e] = (beta ? beta[c + e] : 0.f) - mu * rs * g;
        }
        const int g0 = c / group, g1 = (c + 7) / group;
        const int first1 = g1 != g0 ? g1 * group - c : 8;   // first element of the vector that belongs to group g1
        const T* src = in + (size_t)px * C + c;
        // The four source vectors (two axes x the two groups a vector can straddle) are loaded UNCONDITIONALLY -- a source outside the
        // map reads this pixel instead and is zeroed afterwards.  Behind `if (inside)` hipcc puts an s_waitcnt vmcnt(0) in front of
        // every load: four memory latencies in a row per vector (the depthwise kernel's disease, profiles/r04_dwconv_variants.txt).
        const int s0 = ksz / 2 - g0, s1 = ksz / 2 - g1;
        const bool okw0 = (unsigned)(w + s0) < (unsigned)W, okw1 = (unsigned)(w + s1) < (unsigned)W;
        const bool okh0 = (unsigned)(h + s0) < (unsigned)H, okh1 = (unsigned)(h + s1) < (unsigned)H;
        float vw0[8], vw1[8], vh0[8], vh1[8];
        ld8<T>(src + (okw0 ? (ptrdiff_t)s0 * C : 0), vw0);
        ld8<T>(src + (okw1 ? (ptrdiff_t)s1 * C : 0), vw1);
        ld8<T>(src + (okh0 ? (ptrdiff_t)s0 * W * C : 0), vh0);
        ld8<T>(src + (okh1 ? (ptrdiff_t)s1 * W * C : 0), vh1);
        float ow[8], oh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool lo = e < first1;                       // element of group g0
            float tw = (lo ? vw0[e] : vw1[e]) * sc[e] + sh[e];
            float th = (lo ? vh0[e] : vh1[e]) * sc[e] + sh[e];
            if (act == MLPK_ACT_GELU) { tw = gelu_t<T>(tw); th = gelu_t<T>(th); }
            ow[e] = (lo ? okw0 : okw1) ? tw : 0.f;
            oh[e] = (lo ? okh0 : okh1) ? th : 0.f;
        }
        st8<T>(out_w + (size_t)px * C + c, ow);
        st8<T>(out_h + (size_t)px * C + c, oh);
    }
}

// The same operation for maps that FIT THE LDS (AS-MLP stages 3 and 4: 14 x 14 x 384 and 7 x 7 x 768 16-bit values = 147 / 74 KiB): one
// workgroup per image.  t = act(norm(u)) is evaluated ONCE per element into an LDS copy of the image (the gather form above evaluates
// it for each of its two outputs, and spends as many instructions on dividing its flat index into (n, h, w, c)); both outputs are then
// 16-byte reads of that copy at the shifted pixel -- zero outside the map, two reads where a vector straddles two shift groups.
template <typename T, bool GELU>
__global__ void __launch_bounds__(512) norm_shift_img_kernel(const T* __restrict__ in, T* __restrict__ out_w, T* __restrict__ out_h, int H, int W, int C,
                                                             int ksz, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) char smem_img[];
    u32x4* const img = reinterpret_cast<u32x4*>(smem_img);
    const int n = blockIdx.x, tid = threadIdx.x;
    const int CV = C / 8, HW = H * W, NV = HW * CV;
    const int group = (C + ksz - 1) / ksz;
    const float mu = mean[n], rs = rstd[n];
    const float inv_cv = 1.0f / (float)CV, inv_w = 1.0f / (float)W;
    const T* src = in + (size_t)n * HW * C;
    // per-channel scale and shift ONCE per workgroup, behind the image in LDS.  (Until round 6 every vector re-read its eight gamma / beta
    // values through sixteen scalar-branch-guarded dword loads in the same loop iteration as its own 16-byte load: one memory latency after
    // the other, 18 iterations per thread -- 72 us for 115 MB moved, profiles/r05_asmlp_t_kernel_stats_v4.csv.)  The expressions stay those
    // of norm_shift_vec_kernel, term for term: the two forms -- and mlpk_as_conv2, which is bit-equal to that one -- must round alike
    float* const scs = reinterpret_cast<float*>(smem_img + (size_t)HW * C * sizeof(T));
    for (int c = tid; c < C; c += 512) {
        const float g = gamma ? gamma[c] : 1.f;
        scs[c] = rs * g;
        scs[C + c] = (beta ? beta[c] : 0.f) - mu * rs * g;
    }
    __syncthreads();
    // (index arithmetic by float reciprocals: exact for these ranges -- NV < 2^17, the quotients are at least 0.5 / CV away from an integer)
    constexpr int U = 6;                                    // vectors of a round: all their loads are in flight before the first is used
    for (int base = 0; base < NV; base += 512 * U) {
        u32x4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = base + u * 512 + tid;
            raw[u] = *reinterpret_cast<const u32x4*>(src + (size_t)(v < NV ? v : NV - 1) * 8);      // (clamped, not predicated: no branch between the loads)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = base + u * 512 + tid;
            if (v >= NV) continue;
            const int px = (int)(((float)v + 0.5f) * inv_cv);
            const int c = (v - px * CV) * 8;
            T xe[8];
            __builtin_memcpy(xe, &raw[u], 16);
            const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scs + c), sc1 = *reinterpret_cast<const f32x4*>(scs + c + 4);
            const f32x4 sh0 = *reinterpret_cast<const f32x4*>(scs + C + c), sh1 = *reinterpret_cast<const f32x4*>(scs + C + c + 4);
            const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
            const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
            T r[8];
            f32x2 g2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g2[e] = f32x2{__builtin_fmaf(to_f32(xe[2 * e]), sc[2 * e], sh[2 * e]), __builtin_fmaf(to_f32(xe[2 * e + 1]), sc[2 * e + 1], sh[2 * e + 1])};
            // (GELU is a template parameter: as a run-time flag hipcc branched around every element's evaluation; four pairs abreast give
            //  the same bits as gelu16_f per element -- mlpk_as_conv2 stages its band the same way)
            if (GELU) gelu_pk_n<T, 4>(g2);
#pragma unroll
            for (int e = 0; e < 4; ++e) { r[2 * e] = from_f32<T>(g2[e].x); r[2 * e + 1] = from_f32<T>(g2[e].y); }
            u32x4 pk;
            __builtin_memcpy(&pk, r, 16);
            img[v] = pk;
        }
    }
    __syncthreads();
    T* const dw = out_w + (size_t)n * HW * C;
    T* const dh = out_h + (size_t)n * HW * C;
    for (int v = tid; v < NV; v += 512) {
        const int px = (int)(((float)v + 0.5f) * inv_cv);
        const int cv = v - px * CV, c = cv * 8;
        const int h = (int)(((float)px + 0.5f) * inv_w);
        const int w = px - h * W;
        const int ga = c / group, gb = (c + 7) / group;
        const int first1 = gb != ga ? gb * group - c : 8;      // first element of the vector that belongs to group gb
        const int s0 = ksz / 2 - ga, s1 = ksz / 2 - gb;
        const bool okw0 = (unsigned)(w + s0) < (unsigned)W, okw1 = (unsigned)(w + s1) < (unsigned)W;
        const bool okh0 = (unsigned)(h + s0) < (unsigned)H, okh1 = (unsigned)(h + s1) < (unsigned)H;
        const u32x4 z = {0u, 0u, 0u, 0u};
        // element e < first1 from the group-ga source, the others from the group-gb source: a 16-bit lane mask over the two vectors
        const u32x4 a_w = okw0 ? img[v + s0 * CV] : z, b_w = okw1 ? img[v + s1 * CV] : z;
        const u32x4 a_h = okh0 ? img[v + s0 * W * CV] : z, b_h = okh1 ? img[v + s1 * W * CV] : z;
        unsigned m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (2 * q < first1 ? 0x0000ffffu : 0u) | (2 * q + 1 < first1 ? 0xffff0000u : 0u);
        const u32x4 ow = {(a_w.x & m[0]) | (b_w.x & ~m[0]), (a_w.y & m[1]) | (b_w.y & ~m[1]), (a_w.z & m[2]) | (b_w.z & ~m[2]), (a_w.w & m[3]) | (b_w.w & ~m[3])};
        const u32x4 oh = {(a_h.x & m[0]) | (b_h.x & ~m[0]), (a_h.y & m[1]) | (b_h.y & ~m[1]), (a_h.z & m[2]) | (b_h.z & ~m[2]), (a_h.w & m[3]) | (b_h.w & ~m[3])};
        *reinterpret_cast<u32x4*>(dw + (size_t)v * 8) = ow;
        *reinterpret_cast<u32x4*>(dh + (size_t)v * 8) = oh;
    }
}

// ================================ ConvMixer depthwise half ================================
// thread = (image, row y, strip of 8 outputs along x, channel c); the k*k taps of channel c and a
// sliding (8 + k - 1)-wide input window live in registers, lanes run along c (coalesced).
template <typename T, int KS>
__global__ void __launch_bounds__(256) dwconv_nhwc_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H,
                                                          int W, int C, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ bns,
                                                          const float* __restrict__ bnh) {
    constexpr int P = (KS - 1) / 2;
    constexpr int STRIP = 8;
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int strips = (W + STRIP - 1) / STRIP;
    const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);    // (b, y, strip)
    const int64_t njobs = (int64_t)B * H * strips;
    if (c >= C || job >= njobs) return;
    const int st = (int)(job % strips);
    const int y = (int)((job / strips) % H);
    const int64_t b = job / ((int64_t)strips * H);
    const int x0 = st * STRIP;
    float wt[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wt[t] = w[(int64_t)t * C + c];
    float acc[STRIP];
    const float bs = bias ? bias[c] : 0.f;
#pragma unroll
    for (int o = 0; o < STRIP; ++o) acc[o] = bs;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
        const int yy = y + dy - P;
        if (yy < 0 || yy >= H) continue;
        float win[STRIP + KS - 1];
#pragma unroll
        for (int t = 0; t < STRIP + KS - 1; ++t) {
            const int xx = x0 + t - P;
            win[t] = (xx >= 0 && xx < W) ? to_f32(x[(((b * H + yy) * W) + xx) * C + c]) : 0.f;
        }
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
            for (int o = 0; o < STRIP; ++o) acc[o] = __builtin_fmaf(wt[dy * KS + dx], win[o + dx], acc[o]);
    }
    const float sc = bns ? bns[c] : 1.f, sh = bnh ? bnh[c] : 0.f;
#pragma unroll
    for (int o = 0; o < STRIP; ++o) {
        const int xx = x0 + o;
        if (xx >= W) break;
        const int64_t i = (((b * H + y) * W) + xx) * C + c;
        out[i] = from_f32<T>(to_f32(x[i]) + gelu_f(acc[o]) * sc + sh);
    }
}

}  // namespace mlpk

using namespace mlpk;

#define DISPATCH_DTYPE(dt, ...)                                              \
    switch (dt) {                                                            \
        case MLPK_F32: { typedef float T; __VA_ARGS__; break; }              \
        case MLPK_F16: { typedef f16_t T; __VA_ARGS__; break; }              \
        case MLPK_BF16: { typedef bf16_t T; __VA_ARGS__; break; }            \
        default: return MLPK_EDTYPE;                                         \
    }

static inline unsigned grid_for(int64_t total) {
    const int64_t g = (total + 255) / 256;
    return (unsigned)(g < 32768 ? g : 32768);
}

static int shift_check(int ksz, int dim) {
    // shift_cuda.py:167-168 / 184-185: odd kernel >= 3, dim in {2,3}
    if (ksz < 3 || (ksz & 1) == 0) return MLPK_ESHAPE;
    if (dim != 2 && dim != 3) return MLPK_EMODE;
    return 0;
}

static int shift_nchw_launch(int dtype, const void* in, void* out, int N, int C, int H, int W, int kernel_size, int dim, int sign,
                             void* stream) {
    if (!in || !out) return MLPK_ENULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return MLPK_ESHAPE;
    if (int e = shift_check(kernel_size, dim)) return e;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)N * C * H * W;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shift_nchw_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s,
                                             (const T*)in, (T*)out, N, C, H, W, kernel_size, dim, sign));
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_shift_nchw(int dtype, const void* in, void* out, int N, int C, int H, int W, int kernel_size,
                               int dim, void* stream) {
    return shift_nchw_launch(dtype, in, out, N, C, H, W, kernel_size, dim, 1, stream);
}

extern "C" int mlpk_shift_nchw_backward(int dtype, const void* grad_out, void* grad_in, int N, int C, int H, int W, int kernel_size,
                                        int dim, void* stream) {
    return shift_nchw_launch(dtype, grad_out, grad_in, N, C, H, W, kernel_size, dim, -1, stream);
}

extern "C" int mlpk_shift_nhwc(int dtype, const void* in, void* out, int N, int H, int W, int C, int kernel_size,
                               int dim, void* stream) {
    if (!in || !out) return MLPK_ENULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return MLPK_ESHAPE;
    if (int e = shift_check(kernel_size, dim)) return e;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)N * C * H * W;
    const int group = (C + kernel_size - 1) / kernel_size;
    if (dtype != MLPK_F32 && C % 8 == 0 && group >= 8 && total / 8 < 0x7fffffffLL && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        // (group >= 8: an 8-channel vector touches at most two shift groups)
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shift_nhwc_vec_kernel<T>), dim3(grid_for(total / 8)), dim3(256), 0, s,
                                                 (const T*)in, (T*)out, N, H, W, C, kernel_size, dim));
    } else {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shift_nhwc_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)in, (T*)out, N, H, W, C, kernel_size, dim));
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_norm_shift_nhwc(int dtype, const void* in, void* out_w, void* out_h, int N, int H, int W, int C, int kernel_size,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                    void* stream) {
    if (!in || !out_w || !out_h || !mean || !rstd) return MLPK_ENULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return MLPK_ESHAPE;
    if (kernel_size < 3 || !(kernel_size & 1)) return MLPK_ESHAPE;
    if (act != MLPK_ACT_NONE && act != MLPK_ACT_GELU) return MLPK_EMODE;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;     // 16-bit channel-last activations (fp32 keeps the unfused path)
    const int64_t total = (int64_t)N * C * H * W;
    const int group = (C + kernel_size - 1) / kernel_size;
    if (C % 8 || group < 8 || total / 8 >= 0x7fffffffLL) return MLPK_ESHAPE;    // (group >= 8: a vector touches at most two shift groups)
    if (((uintptr_t)in | (uintptr_t)out_w | (uintptr_t)out_h) & 15) return MLPK_EALIGN;
    if (in == out_w || in == out_h) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // maps that fit the LDS: one workgroup per image, the activation evaluated once per element (MLPK_NORM_SHIFT_IMG=0: the gather form, A/B aid)
    const size_t img_bytes = (size_t)H * W * C * 2;
    static const bool img_off = getenv("MLPK_NORM_SHIFT_IMG") && atoi(getenv("MLPK_NORM_SHIFT_IMG")) == 0;
    const size_t img_lds = img_bytes + (size_t)C * 8;                // + the per-channel scale / shift tables
    if (!img_off && img_lds <= 160 * 1024 && (size_t)H * W * (C / 8) < (1u << 17)) {
        hipError_t e = hipSuccess;
#define NSI_LAUNCH(TT, GG)                                                                                                              \
    {                                                                                                                                   \
        auto k = norm_shift_img_kernel<TT, GG>;                                                                                         \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds);           \
        if (e != hipSuccess) return (int)e;                                                                                             \
        hipLaunchKernelGGL(k, dim3(N), dim3(512), img_lds, s, (const TT*)in, (TT*)out_w, (TT*)out_h, H, W, C, kernel_size, mean, rstd, gamma, beta); \
    }
        const bool gelu = act == MLPK_ACT_GELU;
        if (dtype == MLPK_BF16) {
            if (gelu) NSI_LAUNCH(bf16_t, true) else NSI_LAUNCH(bf16_t, false)
        } else {
            if (gelu) NSI_LAUNCH(f16_t, true) else NSI_LAUNCH(f16_t, false)
        }
#undef NSI_LAUNCH
        MLPK_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == MLPK_BF16) {
        hipLaunchKernelGGL((norm_shift_vec_kernel<bf16_t>), dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out_w,
                           (bf16_t*)out_h, N, H, W, C, kernel_size, mean, rstd, gamma, beta, act);
    } else {
        hipLaunchKernelGGL((norm_shift_vec_kernel<f16_t>), dim3(grid_for(total / 8)), dim3(256), 0, s, (const f16_t*)in, (f16_t*)out_w,
                           (f16_t*)out_h, N, H, W, C, kernel_size, mean, rstd, gamma, beta, act);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_cycle_shift_ln(int dtype, const void* in, const float* mean, const float* rstd, const float* gamma, const float* beta, void* out_h,
                                   void* out_w, int B, int H, int W, int C, int k, int ldi, int ldo, void* stream);

extern "C" int mlpk_cycle_shift(int dtype, const void* in, void* out_h, void* out_w, int B, int H, int W, int C, int k, int ldi,
                                int ldo, void* stream) {
    return mlpk_cycle_shift_ln(dtype, in, nullptr, nullptr, nullptr, nullptr, out_h, out_w, B, H, W, C, k, ldi, ldo, stream);
}

extern "C" int mlpk_cycle_shift_ln(int dtype, const void* in, const float* mean, const float* rstd, const float* gamma, const float* beta, void* out_h,
                                   void* out_w, int B, int H, int W, int C, int k, int ldi, int ldo, void* stream) {
    if (!in || (!out_h && !out_w)) return MLPK_ENULL;
    if ((mean != nullptr) != (rstd != nullptr) || (mean != nullptr) != (gamma != nullptr) || (mean != nullptr) != (beta != nullptr)) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ldi < C || ldo < C) return MLPK_ESHAPE;
    if (k < 1 || !(k & 1)) return MLPK_ESHAPE;
    if (in == out_h || in == out_w) return MLPK_ESHAPE;         // a gather cannot run in place
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * W * C;
    const bool vec = dtype != MLPK_F32 && C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && total / 8 < 0x7fffffffLL && (k == 3 || k == 5 || k == 7) &&
                     (((uintptr_t)in | (uintptr_t)out_h | (uintptr_t)out_w) & 15) == 0;
    if (vec) {
        const dim3 grid(grid_for(total / 8));
        if (k == 3) {
            DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cycle_shift_vec_kernel<T, 3>), grid, dim3(256), 0, s, (const T*)in, (T*)out_h, (T*)out_w, B, H, W, C, ldi, ldo, mean, rstd, gamma, beta));
        } else if (k == 5) {
            DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cycle_shift_vec_kernel<T, 5>), grid, dim3(256), 0, s, (const T*)in, (T*)out_h, (T*)out_w, B, H, W, C, ldi, ldo, mean, rstd, gamma, beta));
        } else {
            DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cycle_shift_vec_kernel<T, 7>), grid, dim3(256), 0, s, (const T*)in, (T*)out_h, (T*)out_w, B, H, W, C, ldi, ldo, mean, rstd, gamma, beta));
        }
    } else {
        if (mean) return MLPK_ESHAPE;                            // (the LayerNorm form runs on the 16-byte-vector kernel only: 16 bit, C % 8 == 0, k = 3 / 5 / 7)
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cycle_shift_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)in, (T*)out_h,
                                                 (T*)out_w, B, H, W, C, k, k, ldi, ldo));
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

// 8-channel vectors are usable when no vector straddles two S2 shift groups (quarters of C) and all
// three branch pointers / strides keep 16-byte alignment (16-bit dtypes; fp32 keeps the scalar kernels)
static bool split_vec_ok(const void* x0, const void* x1, const void* x2, int ld0, int ld1, int ld2, int C, int mode) {
    if (C % 8 || ld0 % 8 || ld1 % 8 || ld2 % 8) return false;
    if (((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)x2) & 15) return false;
    if (mode != MLPK_SHIFT_NONE && C % 32) return false;
    return true;
}

static int split_check(const void* x0, const void* x1, const void* x2, int ld0, int ld1, int ld2, int B, int H, int W,
                       int C, int mode) {
    if (!x0 || !x1 || !x2) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld0 < C || ld1 < C || ld2 < C) return MLPK_ESHAPE;
    if (mode < MLPK_SHIFT_NONE || mode > MLPK_SHIFT_S2_REF) return MLPK_EMODE;
    return 0;
}

extern "C" int mlpk_split_sum(int dtype, const void* x0, const void* x1, const void* x2, int ld0, int ld1, int ld2,
                              int B, int H, int W, int C, int shift_mode, float scale, float* a, void* stream) {
    if (int e = split_check(x0, x1, x2, ld0, ld1, ld2, B, H, W, C, shift_mode)) return e;
    if (!a) return MLPK_ENULL;
    SplitArgs p{x0, x1, x2, ld0, ld1, ld2, B, H, W, C, shift_mode};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)B, (unsigned)((C + 63) / 64));
    if (dtype != MLPK_F32 && split_vec_ok(x0, x1, x2, ld0, ld1, ld2, C, shift_mode)) {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((split_sum_vec_kernel<T>), grid, dim3(256), 0, s, p, a, scale));
    } else {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((split_sum_kernel<T>), grid, dim3(256), 0, s, p, a, scale));
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_split_softmax(const float* hat, float* bar, int B, int C, void* stream) {
    if (!hat || !bar) return MLPK_ENULL;
    if (B <= 0 || C <= 0) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(split_softmax_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, hat, bar, B, C);
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_split_apply(int dtype, const void* x0, const void* x1, const void* x2, int ld0, int ld1, int ld2,
                                int B, int H, int W, int C, int shift_mode, const float* bar, void* out, int ldo,
                                void* stream) {
    if (int e = split_check(x0, x1, x2, ld0, ld1, ld2, B, H, W, C, shift_mode)) return e;
    if (!bar || !out) return MLPK_ENULL;
    if (ldo < C) return MLPK_ESHAPE;
    SplitArgs p{x0, x1, x2, ld0, ld1, ld2, B, H, W, C, shift_mode};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * W * C;
    if (dtype != MLPK_F32 && split_vec_ok(x0, x1, x2, ld0, ld1, ld2, C, shift_mode) && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0 &&
        ((uintptr_t)bar & 15) == 0) {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((split_apply_vec_kernel<T>), dim3(grid_for(total / 8)), dim3(256), 0, s, p,
                                                 bar, (T*)out, ldo));
    } else {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((split_apply_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, p, bar,
                                                 (T*)out, ldo));
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

static int vip_split_check(int dtype, const void* zh, const void* zw, const void* xc, int ldh, int ldw, int ldc, int B, int H, int W, int C,
                           int seg) {
    if (!zh || !zw || !xc) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || seg <= 0 || C % seg) return MLPK_ESHAPE;
    // 8-byte pieces: 4 channels never straddle a segment, every piece is 8-byte aligned
    if (C % 8 || seg % 4 || ldh % 4 || ldw % 4 || ldc % 8 || ldh < H * seg || ldw < W * seg || ldc < C) return MLPK_ESHAPE;
    if (((uintptr_t)zh | (uintptr_t)zw) & 7 || ((uintptr_t)xc & 15)) return MLPK_EALIGN;
    return 0;
}

extern "C" int mlpk_vip_split_apply(int dtype, const void* zh, const void* zw, const void* xc, int ldh, int ldw, int ldc, int B, int H, int W,
                                    int C, int seg, const float* bar, void* out, int ldo, void* stream) {
    if (int e = vip_split_check(dtype, zh, zw, xc, ldh, ldw, ldc, B, H, W, C, seg)) return e;
    if (!bar || !out) return MLPK_ENULL;
    if (ldo < C || ldo % 8) return MLPK_ESHAPE;
    if (((uintptr_t)out | (uintptr_t)bar) & 15) return MLPK_EALIGN;
    VipSplitArgs p{zh, zw, xc, ldh, ldw, ldc, B, H, W, C, seg};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * W * C;
    // 8 x 8 pixel tiles when the map is made of whole tiles, the runs are 16-byte aligned and the two staged tiles fit in LDS
    // channel split (round 6): as many workgroups per pixel tile as it takes for three of them to fit a CU's LDS (each takes whole groups,
    // a multiple of 8 channels); MLPK_VIP_APPLY_PARTS overrides (A/B aid)
    const int G = C / seg;
    static const int parts_env = getenv("MLPK_VIP_APPLY_PARTS") ? atoi(getenv("MLPK_VIP_APPLY_PARTS")) : 0;
    int parts = 1;
    auto lds_of = [&](int pp) { return (size_t)2 * 8 * (G / pp) * 8 * seg * 2 + (size_t)3 * (C / pp) * 4; };
    auto parts_ok = [&](int pp) { return pp >= 1 && G % pp == 0 && ((G / pp) * seg) % 8 == 0; };
    if (parts_env > 0 && parts_ok(parts_env)) parts = parts_env;
    else
        for (int pp = 1; pp <= 4; ++pp)
            if (parts_ok(pp)) { parts = pp; if (lds_of(pp) <= 52 * 1024) break; }
    const size_t lds_tile = lds_of(parts);
    static const bool no_tile = getenv("MLPK_VIP_APPLY_NO_TILE") != nullptr;       // A/B aid
    if (!no_tile && H % 8 == 0 && W % 8 == 0 && ldh % 8 == 0 && ldw % 8 == 0 && lds_tile <= 150 * 1024 && (int64_t)B * (H / 8) * (W / 8) * parts < 0x7fffffff &&
        (((uintptr_t)zh | (uintptr_t)zw) & 15) == 0) {
        const unsigned grid = (unsigned)((int64_t)B * (H / 8) * (W / 8) * parts);
        hipError_t e = hipSuccess;
        if (dtype == MLPK_BF16) {
            auto k = vip_split_apply_tile_kernel<bf16_t>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds_tile, s, p, bar, (bf16_t*)out, ldo, parts);
        } else {
            auto k = vip_split_apply_tile_kernel<f16_t>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds_tile, s, p, bar, (f16_t*)out, ldo, parts);
        }
        MLPK_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == MLPK_BF16) hipLaunchKernelGGL((vip_split_apply_kernel<bf16_t>), dim3(grid_for(total / 8)), dim3(256), 0, s, p, bar, (bf16_t*)out, ldo);
    else hipLaunchKernelGGL((vip_split_apply_kernel<f16_t>), dim3(grid_for(total / 8)), dim3(256), 0, s, p, bar, (f16_t*)out, ldo);
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_s2_shift(int dtype, const void* in, void* out, int B, int H, int W, int C, int ldi, int ldo,
                             int shift_mode, void* stream) {
    if (!in || !out) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ldi < C || ldo < C) return MLPK_ESHAPE;
    if (shift_mode < MLPK_SHIFT_NONE || shift_mode > MLPK_SHIFT_S2_REF) return MLPK_EMODE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * H * W * C;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((s2_shift_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)in,
                                             (T*)out, B, H, W, C, ldi, ldo, shift_mode));
    MLPK_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int dwconv_launch(int k, const void* x, void* out, int B, int H, int W, int C, const float* w, const float* bias,
                         const float* bns, const float* bnh, hipStream_t s) {
    const int strips = (W + 7) / 8;
    const int64_t njobs = (int64_t)B * H * strips;
    const dim3 grid((unsigned)((njobs + 3) / 4), (unsigned)((C + 63) / 64));
#define DW_CASE(KS)                                                                                                   \
    case KS:                                                                                                          \
        hipLaunchKernelGGL((dwconv_nhwc_kernel<T, KS>), grid, dim3(256), 0, s, (const T*)x, (T*)out, B, H, W, C, w, bias, \
                           bns, bnh);                                                                                 \
        break;
    switch (k) {
        DW_CASE(1) DW_CASE(3) DW_CASE(5) DW_CASE(7) DW_CASE(9) DW_CASE(11) DW_CASE(13)
        default: return MLPK_ESHAPE;
    }
#undef DW_CASE
    MLPK_LAUNCH_CHECK();
    return 0;
}

// generic fallback (any H, W, C); the LDS-tiled kernel in mlpk_dwconv.hip is the fast path
int mlpk_dwconv_direct(int dtype, const void* x, void* out, int B, int H, int W, int C, int k, const float* w,
                       const float* bias, const float* bn_scale, const float* bn_shift, void* stream) {
    if (!x || !out || !w) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return MLPK_ESHAPE;
    if ((int64_t)B * H * ((W + 7) / 8) / 4 + 1 > 0x7fffffffLL) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case MLPK_F32: return dwconv_launch<float>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s);
        case MLPK_F16: return dwconv_launch<f16_t>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s);
        case MLPK_BF16: return dwconv_launch<bf16_t>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s);
        default: return MLPK_EDTYPE;
    }
}
