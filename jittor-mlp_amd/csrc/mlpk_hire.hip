// Hire-MLP region remaps (hire_mlp.py:44-152): the reference pads the map circularly to whole
// regions, rolls it (cross-region blocks), folds the h (w) rows (columns) that sit one region-count apart into the
// channel axis with einops, runs a two-layer 1x1-conv MLP on that, and undoes all three.  Here the three steps are
// index arithmetic inside one gather (building the GEMM operand rows directly from the channel-last LayerNorm output)
// and one combine (adding both branch results back onto the map); nothing padded, rolled or permuted is ever stored.
// Both kernels are HBM-bound element moves on 16-byte channel vectors.
#include "mlpk_common.h"

namespace mlpk {

struct HireArgs {
    const void* xn;      // (B, H, W, C) channel-last
    void* a_h;           // (B * gh * W, ld_h): row (b, g, x), columns hh * C + c  <-  xn[b, src_h((hh * gh + g - step) mod Hp), x, c]
    void* a_w;           // (B * H * gw, ld_w): row (b, y, g), columns ww * C + c  <-  xn[b, y, src_w((ww * gw + g - step) mod Wp), c]
    int B, H, W, C, h, w, step, Hp, Wp, gh, gw, ld_h, ld_w;
    // round 5 (mlpk_hire_gather_ln): xn is the UN-normalised x and the block's LayerNorm (hire_mlp.py:176, per pixel over C) is applied to every
    // vector on the way -- (x - mean) rstd gamma + beta, one rounding, the expression of mlpk_norm_apply: no normalised tensor is stored
    const float* mean;   // per pixel (B*H*W), or NULL: xn is used as it is
    const float* rstd;
    const float* gamma;  // per channel
    const float* beta;
};

template <typename T>
__device__ __forceinline__ u32x4 hire_ln(const HireArgs& p, u32x4 v, int64_t pix, int c0) {
    constexpr int EPV = 16 / (int)sizeof(T);
    if (!p.mean) return v;
    const float rs = p.rstd[pix], mu = p.mean[pix];
    T e[EPV];
    __builtin_memcpy(e, &v, 16);
#pragma unroll
    for (int k = 0; k < EPV; ++k) e[k] = from_f32<T>(__builtin_fmaf((to_f32(e[k]) - mu) * rs, p.gamma[c0 + k], p.beta[c0 + k]));   // (norm_apply_tile_kernel's form)
    u32x4 o;
    __builtin_memcpy(&o, e, 16);
    return o;
}

template <typename T>
__global__ void __launch_bounds__(256) hire_gather_kernel(const HireArgs p) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const T* __restrict__ xn = reinterpret_cast<const T*>(p.xn);
    const int cv = p.C / EPV;
    const int64_t n_h = (int64_t)p.B * p.gh * p.W * p.h * cv;
    const int64_t n_w = (int64_t)p.B * p.H * p.gw * p.w * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_h + n_w; idx += (int64_t)gridDim.x * 256) {
        if (idx < n_h) {
            const int c0 = (int)(idx % cv) * EPV;
            int64_t r = idx / cv;
            const int hh = (int)(r % p.h); r /= p.h;
            const int x = (int)(r % p.W); r /= p.W;
            const int g = (int)(r % p.gh);
            const int64_t b = r / p.gh;
            int q = (hh * p.gh + g - p.step) % p.Hp;
            if (q < 0) q += p.Hp;
            const int y = q < p.H ? q : q - p.H;                       // circular padding: appended rows repeat the first ones
            const u32x4 v = hire_ln<T>(p, *reinterpret_cast<const u32x4*>(xn + ((b * p.H + y) * p.W + x) * p.C + c0), (b * p.H + y) * p.W + x, c0);
            T* o = reinterpret_cast<T*>(p.a_h) + ((b * p.gh + g) * p.W + x) * p.ld_h + hh * p.C + c0;
            *reinterpret_cast<u32x4*>(o) = v;
        } else {
            const int64_t j = idx - n_h;
            const int c0 = (int)(j % cv) * EPV;
            int64_t r = j / cv;
            const int ww = (int)(r % p.w); r /= p.w;
            const int g = (int)(r % p.gw); r /= p.gw;
            const int y = (int)(r % p.H);
            const int64_t b = r / p.H;
            int q = (ww * p.gw + g - p.step) % p.Wp;
            if (q < 0) q += p.Wp;
            const int x = q < p.W ? q : q - p.W;
            const u32x4 v = hire_ln<T>(p, *reinterpret_cast<const u32x4*>(xn + ((b * p.H + y) * p.W + x) * p.C + c0), (b * p.H + y) * p.W + x, c0);
            T* o = reinterpret_cast<T*>(p.a_w) + ((b * p.H + y) * p.gw + g) * p.ld_w + ww * p.C + c0;
            *reinterpret_cast<u32x4*>(o) = v;
        }
    }
}

struct HireCombineArgs {
    void* x;             // (B, H, W, C): x = src + y_h(restored) + y_w(restored)
    const void* src;     // (B, H, W, C); == x: in place (mlpk_hire_combine)
    const void* y_h;     // (B * gh * W, ld_h), columns hh * C + c
    const void* y_w;     // (B * H * gw, ld_w), columns ww * C + c
    int B, H, W, C, h, w, step, Hp, Wp, gh, gw, ld_h, ld_w;
};

template <typename T>
__global__ void __launch_bounds__(256) hire_combine_kernel(const HireCombineArgs p) {
    constexpr int EPV = 16 / (int)sizeof(T);
    T* xo = reinterpret_cast<T*>(p.x);
    const T* xs = reinterpret_cast<const T*>(p.src);
    const T* __restrict__ yh = reinterpret_cast<const T*>(p.y_h);
    const T* __restrict__ yw = reinterpret_cast<const T*>(p.y_w);
    const int cv = p.C / EPV;
    const int64_t total = (int64_t)p.B * p.H * p.W * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(idx % cv) * EPV;
        int64_t r = idx / cv;
        const int x = (int)(r % p.W); r /= p.W;
        const int y = (int)(r % p.H);
        const int64_t b = r / p.H;
        // restore roll: out[i] = full[(i + step) mod Hp]; full[hh * gh + g] = branch output row g, column block hh
        int qh = (y + p.step) % p.Hp;
        if (qh < 0) qh += p.Hp;
        int qw = (x + p.step) % p.Wp;
        if (qw < 0) qw += p.Wp;
        const T* ph = yh + ((b * p.gh + qh % p.gh) * p.W + x) * p.ld_h + (qh / p.gh) * p.C + c0;
        const T* pw = yw + ((b * p.H + y) * p.gw + qw % p.gw) * p.ld_w + (qw / p.gw) * p.C + c0;
        T* po = xo + ((b * p.H + y) * p.W + x) * p.C + c0;
        T a[EPV], u[EPV], v[EPV];
        *reinterpret_cast<u32x4*>(a) = *reinterpret_cast<const u32x4*>(xs + ((b * p.H + y) * p.W + x) * p.C + c0);
        *reinterpret_cast<u32x4*>(u) = *reinterpret_cast<const u32x4*>(ph);
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(pw);
#pragma unroll
        for (int e = 0; e < EPV; ++e) a[e] = from_f32<T>(to_f32(a[e]) + (to_f32(u[e]) + to_f32(v[e])));
        *reinterpret_cast<u32x4*>(po) = *reinterpret_cast<const u32x4*>(a);
    }
}

// The same combine with the LayerNorm statistics of the rows it writes (round 6, mlpk_hire_combine_stats): what the block's second PreNormResidual
// needs next (hire_mlp.py:181), so that no statistics pass reads x again.  Eight lanes own one pixel (C / 8 16-byte vectors in strides of 8: 128
// contiguous bytes per step), the pixel's index arithmetic is done once; sums and sums of squares of the ROUNDED results in fp32, reduced over the eight
// lanes in a fixed order (deterministic, independent of the batch).
template <typename T>
__global__ void __launch_bounds__(256) hire_combine_stats_kernel(const HireCombineArgs p, float* __restrict__ out_mean, float* __restrict__ out_rstd, const float eps) {
    constexpr int EPV = 16 / (int)sizeof(T);
    T* xo = reinterpret_cast<T*>(p.x);
    const T* xs = reinterpret_cast<const T*>(p.src);
    const T* __restrict__ yh = reinterpret_cast<const T*>(p.y_h);
    const T* __restrict__ yw = reinterpret_cast<const T*>(p.y_w);
    const int cv = p.C / EPV;
    const int64_t npix = (int64_t)p.B * p.H * p.W;
    const int sub = threadIdx.x & 7;
    for (int64_t px = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); px < npix; px += (int64_t)gridDim.x * 32) {
        int64_t r = px;
        const int x = (int)(r % p.W); r /= p.W;
        const int y = (int)(r % p.H);
        const int64_t b = r / p.H;
        int qh = (y + p.step) % p.Hp;
        if (qh < 0) qh += p.Hp;
        int qw = (x + p.step) % p.Wp;
        if (qw < 0) qw += p.Wp;
        const T* ph = yh + ((b * p.gh + qh % p.gh) * p.W + x) * p.ld_h + (qh / p.gh) * p.C;
        const T* pw = yw + ((b * p.H + y) * p.gw + qw % p.gw) * p.ld_w + (qw / p.gw) * p.C;
        const T* ps = xs + px * p.C;
        T* po = xo + px * p.C;
        float s = 0.f, ss = 0.f;
        for (int v = sub; v < cv; v += 8) {
            T a[EPV], u[EPV], w[EPV];
            *reinterpret_cast<u32x4*>(a) = *reinterpret_cast<const u32x4*>(ps + v * EPV);
            *reinterpret_cast<u32x4*>(u) = *reinterpret_cast<const u32x4*>(ph + v * EPV);
            *reinterpret_cast<u32x4*>(w) = *reinterpret_cast<const u32x4*>(pw + v * EPV);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                a[e] = from_f32<T>(to_f32(a[e]) + (to_f32(u[e]) + to_f32(w[e])));
                const float f = to_f32(a[e]);
                s += f;
                ss = __builtin_fmaf(f, f, ss);
            }
            *reinterpret_cast<u32x4*>(po + v * EPV) = *reinterpret_cast<const u32x4*>(a);
        }
        s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
        s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
        s += __shfl_xor(s, 4); ss += __shfl_xor(ss, 4);
        if (sub == 0) {
            const float mean = s / (float)p.C;
            const float var = ss / (float)p.C - mean * mean;
            out_mean[px] = mean;
            out_rstd[px] = 1.0f / __builtin_sqrtf((var > 0.f ? var : 0.f) + eps);
        }
    }
}

static int hire_geometry(int B, int H, int W, int C, int h, int w, int dtype, int ld_h, int ld_w, int* Hp, int* Wp) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || h <= 0 || w <= 0) return MLPK_ESHAPE;
    if (dtype != MLPK_F32 && dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    const int epv = dtype == MLPK_F32 ? 4 : 8;
    if (C % epv || ld_h % epv || ld_w % epv || ld_h < h * C || ld_w < w * C) return MLPK_ESHAPE;
    *Hp = H + (h - H % h);                                  // hire_mlp.py:131-133: a whole extra region when H % h == 0
    *Wp = W + (w - W % w);
    if (*Hp - H > H || *Wp - W > W) return MLPK_ESHAPE;     // circular padding cannot wrap more than once (torch raises too)
    return 0;
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_hire_gather_ln(int dtype, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* a_h,
                                   void* a_w, int B, int H, int W, int C, int h, int w, int step, int ld_h, int ld_w, void* stream);

extern "C" int mlpk_hire_gather(int dtype, const void* xn, void* a_h, void* a_w, int B, int H, int W, int C, int h, int w,
                                int step, int ld_h, int ld_w, void* stream) {
    return mlpk_hire_gather_ln(dtype, xn, nullptr, nullptr, nullptr, nullptr, a_h, a_w, B, H, W, C, h, w, step, ld_h, ld_w, stream);
}

extern "C" int mlpk_hire_gather_ln(int dtype, const void* xn, const float* mean, const float* rstd, const float* gamma, const float* beta, void* a_h,
                                   void* a_w, int B, int H, int W, int C, int h, int w, int step, int ld_h, int ld_w, void* stream) {
    if (!xn || !a_h || !a_w) return MLPK_ENULL;
    if ((mean != nullptr) != (rstd != nullptr) || (mean != nullptr) != (gamma != nullptr) || (mean != nullptr) != (beta != nullptr)) return MLPK_ENULL;
    HireArgs a;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
    int rc = hire_geometry(B, H, W, C, h, w, dtype, ld_h, ld_w, &a.Hp, &a.Wp);
    if (rc) return rc;
    if (((uintptr_t)xn & 15) || ((uintptr_t)a_h & 15) || ((uintptr_t)a_w & 15)) return MLPK_EALIGN;
    a.xn = xn; a.a_h = a_h; a.a_w = a_w; a.B = B; a.H = H; a.W = W; a.C = C; a.h = h; a.w = w; a.step = step;
    a.gh = a.Hp / h; a.gw = a.Wp / w; a.ld_h = ld_h; a.ld_w = ld_w;
    const int epv = dtype == MLPK_F32 ? 4 : 8;
    const long long total = ((long long)B * a.gh * W * h + (long long)B * H * a.gw * w) * (C / epv);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case MLPK_F32: hipLaunchKernelGGL(hire_gather_kernel<float>, dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_F16: hipLaunchKernelGGL(hire_gather_kernel<f16_t>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(hire_gather_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a); break;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_hire_combine_from(int dtype, void* x, const void* src, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                                      int step, int ld_h, int ld_w, void* stream);

extern "C" int mlpk_hire_combine(int dtype, void* x, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                                 int step, int ld_h, int ld_w, void* stream) {
    return mlpk_hire_combine_from(dtype, x, x, y_h, y_w, B, H, W, C, h, w, step, ld_h, ld_w, stream);
}

extern "C" int mlpk_hire_combine_from(int dtype, void* x, const void* src, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                                      int step, int ld_h, int ld_w, void* stream) {
    if (!x || !src || !y_h || !y_w) return MLPK_ENULL;
    if ((uintptr_t)src & 15) return MLPK_EALIGN;
    HireCombineArgs a;
    a.src = src;
    int rc = hire_geometry(B, H, W, C, h, w, dtype, ld_h, ld_w, &a.Hp, &a.Wp);
    if (rc) return rc;
    if (((uintptr_t)x & 15) || ((uintptr_t)y_h & 15) || ((uintptr_t)y_w & 15)) return MLPK_EALIGN;
    a.x = x; a.y_h = y_h; a.y_w = y_w; a.B = B; a.H = H; a.W = W; a.C = C; a.h = h; a.w = w; a.step = step;
    a.gh = a.Hp / h; a.gw = a.Wp / w; a.ld_h = ld_h; a.ld_w = ld_w;
    const int epv = dtype == MLPK_F32 ? 4 : 8;
    const long long total = (long long)B * H * W * (C / epv);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case MLPK_F32: hipLaunchKernelGGL(hire_combine_kernel<float>, dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_F16: hipLaunchKernelGGL(hire_combine_kernel<f16_t>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(hire_combine_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a); break;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_hire_combine_stats(int dtype, void* x, const void* src, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                                       int step, int ld_h, int ld_w, float* out_mean, float* out_rstd, float eps, void* stream);

extern "C" int mlpk_hire_combine_stats(int dtype, void* x, const void* src, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                                       int step, int ld_h, int ld_w, float* out_mean, float* out_rstd, float eps, void* stream) {
    if (!x || !src || !y_h || !y_w || !out_mean || !out_rstd) return MLPK_ENULL;
    if ((uintptr_t)src & 15) return MLPK_EALIGN;
    HireCombineArgs a;
    a.src = src;
    int rc = hire_geometry(B, H, W, C, h, w, dtype, ld_h, ld_w, &a.Hp, &a.Wp);
    if (rc) return rc;
    if (((uintptr_t)x | (uintptr_t)y_h | (uintptr_t)y_w) & 15) return MLPK_EALIGN;
    if (!(eps > 0.f)) return MLPK_ESHAPE;
    a.x = x; a.y_h = y_h; a.y_w = y_w; a.B = B; a.H = H; a.W = W; a.C = C; a.h = h; a.w = w; a.step = step;
    a.gh = a.Hp / h; a.gw = a.Wp / w; a.ld_h = ld_h; a.ld_w = ld_w;
    const long long npix = (long long)B * H * W;
    const unsigned grid = (unsigned)((npix + 31) / 32 < 65536 ? (npix + 31) / 32 : 65536);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case MLPK_F32: hipLaunchKernelGGL(hire_combine_stats_kernel<float>, dim3(grid), dim3(256), 0, s, a, out_mean, out_rstd, eps); break;
        case MLPK_F16: hipLaunchKernelGGL(hire_combine_stats_kernel<f16_t>, dim3(grid), dim3(256), 0, s, a, out_mean, out_rstd, eps); break;
        default: hipLaunchKernelGGL(hire_combine_stats_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a, out_mean, out_rstd, eps); break;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

// ---- MS-MLP mix-shift (ms_mlp.py:48-66) -------------------------------------------------
// out[b,y,x,c] = b_lr[c] + sum_{dy,dx} w_lr[c][dy][dx] * R_w(y+dy-p, x+dx-p)  +  b_td[c] + sum w_td[c][dy][dx] * R_h(y+dy-p, x+dx-p)
// with k = ksize[g(c)], p = k/2, zero outside [0,H)x[0,W), R_w(y,x) = in[b, y, (x - s) mod W, c] (the chunk rolled along W),
// R_h(y,x) = in[b, (y - s) mod H, x, c], s = shift[g(c)], g(c) = c / chunk0 (torch.chunk: ceil(C / groups) channels per chunk).
// One thread per (pixel, channel): consecutive threads = consecutive channels (coalesced); chunks need not be whole vectors.
namespace mlpk {

struct MixShiftArgs {
    const void* x;
    void* out;
    const float* w_lr;   // [kmax*kmax][C] tap-major (row dy*k + dx of the chunk's own k), unused taps absent
    const float* w_td;
    const float* b_lr;   // [C]
    const float* b_td;
    int B, H, W, C, groups, chunk0;
    int shift[8], ksize[8];
};

template <typename T>
__global__ void __launch_bounds__(256) mixshift_kernel(const MixShiftArgs p) {
    const T* __restrict__ in = reinterpret_cast<const T*>(p.x);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int64_t total = (int64_t)p.B * p.H * p.W * p.C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % p.C);
        int64_t r = idx / p.C;
        const int x = (int)(r % p.W); r /= p.W;
        const int y = (int)(r % p.H);
        const int64_t b = r / p.H;
        const int g = c / p.chunk0;
        const int k = p.ksize[g], s = p.shift[g], P = k >> 1;
        float acc = p.b_lr[c] + p.b_td[c];
        const T* img = in + b * p.H * p.W * p.C + c;
        for (int dy = 0; dy < k; ++dy) {
            const int yy = y + dy - P;
            if (yy < 0 || yy >= p.H) continue;
            int ys = (yy - s) % p.H;
            if (ys < 0) ys += p.H;
            for (int dx = 0; dx < k; ++dx) {
                const int xx = x + dx - P;
                if (xx < 0 || xx >= p.W) continue;
                int xs = (xx - s) % p.W;
                if (xs < 0) xs += p.W;
                const int t = dy * k + dx;
                acc = fmaf(p.w_lr[(size_t)t * p.C + c], to_f32(img[((int64_t)yy * p.W + xs) * p.C]), acc);
                acc = fmaf(p.w_td[(size_t)t * p.C + c], to_f32(img[((int64_t)ys * p.W + xx) * p.C]), acc);
            }
        }
        out[idx] = from_f32<T>(acc);
    }
}

// 16-byte channel vectors: a vector whose channels share one chunk (same shift, same kernel size) moves as whole vectors
// -- per tap two 16-byte data loads and the weight vectors -- instead of one 2-byte load per channel and tap; a vector
// that straddles a chunk boundary (chunks of 20 channels: every fifth one) falls back to the per-channel arithmetic.
// Measured on MS-MLP-T (bs 256, bf16): 950 us per call per-channel -> 805 us; still texture-address bound (every input vector is
// fetched once per tap and branch, and a wave runs the longest kernel of its lanes).  A variant with 4-pixel output strips
// and the kernel size as a template parameter (each loaded vector reused for 4 outputs) spilled its window registers
// and ran at 2650 us; the LDS-tiled mixshift_band_kernel below is what the models use, this one is the fallback for the
// shapes it does not cover (kernel sizes other than 1/3/5/7, maps wider than 56 columns).
template <typename T>
__global__ void __launch_bounds__(256) mixshift_vec_kernel(const MixShiftArgs p) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const T* __restrict__ in = reinterpret_cast<const T*>(p.x);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int cv = p.C / EPV;
    const int64_t total = (int64_t)p.B * p.H * p.W * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(idx % cv) * EPV;
        int64_t r = idx / cv;
        const int x = (int)(r % p.W); r /= p.W;
        const int y = (int)(r % p.H);
        const int64_t b = r / p.H;
        const T* img = in + b * p.H * p.W * p.C;
        float acc[EPV];
        const int g0 = c0 / p.chunk0;
        if (g0 == (c0 + EPV - 1) / p.chunk0) {
            const int k = p.ksize[g0], P = k >> 1;
            // roll distances reduced once (two divisions per thread, none per tap): 0 <= sh < H, 0 <= sw < W
            int sh = p.shift[g0] % p.H, sw = p.shift[g0] % p.W;
            if (sh < 0) sh += p.H;
            if (sw < 0) sw += p.W;
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                const f32x4 bl = *reinterpret_cast<const f32x4*>(p.b_lr + c0 + e), bt = *reinterpret_cast<const f32x4*>(p.b_td + c0 + e);
                acc[e] = bl.x + bt.x; acc[e + 1] = bl.y + bt.y; acc[e + 2] = bl.z + bt.z; acc[e + 3] = bl.w + bt.w;
            }
            for (int dy = 0; dy < k; ++dy) {
                const int yy = y + dy - P;
                if (yy < 0 || yy >= p.H) continue;
                int ys = yy - sh;
                if (ys < 0) ys += p.H;
                for (int dx = 0; dx < k; ++dx) {
                    const int xx = x + dx - P;
                    if (xx < 0 || xx >= p.W) continue;
                    int xs = xx - sw;
                    if (xs < 0) xs += p.W;
                    const int t = dy * k + dx;
                    T vl[EPV], vt[EPV];
                    *reinterpret_cast<u32x4*>(vl) = *reinterpret_cast<const u32x4*>(img + ((int64_t)yy * p.W + xs) * p.C + c0);
                    *reinterpret_cast<u32x4*>(vt) = *reinterpret_cast<const u32x4*>(img + ((int64_t)ys * p.W + xx) * p.C + c0);
                    const float* wl = p.w_lr + (size_t)t * p.C + c0;
                    const float* wt = p.w_td + (size_t)t * p.C + c0;
#pragma unroll
                    for (int e = 0; e < EPV; e += 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(wl + e), d = *reinterpret_cast<const f32x4*>(wt + e);
                        acc[e] = fmaf(d.x, to_f32(vt[e]), fmaf(a.x, to_f32(vl[e]), acc[e]));
                        acc[e + 1] = fmaf(d.y, to_f32(vt[e + 1]), fmaf(a.y, to_f32(vl[e + 1]), acc[e + 1]));
                        acc[e + 2] = fmaf(d.z, to_f32(vt[e + 2]), fmaf(a.z, to_f32(vl[e + 2]), acc[e + 2]));
                        acc[e + 3] = fmaf(d.w, to_f32(vt[e + 3]), fmaf(a.w, to_f32(vl[e + 3]), acc[e + 3]));
                    }
                }
            }
        } else {
            for (int e = 0; e < EPV; ++e) {
                const int c = c0 + e;
                const int g = c / p.chunk0;
                const int k = p.ksize[g], s = p.shift[g], P = k >> 1;
                float a = p.b_lr[c] + p.b_td[c];
                for (int dy = 0; dy < k; ++dy) {
                    const int yy = y + dy - P;
                    if (yy < 0 || yy >= p.H) continue;
                    int ys = (yy - s) % p.H;
                    if (ys < 0) ys += p.H;
                    for (int dx = 0; dx < k; ++dx) {
                        const int xx = x + dx - P;
                        if (xx < 0 || xx >= p.W) continue;
                        int xs = (xx - s) % p.W;
                        if (xs < 0) xs += p.W;
                        const int t = dy * k + dx;
                        a = fmaf(p.w_lr[(size_t)t * p.C + c], to_f32(img[((int64_t)yy * p.W + xs) * p.C + c]), a);
                        a = fmaf(p.w_td[(size_t)t * p.C + c], to_f32(img[((int64_t)ys * p.W + xx) * p.C + c]), a);
                    }
                }
                acc[e] = a;
            }
        }
        T o[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = from_f32<T>(acc[e]);
        *reinterpret_cast<u32x4*>(out + ((b * p.H + y) * p.W + x) * p.C + c0) = *reinterpret_cast<const u32x4*>(o);
    }
}

// LDS-tiled form (the design of mlpk_dwconv.hip): workgroup = one image x a band of 8 output rows x 32 channels of ONE chunk
// (uniform kernel size KS and shift), both branches.  Per branch the band (+ halo) is staged TRANSPOSED in LDS as
// [channel][row][column] with the roll folded into the source index and zeros outside the map; a thread is one channel
// and walks strips of 8 outputs, its taps in registers as pairs feeding v_pk_fma_f32.  The partial sums of the first
// branch stay in registers while the second branch's band replaces the first in LDS.  Every input element is fetched
// from global (R + KS - 1) / R times per branch instead of KS * KS times.
constexpr int MS_CT = 32, MS_R = 8, MS_NT = 256, MS_TMAX = 7;

template <typename T, int KS>
__global__ void __launch_bounds__(MS_NT) mixshift_band_kernel(const MixShiftArgs p, const int cb, const int cn, const int sh, const int sw,
                                                              const int pitch, const int plane) {
    constexpr int P = KS / 2;
    constexpr int STRIP = 8;
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int WIN = STRIP + KS - 1;
    constexpr int NV = (WIN + EPV - 1) / EPV;
    constexpr int GROUPS = MS_NT / MS_CT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);              // [MS_CT][MS_R + 2P][pitch], plane stride `plane` elements
    const T* __restrict__ in = reinterpret_cast<const T*>(p.x);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int tid = threadIdx.x;
    const int nbands = (p.H + MS_R - 1) / MS_R;
    const int b = blockIdx.x / nbands;
    const int y0 = (blockIdx.x - b * nbands) * MS_R;
    const int c0 = cb + blockIdx.y * MS_CT;
    const int nch = min(MS_CT, cb + cn - c0);              // channels of this tile
    const int rows_t = MS_R + 2 * P;
    const int cols_t = p.W + 2 * P;
    const T* img = in + (size_t)b * p.H * p.W * p.C;

    const int cl = tid % MS_CT;
    const int c = c0 + cl;
    const bool live = cl < nch;
    const int strips = (p.W + STRIP - 1) / STRIP;
    const int ntask = MS_R * strips;
    const T* cplane = tile + cl * plane;
    f32x2 acc2[MS_TMAX][STRIP / 2];
    {
        const float bs = live ? p.b_lr[c] + p.b_td[c] : 0.f;
#pragma unroll
        for (int k = 0; k < MS_TMAX; ++k)
#pragma unroll
            for (int o = 0; o < STRIP / 2; ++o) acc2[k][o] = f32x2{bs, bs};
    }
#pragma unroll 1
    for (int branch = 0; branch < 2; ++branch) {
        __syncthreads();                                   // the previous branch's reads of the tile are done
        // ---- stage: element (channel ch, tile row r, tile column q) <- rolled source or zero.  16-byte loads (8 channels of one
        // pixel per thread, 4 consecutive threads = the 64 contiguous bytes of a pixel), UNCONDITIONAL: a position outside the map
        // reads pixel 0 and stores zeros (behind `if (inside)` every load waits for the one before it -- one memory latency per
        // element, which is what this kernel spent its time on: 0.5-0.9 TB/s, profiles/r04_traffic_models.txt) ----
        {
            constexpr int VPP = MS_CT / EPV;                   // vectors per pixel
            const bool vec = sizeof(T) == 2 && (p.C % EPV) == 0 && (c0 % EPV) == 0 && (nch % EPV) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0;
            if (vec) {
                const int total = rows_t * cols_t * VPP;
                for (int i = tid; i < total; i += MS_NT) {
                    const int vq = i % VPP;
                    const int px = i / VPP;
                    const int q = px % cols_t, r = px / cols_t;
                    const int yy = y0 + r - P, xx = q - P;
                    const bool inside = vq * EPV < nch && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    int ys = inside ? yy : 0, xs = inside ? xx : 0;
                    if (branch == 0) { xs -= sw; if (xs < 0) xs += p.W; }
                    else { ys -= sh; if (ys < 0) ys += p.H; }
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(img + ((size_t)ys * p.W + xs) * p.C + c0 + (inside ? vq * EPV : 0));
                    T e[EPV];
                    __builtin_memcpy(e, &raw, 16);
#pragma unroll
                    for (int k = 0; k < EPV; ++k) tile[(vq * EPV + k) * plane + r * pitch + q] = inside ? e[k] : from_f32<T>(0.f);
                }
            } else {
                const int total = rows_t * cols_t * MS_CT;
                for (int i = tid; i < total; i += MS_NT) {
                    const int ch = i % MS_CT;
                    const int px = i / MS_CT;
                    const int q = px % cols_t, r = px / cols_t;
                    const int yy = y0 + r - P, xx = q - P;
                    const bool inside = ch < nch && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    int ys = inside ? yy : 0, xs = inside ? xx : 0;
                    if (branch == 0) { xs -= sw; if (xs < 0) xs += p.W; }
                    else { ys -= sh; if (ys < 0) ys += p.H; }
                    const T v = img[((size_t)ys * p.W + xs) * p.C + c0 + (inside ? ch : 0)];
                    tile[ch * plane + r * pitch + q] = inside ? v : from_f32<T>(0.f);
                }
            }
            // the window reads run up to NV * EPV columns past a strip's start: zero the slack columns of every row
            const int slack = pitch - cols_t;
            if (slack > 0) {
                const int tot2 = rows_t * slack * MS_CT;
                for (int i = tid; i < tot2; i += MS_NT) {
                    const int ch = i % MS_CT;
                    const int px = i / MS_CT;
                    tile[ch * plane + (px / slack) * pitch + cols_t + px % slack] = from_f32<T>(0.f);
                }
            }
        }
        __syncthreads();
        if (live) {
            const float* wsrc = branch == 0 ? p.w_lr : p.w_td;
            constexpr int NT2 = (KS * KS + 1) / 2;
            f32x2 wt2[NT2];
#pragma unroll
            for (int q = 0; q < NT2; ++q) {
                wt2[q].x = wsrc[(size_t)(2 * q) * p.C + c];
                wt2[q].y = 2 * q + 1 < KS * KS ? wsrc[(size_t)(2 * q + 1) * p.C + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < MS_TMAX; ++k) {
                const int task = tid / MS_CT + k * GROUPS;
                if (task >= ntask) break;
                const int st = task % strips, ly = task / strips;
                const int x0 = st * STRIP;
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    const T* row = cplane + (ly + dy) * pitch + x0;
                    float win[NV * EPV + 1];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const u32x4 raw = *reinterpret_cast<const u32x4*>(row + v * EPV);
                        T e[EPV];
                        __builtin_memcpy(e, &raw, 16);
#pragma unroll
                        for (int kk = 0; kk < EPV; ++kk) win[v * EPV + kk] = to_f32(e[kk]);
                    }
                    win[NV * EPV] = 0.f;
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) {
                        const int tap = dy * KS + dx;
#pragma unroll
                        for (int o = 0; o < STRIP / 2; ++o) {
                            const f32x2 xin = {win[2 * o + dx], win[2 * o + dx + 1]};
                            if (tap & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc2[k][o]) : "v"(xin), "v"(wt2[tap >> 1]));
                            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[k][o]) : "v"(xin), "v"(wt2[tap >> 1]));
                        }
                    }
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int k = 0; k < MS_TMAX; ++k) {
        const int task = tid / MS_CT + k * GROUPS;
        if (task >= ntask) break;
        const int st = task % strips, ly = task / strips;
        const int y = y0 + ly;
        if (y >= p.H) continue;
#pragma unroll
        for (int o = 0; o < STRIP; ++o) {
            const int xx = st * STRIP + o;
            if (xx < p.W) out[(((size_t)b * p.H + y) * p.W + xx) * p.C + c] = from_f32<T>((o & 1) ? acc2[k][o >> 1].y : acc2[k][o >> 1].x);
        }
    }
}

// A chunk with kernel size 1 is no stencil: out = b_lr + b_td + w_lr x[.., x - s] + w_td x[y - s, ..] -- two rolled reads and a store per
// element.  Through the band kernel (LDS tile, barriers, strips) it ran at 0.6 TB/s; here a thread is one (pixel, channel of the chunk),
// LPP = the chunk width rounded up to a power of two lanes per pixel, a workgroup walks 256 / LPP pixels of one image row.  Same fused
// multiply-adds in the same order as the band kernel.
template <typename T>
__global__ void __launch_bounds__(256) mixshift_k1_kernel(const MixShiftArgs p, const int cb, const int cn, const int sh, const int sw, const int lpp_log2) {
    const T* __restrict__ in = reinterpret_cast<const T*>(p.x);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int tid = threadIdx.x;
    const int cc = tid & ((1 << lpp_log2) - 1), pl = tid >> lpp_log2;
    const int xx = blockIdx.y * (256 >> lpp_log2) + pl;
    const int row = blockIdx.x;                               // (b, y)
    const int b = row / p.H, y = row - b * p.H;
    if (cc >= cn || xx >= p.W) return;
    const int c = cb + cc;
    int xs = xx - sw; if (xs < 0) xs += p.W;
    int ys = y - sh; if (ys < 0) ys += p.H;
    const size_t img = (size_t)b * p.H * p.W;
    const float a_lr = to_f32(in[(img + (size_t)y * p.W + xs) * p.C + c]);
    const float a_td = to_f32(in[(img + (size_t)ys * p.W + xx) * p.C + c]);
    float acc = p.b_lr[c] + p.b_td[c];
    acc = fmaf(a_lr, p.w_lr[c], acc);
    acc = fmaf(a_td, p.w_td[c], acc);
    out[(img + (size_t)y * p.W + xx) * p.C + c] = from_f32<T>(acc);
}

template <typename T>
static int mixshift_band_launch(const MixShiftArgs& a, hipStream_t s) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const int strips = (a.W + 7) / 8;
    if (strips > MS_TMAX) return 1;
    for (int g = 0; g < a.groups; ++g)
        if (a.ksize[g] != 1 && a.ksize[g] != 3 && a.ksize[g] != 5 && a.ksize[g] != 7) return 1;
    const int nbands = (a.H + MS_R - 1) / MS_R;
    for (int g = 0; g < a.groups; ++g) {
        const int k = a.ksize[g], P = k / 2;
        const int cb = g * a.chunk0;
        const int cn = (cb + a.chunk0 <= a.C ? a.chunk0 : a.C - cb);
        if (k == 1 && cn <= 256) {
            int lg = 5;
            while ((1 << lg) < cn) ++lg;
            int sh1 = a.shift[g] % a.H, sw1 = a.shift[g] % a.W;
            if (sh1 < 0) sh1 += a.H;
            if (sw1 < 0) sw1 += a.W;
            const int ppw = 256 >> lg;
            const dim3 grid1((unsigned)(a.B * a.H), (unsigned)((a.W + ppw - 1) / ppw));
            hipLaunchKernelGGL((mixshift_k1_kernel<T>), grid1, dim3(256), 0, s, a, cb, cn, sh1, sw1, lg);
            continue;
        }
        int pitch = (strips - 1) * 8 + ((8 + k - 1 + EPV - 1) / EPV) * EPV;
        if (pitch < a.W + 2 * P) pitch = a.W + 2 * P;
        pitch = (pitch + EPV - 1) / EPV * EPV;
        int plane = (MS_R + 2 * P) * pitch;
        plane = (plane + EPV - 1) / EPV * EPV;
        if (((plane / EPV) & 1) == 0) plane += EPV;            // odd number of 16-byte slots per channel plane: conflict-free across channels
        const size_t lds = (size_t)MS_CT * plane * sizeof(T);
        if (lds > 150 * 1024) return 1;
        int sh = a.shift[g] % a.H, sw = a.shift[g] % a.W;
        if (sh < 0) sh += a.H;
        if (sw < 0) sw += a.W;
        const dim3 grid((unsigned)(a.B * nbands), (unsigned)((cn + MS_CT - 1) / MS_CT));
#define MS_CASE(KS)                                                                                                         \
    case KS: {                                                                                                              \
        auto kern = mixshift_band_kernel<T, KS>;                                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(MS_NT), lds, s, a, cb, cn, sh, sw, pitch, plane);                                \
        break;                                                                                                              \
    }
        switch (k) { MS_CASE(1) MS_CASE(3) MS_CASE(5) MS_CASE(7) default: return 1; }
#undef MS_CASE
    }
    return 0;
}

// ---- round 6: the whole mix-shift as ONE launch over aligned channel blocks (mixshift_tile_kernel) ----------------------------------
// The band kernel above runs once per chunk, and MS-MLP's chunks are 20 / 39 / 77 / 154 channels (torch.chunk of 96 / 192 / 384 / 768
// into 5): no chunk starts or ends on a 16-byte vector, so its staging fell back to 2-byte loads, its results left as 2-byte stores of
// 40-byte pieces (partial 64-byte sectors), up to 12 of a wave's 32 channel lanes idled, and a call was five launches: 780 us for the
// 154 MB + 154 MB of the 56 x 56 x 96 stage (0.4 TB/s).  Here a workgroup owns one image x a band of R rows x 32 ALIGNED channels
// (64 contiguous bytes of every pixel) and walks the chunks that meet its block: per chunk and branch the band (+ halo) of the chunk's
// channels is staged as in the band kernel -- but from aligned 16-byte loads, the elements of other chunks masked out, four loads in
// flight per thread -- a thread is one channel of the chunk and one of 256 / nch task slots (every lane works whatever the chunk
// width), the stencil arithmetic is the band kernel's (same fused multiply-adds in the same order: bit-equal results), and the
// results go through an LDS image of the block's output rows, which leaves as whole 64-byte pixel pieces.
constexpr int MT_CB = 32, MT_NT = 256, MT_TMAX = 7;

struct MtDiv {                                   // x / n for x < 2^32 / n by one multiply (n fixed per chunk)
    unsigned inv, n;
    __device__ explicit MtDiv(unsigned n_) : inv(n_ > 1 ? (unsigned)(((1ull << 32) + n_ - 1) / n_) : 0u), n(n_) {}
    __device__ __forceinline__ unsigned operator()(unsigned x) const { return n > 1 ? __umulhi(x, inv) : x; }
};

template <typename T, int KS>
__device__ __forceinline__ void mixtile_chunk(const MixShiftArgs& p, T* __restrict__ tile, T* __restrict__ otile, const T* __restrict__ img,
                                              const int g, const int c_lo, const int nch, const int cb0, const int y0, const int R,
                                              const int pitch, const int plane, const int tid) {
    constexpr int P = KS / 2;
    constexpr int STRIP = 8;
    constexpr int EPV = 8;
    constexpr int WIN = STRIP + KS - 1;
    constexpr int NV = (WIN + EPV - 1) / EPV;
    int sh = p.shift[g] % p.H, sw = p.shift[g] % p.W;
    if (sh < 0) sh += p.H;
    if (sw < 0) sw += p.W;
    const int slots = MT_NT / nch;                          // nch <= 32: at least 8 task slots
    const int cl = tid % nch, slot = tid / nch;
    const bool live = slot < slots;
    const int c = c_lo + cl;
    if constexpr (KS == 1) {
        // no stencil: two rolled reads and two multiply-adds per element (the k = 1 kernel's arithmetic), straight from global
        if (!live) return;
        const int rows_here = min(R, p.H - y0);
        const int npx = rows_here * p.W;
        const float wl = p.w_lr[c], wt = p.w_td[c], bs = p.b_lr[c] + p.b_td[c];
        const MtDiv by_w((unsigned)p.W);
        T* oc = otile + (c - cb0);
        const T* ic = img + c;
#pragma unroll 4
        for (int px = slot; px < npx; px += slots) {
            const int ly = (int)by_w((unsigned)px), xx = px - ly * p.W, y = y0 + ly;
            int xs = xx - sw; if (xs < 0) xs += p.W;
            int ys = y - sh; if (ys < 0) ys += p.H;
            const float a_lr = to_f32(ic[((size_t)y * p.W + xs) * p.C]);
            const float a_td = to_f32(ic[((size_t)ys * p.W + xx) * p.C]);
            float acc = bs;
            acc = fmaf(a_lr, wl, acc);
            acc = fmaf(a_td, wt, acc);
            oc[px * MT_CB] = from_f32<T>(acc);
        }
        return;
    }
    const int rows_t = R + 2 * P;
    const int strips = (p.W + STRIP - 1) / STRIP;
    const int ntask = R * strips;                           // <= 56 = MT_TMAX * 8
    const T* cplane = tile + cl * plane;
    const int rel0 = c_lo - cb0;                            // the chunk's first channel inside the block
    const int v_lo = rel0 / EPV, nvec = (rel0 + nch - 1) / EPV - v_lo + 1;
    const int quads = pitch / 4;                            // four tile columns per staging item: one 8-byte LDS write per channel
    const int row_items = quads * nvec, total = rows_t * row_items;
    const MtDiv by_items((unsigned)row_items), by_nvec((unsigned)nvec), by_strips((unsigned)strips);
    f32x2 acc2[MT_TMAX][STRIP / 2];
    {
        const float bs = live ? p.b_lr[c] + p.b_td[c] : 0.f;
#pragma unroll
        for (int k = 0; k < MT_TMAX; ++k)
#pragma unroll
            for (int o = 0; o < STRIP / 2; ++o) acc2[k][o] = f32x2{bs, bs};
    }
#pragma unroll 1
    for (int branch = 0; branch < 2; ++branch) {
        __syncthreads();                                    // the reads of the tile (previous branch / previous chunk) are done
        // ---- stage: (tile row r, four tile columns 4 qq .., vector v) <- the rolled source pixels' 8 channels, zeros outside the map and in
        // the slack columns up to the pitch (the window reads run past a strip's end).  The loads are unconditional (a position outside reads
        // pixel 0), eight are in flight per thread before the first LDS write; the transposition is one 8-byte write per channel ----
#pragma unroll 1
        for (int base = tid; base < total; base += 2 * MT_NT) {
            u32x4 raw[2][4];
            int dst[2], vv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = base + u * MT_NT;
                const int id = idx < total ? idx : total - 1;
                const int r = (int)by_items((unsigned)id);
                const int rem = id - r * row_items;
                const int qq = (int)by_nvec((unsigned)rem);
                const int v = v_lo + rem - qq * nvec;
                const int yy = y0 + r - P;
                const bool row_in = yy >= 0 && yy < p.H;
                int ys = row_in ? yy : 0;
                if (branch == 1) { ys -= sh; if (ys < 0) ys += p.H; }
                const T* srow = img + (size_t)ys * p.W * p.C + cb0 + v * EPV;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = 4 * qq + j - P;
                    const bool inside = row_in && xx >= 0 && xx < p.W;
                    int xs = inside ? xx : 0;
                    if (branch == 0) { xs -= sw; if (xs < 0) xs += p.W; }
                    const u32x4 t = *reinterpret_cast<const u32x4*>(srow + (size_t)xs * p.C);
                    raw[u][j] = inside ? t : u32x4{0u, 0u, 0u, 0u};
                }
                dst[u] = idx < total ? r * pitch + 4 * qq : -1;
                vv[u] = v * EPV - rel0;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (dst[u] < 0) continue;
#pragma unroll
                for (int k = 0; k < EPV; ++k) {
                    const int rel = vv[u] + k;              // channel inside the chunk
                    if (rel < 0 || rel >= nch) continue;
                    const int w = k >> 1;
                    u32x2 pk;
                    if (k & 1) {
                        pk.x = (raw[u][0][w] >> 16) | (raw[u][1][w] & 0xffff0000u);
                        pk.y = (raw[u][2][w] >> 16) | (raw[u][3][w] & 0xffff0000u);
                    } else {
                        pk.x = (raw[u][0][w] & 0xffffu) | (raw[u][1][w] << 16);
                        pk.y = (raw[u][2][w] & 0xffffu) | (raw[u][3][w] << 16);
                    }
                    *reinterpret_cast<u32x2*>(tile + rel * plane + dst[u]) = pk;
                }
            }
        }
        __syncthreads();
        if (live) {
            const float* wsrc = branch == 0 ? p.w_lr : p.w_td;
            constexpr int NT2 = (KS * KS + 1) / 2;
            f32x2 wt2[NT2];
#pragma unroll
            for (int q = 0; q < NT2; ++q) {
                wt2[q].x = wsrc[(size_t)(2 * q) * p.C + c];
                wt2[q].y = 2 * q + 1 < KS * KS ? wsrc[(size_t)(2 * q + 1) * p.C + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < MT_TMAX; ++k) {
                const int task = slot + k * slots;
                if (task >= ntask) break;
                const int ly = (int)by_strips((unsigned)task), st = task - ly * strips;
                const int x0 = st * STRIP;
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    const T* row = cplane + (ly + dy) * pitch + x0;
                    float win[NV * EPV + 1];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const u32x4 rw = *reinterpret_cast<const u32x4*>(row + v * EPV);
                        T e[EPV];
                        __builtin_memcpy(e, &rw, 16);
#pragma unroll
                        for (int kk = 0; kk < EPV; ++kk) win[v * EPV + kk] = to_f32(e[kk]);
                    }
                    win[NV * EPV] = 0.f;
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) {
                        const int tap = dy * KS + dx;
#pragma unroll
                        for (int o = 0; o < STRIP / 2; ++o) {
                            const f32x2 xin = {win[2 * o + dx], win[2 * o + dx + 1]};
                            if (tap & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc2[k][o]) : "v"(xin), "v"(wt2[tap >> 1]));
                            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[k][o]) : "v"(xin), "v"(wt2[tap >> 1]));
                        }
                    }
                }
            }
        }
    }
    if (!live) return;
    // results -> the block's output image in LDS, [row][column][channel of the block]
    T* oc = otile + (c - cb0);
#pragma unroll
    for (int k = 0; k < MT_TMAX; ++k) {
        const int task = slot + k * slots;
        if (task >= ntask) break;
        const int ly = (int)by_strips((unsigned)task), st = task - ly * strips;
#pragma unroll
        for (int o = 0; o < STRIP; ++o) {
            const int xx = st * STRIP + o;
            if (xx < p.W) oc[(ly * p.W + xx) * MT_CB] = from_f32<T>((o & 1) ? acc2[k][o >> 1].y : acc2[k][o >> 1].x);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(MT_NT) mixshift_tile_kernel(const MixShiftArgs p, const int R, const int pitch, const int plane, const int nbands,
                                                               float* __restrict__ row_part, const long long row_part_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);               // [<= MT_CB channels of one chunk][R + 2P rows][pitch]
    T* otile = tile + MT_CB * plane;                        // [R][W][MT_CB]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / nbands;
    const int y0 = (blockIdx.x - b * nbands) * R;
    const int cb0 = blockIdx.y * MT_CB;
    const int cbn = min(MT_CB, p.C - cb0);
    const T* img = reinterpret_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.C;
    for (int g = cb0 / p.chunk0; g < p.groups; ++g) {
        const int glo = g * p.chunk0;
        if (glo >= cb0 + cbn) break;
        const int c_lo = max(glo, cb0);
        const int c_hi = min(min(glo + p.chunk0, p.C), cb0 + cbn);
        const int nch = c_hi - c_lo;
        switch (p.ksize[g]) {
            case 1: mixtile_chunk<T, 1>(p, tile, otile, img, g, c_lo, nch, cb0, y0, R, pitch, plane, tid); break;
            case 3: mixtile_chunk<T, 3>(p, tile, otile, img, g, c_lo, nch, cb0, y0, R, pitch, plane, tid); break;
            case 5: mixtile_chunk<T, 5>(p, tile, otile, img, g, c_lo, nch, cb0, y0, R, pitch, plane, tid); break;
            default: mixtile_chunk<T, 7>(p, tile, otile, img, g, c_lo, nch, cb0, y0, R, pitch, plane, tid); break;
        }
    }
    __syncthreads();
    // the block's rows leave as whole 16-byte vectors: cbn * 2 contiguous bytes per pixel
    const int vpp = cbn / 8;
    const int rows_here = min(R, p.H - y0);
    const int total = rows_here * p.W * vpp;
    T* o = reinterpret_cast<T*>(p.out) + (((size_t)b * p.H + y0) * p.W) * p.C + cb0;
    for (int i = tid; i < total; i += MT_NT) {
        const int px = i / vpp, v = i - px * vpp;
        const u32x4 ov = *reinterpret_cast<const u32x4*>(otile + px * MT_CB + v * 8);
        *reinterpret_cast<u32x4*>(o + (size_t)px * p.C + v * 8) = ov;
        if (row_part) {
            // by-product statistics (mlpk_mixshift_nhwc_stats): (sum, sum of squares) of the block's 32 stored channels of this pixel -- plane
            // blockIdx.y of the planar pairs mlpk_stats_finalize_planar reads; the pixel's four lanes (vpp == 4) add in a fixed order
            T e[8];
            __builtin_memcpy(e, &ov, 16);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float f = to_f32(e[q]); s1 += f; s2 = __builtin_fmaf(f, f, s2); }
            s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
            s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
            if (v == 0) {
                const long long row = ((long long)b * p.H + y0) * p.W + px;
                *reinterpret_cast<f32x2*>(row_part + ((long long)blockIdx.y * row_part_ld + row) * 2) = f32x2{s1, s2};
            }
        }
    }
}

// 0 = launched, MT_NOT_TAKEN = shape outside this kernel (the per-chunk band kernel / the gather kernels take it; distinct from every hipError_t >= 0
// and MLPK_E* value), else a HIP error
constexpr int MT_NOT_TAKEN = -1000;
template <typename T>
static int mixshift_tile_launch(const MixShiftArgs& a, hipStream_t s, float* row_part = nullptr, long long row_part_ld = 0, bool query = false) {
    if (sizeof(T) != 2 || (a.C & 7) || (!query && (((uintptr_t)a.x | (uintptr_t)a.out) & 15))) return MT_NOT_TAKEN;
    if ((row_part || query) && (a.C % MT_CB)) return MT_NOT_TAKEN;        // the statistics planes are whole 32-channel blocks
    const int strips = (a.W + 7) / 8;
    if (strips > 7) return MT_NOT_TAKEN;
    int kmax = 1;
    for (int g = 0; g < a.groups; ++g) {
        if (a.ksize[g] != 1 && a.ksize[g] != 3 && a.ksize[g] != 5 && a.ksize[g] != 7) return MT_NOT_TAKEN;
        kmax = a.ksize[g] > kmax ? a.ksize[g] : kmax;
    }
    const int P = kmax / 2;
    int pitch = (strips - 1) * 8 + ((8 + kmax - 1 + 7) / 8) * 8;
    if (pitch < a.W + 2 * P) pitch = a.W + 2 * P;
    pitch = (pitch + 7) / 8 * 8;
    static const int r_env = getenv("MLPK_MIXSHIFT_R") ? atoi(getenv("MLPK_MIXSHIFT_R")) : 0;       // tuning hook
    // rows per band: a thread holds at most MT_TMAX tasks of its >= 8 slots; among the heights whose tile leaves room for two workgroups
    // per CU, the one that stages the fewest rows (bands x (R + halo)) -- measured: 14 rows against 7 on the 28 x 28 and 14 x 14 maps
    // 188 -> 155 and 104 -> 79 us, 8 rows on the 56 x 56 map (one workgroup per CU) 362 -> 530
    const int r_cap = 8 * MT_TMAX / strips < a.H ? 8 * MT_TMAX / strips : a.H;
    auto geometry = [&](int R_, int* plane_) {
        int pl = (R_ + 2 * P) * pitch;                         // pitch is a multiple of 8: whole 16-byte slots
        if (((pl / 8) & 1) == 0) pl += 8;                      // odd number of 16-byte slots per channel plane: conflict-free across channels
        *plane_ = pl;
        return ((size_t)MT_CB * pl + (size_t)R_ * a.W * MT_CB) * sizeof(T);
    };
    int R = 1, plane = 0;
    if (r_env >= 1) {
        R = r_env < r_cap ? r_env : r_cap;
    } else {
        long best = -1;
        for (int r = r_cap; r >= 1; --r) {
            int pl;
            if (geometry(r, &pl) > 79 * 1024 && r > 1) continue;
            const long cost = (long)((a.H + r - 1) / r) * (r + 2 * P);
            if (best < 0 || cost < best) { best = cost; R = r; }
        }
    }
    const size_t lds = geometry(R, &plane);
    if (lds > 150 * 1024) return MT_NOT_TAKEN;
    const int nbands = (a.H + R - 1) / R;
    if ((long long)a.B * nbands > 0x7fffffffll) return MT_NOT_TAKEN;
    if (query) return 0;
    auto kern = mixshift_tile_kernel<T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((unsigned)(a.B * nbands), (unsigned)((a.C + MT_CB - 1) / MT_CB));
    hipLaunchKernelGGL(kern, grid, dim3(MT_NT), lds, s, a, R, pitch, plane, nbands, row_part, row_part_ld);
    return 0;
}

// ================================ Swin-MLP: the whole spatial-MLP half of a block in one kernel ================================
// swin_mlp.py:97-151: x <- x + merge(crop(spatial_mlp(partition(pad(LayerNorm(x)))))) where spatial_mlp is a grouped Conv1d over the ws^2
// positions of a window, one (ws^2 x ws^2) matrix per head, a head = C / heads consecutive channels.  As separate passes this was
// normalise, window gather, per-window transpose, a block-diagonal GEMM (heads x the useful flops, on the generic tile) and the
// scatter-add: five trips over the tensor for 0.5 MFLOP per window.  Here a workgroup owns ONE window:
//   * its ws^2 tokens are read once (16-byte loads from clamped addresses; a padded position is zeros, swin_mlp.py:101-102), normalised
//     with the given row statistics and written TRANSPOSED into LDS as [channel][token] (the matrix cores want a lane's 8 k-values --
//     tokens -- contiguous); the raw values stay in registers for the residual;
//   * head h: out[t][c] = sum_s W_h[t][s] xn[s][c] + b_h[t] for its 32 channels -- v_mfma_f32_32x32x16 with W_h (zero-padded to 64 x 64,
//     fragments straight from global: 8 KiB per head, cache-resident) as A and the LDS image as B: 8 MFMAs per head;
//   * the results go back through the same LDS bytes as [token][channel], are added to the raw values and stored in place.
// Requires C / heads == 32 (every Swin-MLP of the reference) and ws^2 <= 64.
struct SwinArgs {
    void* x;                 // (B, H, W, C) in place
    const float* mean;       // LayerNorm statistics of x's rows
    const float* rstd;
    const float* gamma;
    const float* beta;
    const void* w;           // (heads, 64, 64) zero-padded [t_out][t_in]
    const float* bias;       // (heads, 64)
    int B, H, W, C, ws, pad_t, pad_l, nWy, nWx, heads;
    // round 5 (mlpk_swin_spatial_stats): the LayerNorm statistics of the rows the kernel WRITES -- what norm2 of the block needs (swin_mlp.py:154) --
    // so that no statistics pass follows: a workgroup holds whole rows (every channel of its window's tokens)
    float* out_mean;         // per row of x, or NULL
    float* out_rstd;
    float eps;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T> struct SwinMma;
template <> struct SwinMma<bf16_t> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SwinMma<f16_t> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

#ifndef SW_RELOAD
#define SW_RELOAD 0                                          // (measured: 4.53 vs 4.43 ms on Swin-MLP-T -- fewer registers, more windows per CU, no gain)
#endif
constexpr int SW_NT = 256;
constexpr int SW_TPITCH = 64 * 2 + 16;                       // LDS row of the transposed image: 64 tokens + 16 bytes (bank spread)
constexpr int SW_MAXI = 20;                                  // (token, 8-channel chunk) items per thread: ws^2 * C / 8 / 256 <= 20 (C <= 768 at ws = 7)

// MAXI = (token, chunk) items per thread, NHW = heads per wave: sized to the width (registers decide how many windows a CU holds at once)
// (RELOAD: the raw values are read again for the residual -- cache hits -- instead of being held in registers across the products)
template <typename T, int MAXI, int NHW, bool RELOAD>
__global__ void __launch_bounds__(SW_NT) swin_spatial_kernel(const SwinArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_sw[];
    T* __restrict__ x = reinterpret_cast<T*>(p.x);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.w);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, CV = C / 8, ws = p.ws, T2 = ws * ws;
    const int win = blockIdx.x;
    const int wx = win % p.nWx, wy = (win / p.nWx) % p.nWy, b = win / (p.nWx * p.nWy);
    const int nitem = T2 * CV;
    const int opitch = 2 * C + 16;                           // LDS row of the result image [token][channel]
    // zero the k-padding columns (tokens T2 .. 63) of every channel row
    for (int i = tid; i < C * (64 - T2); i += SW_NT) {
        const int ch = i / (64 - T2), t = T2 + i % (64 - T2);
        *reinterpret_cast<T*>(smem_sw + ch * SW_TPITCH + t * 2) = from_f32<T>(0.f);
    }
    u32x4 raw[RELOAD ? 1 : MAXI];
    const float inv_cv = 1.0f / (float)CV, inv_ws = 1.0f / (float)ws;
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int it = tid + k * SW_NT;
        if (!RELOAD) raw[RELOAD ? 0 : k] = u32x4{0u, 0u, 0u, 0u};
        if (it < nitem) {
            const int t = (int)(((float)it + 0.5f) * inv_cv), cq = it - t * CV;
            const int ty = (int)(((float)t + 0.5f) * inv_ws), tx = t - ty * ws;
            const int yy = wy * ws + ty - p.pad_t, xx = wx * ws + tx - p.pad_l;
            const bool inside = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            const size_t row = ((size_t)b * p.H + (inside ? yy : 0)) * p.W + (inside ? xx : 0);
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + row * C + cq * 8);
            if (!RELOAD) raw[RELOAD ? 0 : k] = v;
            const float mu = p.mean[row], rs = p.rstd[row];
            T e[8];
            __builtin_memcpy(e, &v, 16);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + cq * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + cq * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + cq * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + cq * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xn = (to_f32(e[q]) - mu) * rs * gg[q] + bb[q];
                *reinterpret_cast<T*>(smem_sw + (cq * 8 + q) * SW_TPITCH + t * 2) = inside ? from_f32<T>(xn) : from_f32<T>(0.f);
            }
        }
    }
    __syncthreads();
    // ---- head by head: out (64 x 32) = W_h (64 x 64) . xn_h (64 tokens x 32 channels) ----
    const int n = lane & 31, kh = lane >> 5;
    f32x16 acc[2];
    // every wave walks its heads (wave, wave + 4, ...); the results are written after a barrier that follows ALL MFMA reads of the image
    const int nh_w = (p.heads - wave + SW_NT / 64 - 1) / (SW_NT / 64);      // heads of this wave: wave, wave + 4, ...
    // results of up to NHW heads per wave (24 heads / 4 waves = 6) are kept in registers as packed 16-bit values: 16 per lane and head
    u32x4 res[NHW][2][2];
#pragma unroll
    for (int hi = 0; hi < NHW; ++hi) {
        if (hi < nh_w) {
            const int h = wave + hi * (SW_NT / 64);
            const T* wh = wgt + (size_t)h * 64 * 64;
            acc[0] = acc[1] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 bfrag = *reinterpret_cast<const u32x4*>(smem_sw + (h * 32 + n) * SW_TPITCH + (16 * ks + 8 * kh) * 2);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const u32x4 afrag = *reinterpret_cast<const u32x4*>(wh + (size_t)(32 * mb + n) * 64 + 16 * ks + 8 * kh);
                    acc[mb] = SwinMma<T>::run(afrag, bfrag, acc[mb]);
                }
            }
            // lane (channel n, kh): acc[mb][r] = out[token 32 mb + 8 (r / 4) + 4 kh + (r % 4)][channel 32 h + n]; + bias, round, pack
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                T e[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + h * 64 + 32 * mb + 8 * q4 + 4 * kh);
                    e[4 * q4 + 0] = from_f32<T>(acc[mb][4 * q4 + 0] + bv.x);
                    e[4 * q4 + 1] = from_f32<T>(acc[mb][4 * q4 + 1] + bv.y);
                    e[4 * q4 + 2] = from_f32<T>(acc[mb][4 * q4 + 2] + bv.z);
                    e[4 * q4 + 3] = from_f32<T>(acc[mb][4 * q4 + 3] + bv.w);
                }
                __builtin_memcpy(&res[hi][mb][0], e, 16);
                __builtin_memcpy(&res[hi][mb][1], e + 8, 16);
            }
        }
    }
    __syncthreads();                                         // every wave is done reading the transposed image: its bytes become the result image
#pragma unroll
    for (int hi = 0; hi < NHW; ++hi) {
        if (hi < nh_w) {
            const int h = wave + hi * (SW_NT / 64);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                T e[16];
                __builtin_memcpy(e, &res[hi][mb][0], 16);
                __builtin_memcpy(e + 8, &res[hi][mb][1], 16);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = 32 * mb + 8 * (r >> 2) + 4 * kh + (r & 3);
                    if (t < T2) *reinterpret_cast<T*>(smem_sw + t * opitch + (h * 32 + n) * 2) = e[r];
                }
            }
        }
    }
    __syncthreads();
    // ---- residual and store (the positions inside the map only: the crop of swin_mlp.py:148-149) ----
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int it = tid + k * SW_NT;
        if (it < nitem) {
            const int t = (int)(((float)it + 0.5f) * inv_cv), cq = it - t * CV;
            const int ty = (int)(((float)t + 0.5f) * inv_ws), tx = t - ty * ws;
            const int yy = wy * ws + ty - p.pad_t, xx = wx * ws + tx - p.pad_l;
            if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                const u32x4 yv = *reinterpret_cast<const u32x4*>(smem_sw + t * opitch + cq * 16);
                T a[8], y8[8], o[8];
                const u32x4 rv = RELOAD ? *reinterpret_cast<const u32x4*>(x + (((size_t)b * p.H + yy) * p.W + xx) * C + cq * 8) : raw[RELOAD ? 0 : k];
                __builtin_memcpy(a, &rv, 16);
                __builtin_memcpy(y8, &yv, 16);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = from_f32<T>(to_f32(a[q]) + to_f32(y8[q]));
                u32x4 ov;
                __builtin_memcpy(&ov, o, 16);
                *reinterpret_cast<u32x4*>(x + (((size_t)b * p.H + yy) * p.W + xx) * C + cq * 8) = ov;
                if (p.out_mean) *reinterpret_cast<u32x4*>(smem_sw + t * opitch + cq * 16) = ov;      // (the item's own bytes: read above by this thread only)
            }
        }
    }
    if (!p.out_mean) return;                                 // (workgroup-uniform)
    // ---- statistics of the stored rows: four threads per token, a quarter of the channels each, in a fixed order (deterministic) ----
    __syncthreads();
    {
        const int t = tid >> 2, part = tid & 3;
        float s = 0.f, ss = 0.f;
        const int ty = t / ws, tx = t - ty * ws;
        const int yy = wy * ws + ty - p.pad_t, xx = wx * ws + tx - p.pad_l;
        const bool live = t < T2 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        if (live) {
            const int nq = C / 32;                           // 16-byte chunks per quarter row
            for (int i = 0; i < nq; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem_sw + t * opitch + (part * nq + i) * 16);
                T e[8];
                __builtin_memcpy(e, &v, 16);
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float f = to_f32(e[q]); s += f; ss = __builtin_fmaf(f, f, ss); }
            }
        }
        s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);       // (p0 + p1) + (p2 + p3): the same order in every lane of the quad
        s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
        if (live && part == 0) {
            const float mean = s / (float)C;
            const float var = ss / (float)C - mean * mean;
            const size_t row = ((size_t)b * p.H + yy) * p.W + xx;
            p.out_mean[row] = mean;
            p.out_rstd[row] = 1.0f / __builtin_sqrtf((var > 0.f ? var : 0.f) + p.eps);
        }
    }
}

// Round 6: the same kernel with the LDS traffic rearranged (swin_spatial_q_kernel; MLPK_SWIN_SPATIAL_Q=0 runs the one above).  Measured on the
// one above: 124 us per call on the 14 x 14 x 384 maps = 0.6 TB/s, 65 k cycles per window of 49 x 384 values.  Its loader wrote the transposed
// image with one 2-byte LDS write per value, consecutive lanes = consecutive 8-channel chunks = rows 8 x SW_TPITCH = 288 dwords apart: TWO banks for a
// whole wave (any 16-byte-aligned row pitch gives 0 or 32 mod 64), and the results went back as 16 2-byte writes per lane and product.  Here
//   * an item is (four consecutive tokens, 8-channel chunk), lanes ordered (chunk & 3, token quad, chunk / 4): four 16-byte loads per item (the four
//     chunks of a 64-byte sector side by side), the normalised values leave as ONE 8-byte write per channel, lanes of a quad of chunks 32 banks apart
//     and token quads 2 banks apart (2-way instead of 32-way);
//   * the product is computed transposed -- A = the image (channel x token), B = W_h^T -- so a lane's accumulators are 4 consecutive CHANNELS of one
//     token: an 8-byte write into the [token][channel] result image instead of four 2-byte ones; the same products summed in the same order;
//   * residual / store / statistics as above, on the new item shape.  Bit-equal to the kernel above.
template <typename T, int MAXQ, int NHW>
__global__ void __launch_bounds__(SW_NT) swin_spatial_q_kernel(const SwinArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_sw[];
    T* __restrict__ x = reinterpret_cast<T*>(p.x);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.w);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C, CV = C / 8, ws = p.ws, T2 = ws * ws, NQ = (T2 + 3) >> 2;
    const int win = blockIdx.x;
    const int wx = win % p.nWx, wy = (win / p.nWx) % p.nWy, b = win / (p.nWx * p.nWy);
    const int nitem = CV * NQ;
    const int opitch = 2 * C + 16;                           // LDS row of the result image [token][channel]
    // zero the k-padding columns the loader does not write (token quads NQ .. 15) of every channel row
    {
        const int zq = 16 - NQ;
        for (int i = tid; i < C * zq; i += SW_NT) {
            const int ch = i / zq, q = NQ + i - ch * zq;
            *reinterpret_cast<u32x2*>(smem_sw + ch * SW_TPITCH + q * 8) = u32x2{0u, 0u};
        }
    }
    u32x4 raw[MAXQ][4];
    const float inv_nq = 1.0f / (float)NQ, inv_ws = 1.0f / (float)ws;
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int it = tid + k * SW_NT;
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[k][j] = u32x4{0u, 0u, 0u, 0u};
        if (it < nitem) {
            const int r_ = it >> 2;
            const int cg = (int)(((float)r_ + 0.5f) * inv_nq), tq = r_ - cg * NQ;
            const int cq = cg * 4 + (it & 3);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + cq * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + cq * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + cq * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + cq * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            size_t rows[4];
            bool ins[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * tq + j;
                const int ty = (int)(((float)t + 0.5f) * inv_ws), tx = t - ty * ws;
                const int yy = wy * ws + ty - p.pad_t, xx = wx * ws + tx - p.pad_l;
                ins[j] = t < T2 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                rows[j] = ((size_t)b * p.H + (ins[j] ? yy : 0)) * p.W + (ins[j] ? xx : 0);
                raw[k][j] = *reinterpret_cast<const u32x4*>(x + rows[j] * C + cq * 8);
            }
            T xn[4][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mu = p.mean[rows[j]], rs = p.rstd[rows[j]];
                T e[8];
                __builtin_memcpy(e, &raw[k][j], 16);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = (to_f32(e[q]) - mu) * rs * gg[q] + bb[q];
                    xn[j][q] = ins[j] ? from_f32<T>(v) : from_f32<T>(0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const T four[4] = {xn[0][q], xn[1][q], xn[2][q], xn[3][q]};
                u32x2 pk;
                __builtin_memcpy(&pk, four, 8);
                *reinterpret_cast<u32x2*>(smem_sw + (cq * 8 + q) * SW_TPITCH + tq * 8) = pk;
            }
        }
    }
    __syncthreads();
    // ---- head by head, transposed: out^T (32 channels x 64 tokens) = xn_h^T (32 x 64 tokens) . W_h^T ----
    const int n = lane & 31, kh = lane >> 5;
    f32x16 acc[2];
    const int nh_w = (p.heads - wave + SW_NT / 64 - 1) / (SW_NT / 64);      // heads of this wave: wave, wave + 4, ...
    u32x4 res[NHW][2][2];
#pragma unroll
    for (int hi = 0; hi < NHW; ++hi) {
        if (hi < nh_w) {
            const int h = wave + hi * (SW_NT / 64);
            const T* wh = wgt + (size_t)h * 64 * 64;
            acc[0] = acc[1] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 xfrag = *reinterpret_cast<const u32x4*>(smem_sw + (h * 32 + n) * SW_TPITCH + (16 * ks + 8 * kh) * 2);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const u32x4 wfrag = *reinterpret_cast<const u32x4*>(wh + (size_t)(32 * nb + n) * 64 + 16 * ks + 8 * kh);
                    acc[nb] = SwinMma<T>::run(xfrag, wfrag, acc[nb]);
                }
            }
            // lane (token 32 nb + n, kh): acc[nb][r] = out[token][channel 32 h + 8 (r / 4) + 4 kh + (r % 4)]; + bias of the token, round, pack
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float bv = p.bias[h * 64 + 32 * nb + n];
                T e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = from_f32<T>(acc[nb][r] + bv);
                __builtin_memcpy(&res[hi][nb][0], e, 16);
                __builtin_memcpy(&res[hi][nb][1], e + 8, 16);
            }
        }
    }
    __syncthreads();                                         // every wave is done reading the transposed image: its bytes become the result image
#pragma unroll
    for (int hi = 0; hi < NHW; ++hi) {
        if (hi < nh_w) {
            const int h = wave + hi * (SW_NT / 64);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int t = 32 * nb + n;
                if (t < T2) {
                    char* o = smem_sw + t * opitch + (h * 32 + 4 * kh) * 2;
                    *reinterpret_cast<u32x2*>(o) = u32x2{res[hi][nb][0].x, res[hi][nb][0].y};
                    *reinterpret_cast<u32x2*>(o + 16) = u32x2{res[hi][nb][0].z, res[hi][nb][0].w};
                    *reinterpret_cast<u32x2*>(o + 32) = u32x2{res[hi][nb][1].x, res[hi][nb][1].y};
                    *reinterpret_cast<u32x2*>(o + 48) = u32x2{res[hi][nb][1].z, res[hi][nb][1].w};
                }
            }
        }
    }
    __syncthreads();
    // ---- residual and store (the positions inside the map only: the crop of swin_mlp.py:148-149) ----
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int it = tid + k * SW_NT;
        if (it < nitem) {
            const int r_ = it >> 2;
            const int cg = (int)(((float)r_ + 0.5f) * inv_nq), tq = r_ - cg * NQ;
            const int cq = cg * 4 + (it & 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * tq + j;
                const int ty = (int)(((float)t + 0.5f) * inv_ws), tx = t - ty * ws;
                const int yy = wy * ws + ty - p.pad_t, xx = wx * ws + tx - p.pad_l;
                if (t < T2 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                    const u32x4 yv = *reinterpret_cast<const u32x4*>(smem_sw + t * opitch + cq * 16);
                    T a[8], y8[8], o[8];
                    __builtin_memcpy(a, &raw[k][j], 16);
                    __builtin_memcpy(y8, &yv, 16);
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = from_f32<T>(to_f32(a[q]) + to_f32(y8[q]));
                    u32x4 ov;
                    __builtin_memcpy(&ov, o, 16);
                    *reinterpret_cast<u32x4*>(x + (((size_t)b * p.H + yy) * p.W + xx) * C + cq * 8) = ov;
                    if (p.out_mean) *reinterpret_cast<u32x4*>(smem_sw + t * opitch + cq * 16) = ov;      // (the item's own bytes: read above by this thread only)
                }
            }
        }
    }
    if (!p.out_mean) return;                                 // (workgroup-uniform)
    // ---- statistics of the stored rows: four threads per token, a quarter of the channels each, in a fixed order (deterministic) ----
    __syncthreads();
    {
        const int t = tid >> 2, part = tid & 3;
        float s = 0.f, ss = 0.f;
        const int ty = t / ws, tx = t - ty * ws;
        const int yy = wy * ws + ty - p.pad_t, xx = wx * ws + tx - p.pad_l;
        const bool live = t < T2 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        if (live) {
            const int nq = C / 32;                           // 16-byte chunks per quarter row
            for (int i = 0; i < nq; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem_sw + t * opitch + (part * nq + i) * 16);
                T e[8];
                __builtin_memcpy(e, &v, 16);
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float f = to_f32(e[q]); s += f; ss = __builtin_fmaf(f, f, ss); }
            }
        }
        s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);       // (p0 + p1) + (p2 + p3): the same order in every lane of the quad
        s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
        if (live && part == 0) {
            const float mean = s / (float)C;
            const float var = ss / (float)C - mean * mean;
            const size_t row = ((size_t)b * p.H + yy) * p.W + xx;
            p.out_mean[row] = mean;
            p.out_rstd[row] = 1.0f / __builtin_sqrtf((var > 0.f ? var : 0.f) + p.eps);
        }
    }
}

}  // namespace mlpk

extern "C" int mlpk_swin_spatial_supported(int dtype, int C, int heads, int ws) {
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && heads > 0 && heads <= 24 && C == heads * 32 && ws >= 1 && ws * ws <= 64 &&
           (ws * ws * (C / 8) + SW_NT - 1) / SW_NT <= SW_MAXI;
}

extern "C" int mlpk_swin_spatial_stats(int dtype, void* x, int B, int H, int W, int C, int ws, int pad_t, int pad_l, int Hp, int Wp, int heads,
                                       const float* mean, const float* rstd, const float* gamma, const float* beta, const void* w, const float* bias,
                                       float* out_mean, float* out_rstd, float eps, void* stream);

extern "C" int mlpk_swin_spatial(int dtype, void* x, int B, int H, int W, int C, int ws, int pad_t, int pad_l, int Hp, int Wp, int heads,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, const void* w, const float* bias,
                                 void* stream) {
    return mlpk_swin_spatial_stats(dtype, x, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, mean, rstd, gamma, beta, w, bias, nullptr, nullptr, 0.f, stream);
}

extern "C" int mlpk_swin_spatial_stats(int dtype, void* x, int B, int H, int W, int C, int ws, int pad_t, int pad_l, int Hp, int Wp, int heads,
                                       const float* mean, const float* rstd, const float* gamma, const float* beta, const void* w, const float* bias,
                                       float* out_mean, float* out_rstd, float eps, void* stream) {
    if (!x || !mean || !rstd || !gamma || !beta || !w || !bias) return MLPK_ENULL;
    if ((out_mean != nullptr) != (out_rstd != nullptr)) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || ws <= 0 || Hp % ws || Wp % ws || Hp < H + pad_t || Wp < W + pad_l || pad_t < 0 || pad_l < 0) return MLPK_ESHAPE;
    if (!mlpk_swin_spatial_supported(dtype, C, heads, ws)) return MLPK_ESHAPE;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias) & 15) return MLPK_EALIGN;
    SwinArgs a;
    a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.w = w; a.bias = bias;
    a.B = B; a.H = H; a.W = W; a.C = C; a.ws = ws; a.pad_t = pad_t; a.pad_l = pad_l; a.nWy = Hp / ws; a.nWx = Wp / ws; a.heads = heads;
    a.out_mean = out_mean; a.out_rstd = out_rstd; a.eps = eps;
    const long long nwin = (long long)B * a.nWy * a.nWx;
    if (nwin > 0x7fffffffLL) return MLPK_ESHAPE;
    const int t2 = ws * ws;
    const int lds_t = C * SW_TPITCH, lds_o = t2 * (2 * C + 16);
    const int lds = lds_t > lds_o ? lds_t : lds_o;
    if (lds > 160 * 1024) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const int items = (t2 * (C / 8) + SW_NT - 1) / SW_NT, hpw = (heads + 3) / 4;
    const char* q_env = getenv("MLPK_SWIN_SPATIAL_Q");                 // "0": the round-4 kernel (A/B runs, the bit-equality test)
    if (!(q_env && q_env[0] == '0')) {
        const int qitems = (((t2 + 3) / 4) * (C / 8) + SW_NT - 1) / SW_NT;
#define SWQ_LAUNCH(TT, MAXQ, NHW)                                                                                          \
    do {                                                                                                                   \
        auto k = swin_spatial_q_kernel<TT, MAXQ, NHW>;                                                                     \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);        \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL(k, dim3((unsigned)nwin), dim3(SW_NT), lds, s, a);                                               \
    } while (0)
#define SWQ_PICK(TT)                                                                                                       \
    do {                                                                                                                   \
        if (qitems <= 1 && hpw <= 1) SWQ_LAUNCH(TT, 1, 1);                                                                 \
        else if (qitems <= 2 && hpw <= 2) SWQ_LAUNCH(TT, 2, 2);                                                            \
        else if (qitems <= 3 && hpw <= 3) SWQ_LAUNCH(TT, 3, 3);                                                            \
        else SWQ_LAUNCH(TT, 6, 6);                                                                                         \
    } while (0)
        if (dtype == MLPK_BF16) SWQ_PICK(bf16_t); else SWQ_PICK(f16_t);
#undef SWQ_PICK
#undef SWQ_LAUNCH
        MLPK_LAUNCH_CHECK();
        return 0;
    }
#define SW_LAUNCH(TT, MAXI, NHW)                                                                                           \
    do {                                                                                                                   \
        auto k = swin_spatial_kernel<TT, MAXI, NHW, (MAXI > 3) && SW_RELOAD>;                                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);        \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL(k, dim3((unsigned)nwin), dim3(SW_NT), lds, s, a);                                               \
    } while (0)
#define SW_PICK(TT)                                                                                                        \
    do {                                                                                                                   \
        if (items <= 3 && hpw <= 1) SW_LAUNCH(TT, 3, 1);                                                                   \
        else if (items <= 5 && hpw <= 2) SW_LAUNCH(TT, 5, 2);                                                              \
        else if (items <= 10 && hpw <= 3) SW_LAUNCH(TT, 10, 3);                                                            \
        else SW_LAUNCH(TT, SW_MAXI, 6);                                                                                    \
    } while (0)
    if (dtype == MLPK_BF16) SW_PICK(bf16_t); else SW_PICK(f16_t);
#undef SW_PICK
#undef SW_LAUNCH
    MLPK_LAUNCH_CHECK();
    return 0;
}

static int mixshift_args(mlpk::MixShiftArgs& a, int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift, const int* ksize,
                         const float* w_lr, const float* b_lr, const float* w_td, const float* b_td) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || groups > 8 || groups > C) return MLPK_ESHAPE;
    if (dtype != MLPK_F32 && dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    a.x = x; a.out = out; a.w_lr = w_lr; a.w_td = w_td; a.b_lr = b_lr; a.b_td = b_td;
    a.B = B; a.H = H; a.W = W; a.C = C; a.groups = groups;
    a.chunk0 = (C + groups - 1) / groups;                       // torch.chunk
    if ((C + a.chunk0 - 1) / a.chunk0 != groups) return MLPK_ESHAPE;     // torch.chunk would return fewer chunks
    for (int g = 0; g < 8; ++g) {
        a.shift[g] = g < groups && shift ? shift[g] : 0;
        a.ksize[g] = g < groups ? ksize[g] : 1;
        if (a.ksize[g] < 1 || !(a.ksize[g] & 1) || a.ksize[g] > 15) return MLPK_ESHAPE;
    }
    return 0;
}

extern "C" int mlpk_mixshift_stats_planes(int dtype, int B, int H, int W, int C, int groups, const int* ksize);
extern "C" int mlpk_mixshift_nhwc_stats(int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift,
                                        const int* ksize, const float* w_lr, const float* b_lr, const float* w_td, const float* b_td,
                                        float* row_part, long long row_part_ld, void* stream);

// planes of by-product statistics mlpk_mixshift_nhwc_stats writes for this shape (C / 32), 0 when it does not take the shape
extern "C" int mlpk_mixshift_stats_planes(int dtype, int B, int H, int W, int C, int groups, const int* ksize) {
    if (!ksize || (dtype != MLPK_F16 && dtype != MLPK_BF16)) return 0;
    mlpk::MixShiftArgs a;
    if (mixshift_args(a, dtype, nullptr, nullptr, B, H, W, C, groups, nullptr, ksize, nullptr, nullptr, nullptr, nullptr)) return 0;
    const int rc = dtype == MLPK_F16 ? mlpk::mixshift_tile_launch<mlpk::f16_t>(a, nullptr, nullptr, 0, true)
                                     : mlpk::mixshift_tile_launch<mlpk::bf16_t>(a, nullptr, nullptr, 0, true);
    return rc == 0 ? C / mlpk::MT_CB : 0;
}

extern "C" int mlpk_mixshift_nhwc_stats(int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift,
                                        const int* ksize, const float* w_lr, const float* b_lr, const float* w_td, const float* b_td,
                                        float* row_part, long long row_part_ld, void* stream) {
    if (!x || !out || !shift || !ksize || !w_lr || !b_lr || !w_td || !b_td || !row_part) return MLPK_ENULL;
    if (x == out) return MLPK_ESHAPE;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (row_part_ld < (long long)B * H * W) return MLPK_ESHAPE;
    if ((uintptr_t)row_part & 7) return MLPK_EALIGN;
    mlpk::MixShiftArgs a;
    int rc = mixshift_args(a, dtype, x, out, B, H, W, C, groups, shift, ksize, w_lr, b_lr, w_td, b_td);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rc = dtype == MLPK_F16 ? mlpk::mixshift_tile_launch<mlpk::f16_t>(a, s, row_part, row_part_ld) : mlpk::mixshift_tile_launch<mlpk::bf16_t>(a, s, row_part, row_part_ld);
    if (rc == mlpk::MT_NOT_TAKEN) return MLPK_ESHAPE;        // ask mlpk_mixshift_stats_planes first
    if (rc) return rc;
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_mixshift_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift,
                                  const int* ksize, const float* w_lr, const float* b_lr, const float* w_td, const float* b_td,
                                  void* stream) {
    if (!x || !out || !shift || !ksize || !w_lr || !b_lr || !w_td || !b_td) return MLPK_ENULL;
    if (x == out) return MLPK_ESHAPE;
    MixShiftArgs a;
    {
        const int rc0 = mixshift_args(a, dtype, x, out, B, H, W, C, groups, shift, ksize, w_lr, b_lr, w_td, b_td);
        if (rc0) return rc0;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const bool no_band = getenv("MLPK_MIXSHIFT_NO_BAND") != nullptr;      // tuning hook
    const char* tile_env = getenv("MLPK_MIXSHIFT_TILE");                         // "0": the per-chunk band kernel (A/B runs, the bit-equality test)
    if (!no_band && !(tile_env && tile_env[0] == '0') && dtype != MLPK_F32) {
        const int rc = dtype == MLPK_F16 ? mlpk::mixshift_tile_launch<mlpk::f16_t>(a, s) : mlpk::mixshift_tile_launch<mlpk::bf16_t>(a, s);
        if (rc != mlpk::MT_NOT_TAKEN) {
            if (rc) return rc;
            MLPK_LAUNCH_CHECK();
            return 0;
        }
    }
    if (!no_band && B <= 0x7fffff) {
        int rc;
        switch (dtype) {
            case MLPK_F32: rc = mlpk::mixshift_band_launch<float>(a, s); break;
            case MLPK_F16: rc = mlpk::mixshift_band_launch<mlpk::f16_t>(a, s); break;
            default: rc = mlpk::mixshift_band_launch<mlpk::bf16_t>(a, s); break;
        }
        if (rc != 1) {                                      // 1 = shape outside the tiled kernel: gather kernels below
            if (rc) return rc;
            MLPK_LAUNCH_CHECK();
            return 0;
        }
    }
    const int epv = dtype == MLPK_F32 ? 4 : 8;
    const bool aligned = !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w_lr | (uintptr_t)w_td | (uintptr_t)b_lr | (uintptr_t)b_td) & 15);
    if (C % epv == 0 && aligned) {
        const long long tv = (long long)B * H * W * (C / epv);
        const unsigned gv = (unsigned)((tv + 255) / 256 < 262144 ? (tv + 255) / 256 : 262144);
        switch (dtype) {
            case MLPK_F32: hipLaunchKernelGGL(mlpk::mixshift_vec_kernel<float>, dim3(gv), dim3(256), 0, s, a); break;
            case MLPK_F16: hipLaunchKernelGGL(mlpk::mixshift_vec_kernel<mlpk::f16_t>, dim3(gv), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL(mlpk::mixshift_vec_kernel<mlpk::bf16_t>, dim3(gv), dim3(256), 0, s, a); break;
        }
        MLPK_LAUNCH_CHECK();
        return 0;
    }
    const long long total = (long long)B * H * W * C;
    const unsigned grid = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
    switch (dtype) {
        case MLPK_F32: hipLaunchKernelGGL(mlpk::mixshift_kernel<float>, dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_F16: hipLaunchKernelGGL(mlpk::mixshift_kernel<mlpk::f16_t>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(mlpk::mixshift_kernel<mlpk::bf16_t>, dim3(grid), dim3(256), 0, s, a); break;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

// ---- Swin-MLP window partition / merge (swin_mlp.py:29-60, 122-151) -----------------------
// gather : rows ((b, wy, wx), (iy, ix)) of the ws x ws windows of the zero-padded map (pad_t rows on top, pad_l columns on
//          the left; the bottom/right padding is whatever completes Hp x Wp), 16-byte channel vectors
// scatter: x[b, y, x', :] += yw[window row of padded position (y + pad_t, x' + pad_l)]   (merge + crop + residual)
namespace mlpk {

struct WinArgs {
    void* x;             // (B, H, W, C): source of the gather / destination of the scatter-add
    void* w;             // (B * nWy * nWx * ws * ws, C)
    int B, H, W, C, ws, pad_t, pad_l, nWy, nWx;
};

template <typename T, bool SCATTER>
__global__ void __launch_bounds__(256) window_kernel(const WinArgs p) {
    constexpr int EPV = 16 / (int)sizeof(T);
    T* __restrict__ xm = reinterpret_cast<T*>(p.x);
    T* __restrict__ wn = reinterpret_cast<T*>(p.w);
    const int cv = p.C / EPV;
    if constexpr (!SCATTER) {
        const int64_t total = (int64_t)p.B * p.nWy * p.nWx * p.ws * p.ws * cv;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int c0 = (int)(idx % cv) * EPV;
            int64_t r = idx / cv;
            const int64_t row = r;
            const int ix = (int)(r % p.ws); r /= p.ws;
            const int iy = (int)(r % p.ws); r /= p.ws;
            const int wx = (int)(r % p.nWx); r /= p.nWx;
            const int wy = (int)(r % p.nWy);
            const int64_t b = r / p.nWy;
            const int y = wy * p.ws + iy - p.pad_t, x = wx * p.ws + ix - p.pad_l;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) v = *reinterpret_cast<const u32x4*>(xm + ((b * p.H + y) * p.W + x) * p.C + c0);
            *reinterpret_cast<u32x4*>(wn + row * p.C + c0) = v;
        }
    } else {
        const int64_t total = (int64_t)p.B * p.H * p.W * cv;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int c0 = (int)(idx % cv) * EPV;
            int64_t r = idx / cv;
            const int x = (int)(r % p.W); r /= p.W;
            const int y = (int)(r % p.H);
            const int64_t b = r / p.H;
            const int py = y + p.pad_t, px = x + p.pad_l;
            const int64_t row = (((b * p.nWy + py / p.ws) * p.nWx + px / p.ws) * p.ws + py % p.ws) * p.ws + px % p.ws;
            T a[EPV], u[EPV];
            T* po = xm + ((b * p.H + y) * p.W + x) * p.C + c0;
            *reinterpret_cast<u32x4*>(a) = *reinterpret_cast<const u32x4*>(po);
            *reinterpret_cast<u32x4*>(u) = *reinterpret_cast<const u32x4*>(wn + row * p.C + c0);
#pragma unroll
            for (int e = 0; e < EPV; ++e) a[e] = from_f32<T>(to_f32(a[e]) + to_f32(u[e]));
            *reinterpret_cast<u32x4*>(po) = *reinterpret_cast<const u32x4*>(a);
        }
    }
}

template <bool SCATTER>
static int window_launch(int dtype, void* x, void* w, int B, int H, int W, int C, int ws, int pad_t, int pad_l, int Hp, int Wp, void* stream) {
    if (!x || !w) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ws <= 0 || pad_t < 0 || pad_l < 0) return MLPK_ESHAPE;
    if (Hp < H + pad_t || Wp < W + pad_l || Hp % ws || Wp % ws) return MLPK_ESHAPE;
    if (dtype != MLPK_F32 && dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    const int epv = dtype == MLPK_F32 ? 4 : 8;
    if (C % epv) return MLPK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return MLPK_EALIGN;
    WinArgs a;
    a.x = x; a.w = w; a.B = B; a.H = H; a.W = W; a.C = C; a.ws = ws; a.pad_t = pad_t; a.pad_l = pad_l; a.nWy = Hp / ws; a.nWx = Wp / ws;
    const long long total = SCATTER ? (long long)B * H * W * (C / epv) : (long long)B * Hp * Wp * (C / epv);
    const unsigned grid = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case MLPK_F32: hipLaunchKernelGGL((window_kernel<float, SCATTER>), dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_F16: hipLaunchKernelGGL((window_kernel<f16_t, SCATTER>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((window_kernel<bf16_t, SCATTER>), dim3(grid), dim3(256), 0, s, a); break;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk

extern "C" int mlpk_window_gather(int dtype, const void* x, void* windows, int B, int H, int W, int C, int ws, int pad_t, int pad_l,
                                  int Hp, int Wp, void* stream) {
    return mlpk::window_launch<false>(dtype, const_cast<void*>(x), windows, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, stream);
}

extern "C" int mlpk_window_scatter_add(int dtype, void* x, const void* windows, int B, int H, int W, int C, int ws, int pad_t,
                                       int pad_l, int Hp, int Wp, void* stream) {
    return mlpk::window_launch<true>(dtype, x, const_cast<void*>(windows), B, H, W, C, ws, pad_t, pad_l, Hp, Wp, stream);
}
