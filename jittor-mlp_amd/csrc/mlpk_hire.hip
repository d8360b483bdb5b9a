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
};

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
            const u32x4 v = *reinterpret_cast<const u32x4*>(xn + ((b * p.H + y) * p.W + x) * p.C + c0);
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
            const u32x4 v = *reinterpret_cast<const u32x4*>(xn + ((b * p.H + y) * p.W + x) * p.C + c0);
            T* o = reinterpret_cast<T*>(p.a_w) + ((b * p.H + y) * p.gw + g) * p.ld_w + ww * p.C + c0;
            *reinterpret_cast<u32x4*>(o) = v;
        }
    }
}

struct HireCombineArgs {
    void* x;             // (B, H, W, C): x += y_h(restored) + y_w(restored)
    const void* y_h;     // (B * gh * W, ld_h), columns hh * C + c
    const void* y_w;     // (B * H * gw, ld_w), columns ww * C + c
    int B, H, W, C, h, w, step, Hp, Wp, gh, gw, ld_h, ld_w;
};

template <typename T>
__global__ void __launch_bounds__(256) hire_combine_kernel(const HireCombineArgs p) {
    constexpr int EPV = 16 / (int)sizeof(T);
    T* __restrict__ xo = reinterpret_cast<T*>(p.x);
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
        *reinterpret_cast<u32x4*>(a) = *reinterpret_cast<const u32x4*>(po);
        *reinterpret_cast<u32x4*>(u) = *reinterpret_cast<const u32x4*>(ph);
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(pw);
#pragma unroll
        for (int e = 0; e < EPV; ++e) a[e] = from_f32<T>(to_f32(a[e]) + (to_f32(u[e]) + to_f32(v[e])));
        *reinterpret_cast<u32x4*>(po) = *reinterpret_cast<const u32x4*>(a);
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

extern "C" int mlpk_hire_gather(int dtype, const void* xn, void* a_h, void* a_w, int B, int H, int W, int C, int h, int w,
                                int step, int ld_h, int ld_w, void* stream) {
    if (!xn || !a_h || !a_w) return MLPK_ENULL;
    HireArgs a;
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

extern "C" int mlpk_hire_combine(int dtype, void* x, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                                 int step, int ld_h, int ld_w, void* stream) {
    if (!x || !y_h || !y_w) return MLPK_ENULL;
    HireCombineArgs a;
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
