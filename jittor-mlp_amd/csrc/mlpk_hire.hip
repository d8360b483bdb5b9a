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
            const bool vec = sizeof(T) == 2 && (p.C % EPV) == 0 && (c0 % EPV) == 0 && (nch % EPV) == 0;
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

}  // namespace mlpk

extern "C" int mlpk_mixshift_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift,
                                  const int* ksize, const float* w_lr, const float* b_lr, const float* w_td, const float* b_td,
                                  void* stream) {
    if (!x || !out || !shift || !ksize || !w_lr || !b_lr || !w_td || !b_td) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || groups > 8 || groups > C) return MLPK_ESHAPE;
    if (x == out) return MLPK_ESHAPE;
    if (dtype != MLPK_F32 && dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    MixShiftArgs a;
    a.x = x; a.out = out; a.w_lr = w_lr; a.w_td = w_td; a.b_lr = b_lr; a.b_td = b_td;
    a.B = B; a.H = H; a.W = W; a.C = C; a.groups = groups;
    a.chunk0 = (C + groups - 1) / groups;                       // torch.chunk
    if ((C + a.chunk0 - 1) / a.chunk0 != groups) return MLPK_ESHAPE;     // torch.chunk would return fewer chunks
    for (int g = 0; g < 8; ++g) {
        a.shift[g] = g < groups ? shift[g] : 0;
        a.ksize[g] = g < groups ? ksize[g] : 1;
        if (a.ksize[g] < 1 || !(a.ksize[g] & 1) || a.ksize[g] > 15) return MLPK_ESHAPE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const bool no_band = getenv("MLPK_MIXSHIFT_NO_BAND") != nullptr;      // tuning hook
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
