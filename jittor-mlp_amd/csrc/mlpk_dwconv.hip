// ConvMixer depthwise half, LDS-tiled:  out = x + BN(gelu(dwconv_kxk_same(x) + b))   (conv_mixer.py:5-11, 24-28)
// on channel-last activations.  81 taps per output at k = 9 make this VALU-bound, not HBM-bound, once every
// input element is fetched once -- so the design is about feeding the vector ALUs:
//   * workgroup = one image x CT channels x the whole H x W plane (+ halo), staged in LDS TRANSPOSED to
//     [channel][row][col] so that a lane can read 16 consecutive columns of its channel with two 16-byte reads;
//     the channel plane pitch is an odd number of 16-byte slots -> conflict-free for lanes on different channels;
//   * thread = one channel (its k*k taps stay in registers for the whole tile) and strips of 8 outputs along x:
//     per tap row 2 LDS reads feed 8*k FMAs (sliding window in registers);
//   * the residual comes from the same LDS tile; bias, exact GELU, BatchNorm(eval) scale/shift fused in the store.
// Global traffic = x read once + out written once (the stencil re-reads stay in LDS).
#include "mlpk_common.h"

namespace mlpk {

template <typename T> struct DwCfg;
template <> struct DwCfg<float> { static constexpr int CT = 16; };
template <> struct DwCfg<f16_t> { static constexpr int CT = 32; };
template <> struct DwCfg<bf16_t> { static constexpr int CT = 32; };

// 512 threads: the tile's LDS footprint allows one workgroup per CU, and a SIMD holding a single wave issues a VALU
// instruction only every ~5.5 cycles against ~2.7 with two (tools/ubench/issue_rate.hip) -- 8 waves, not 4.
constexpr int DW_NT = 512;

template <typename T, int KS>
__global__ void __launch_bounds__(DW_NT) dwconv_lds_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W,
                                                         int C, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ bns, const float* __restrict__ bnh,
                                                         int pitch, int plane) {
    constexpr int CT = DwCfg<T>::CT;
    constexpr int P = (KS - 1) / 2;
    constexpr int STRIP = 8;
    constexpr int EPV = 16 / (int)sizeof(T);           // elements per 16-byte vector
    constexpr int WIN = STRIP + KS - 1;                // input window of a strip (<= 16 elements)
    constexpr int NV = (WIN + EPV - 1) / EPV;          // 16-byte reads per window row
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);          // [CT][H + 2P][pitch], plane stride `plane` elements
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int c0 = blockIdx.y * CT;
    const int HP = H + 2 * P;

    // ---- zero the tile (halo + padding), then scatter the image in: 16-byte global reads along channels ----
    {
        const int nvec = CT * plane / EPV;
        u32x4* t4 = reinterpret_cast<u32x4*>(tile);
        for (int i = tid; i < nvec; i += DW_NT) t4[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    {
        constexpr int CV = CT / EPV;                    // channel vectors per pixel
        const int total = H * W * CV;
        for (int i = tid; i < total; i += DW_NT) {
            const int cv = i % CV;
            const int px = i / CV;
            const int xx = px % W, yy = px / W;
            const int c = c0 + cv * EPV;
            if (c < C) {                                // C % EPV == 0 is checked by the launcher
                const u32x4 raw = *reinterpret_cast<const u32x4*>(x + (((size_t)b * H + yy) * W + xx) * C + c);
                T e[EPV];
                __builtin_memcpy(e, &raw, 16);
                T* dst = tile + (cv * EPV) * plane + (yy + P) * pitch + (xx + P);
#pragma unroll
                for (int k = 0; k < EPV; ++k) dst[k * plane] = e[k];
            }
        }
    }
    __syncthreads();

    // ---- compute: thread -> channel cl = tid % CT, tasks (row, strip) = tid / CT + k * (DW_NT / CT) ----
    const int cl = tid % CT;
    const int c = c0 + cl;
    if (c >= C) return;
    // taps as register PAIRS (tap 2q, tap 2q+1): v_pk_fma_f32 broadcasts either half to both of its lanes through
    // op_sel, so no tap is duplicated (81 duplicated pairs do not fit next to the window at two waves per SIMD)
    constexpr int NT2 = (KS * KS + 1) / 2;
    f32x2 wt2[NT2];
#pragma unroll
    for (int q = 0; q < NT2; ++q) {
        wt2[q].x = w[(size_t)(2 * q) * C + c];
        wt2[q].y = 2 * q + 1 < KS * KS ? w[(size_t)(2 * q + 1) * C + c] : 0.f;
    }
    const float bs = bias ? bias[c] : 0.f;
    const float sc = bns ? bns[c] : 1.f, sh = bnh ? bnh[c] : 0.f;
    const int strips = (W + STRIP - 1) / STRIP;
    const int ntask = H * strips;
    const T* cplane = tile + cl * plane;
    for (int task = tid / CT; task < ntask; task += DW_NT / CT) {
        const int st = task % strips, y = task / strips;
        const int x0 = st * STRIP;
        // outputs as pairs (o, o+1); the window in two views, pairs starting at even and at odd columns
        f32x2 acc2[STRIP / 2];
#pragma unroll
        for (int o = 0; o < STRIP / 2; ++o) acc2[o] = f32x2{bs, bs};
        float centre[STRIP];
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
            // window row: padded columns x0 .. x0 + WIN - 1 of padded row y + dy (16-byte aligned: pitch % EPV == 0)
            const T* row = cplane + (y + dy) * pitch + x0;
            float win[NV * EPV + 1];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(row + v * EPV);
                T e[EPV];
                __builtin_memcpy(e, &raw, 16);
#pragma unroll
                for (int k = 0; k < EPV; ++k) win[v * EPV + k] = to_f32(e[k]);
            }
            win[NV * EPV] = 0.f;
            if (dy == P) {
#pragma unroll
                for (int o = 0; o < STRIP; ++o) centre[o] = win[o + P];
            }
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const int tap = dy * KS + dx;
#pragma unroll
                for (int o = 0; o < STRIP / 2; ++o) {
                    const f32x2 xin = {win[2 * o + dx], win[2 * o + dx + 1]};
                    if (tap & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc2[o]) : "v"(xin), "v"(wt2[tap >> 1]));
                    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[o]) : "v"(xin), "v"(wt2[tap >> 1]));
                }
            }
        }
        float acc[STRIP];
#pragma unroll
        for (int o = 0; o < STRIP / 2; ++o) { acc[2 * o] = acc2[o].x; acc[2 * o + 1] = acc2[o].y; }
        if constexpr (sizeof(T) == 2) {
            // 16-bit storage: the packed rcp-only GELU on four interleaved pairs (17 instructions per pair)
            f32x2 g[STRIP / 2];
#pragma unroll
            for (int o = 0; o < STRIP / 2; ++o) g[o] = f32x2{acc[2 * o], acc[2 * o + 1]};
            gelu_pk_n<STRIP / 2>(g);
#pragma unroll
            for (int o = 0; o < STRIP / 2; ++o) { acc[2 * o] = g[o].x; acc[2 * o + 1] = g[o].y; }
        } else {
#pragma unroll
            for (int o = 0; o < STRIP; ++o) acc[o] = gelu_f(acc[o]);
        }
#pragma unroll
        for (int o = 0; o < STRIP; ++o) {
            const int xx = x0 + o;
            if (xx < W) out[(((size_t)b * H + y) * W + xx) * C + c] = from_f32<T>(centre[o] + acc[o] * sc + sh);
        }
    }
}

// "the staged plane does not fit the LDS": distinct from every hipError_t (>= 0) and MLPK_E* (-1 .. -5) value, so a real
// HIP error can never be mistaken for it and swallowed by the fallback
static constexpr int DW_NOFIT = -1000;

template <typename T>
static int dwconv_lds_launch(int k, const void* x, void* out, int B, int H, int W, int C, const float* w, const float* bias,
                             const float* bns, const float* bnh, hipStream_t s) {
    constexpr int CT = DwCfg<T>::CT;
    constexpr int EPV = 16 / (int)sizeof(T);
    const int P = (k - 1) / 2;
    // row pitch: W + 2P columns, plus room for the last strip's 16-element window, rounded to whole vectors
    const int strips = (W + 7) / 8;
    int pitch = (strips - 1) * 8 + ((8 + k - 1 + EPV - 1) / EPV) * EPV;
    if (pitch < W + 2 * P) pitch = W + 2 * P;
    pitch = (pitch + EPV - 1) / EPV * EPV;
    int plane = (H + 2 * P) * pitch;                    // elements; make it an ODD number of 16-byte slots
    plane = (plane + EPV - 1) / EPV * EPV;
    if (((plane / EPV) & 1) == 0) plane += EPV;
    const size_t lds = (size_t)CT * plane * sizeof(T);
    if (lds > 160 * 1024) return DW_NOFIT;              // caller falls back (a value no hipError_t / MLPK_E* takes)
    const dim3 grid((unsigned)B, (unsigned)((C + CT - 1) / CT));
#define DW_CASE(KS)                                                                                                    \
    case KS: {                                                                                                         \
        auto kern = dwconv_lds_kernel<T, KS>;                                                                          \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(DW_NT), lds, s, (const T*)x, (T*)out, B, H, W, C, w, bias, bns, bnh, pitch, plane); \
        break;                                                                                                         \
    }
    switch (k) {
        DW_CASE(3) DW_CASE(5) DW_CASE(7) DW_CASE(9)
        default: return MLPK_ESHAPE;
    }
#undef DW_CASE
    MLPK_LAUNCH_CHECK();
    return 0;
}

// Sparse-MLP depthwise step (sparse_mlp.py:84-87): out = x + dwconv_same(pre_scale * x + pre_shift) + bias, zero padding
// applied AFTER the per-channel affine (BatchNorm in eval mode in front of the convolution: a padded tap contributes
// nothing, so the shift cannot be folded into the bias near the border).  Channel-last, one thread per pixel and
// 16-byte channel vector, k*k neighbour vectors from L1/L2; HBM-bound (read + write of x once).
template <typename T>
__global__ void __launch_bounds__(256) dwconv_affine_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C,
                                                            int k, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ ps, const float* __restrict__ ph) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const int cv = C / EPV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W * cv;
    if (idx >= total) return;
    const int c0 = (int)(idx % cv) * EPV;
    const long long pix = idx / cv;
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    const int P = (k - 1) / 2;
    // per-channel parameters as 16-byte vectors (EPV scalar loads per tap made the texture-address unit the bottleneck)
    float a[EPV], sh[EPV], acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; e += 4) {
        const f32x4 av = ps ? *reinterpret_cast<const f32x4*>(ps + c0 + e) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sv = ph ? *reinterpret_cast<const f32x4*>(ph + c0 + e) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c0 + e) : f32x4{0.f, 0.f, 0.f, 0.f};
        a[e] = av.x; a[e + 1] = av.y; a[e + 2] = av.z; a[e + 3] = av.w;
        sh[e] = sv.x; sh[e + 1] = sv.y; sh[e + 2] = sv.z; sh[e + 3] = sv.w;
        acc[e] = bv.x; acc[e + 1] = bv.y; acc[e + 2] = bv.z; acc[e + 3] = bv.w;
    }
    T ctr[EPV];
    for (int dy = 0; dy < k; ++dy) {
        const int yy = yh + dy - P;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = xw + dx - P;
            if (xx < 0 || xx >= W) continue;
            T v[EPV];
            *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + ((b * H + yy) * W + xx) * C + c0);
            if (dy == P && dx == P) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) ctr[e] = v[e];
            }
            const float* wt = w + (size_t)(dy * k + dx) * C + c0;
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + e);
                acc[e] = fmaf(wv.x, fmaf(a[e], to_f32(v[e]), sh[e]), acc[e]);
                acc[e + 1] = fmaf(wv.y, fmaf(a[e + 1], to_f32(v[e + 1]), sh[e + 1]), acc[e + 1]);
                acc[e + 2] = fmaf(wv.z, fmaf(a[e + 2], to_f32(v[e + 2]), sh[e + 2]), acc[e + 2]);
                acc[e + 3] = fmaf(wv.w, fmaf(a[e + 3], to_f32(v[e + 3]), sh[e + 3]), acc[e + 3]);
            }
        }
    }
    T o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = from_f32<T>(to_f32(ctr[e]) + acc[e]);
    *reinterpret_cast<u32x4*>(out + pix * C + c0) = *reinterpret_cast<const u32x4*>(o);
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_dwconv_affine_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int k, const float* w,
                                       const float* bias, const float* pre_scale, const float* pre_shift, void* stream) {
    if (!x || !out || !w) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || !(k & 1)) return MLPK_ESHAPE;
    if (x == out) return MLPK_ESHAPE;                    // a stencil cannot run in place
    if (dtype != MLPK_F32 && dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    const int epv = dtype == MLPK_F32 ? 4 : 8;
    if (C % epv) return MLPK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)w & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)pre_scale & 15) ||
        ((uintptr_t)pre_shift & 15))
        return MLPK_EALIGN;
    const long long total = (long long)B * H * W * (C / epv);
    if ((total + 255) / 256 > 0x7fffffffLL) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((total + 255) / 256));
    switch (dtype) {
        case MLPK_F32:
            hipLaunchKernelGGL(dwconv_affine_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)out, B, H, W, C, k, w, bias, pre_scale, pre_shift);
            break;
        case MLPK_F16:
            hipLaunchKernelGGL(dwconv_affine_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, (f16_t*)out, B, H, W, C, k, w, bias, pre_scale, pre_shift);
            break;
        default:
            hipLaunchKernelGGL(dwconv_affine_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, C, k, w, bias, pre_scale, pre_shift);
            break;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_dwconv_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int k, const float* w,
                                const float* bias, const float* bn_scale, const float* bn_shift, void* stream) {
    if (!x || !out || !w) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return MLPK_ESHAPE;
    if (x == out) return MLPK_ESHAPE;                    // a stencil cannot run in place
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int es = dtype == MLPK_F32 ? 4 : 2;
    const bool fast_ok = (C % (16 / es) == 0) && (((uintptr_t)x & 15) == 0) && B <= 0x7fffffff;
    if (fast_ok) {
        int rc;
        switch (dtype) {
            case MLPK_F32: rc = dwconv_lds_launch<float>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s); break;
            case MLPK_F16: rc = dwconv_lds_launch<f16_t>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s); break;
            case MLPK_BF16: rc = dwconv_lds_launch<bf16_t>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s); break;
            default: return MLPK_EDTYPE;
        }
        if (rc != DW_NOFIT) return rc;                   // tile does not fit the LDS: use the generic kernel
    }
    return mlpk_dwconv_direct(dtype, x, out, B, H, W, C, k, w, bias, bn_scale, bn_shift, stream);
}
