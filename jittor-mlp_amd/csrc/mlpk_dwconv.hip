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
#include <stdlib.h>
#include <type_traits>

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
            gelu_pk_n<T, STRIP / 2>(g);
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
        default: return DW_NOFIT;                        // other sizes: the generic kernel (odd k <= 13, round 5)
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

// The same for a compile-time k (3: Sparse-MLP): the k * k neighbour vectors and their taps are loaded UNCONDITIONALLY (a position outside
// the map reads the centre pixel and contributes zero) -- behind `continue` each load waited for the one before it, 27 memory round
// trips in a row per output vector (the depthwise kernel's finding, profiles/r04_dwconv_variants.txt).
template <typename T, int KS>
__global__ void __launch_bounds__(256) dwconv_affine_k_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ ps, const float* __restrict__ ph) {
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int P = (KS - 1) / 2;
    const int cv = C / EPV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W * cv;
    if (idx >= total) return;
    const int c0 = (int)(idx % cv) * EPV;
    const long long pix = idx / cv;
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
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
    const T* ctr_p = x + pix * C + c0;
    u32x4 v[KS * KS];
    bool ok[KS * KS];
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
            const int yy = yh + dy - P, xx = xw + dx - P;
            ok[dy * KS + dx] = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            v[dy * KS + dx] = *reinterpret_cast<const u32x4*>(ok[dy * KS + dx] ? ctr_p + ((ptrdiff_t)(dy - P) * W + (dx - P)) * C : ctr_p);
        }
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) {
        T e8[EPV];
        __builtin_memcpy(e8, &v[t], 16);
        const float* wt = w + (size_t)t * C + c0;
#pragma unroll
        for (int e = 0; e < EPV; e += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + e);
            const float w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tap = ok[t] ? fmaf(a[e + q], to_f32(e8[e + q]), sh[e + q]) : 0.f;
                acc[e + q] = fmaf(w4[q], tap, acc[e + q]);
            }
        }
    }
    T c8[EPV], o[EPV];
    __builtin_memcpy(c8, &v[P * KS + P], 16);
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = from_f32<T>(to_f32(c8[e]) + acc[e]);
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
    if (k == 3 && dtype != MLPK_F32) {
        if (dtype == MLPK_F16)
            hipLaunchKernelGGL((dwconv_affine_k_kernel<f16_t, 3>), grid, dim3(256), 0, s, (const f16_t*)x, (f16_t*)out, B, H, W, C, w, bias, pre_scale, pre_shift);
        else
            hipLaunchKernelGGL((dwconv_affine_k_kernel<bf16_t, 3>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, C, w, bias, pre_scale, pre_shift);
        MLPK_LAUNCH_CHECK();
        return 0;
    }
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

// ---------------------------------------------------------------------------------------------------------------------
// Depthwise k x k convolution on the MATRIX pipe (16-bit dtypes, maps of up to 32 x 32 pixels: every ConvMixer of the
// reference at 224^2 with patch >= 7).  For one channel and one tap row i the stencil is a banded-Toeplitz product: with the
// input plane stored with a 4-pixel zero halo, a 16 x 16 block of outputs at (16 ty, 16 tx) is
//     out[16ty + m][16tx + n] += sum_{kk < 32}  plane[16ty + m + i - P + 4][16tx + kk] * Tb_i[kk][n],    Tb_i[kk][n] = w[i][kk - 4 - n + P]
// (zero outside 0 <= kk - 4 - n + P < k), i.e. v_mfma_f32_16x16x32 with A = 16 rows x 32 consecutive plane columns and a B
// operand that depends on the tap row ONLY -- not on the block -- so one channel needs k fragments (4 registers each) and
// 4 k MFMAs per 32 x 32 plane instead of k^2 = 81 VALU FMAs per output: 3.2x the arithmetic at 16x the rate, which takes the
// kernel off the VALU wall (1.66 ms per ConvMixer-1536/20 layer, 39 TFLOP/s fp32) towards the HBM time of its 1.6 GB.
// The taps are rounded to the activation dtype (they are MFMA operands); products and sums are fp32.
//   * workgroup = 8 waves x 4 channels x a range of images.  32 channels = 64 contiguous, 64-byte-ALIGNED bytes of every
//     channel-last pixel: at 24 channels (48-byte pieces, what the register file holds at k = 9) the same data movement alone
//     took 1.12 ms per ConvMixer-1536/20 layer against 0.40 ms at 32 (tools/gpu_dwconv_variants.sh, profiles/r04_dwconv_*):
//     pieces that end inside a 64-byte sector reach HBM as partial writes;
//   * a wave keeps the k tap fragments of CREG of its channels in registers for the whole range; those of the other 4 - CREG
//     live in the LDS as a table of the 16 distinct lane patterns (+ one of zeros) per (channel, tap row) -- a fragment only
//     depends on 8 (lane >> 4) - (lane & 15) -- and are read into registers channel by channel;
//   * LDS = 32 planes (+ one all-zero plane for the fragment lanes that only meet zero taps) of [40 rows][40 columns] (80-byte rows)
//     zeroed once; a plane is input AND output: when a channel's MFMAs are done its wave applies bias, GELU, BatchNorm
//     scale / shift and the residual (the plane's own centre value) and writes the result in place;
//   * a 16-byte fragment is read as TWO ds_read_b64, odd (lane >> 4) groups upper half first (the taps' fragments are stored in
//     the same order): ds_read_b128 serves the lanes in groups of 16 that mix two (lane >> 4) values, whose 16-byte slots collide
//     for any row pitch (2-way: SQ_LDS_BANK_CONFLICT was 48 % of the LDS cycles); ds_read_b64 serves 32 lanes per cycle from
//     64 banks and the half swap puts the two (lane >> 4) values of a group on disjoint banks;
//   * between two images every thread stores its 4 pixel-pair chunks (8 channels x 2 pixels, two 16-byte stores) and puts the
//     next image's chunks (prefetched into registers during the MFMA phase) into the SAME dwords -- no double buffer, two
//     barriers per image.  Every 8th plane starts 32 bytes later: the 4 chunks of a pixel pair (planes 8 apart) then fall
//     on different banks in these 4-byte transposing accesses;
//   * workgroups are dealt round-robin to the 8 XCDs, each with its own L2: the 8 workgroups of 8 NEIGHBOURING channel groups
//     (4 whole 128-byte lines) go to ONE XCD at the same time, whose L2 then fetches a line once and writes whole lines back.
#ifndef DWM_LOADS
#define DWM_LOADS 3                                                 // experiment knobs of tools/gpu_dwconv_variants.sh
#endif
#ifndef DWM_CREG9
#define DWM_CREG9 2
#endif
#ifndef DWM_CREG7
#define DWM_CREG7 3
#endif
#ifndef DWM_XCD
#define DWM_XCD 1
#endif
#ifndef DWM_STAGES
#define DWM_STAGES 2
#endif
#ifndef DWM_SKIP
#define DWM_SKIP 0                                                  // 1: no MFMA phase; 2: no global loads / stores; 4: no LDS transposes
#endif
constexpr int DWM_PITCH = 80;
constexpr int DWM_PLANE = 40 * DWM_PITCH;
constexpr int DWM_NPAT = 17;                                        // lane patterns of a tap fragment: 16 + all zeros
__host__ __device__ constexpr int dwm_plane_off(int c) { return c * DWM_PLANE + (c >> 3) * 32; }

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// A 16-byte fragment as two ds_read_b64 issued by hand (see the kernel's header): registers 0, 1 from `lo` + IMM, registers 2, 3
// from `hi` + IMM, where lo / hi = the fragment's address + h0 / + 8 - h0.  The reads are asynchronous: the values may only be
// used after a dwm_wait<N> that names them (N = LDS reads issued after theirs).  Inline assembly because hipcc pairs plain 8-byte
// reads of two fragments into ds_read2_b64 (half the LDS rate) and then moves the halves between registers.
struct DwmFrag { u32x2 lo, hi; };
template <int IMM> static __device__ __forceinline__ void dwm_issue(DwmFrag& f, unsigned lo, unsigned hi) {
    asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4" : "=&v"(f.lo), "=&v"(f.hi) : "v"(lo), "v"(hi), "i"(IMM));
}
template <int N> static __device__ __forceinline__ void dwm_wait(DwmFrag& a, DwmFrag& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "i"(N));
}
static __device__ __forceinline__ void dwm_tie(DwmFrag& f) { asm volatile("" : "+v"(f.lo), "+v"(f.hi)); }   // "defined from here on"
static __device__ __forceinline__ u32x4 dwm_val(const DwmFrag& f) { return u32x4{f.lo.x, f.lo.y, f.hi.x, f.hi.y}; }
template <int I, int N, typename F> static __device__ __forceinline__ void dwm_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dwm_for<I + 1, N>(f);
    }
}

template <typename T, int KS, int CPW, int CREG, int NW, bool FULL>
__global__ void __launch_bounds__(NW * 64) dwconv_mfma_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ bns, const float* __restrict__ bnh, int img_per_wg) {
    constexpr int P = KS / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int CG = NW * CPW;                                    // channels per workgroup (a multiple of 8)
    constexpr int NT = NW * 64;
    constexpr int CH8 = CG / 8;                                     // 16-byte chunks per pixel
    constexpr int NSLOT = 512 * CH8 / NT;                           // (pixel pair, chunk) slots per thread
    constexpr int CLDS = CPW - CREG;                                // channels of a wave whose fragments live in the LDS
    constexpr int PLANES_END = dwm_plane_off(CG) + DWM_PLANE;       // planes + the zero plane; the fragment tables follow
    constexpr int TBL_WAVE = CLDS * KS * DWM_NPAT * 16;
    static_assert(CG % 8 == 0 && (512 * CH8) % NT == 0 && CREG >= 1 && CREG <= CPW, "workgroup geometry");
    static_assert(KS * KS * CG * 4 <= PLANES_END, "tap staging fits the planes");
    int gx = blockIdx.x, gy = blockIdx.y;
#if DWM_XCD
    if (gridDim.x % 8 == 0 && (gridDim.x * gridDim.y) % 64 == 0) {     // (else the plain order: the deal below needs whole rounds)
        const int id = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = id & 7, l = id >> 3;                        // l: this XCD's l-th workgroup
        const int oct = (l >> 3) * 8 + xcd, octs = gridDim.x >> 3;  // octet = 8 neighbouring channel groups of one image range
        gx = (oct % octs) * 8 + (l & 7);
        gy = oct / octs;
    }
#endif
    const int c0 = gx * CG;
    const int b0 = gy * img_per_wg;
    const int b1 = b0 + img_per_wg < B ? b0 + img_per_wg : B;
    if (b0 >= B) return;

    // ---- the workgroup's taps through the LDS first: [k * k][CG] floats, coalesced, before the planes are zeroed.  (Built from global
    // memory directly, the 8 k CPW scalar loads of a lane came out of hipcc one at a time, an s_waitcnt vmcnt(0) between them.) ----
    float* const wl = reinterpret_cast<float*>(smem);
    for (int idx = tid; idx < KS * KS * CG; idx += NT) {
        const int t = idx / CG, cc = idx - t * CG;
        const bool live = c0 + cc < C;
        const float tap = w[(size_t)t * C + (live ? c0 + cc : C - 1)];
        wl[idx] = live ? tap : 0.f;
    }
    __syncthreads();

    // ---- tap fragments: A operand of 16x16x32 = lane (n = lane & 15, kk = (lane >> 4) * 8 + e); element e of the fragment is
    // tap d = E + e of the row with E = 8 kq - n + P - 4 (zero outside 0 <= d < k).  Odd kq: halves swapped (see dwm_read16). ----
    const int n = lane & 15;
    const int kq = lane >> 4;
    const int h0 = (kq & 1) * 8;
    const int eswap = (kq & 1) * 4;
    u32x4 tf[CREG > 0 ? CREG : 1][KS];
    float bz[CPW], sc[CPW], sh[CPW];                                // wave-uniform: kept in SGPRs
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int c = c0 + wave * CPW + j;
        const bool live = c < C;
        if (j < CREG) {
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                T e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    // (unconditional reads from a clamped address + a select: no branch per tap)
                    const int d = kq * 8 + (q ^ eswap) - 4 - n + P;
                    const int dc = d < 0 ? 0 : (d >= KS ? KS - 1 : d);
                    const float tap = wl[(i * KS + dc) * CG + wave * CPW + j];
                    e[q] = from_f32<T>((d >= 0 && d < KS) ? tap : 0.f);
                }
                __builtin_memcpy(&tf[j][i], e, 16);
            }
        }
        const int cl = live ? c : C - 1;
        const float bzv = bias ? bias[cl] : 0.f, scv = bns ? bns[cl] : 1.f, shv = bnh ? bnh[cl] : 0.f;
        bz[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, live ? bzv : 0.f)));
        sc[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, live ? scv : 1.f)));
        sh[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, live ? shv : 0.f)));
    }
    // the table of this wave: [CLDS channels][k tap rows][17 patterns] x 16 bytes; pattern p < 16 holds taps d = p - 7 + e, e < 8
    char* const tbl = smem + PLANES_END + wave * TBL_WAVE;
    for (int idx = lane; idx < CLDS * KS * DWM_NPAT; idx += 64) {
        const int ji = idx / DWM_NPAT, pat = idx - ji * DWM_NPAT;
        const int jl = ji / KS, i = ji - jl * KS;
        T e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int d = pat - 7 + q;
            const int dc = d < 0 ? 0 : (d >= KS ? KS - 1 : d);
            const float tap = wl[(i * KS + dc) * CG + wave * CPW + CREG + jl];
            e[q] = from_f32<T>((pat < 16 && d >= 0 && d < KS) ? tap : 0.f);
        }
        u32x4 v;
        __builtin_memcpy(&v, e, 16);
        *reinterpret_cast<u32x4*>(tbl + idx * 16) = v;
    }
    const int epat = 8 * kq - n + P - 4;                            // this lane's pattern: E in [-7, 8] -> E + 7, else the zeros
    const unsigned tf_rd = (unsigned)(uintptr_t)(tbl + ((epat >= -7 && epat <= 8) ? epat + 7 : 16) * 16);
    const unsigned tf_lo = tf_rd + h0, tf_hi = tf_rd + 8 - h0;
    DwmFrag tfl[KS];                                                // the fragments of the channel at work when they are the table's
    __syncthreads();                                                // every wave has its fragments: the taps' bytes become planes
    for (int i = tid * 16; i < PLANES_END; i += NT * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0u, 0u, 0u, 0u};

    // ---- staging slots of this thread: slot = tid + NT q, q < NSLOT -> (pixel pair slot / CH8, 8-channel chunk slot % CH8):
    // consecutive lanes move the consecutive chunks of one pixel pair ----
    auto slot_of = [&](const int q, int& cq, int& yy, int& xx) {
        const int slot = tid + NT * q;
        const int pp = slot / CH8;                                  // pixel pair 0 .. 511
        cq = slot - pp * CH8;
        yy = pp >> 4;
        xx = (pp & 15) * 2;
    };
    u32x4 r0[NSLOT], r1[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) r0[q] = r1[q] = u32x4{0u, 0u, 0u, 0u};
    // The loads are UNCONDITIONAL (an address clamped to the tensor's first bytes and a select): behind a branch each of them gets
    // an s_waitcnt vmcnt(0) in front from hipcc, one memory latency after the other.  FULL (32 x 32 map, whole channel groups):
    // no conditions at all.
    auto gload_slot = [&](const int b, const int q) {
        int cq, yy, xx;
        slot_of(q, cq, yy, xx);
        const T* src = x + (((size_t)b * H + yy) * W + xx) * C + c0 + 8 * cq;
        if (FULL) {
            r0[q] = *reinterpret_cast<const u32x4*>(src);
            r1[q] = *reinterpret_cast<const u32x4*>(src + C);
        } else {
            const bool ok0 = c0 + 8 * cq < C && yy < H && xx < W;   // (C % 8 == 0: a chunk is whole or absent)
            const bool ok1 = ok0 && xx + 1 < W;
            const u32x4 a = *reinterpret_cast<const u32x4*>(ok0 ? src : x);
            const u32x4 bb = *reinterpret_cast<const u32x4*>(ok1 ? src + C : x);
            r0[q] = ok0 ? a : u32x4{0u, 0u, 0u, 0u};
            r1[q] = ok1 ? bb : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto gload = [&](const int b) {
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) gload_slot(b, q);
    };
    auto lds_put = [&]() {
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            int cq, yy, xx;
            slot_of(q, cq, yy, xx);
            if (!FULL && !(c0 + 8 * cq < C && yy < H && xx < W)) continue;
            char* dst = smem + dwm_plane_off(8 * cq) + (yy + 4) * DWM_PITCH + (xx + 4) * 2;
            const unsigned a[4] = {r0[q].x, r0[q].y, r0[q].z, r0[q].w}, bb[4] = {r1[q].x, r1[q].y, r1[q].z, r1[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // channels 2j, 2j+1 of both pixels: (px0.lo | px1.lo << 16), (px0.hi | px1.hi << 16)
                *reinterpret_cast<unsigned*>(dst + (2 * j) * DWM_PLANE) = __builtin_amdgcn_perm(bb[j], a[j], 0x05040100);
                *reinterpret_cast<unsigned*>(dst + (2 * j + 1) * DWM_PLANE) = __builtin_amdgcn_perm(bb[j], a[j], 0x07060302);
            }
        }
    };
    // fragment reads: lane = (row m = lane & 15, 8 columns at kq * 8); the last column group (kk >= 24) only meets zero taps and
    // would run past the plane row for the right-hand blocks: it reads the zero plane instead (at the 16-byte slot a kq = 2 lane
    // would read: a bank none of its group's other lanes is on)
    const int a_rd = (n + 4 - P) * DWM_PITCH + (kq < 3 ? kq : 2) * 16;
    const char* const zplane = smem + dwm_plane_off(CG);            // one more plane that is never written: all zeros

    gload(b0);
    __syncthreads();                                                // planes zeroed
    lds_put();
    for (int b = b0; b < b1; ++b) {
#if DWM_LOADS != 3
        if (!(DWM_SKIP & 2) && b + 1 < b1) gload(b + 1);
#endif
        __syncthreads();                                            // image b is in the planes
#pragma unroll
        for (int j = 0; j < ((DWM_SKIP & 1) ? 0 : CPW); ++j) {
#if DWM_LOADS == 3
            if (b + 1 < b1) {
#pragma unroll
                for (int q = j * NSLOT / CPW; q < (j + 1) * NSLOT / CPW; ++q) gload_slot(b + 1, q);
            }
#endif
            char* const plane = smem + dwm_plane_off(wave * CPW + j);
            const unsigned ab = (unsigned)(uintptr_t)((kq < 3 ? plane : zplane) + a_rd);
            const unsigned alo = ab + h0, ahi = ab + 8 - h0;
            if (j >= CREG) {                                        // this channel's fragments were requested before the last epilogue
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dwm_for<0, KS>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dwm_tie(tfl[i]);
                });
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (counted waits below: nothing else may be in flight)
            }
            // 2 k steps (the upper 16 rows' tap rows, then the lower 16 rows'), the plane rows of step s + 1 requested before the
            // MFMAs of step s.  Operands swapped (taps as A, plane rows as B): the accumulator then holds 4 consecutive x of ONE
            // row per lane -- 8 contiguous bytes of the plane -- instead of 4 rows of one column.
            f32x4 res[2][2];
            const f32x4 binit = {bz[j], bz[j], bz[j], bz[j]};
            constexpr int NST = DWM_STAGES;                     // plane-row fragment pairs in flight: requested NST - 1 steps ahead
            DwmFrag av[NST][2];
            dwm_for<0, NST - 1>([&](auto pc) {
                constexpr int s0 = decltype(pc)::value, ty0 = s0 / KS, i0 = s0 - ty0 * KS, off0 = ty0 * (16 * DWM_PITCH) + i0 * DWM_PITCH;
                dwm_issue<off0>(av[s0][0], alo, ahi);
                dwm_issue<off0 + 32>(av[s0][1], alo, ahi);
            });
            dwm_for<0, 2 * KS>([&](auto sc) {
                constexpr int st = decltype(sc)::value, ty = st / KS, i = st - ty * KS;
                constexpr int ahead = st + NST - 1;
                if constexpr (ahead < 2 * KS) {
                    constexpr int ty1 = ahead / KS, i1 = ahead - ty1 * KS, off = ty1 * (16 * DWM_PITCH) + i1 * DWM_PITCH;
                    dwm_issue<off>(av[ahead % NST][0], alo, ahi);
                    dwm_issue<off + 32>(av[ahead % NST][1], alo, ahi);
                }
                // reads issued after this step's: 4 per step still ahead
                constexpr int pending = (2 * KS - 1 - st < NST - 1 ? 2 * KS - 1 - st : NST - 1) * 4;
                dwm_wait<pending>(av[st % NST][0], av[st % NST][1]);
                const u32x4 tfi = j < CREG ? tf[j < CREG ? j : 0][i] : dwm_val(tfl[i]);
                if constexpr (i == 0) {                        // the accumulators start from the bias (one splat per channel, not 16 adds)
                    res[ty][0] = Mfma16<T>::run(tfi, dwm_val(av[st % NST][0]), binit);
                    res[ty][1] = Mfma16<T>::run(tfi, dwm_val(av[st % NST][1]), binit);
                } else {
                    res[ty][0] = Mfma16<T>::run(tfi, dwm_val(av[st % NST][0]), res[ty][0]);
                    res[ty][1] = Mfma16<T>::run(tfi, dwm_val(av[st % NST][1]), res[ty][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // the next channel's fragments out of the table, under this channel's epilogue
            if (j + 1 >= CREG && j + 1 < CPW) {
                dwm_for<0, KS>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dwm_issue<0>(tfl[i], tf_lo + ((j + 1 - CREG) * KS + i) * (DWM_NPAT * 16), tf_hi + ((j + 1 - CREG) * KS + i) * (DWM_NPAT * 16));
                });
            }
            // epilogue: res[ty][tx][r] = out[y = 16 ty + n][x = 16 tx + 4 kq + r]; results go back into the plane.  (The rows one half
            // writes are read by the OTHER half's fragments through the tap rows that reach across the boundary: both halves'
            // MFMAs are done before any result is written.)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 a4 = res[t >> 1][t & 1];
                const int y = 16 * (t >> 1) + n, x0 = 16 * (t & 1) + 4 * kq;
                char* const cell = plane + (y + 4) * DWM_PITCH + (x0 + 4) * 2;     // 8-byte aligned: 4 consecutive pixels of row y
                const u32x2 cw = *reinterpret_cast<const u32x2*>(cell);
                T c4[4];
                __builtin_memcpy(c4, &cw, 8);
                f32x2 v[2] = {f32x2{a4.x, a4.y}, f32x2{a4.z, a4.w}};
                gelu_pk_n<T, 2>(v);
                const float g[4] = {v[0].x, v[0].y, v[1].x, v[1].y};
                T o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = from_f32<T>(to_f32(c4[r]) + g[r] * sc[j] + sh[j]);
                if (FULL || (y < H && x0 + 3 < W)) {                // never write outside the map: the halo must stay zero
                    u32x2 ow;
                    __builtin_memcpy(&ow, o4, 8);
                    *reinterpret_cast<u32x2*>(cell) = ow;
                } else if (y < H) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (x0 + r < W) *reinterpret_cast<T*>(cell + 2 * r) = o4[r];
                }
            }
        }
        __syncthreads();                                            // results of image b are in the planes
        // ---- results out (two 16-byte stores per slot), next image in (same dwords) ----
#pragma unroll
        for (int q = 0; q < ((DWM_SKIP & 4) ? 0 : NSLOT); ++q) {
            int cq, yy, xx;
            slot_of(q, cq, yy, xx);
            if (!FULL && !(c0 + 8 * cq < C && yy < H && xx < W)) continue;
            const char* src = smem + dwm_plane_off(8 * cq) + (yy + 4) * DWM_PITCH + (xx + 4) * 2;
            unsigned d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = *reinterpret_cast<const unsigned*>(src + j * DWM_PLANE);
            u32x4 o0, o1;
            o0.x = __builtin_amdgcn_perm(d[1], d[0], 0x05040100); o1.x = __builtin_amdgcn_perm(d[1], d[0], 0x07060302);
            o0.y = __builtin_amdgcn_perm(d[3], d[2], 0x05040100); o1.y = __builtin_amdgcn_perm(d[3], d[2], 0x07060302);
            o0.z = __builtin_amdgcn_perm(d[5], d[4], 0x05040100); o1.z = __builtin_amdgcn_perm(d[5], d[4], 0x07060302);
            o0.w = __builtin_amdgcn_perm(d[7], d[6], 0x05040100); o1.w = __builtin_amdgcn_perm(d[7], d[6], 0x07060302);
            T* dstg = out + (((size_t)b * H + yy) * W + xx) * C + c0 + 8 * cq;
#if DWM_SKIP & 2
            if (o0.x == 0x12345678u && b == -1)
#endif
            {
                *reinterpret_cast<u32x4*>(dstg) = o0;
                if (FULL || xx + 1 < W) *reinterpret_cast<u32x4*>(dstg + C) = o1;
            }
        }
        if (!(DWM_SKIP & 4) && b + 1 < b1) lds_put();
    }
}

template <typename T, int cpw, int NWV = 8>
static int dwconv_mfma_launch_cpw(int k, const void* x, void* out, int B, int H, int W, int C, const float* w, const float* bias,
                                  const float* bns, const float* bnh, hipStream_t s) {
    // 8 waves x 4 channels = 32 channels per workgroup (64 aligned bytes of every pixel); fragments of 2 (k = 9) or 3 (k = 7) of a
    // wave's channels in registers, the others' in the LDS.  History on ConvMixer-1536/20 (k = 9, per layer): 8 x 3 channels, all
    // fragments in registers (48-byte pieces): 0.88 ms; 4 waves x 8 channels with the whole 512-register file per wave: 1.33 ms;
    // 8 waves x 1 channel (16-byte pieces): 1.41 ms; the VALU stencil it replaces: 1.66 ms.
    // Round 6 experiment (MLPK_DWM_CPW=2): 8 waves x 2 channels -- 16 channels per workgroup, every fragment in registers, planes of
    // 55 KB: two workgroups per CU, one's image hand-over (store, load, two barriers) under the other's matrix phase.
    constexpr int cg = NWV * cpw;
    constexpr int creg_cap = cpw;
    const int creg_want = k <= 5 ? 4 : (k == 7 ? DWM_CREG7 : DWM_CREG9);
    const int creg = creg_want < creg_cap ? creg_want : creg_cap;
    const int groups = (C + cg - 1) / cg;
    const int lds = dwm_plane_off(cg) + DWM_PLANE + NWV * (cpw - creg) * k * DWM_NPAT * 16;
    const int per_cu = (160 * 1024) / (lds + 1024) >= 2 ? 2 : 1;
    // images per workgroup: rounds of 256 (x workgroups per CU) workgroups x (images + about 2 images' worth of set-up: taps, fragments,
    // zeroing) -- the fewest image-times wins (1536 channels, 256 images: 48 groups x 16 ranges of 16 = 3 whole rounds)
    int per = 4;
    long long best = -1;
    const long long slots = 256LL * per_cu;
    for (int cand = 32; cand >= 4; cand /= 2) {
        const long long wgs = (long long)groups * ((B + cand - 1) / cand);
        const long long cost = ((wgs + slots - 1) / slots) * (cand + 2);
        if (best < 0 || cost < best) best = cost, per = cand;
    }
    const dim3 grid((unsigned)groups, (unsigned)((B + per - 1) / per));
    const bool full = H == 32 && W == 32 && C % cg == 0;
    hipError_t e = hipSuccess;
#define DWM_CASE(KS, CREG)                                                                                              \
    case KS: {                                                                                                         \
        constexpr int CR = (CREG) < cpw ? (CREG) : cpw;                                                                \
        auto kern = full ? dwconv_mfma_kernel<T, KS, cpw, CR, NWV, true> : dwconv_mfma_kernel<T, KS, cpw, CR, NWV, false>; \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e != hipSuccess) return (int)e;                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, s, (const T*)x, (T*)out, B, H, W, C, w, bias, bns, bnh, per); \
        break;                                                                                                         \
    }
    switch (k) {
        DWM_CASE(3, 4) DWM_CASE(5, 4) DWM_CASE(7, DWM_CREG7) DWM_CASE(9, DWM_CREG9)
        default: return DW_NOFIT;
    }
#undef DWM_CASE
    MLPK_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int dwconv_mfma_launch(int k, const void* x, void* out, int B, int H, int W, int C, const float* w, const float* bias,
                              const float* bns, const float* bnh, hipStream_t s) {
    static const int cpw_env = getenv("MLPK_DWM_CPW") ? atoi(getenv("MLPK_DWM_CPW")) : 0;
    if (cpw_env == 2 && k >= 7 && C % 16 == 0) return dwconv_mfma_launch_cpw<T, 2>(k, x, out, B, H, W, C, w, bias, bns, bnh, s);
    // MLPK_DWM_CPW=44: 4 waves x 4 channels -- two half-size workgroups per CU with the SAME registers per wave as the default
    if (cpw_env == 44 && k >= 7 && C % 16 == 0) return dwconv_mfma_launch_cpw<T, 4, 4>(k, x, out, B, H, W, C, w, bias, bns, bnh, s);
    return dwconv_mfma_launch_cpw<T, 4>(k, x, out, B, H, W, C, w, bias, bns, bnh, s);
}

extern "C" int mlpk_dwconv_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int k, const float* w,
                                const float* bias, const float* bn_scale, const float* bn_shift, void* stream) {
    if (!x || !out || !w) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return MLPK_ESHAPE;
    if (x == out) return MLPK_ESHAPE;                    // a stencil cannot run in place
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int es = dtype == MLPK_F32 ? 4 : 2;
    const bool fast_ok = (C % (16 / es) == 0) && (((uintptr_t)x & 15) == 0) && B <= 0x7fffffff;
    // matrix-pipe form: 16-bit dtypes, maps of up to 32 x 32, whole 8-channel groups (MLPK_DWCONV_NO_MFMA: tuning hook for A/B runs)
    static const bool no_mfma = getenv("MLPK_DWCONV_NO_MFMA") != nullptr;
    if (fast_ok && !no_mfma && es == 2 && H <= 32 && W <= 32 && C % 8 == 0 && (((uintptr_t)out & 15) == 0)) {
        const int rc = dtype == MLPK_F16 ? dwconv_mfma_launch<f16_t>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s)
                                         : dwconv_mfma_launch<bf16_t>(k, x, out, B, H, W, C, w, bias, bn_scale, bn_shift, s);
        if (rc != DW_NOFIT) return rc;
    }
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
