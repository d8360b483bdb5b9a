// Row statistics, normalise+affine(+GELU) with the layout change the next GEMM needs, ViP
// (un)permutes and token mean-pooling.  All of these are HBM-bound: 16-byte vector accesses,
// rows staged through LDS whenever the output order differs from the input order so that both
// the global reads and the global writes stay fully coalesced.
#include "mlpk_common.h"

namespace mlpk {

// ---- 8-element vector access (16 B for bf16/f16, 2 x 16 B for f32) ----------------------------
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p)[0];
        const f32x4 b = reinterpret_cast<const f32x4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const u32x4 t = *reinterpret_cast<const u32x4*>(p);
        T e[8];
        __builtin_memcpy(e, &t, 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = to_f32(e[i]);
    }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<f32x4*>(p)[0] = f32x4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4*>(p)[1] = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        T e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = from_f32<T>(v[i]);
        u32x4 t;
        __builtin_memcpy(&t, e, 16);
        *reinterpret_cast<u32x4*>(p) = t;
    }
}

// ================================ row statistics ================================
// TPR threads per row (64 = one wave per row, 1024 = one 16-wave workgroup per row for GroupNorm-sized rows).  Two passes over the
// row: mean, then centred sum of squares -- no E[x^2]-mu^2 cancellation, which matters for the fp32 1e-5 parity
// gate.  Rows of up to 2048 elements are held in registers between the passes; rows up to 4096 are re-read (L1/L2);
// longer ones (one workgroup per row) are read once with pivoted sums, see below.
// Short rows (<= 128 elements: the 64..128-channel LayerNorms of the staged models): 16 lanes per row, 16 rows per
// workgroup -- with a whole wave per row 3/4 or more of the lanes had nothing to load (96 channels = 12 lanes).
template <typename T>
__global__ void __launch_bounds__(256) row_stats_short_kernel(const T* __restrict__ x, int64_t rows, int64_t len, int64_t ldx, float eps,
                                                              float* __restrict__ mean, float* __restrict__ rstd) {
    const int tid = threadIdx.x;
    const int t = tid & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (tid >> 4);
    const bool live = row < rows;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const bool has = live && (int64_t)t * 8 < len;
    if (has) load8<T>(x + row * ldx + t * 8, v);
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s1 += v[e];
    const float mu = row16_sum(s1) / (float)len;
    float q1 = 0.f;
    if (has) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q1 += d * d; }
    }
    const float var = row16_sum(q1) / (float)len;
    if (live && t == 0) {
        mean[row] = mu;
        rstd[row] = 1.0f / __builtin_sqrtf(var + eps);
    }
}

template <typename T, int TPR>
__global__ void __launch_bounds__(TPR == 64 ? 256 : TPR) row_stats_kernel(const T* __restrict__ x, int64_t rows, int64_t len,
                                                        int64_t ldx, float eps, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int vec) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const int sub = TPR == 64 ? (tid >> 6) : 0;
    const int t = TPR == 64 ? (tid & 63) : tid;
    const int64_t row = (int64_t)blockIdx.x * (TPR == 64 ? 4 : 1) + sub;
    if (TPR == 64 && row >= rows) return;   // whole wave exits together
    const T* xr = x + row * ldx;

    auto block_sum = [&](float v) -> float {
        v = wave_sum(v);
        if constexpr (TPR != 64) {
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = v;
            __syncthreads();
            v = 0.f;
#pragma unroll
            for (int i = 0; i < TPR / 64; ++i) v += red[i];
        }
        return v;
    };

    if constexpr (TPR == 64) {
        // rows of up to 2048 elements stay in registers between the two passes: ONE trip to memory per element
        if (vec && len <= 4 * 64 * 8) {
            float v[4][8];
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = (int64_t)t * 8 + k * 512;
                if (i < len) {
                    load8<T>(xr + i, v[k]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s1 += v[k][e];
                }
            }
            const float mu1 = wave_sum(s1) / (float)len;
            float q1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = (int64_t)t * 8 + k * 512;
                if (i < len) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mu1; q1 += d * d; }
                }
            }
            const float var1 = wave_sum(q1) / (float)len;
            if (t == 0) {
                mean[row] = mu1;
                rstd[row] = 1.0f / __builtin_sqrtf(var1 + eps);
            }
            return;
        }
    }
    if constexpr (TPR != 64) {
        // Long rows (GroupNorm over C*H*W: 0.6 MB per row, 154 MB per call) are HBM-bound and do not fit the L2, so they are
        // read ONCE: sums of d = x - K and d^2 around a pivot K = x[row][0] (var = (q - s^2/n)/n cancels only as far as the
        // pivot is from the mean, i.e. by ~1 + (mean-K)^2/var ulps), four independent 16-byte loads in flight per thread.
        if (vec) {
            const float K = to_f32(xr[0]);
            float s1 = 0.f, q1 = 0.f;
            int64_t i = (int64_t)t * 8;
            for (; i + 3 * (int64_t)TPR * 8 < len; i += 4 * (int64_t)TPR * 8) {
                float v0[8], v1[8], v2[8], v3[8];
                load8<T>(xr + i, v0);
                load8<T>(xr + i + (int64_t)TPR * 8, v1);
                load8<T>(xr + i + 2 * (int64_t)TPR * 8, v2);
                load8<T>(xr + i + 3 * (int64_t)TPR * 8, v3);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d0 = v0[e] - K, d1 = v1[e] - K, d2 = v2[e] - K, d3 = v3[e] - K;
                    s1 += (d0 + d1) + (d2 + d3);
                    q1 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
            for (; i < len; i += TPR * 8) {
                float v[8];
                load8<T>(xr + i, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - K; s1 += d; q1 += d * d; }
            }
            const float S1 = block_sum(s1);
            const float Q1 = block_sum(q1);
            const float n = (float)len;
            const float dm = S1 / n;
            float var1 = (Q1 - S1 * dm) / n;
            var1 = var1 > 0.f ? var1 : 0.f;
            if (t == 0) {
                mean[row] = K + dm;
                rstd[row] = 1.0f / __builtin_sqrtf(var1 + eps);
            }
            return;
        }
    }
    float s = 0.f;
    if (vec) {
        // four independent 16-byte loads in flight per thread: one workgroup per (long) row means one workgroup per
        // CU, and with a single load per thread the row streams at memory LATENCY (measured 2 TB/s on GroupNorm rows)
        int64_t i = (int64_t)t * 8;
        for (; i + 3 * (int64_t)TPR * 8 < len; i += 4 * (int64_t)TPR * 8) {
            float v0[8], v1[8], v2[8], v3[8];
            load8<T>(xr + i, v0);
            load8<T>(xr + i + (int64_t)TPR * 8, v1);
            load8<T>(xr + i + 2 * (int64_t)TPR * 8, v2);
            load8<T>(xr + i + 3 * (int64_t)TPR * 8, v3);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; i < len; i += TPR * 8) {
            float v[8];
            load8<T>(xr + i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
        }
    } else {
        for (int64_t i = t; i < len; i += TPR) s += to_f32(xr[i]);
    }
    const float mu = block_sum(s) / (float)len;
    float q = 0.f;
    if (vec) {
        int64_t i = (int64_t)t * 8;
        for (; i + 3 * (int64_t)TPR * 8 < len; i += 4 * (int64_t)TPR * 8) {
            float v0[8], v1[8], v2[8], v3[8];
            load8<T>(xr + i, v0);
            load8<T>(xr + i + (int64_t)TPR * 8, v1);
            load8<T>(xr + i + 2 * (int64_t)TPR * 8, v2);
            load8<T>(xr + i + 3 * (int64_t)TPR * 8, v3);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d0 = v0[e] - mu, d1 = v1[e] - mu, d2 = v2[e] - mu, d3 = v3[e] - mu;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        for (; i < len; i += TPR * 8) {
            float v[8];
            load8<T>(xr + i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q += d * d; }
        }
    } else {
        for (int64_t i = t; i < len; i += TPR) { const float d = to_f32(xr[i]) - mu; q += d * d; }
    }
    const float var = block_sum(q) / (float)len;
    if (t == 0) {
        mean[row] = mu;
        rstd[row] = 1.0f / __builtin_sqrtf(var + eps);
    }
}

// ================================ norm apply: row-major / token-transposed ================================
// Tile = 64 rows (tokens of ONE image) x 64 channels, 256 threads, each thread 2 x (8 channels).
struct NormArgs {
    const void* x;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    void* out_rm;
    void* out_tt;
    int64_t rows;
    int C, ldx, stat_group, S, ld_rm, ld_tt, act;
    int s_tiles;   // tiles along the token axis per image
};

template <typename T>
__global__ void __launch_bounds__(256) norm_apply_tile_kernel(const NormArgs p) {
    constexpr int PITCH = 64 + 8;                 // elements; keeps 16-byte alignment of every row
    __shared__ __attribute__((aligned(16))) T tile[64 * PITCH];
    // 16-bit types stage the transposed copy as [channel pair][token] 32-bit words, row pitch 66 words: the four
    // 4-byte writes of a thread and the 8-byte reads of the transposed side are both bank-conflict free (the
    // element-wise [token][channel] reads used for float measured 9.4M conflict cycles per launch)
    constexpr int WP = 66;
    uint32_t* tw = reinterpret_cast<uint32_t*>(tile);   // 32 pairs x 66 words = 8448 B <= sizeof(tile)
    const int tid = threadIdx.x;
    const int img = blockIdx.x / p.s_tiles;
    const int s0 = (blockIdx.x % p.s_tiles) * 64;
    const int c0 = blockIdx.y * 64;
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    T* out_rm = reinterpret_cast<T*>(p.out_rm);
    T* out_tt = reinterpret_cast<T*>(p.out_tt);
    const int cl = (tid & 7) * 8;                 // channel offset within the tile
    const int c = c0 + cl;
    const bool c_ok = c < p.C;                    // C % 8 == 0, so a chunk is all-in or all-out
    float g[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[e] = 1.0f; b[e] = 0.0f; }
    if (c_ok && p.gamma) {                           // two 16-byte loads each (torch allocations are 16-byte aligned, c % 8 == 0)
        if ((reinterpret_cast<uintptr_t>(p.gamma) & 15) == 0) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + c), g1 = *reinterpret_cast<const f32x4*>(p.gamma + c + 4);
            g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = p.gamma[c + e];
        }
    }
    if (c_ok && p.beta) {
        if ((reinterpret_cast<uintptr_t>(p.beta) & 15) == 0) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + c), b1 = *reinterpret_cast<const f32x4*>(p.beta + c + 4);
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = p.beta[c + e];
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int sl = (tid >> 3) + it * 32;
        const int s = s0 + sl;
        float v[8];
        if (s < p.S && c_ok) {
            const int64_t r = (int64_t)img * p.S + s;
            load8<T>(x + r * p.ldx + c, v);
            float mu = 0.f, rs = 1.f;
            if (p.mean) {
                const int64_t sr = r / p.stat_group;
                mu = p.mean[sr];
                rs = p.rstd[sr];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (v[e] - mu) * rs * g[e] + b[e];
                if (p.act == MLPK_ACT_GELU) t = gelu_f(t);
                v[e] = t;
            }
            if (out_rm) store8<T>(out_rm + r * p.ld_rm + c, v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;   // token padding columns of out_tt are zeros
        }
        if (out_tt) {
            if constexpr (sizeof(T) == 2) {
                T e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = from_f32<T>(v[k]);
                uint32_t w4[4];
                __builtin_memcpy(w4, e, 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) tw[((tid & 7) * 4 + k) * WP + sl] = w4[k];
            } else {
                store8<T>(&tile[sl * PITCH + cl], v);
            }
        }
    }
    if (!out_tt) return;
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
        // transposed write: thread -> (channel pair tid >> 3, 8 consecutive tokens (tid & 7) * 8): four 8-byte LDS
        // reads, the low / high halves of the 8 words are the 8 tokens of the even / odd channel
        const int pr = tid >> 3;
        const int sc = (tid & 7) * 8;
        const int cch = c0 + pr * 2;
        if (cch < p.C && s0 + sc < p.ld_tt) {
            uint32_t w8[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x2 t2 = *reinterpret_cast<const u32x2*>(tw + pr * WP + sc + 2 * k);
                w8[2 * k] = t2.x;
                w8[2 * k + 1] = t2.y;
            }
            u32x4 lo, hi;
            lo.x = __builtin_amdgcn_perm(w8[1], w8[0], 0x05040100u); hi.x = __builtin_amdgcn_perm(w8[1], w8[0], 0x07060302u);
            lo.y = __builtin_amdgcn_perm(w8[3], w8[2], 0x05040100u); hi.y = __builtin_amdgcn_perm(w8[3], w8[2], 0x07060302u);
            lo.z = __builtin_amdgcn_perm(w8[5], w8[4], 0x05040100u); hi.z = __builtin_amdgcn_perm(w8[5], w8[4], 0x07060302u);
            lo.w = __builtin_amdgcn_perm(w8[7], w8[6], 0x05040100u); hi.w = __builtin_amdgcn_perm(w8[7], w8[6], 0x07060302u);
            T* dst = out_tt + ((int64_t)img * p.C + cch) * p.ld_tt + s0 + sc;
            *reinterpret_cast<u32x4*>(dst) = lo;
            *reinterpret_cast<u32x4*>(dst + p.ld_tt) = hi;      // C % 8 == 0: the odd channel exists whenever the even one does
        }
        return;
    }
    // transposed write: thread -> (channel cc, 8 consecutive tokens)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;           // 0..511 = 64 channels x 8 token chunks
        const int cc = idx >> 3;
        const int sc = (idx & 7) * 8;
        if (c0 + cc >= p.C) continue;
        if (s0 + sc >= p.ld_tt) continue;         // ld_tt % 8 == 0
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = to_f32(tile[(sc + e) * PITCH + cc]);
        store8<T>(out_tt + ((int64_t)img * p.C + c0 + cc) * p.ld_tt + s0 + sc, v);
    }
}

// ================================ token LayerNorm, statistics + apply + transpose in one pass ================================
// The token-mixing PreNormResidual of MLP-Mixer (mlp_mixer.py:34 with :6-13): xt[b*C + c, s] = LayerNorm_C(x[b, s, :])[c].
// One workgroup = 32 tokens of one image with ALL their channels: 16 lanes per token hold the row in registers (C / 128 16-byte
// vectors each, coalesced), so mean and the centred sum of squares cost two DPP row reductions and no second read; the normalised
// values go through LDS as [channel pair][token] words and leave as 16-byte stores of 8 consecutive tokens of one channel.
// x is read once and xt written once: the separate statistics kernel (one more read of x per block) disappears.
struct LnTtArgs {
    const void* x;
    const float* gamma;
    const float* beta;
    void* out;
    int S, C, ldx, ld_tt, s_tiles;
    float eps;
};
template <typename T, int NVMAX, int LPT>
__global__ void __launch_bounds__(32 * LPT) layernorm_transpose_kernel(const LnTtArgs p) {
    static_assert(LPT == 16 || LPT == 32, "lanes per token");
    constexpr int THREADS = 32 * LPT;
    static_assert(sizeof(T) == 2, "16-bit storage types");
    constexpr int WP = 34;                                   // words per channel pair: 32 tokens + 2 (8-byte aligned rows)
    extern __shared__ __attribute__((aligned(16))) uint32_t ltw[];      // (C / 2) x WP words
    const int tid = threadIdx.x;
    const int img = blockIdx.x / p.s_tiles;
    const int s0 = (blockIdx.x % p.s_tiles) * 32;
    const int sl = tid / LPT, l16 = tid % LPT;               // token slot, lane within the token's row (wide rows: 32 lanes per
                                                             // token, 1024 threads -- twice the waves in flight behind the one
                                                             // workgroup per CU that the staging tile allows)
    const int s = s0 + sl;
    const int nv = p.C / (LPT * 8);                          // vectors per lane (C % (LPT * 8) == 0, <= NVMAX)
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const bool live = s < p.S;
    u32x4 raw[NVMAX];
#pragma unroll
    for (int k = 0; k < NVMAX; ++k) {
        raw[k] = u32x4{0u, 0u, 0u, 0u};
        if (k < nv && live) raw[k] = *reinterpret_cast<const u32x4*>(x + ((int64_t)img * p.S + s) * p.ldx + (l16 + LPT * k) * 8);
    }
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NVMAX; ++k) {
        if (k < nv) {
            T e[8];
            __builtin_memcpy(e, &raw[k], 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) s1 += to_f32(e[i]);
        }
    }
    const float inv = 1.0f / (float)p.C;
    auto row_sum = [&](float v) {
        v = row16_sum(v);
        if constexpr (LPT == 32) v += __shfl_xor(v, 16);
        return v;
    };
    const float mu = row_sum(s1) * inv;
    float q1 = 0.f;
#pragma unroll
    for (int k = 0; k < NVMAX; ++k) {
        if (k < nv) {
            T e[8];
            __builtin_memcpy(e, &raw[k], 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = to_f32(e[i]) - mu;
                q1 += d * d;
            }
        }
    }
    const float rs = 1.0f / __builtin_sqrtf(row_sum(q1) * inv + p.eps);
#pragma unroll
    for (int k = 0; k < NVMAX; ++k) {
        if (k < nv) {
            const int c = (l16 + LPT * k) * 8;
            T e[8];
            __builtin_memcpy(e, &raw[k], 16);
            if (live) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + c), g1 = *reinterpret_cast<const f32x4*>(p.gamma + c + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + c), b1 = *reinterpret_cast<const f32x4*>(p.beta + c + 4);
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = from_f32<T>((to_f32(e[i]) - mu) * rs * g[i] + b[i]);
            }                                                // tokens past S: the zero K-padding of xt
            uint32_t w4[4];
            __builtin_memcpy(w4, e, 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) ltw[((c >> 1) + i) * WP + sl] = w4[i];
        }
    }
    __syncthreads();
    // transposed side: item = (channel pair, 8 consecutive tokens); 4 items cover a pair's 32 tokens = 64 contiguous bytes per channel
    T* out = reinterpret_cast<T*>(p.out);
    const int items = (p.C >> 1) * 4;
    for (int idx = tid; idx < items; idx += THREADS) {
        const int pr = idx >> 2;
        const int sc = (idx & 3) * 8;
        if (s0 + sc >= p.ld_tt) continue;                    // ld_tt % 8 == 0
        uint32_t w8[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x2 t2 = *reinterpret_cast<const u32x2*>(ltw + pr * WP + sc + 2 * k);
            w8[2 * k] = t2.x;
            w8[2 * k + 1] = t2.y;
        }
        u32x4 lo, hi;
        lo.x = __builtin_amdgcn_perm(w8[1], w8[0], 0x05040100u); hi.x = __builtin_amdgcn_perm(w8[1], w8[0], 0x07060302u);
        lo.y = __builtin_amdgcn_perm(w8[3], w8[2], 0x05040100u); hi.y = __builtin_amdgcn_perm(w8[3], w8[2], 0x07060302u);
        lo.z = __builtin_amdgcn_perm(w8[5], w8[4], 0x05040100u); hi.z = __builtin_amdgcn_perm(w8[5], w8[4], 0x07060302u);
        lo.w = __builtin_amdgcn_perm(w8[7], w8[6], 0x05040100u); hi.w = __builtin_amdgcn_perm(w8[7], w8[6], 0x07060302u);
        T* dst = out + ((int64_t)img * p.C + 2 * pr) * p.ld_tt + s0 + sc;
        *reinterpret_cast<u32x4*>(dst) = lo;
        *reinterpret_cast<u32x4*>(dst + p.ld_tt) = hi;
    }
}

// ================================ norm apply: ViP rearranges ================================
// One workgroup per (image, fixed w) [which = 0, all h] or (image, fixed h) [which = 1, all w]:
// the L x C slab (L = H or W) is staged in LDS in pixel order and written back in (g, l, j)
// order -- both sides are contiguous runs in global memory.
struct VipArgs {
    const void* x;      // (B,H,W,C) pixel stride ldx
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    void* out;
    float* sums;        // optional (fast kernel): per (image, group, fixed coordinate, j) sum over the walked axis, row stride ld_sum
    int B, H, W, C, seg, ldx, ld_p, which, act, ld_sum;
};

template <typename T>
__global__ void __launch_bounds__(256) vip_permute_kernel(const VipArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* slab = reinterpret_cast<T*>(smem_raw);          // [L][C]
    const int tid = threadIdx.x;
    const int L = p.which == 0 ? p.H : p.W;            // length of the mixed axis
    const int O = p.which == 0 ? p.W : p.H;            // the fixed axis
    const int img = blockIdx.x / O;
    const int o = blockIdx.x % O;
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const int cv = p.C / 8;
    for (int idx = tid; idx < L * cv; idx += 256) {
        const int l = idx / cv;
        const int c = (idx % cv) * 8;
        const int h = p.which == 0 ? l : o;
        const int w = p.which == 0 ? o : l;
        const int64_t r = ((int64_t)img * p.H + h) * p.W + w;
        float v[8];
        load8<T>(x + r * p.ldx + c, v);
        float mu = 0.f, rs = 1.f;
        if (p.mean) { mu = p.mean[r]; rs = p.rstd[r]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (v[e] - mu) * rs * (p.gamma ? p.gamma[c + e] : 1.0f) + (p.beta ? p.beta[c + e] : 0.0f);
            if (p.act == MLPK_ACT_GELU) t = gelu_f(t);
            v[e] = t;
        }
        store8<T>(slab + l * p.C + c, v);
    }
    __syncthreads();
    const int G = p.C / p.seg;
    T* out = reinterpret_cast<T*>(p.out) + ((int64_t)img * O + o) * G * p.ld_p;
    const int total = G * p.ld_p;
    if ((p.ld_p & 7) == 0 && ((uintptr_t)out & 15) == 0) {
        // 8 consecutive output elements per thread: one division pair per vector, then the (l, j) cursor is
        // advanced incrementally; LDS is read element-wise (cheap), global memory is written 16 bytes per lane
        for (int v8 = tid; v8 < total / 8; v8 += 256) {
            const int idx = v8 * 8;
            const int g = idx / p.ld_p;
            int col = idx - g * p.ld_p;
            int l = col / p.seg;
            int j = col - l * p.seg;
            const T* src = slab + g * p.seg;
            T e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                e[k] = l < L ? src[l * p.C + j] : from_f32<T>(0.f);
                if (++j == p.seg) { j = 0; ++l; }
            }
            if constexpr (sizeof(T) == 2) {
                u32x4 t;
                __builtin_memcpy(&t, e, 16);
                *reinterpret_cast<u32x4*>(out + idx) = t;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) out[idx + k] = e[k];
            }
        }
        return;
    }
    for (int idx = tid; idx < total; idx += 256) {
        const int g = idx / p.ld_p;
        const int col = idx - g * p.ld_p;
        T val = from_f32<T>(0.f);
        if (col < L * p.seg) {
            const int l = col / p.seg;
            const int j = col - l * p.seg;
            val = slab[l * p.C + g * p.seg + j];
        }
        out[idx] = val;
    }
}

// Fast path (seg % 4 == 0, C % 8 == 0, ld_p % 8 == 0): the slab is staged in LDS already in OUTPUT order
// ([g][l][j], one padded row per group g), so the write-back is pure 16-byte LDS reads -> 16-byte global stores and the
// read side is one 16-byte global load per 8 channels, split into two 4-channel pieces (a piece never straddles a
// group because seg % 4 == 0).  gamma / beta sit in LDS; rows are padded by 32 bytes so that the pieces of neighbouring
// groups fall on different banks.
template <typename T>
__global__ void __launch_bounds__(256) vip_permute_fast_kernel(const VipArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int L = p.which == 0 ? p.H : p.W;            // length of the mixed axis
    const int O = p.which == 0 ? p.W : p.H;            // the fixed axis
    const int img = blockIdx.x / O;
    const int o = blockIdx.x % O;
    const int G = p.C / p.seg;
    const int rowb = p.ld_p * (int)sizeof(T) + 32;     // LDS bytes per group row
    char* const slab = smem_raw;                       // [G][rowb]
    float* const gb = reinterpret_cast<float*>(smem_raw + (size_t)G * rowb);   // gamma[C], beta[C]
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    for (int i = tid; i < p.C; i += 256) {
        gb[i] = p.gamma ? p.gamma[i] : 1.0f;
        gb[p.C + i] = p.beta ? p.beta[i] : 0.0f;
    }
    // zero the pad columns [L*seg, ld_p) of every group row
    const int padn = p.ld_p - L * p.seg;
    if (padn > 0) {
        for (int i = tid; i < G * padn; i += 256) {
            const int g = i / padn, k = i - g * padn;
            *reinterpret_cast<T*>(slab + (size_t)g * rowb + (size_t)(L * p.seg + k) * sizeof(T)) = from_f32<T>(0.f);
        }
    }
    __syncthreads();
    const int cv = p.C / 8;
    for (int idx = tid; idx < L * cv; idx += 256) {
        const int l = idx / cv;
        const int c = (idx - l * cv) * 8;
        const int h = p.which == 0 ? l : o;
        const int w = p.which == 0 ? o : l;
        const int64_t r = ((int64_t)img * p.H + h) * p.W + w;
        float v[8];
        load8<T>(x + r * p.ldx + c, v);
        float mu = 0.f, rs = 1.f;
        if (p.mean) { mu = p.mean[r]; rs = p.rstd[r]; }
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + c), g1 = *reinterpret_cast<const f32x4*>(gb + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + p.C + c), b1 = *reinterpret_cast<const f32x4*>(gb + p.C + c + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        T e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = (v[k] - mu) * rs * gm[k] + bt[k];
            if (p.act == MLPK_ACT_GELU) t = gelu_t<T>(t);
            e[k] = from_f32<T>(t);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cc = c + half * 4;
            const int g = cc / p.seg;
            const int j = cc - g * p.seg;
            char* dst = slab + (size_t)g * rowb + (size_t)(l * p.seg + j) * sizeof(T);
            if constexpr (sizeof(T) == 2) {
                u32x2 t;
                __builtin_memcpy(&t, e + half * 4, 8);
                *reinterpret_cast<u32x2*>(dst) = t;
            } else {
                u32x4 t;
                __builtin_memcpy(&t, e + half * 4, 16);
                *reinterpret_cast<u32x4*>(dst) = t;
            }
        }
    }
    __syncthreads();
    if (p.sums) {
        // by-product: sum over the walked axis of the ROUNDED values (what the branch GEMM will read), per (group, j)
        for (int i = tid; i < p.C; i += 256) {
            const int g = i / p.seg, j = i - g * p.seg;
            const T* row = reinterpret_cast<const T*>(slab + (size_t)g * rowb) + j;
            float acc = 0.f;
            for (int l = 0; l < L; ++l) acc += to_f32(row[l * p.seg]);
            p.sums[((int64_t)img * G + g) * p.ld_sum + o * p.seg + j] = acc;
        }
    }
    T* out = reinterpret_cast<T*>(p.out) + ((int64_t)img * O + o) * G * p.ld_p;
    constexpr int EPV = 16 / (int)sizeof(T);
    const int vpr = p.ld_p / EPV;                       // 16-byte vectors per group row
    for (int i = tid; i < G * vpr; i += 256) {
        const int g = i / vpr, k = i - g * vpr;
        const u32x4 t = *reinterpret_cast<const u32x4*>(slab + (size_t)g * rowb + (size_t)k * 16);
        *reinterpret_cast<u32x4*>(out + (size_t)g * p.ld_p + k * EPV) = t;
    }
}

// inverse rearrange of the GEMM output z back to (B,H,W,C)
template <typename T>
__global__ void __launch_bounds__(256) vip_unpermute_kernel(const T* __restrict__ z, T* __restrict__ out, int B, int H,
                                                            int W, int C, int seg, int ldz, int which) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* slab = reinterpret_cast<T*>(smem_raw);          // [G][L*seg] compact
    const int tid = threadIdx.x;
    const int L = which == 0 ? H : W;
    const int O = which == 0 ? W : H;
    const int img = blockIdx.x / O;
    const int o = blockIdx.x % O;
    const int G = C / seg;
    const int rowlen = L * seg;
    const T* zin = z + ((int64_t)img * O + o) * G * ldz;
    if ((rowlen & 7) == 0 && (ldz & 7) == 0 && sizeof(T) == 2 && ((uintptr_t)zin & 15) == 0) {
        const int rv = rowlen / 8;
        for (int v8 = tid; v8 < G * rv; v8 += 256) {
            const int g = v8 / rv;
            const int col = (v8 - g * rv) * 8;
            *reinterpret_cast<u32x4*>(slab + g * rowlen + col) = *reinterpret_cast<const u32x4*>(zin + (int64_t)g * ldz + col);
        }
    } else {
        for (int idx = tid; idx < G * rowlen; idx += 256) {
            const int g = idx / rowlen;
            const int col = idx - g * rowlen;
            slab[idx] = zin[(int64_t)g * ldz + col];
        }
    }
    __syncthreads();
    if ((C & 7) == 0 && sizeof(T) == 2 && ((uintptr_t)out & 15) == 0) {
        // 8 consecutive channels per thread -> one 16-byte store; (g, q) cursor advanced incrementally
        const int cv = C / 8;
        for (int v8 = tid; v8 < L * cv; v8 += 256) {
            const int l = v8 / cv;
            const int c = (v8 - l * cv) * 8;
            int g = c / seg;
            int q = c - g * seg;
            T e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                e[k] = slab[g * rowlen + l * seg + q];
                if (++q == seg) { q = 0; ++g; }
            }
            const int h = which == 0 ? l : o;
            const int w = which == 0 ? o : l;
            u32x4 t;
            __builtin_memcpy(&t, e, 16);
            *reinterpret_cast<u32x4*>(out + (((int64_t)img * H + h) * W + w) * C + c) = t;
        }
        return;
    }
    for (int idx = tid; idx < L * C; idx += 256) {
        const int l = idx / C;
        const int c = idx - l * C;
        const int g = c / seg;
        const int q = c - g * seg;
        const int h = which == 0 ? l : o;
        const int w = which == 0 ? o : l;
        out[(((int64_t)img * H + h) * W + w) * C + c] = slab[g * rowlen + l * seg + q];
    }
}

// ================================ mean over tokens ================================
// workgroup = (image, 64 channels): 32 token rows in flight x 8 lanes x 8 channels.
template <typename T>
__global__ void __launch_bounds__(256) pool_mean_kernel(const T* __restrict__ x, int B, int S, int C, int ldx,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int stat_group, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int ldo) {
    __shared__ float red[32][64 + 1];
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const int cl = (tid & 7) * 8;
    const int c = blockIdx.y * 64 + cl;
    const int sl = tid >> 3;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c < C) {
        for (int s = sl; s < S; s += 32) {
            const int64_t r = (int64_t)img * S + s;
            float v[8];
            load8<T>(x + r * ldx + c, v);
            float mu = 0.f, rs = 1.f;
            if (mean) { const int64_t sr = r / stat_group; mu = mean[sr]; rs = rstd[sr]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (v[e] - mu) * rs;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[sl][cl + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        const int cc = blockIdx.y * 64 + tid;
        if (cc < C) {
            float s = 0.f;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) s += red[i][tid];
            float y = s / (float)S;
            if (gamma) y *= gamma[cc];
            if (beta) y += beta[cc];
            out[(int64_t)img * ldo + cc] = from_f32<T>(y);
        }
    }
}

template <typename TS, typename TD>
__global__ void __launch_bounds__(256) convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = from_f32<TD>(to_f32(src[i]));
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

// mean / rstd from partial (sum, sum of squares) pairs written by a producer's epilogue (mlpk_token_mlp, mlpk_gemm_nt):
// mean = S1 / count, var = S2 / count - mean^2 (>= 0), rstd = 1 / sqrt(var + eps).  fp32 sums of values that were rounded to
// 16 bits: the cancellation error of the E[x^2] - mean^2 form (~1e-7 (1 + mean^2 / var)) is far below the storage rounding of
// the tensor being normalised.
// planar pairs of a GEMM's by-product statistics (mlpk.h row_part): pair (q, m) at part[(q * plane_stride + m) * 2].
// group = 1: one thread per row, the planes read coalesced.
template <int LPR>      // lanes per row: 1 (default) or 4 (MLPK_FINALIZE_LANES=4: A/B aid)
__global__ void __launch_bounds__(256) stats_finalize_planar_kernel(const float* __restrict__ part, int64_t rows, int nplanes, int64_t plane_stride,
                                                                    float inv_count, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    // LPR lanes per row, each adding a contiguous share of the planes (eight loads in flight), then (q0 + q1) + (q2 + q3).  Round 4 tried
    // LPR = 4 against one thread per row (23 launches per Mixer forward, 59 per gMLP forward, 7-10 us each): measured NOT faster -- Mixer-B/16
    // 7.600 vs 7.592 ms, gMLP-S 9.78 vs 9.67, ViP-S7 29.34 vs 29.32 (profiles/r04_finalize_lanes_ab.txt) -- the launches are bound by their
    // launch latency, not by the chain of loads; one lane per row stays the default.
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int sub = threadIdx.x % LPR;
    const bool live = r < rows;
    const f32x2* pp = reinterpret_cast<const f32x2*>(part) + (live ? r : rows - 1);
    // the planes are combined in fp64: with rows whose mean is large against their spread (deep residual streams) the fp32 form
    // S2 / n - mean^2 loses the variance to cancellation; what remains is the rounding of the per-plane fp32 sums themselves
    const int per = (nplanes + LPR - 1) / LPR;
    const int q0 = sub * per, q1 = q0 + per < nplanes ? q0 + per : nplanes;
    double s1 = 0.0, s2 = 0.0;
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
        f32x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = pp[(int64_t)(q + k) * plane_stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += (double)v[k].x; s2 += (double)v[k].y; }
    }
    for (; q < q1; ++q) {
        const f32x2 v = pp[(int64_t)q * plane_stride];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    // (q0 + q1) + (q2 + q3) over the four lanes of the row
    if (LPR == 4) {
        s1 += __shfl_xor(s1, 1);
        s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2);
        s2 += __shfl_xor(s2, 2);
    }
    if (!live || sub) return;
    const double mu = s1 * (double)inv_count;
    double var = s2 * (double)inv_count - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[r] = (float)mu;
    rstd[r] = (float)(1.0 / __builtin_sqrt(var + (double)eps));
}

// group > 1 (per-sample GroupNorm(1,C) statistics from per-pixel-row pairs): one workgroup per statistic, fp64 for the combine
// (the pairs are sums of <= 128 values each; what is left to lose is the E[x^2] - mean^2 cancellation of a whole sample)
__global__ void __launch_bounds__(256) stats_finalize_group_kernel(const float* __restrict__ part, int nplanes, int64_t plane_stride, int group,
                                                                   double inv_count, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[2][4];
    const int64_t r = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int q = 0; q < nplanes; ++q) {
        const f32x2* pp = reinterpret_cast<const f32x2*>(part) + (int64_t)q * plane_stride + r * group;
        for (int i = threadIdx.x; i < group; i += 256) {
            const f32x2 v = pp[i];
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mu = s1 * inv_count;
        double var = s2 * inv_count - mu * mu;
        var = var > 0.0 ? var : 0.0;
        mean[r] = (float)mu;
        rstd[r] = 1.0f / __builtin_sqrtf((float)var + eps);
    }
}

static inline int esize(int dt) { return dt == MLPK_F32 ? 4 : 2; }

extern "C" int mlpk_row_stats(int dtype, const void* x, int64_t rows, int64_t len, int64_t ldx, float eps,
                              float* mean, float* rstd, void* stream) {
    if (!x || !mean || !rstd) return MLPK_ENULL;
    if (rows <= 0 || len <= 0 || ldx < len) return MLPK_ESHAPE;
    if (dtype < 0 || dtype > 2) return MLPK_EDTYPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = (len % 8 == 0) && (ldx % 8 == 0) && (((uintptr_t)x & 15) == 0);
    if (vec && len <= 128) {
        const unsigned grid = (unsigned)((rows + 15) / 16);
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((row_stats_short_kernel<T>), dim3(grid), dim3(256), 0, s, (const T*)x, rows, len, ldx, eps,
                                                 mean, rstd));
    } else if (len <= 4096) {
        const unsigned grid = (unsigned)((rows + 3) / 4);
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((row_stats_kernel<T, 64>), dim3(grid), dim3(256), 0, s,
                                                 (const T*)x, rows, len, ldx, eps, mean, rstd, vec));
    } else {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((row_stats_kernel<T, 1024>), dim3((unsigned)rows), dim3(1024), 0, s,
                                                 (const T*)x, rows, len, ldx, eps, mean, rstd, vec));
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_stats_finalize_planar(const float* part, int64_t rows, int nplanes, int64_t plane_stride, int group, int64_t count, float eps,
                                          float* mean, float* rstd, void* stream) {
    if (!part || !mean || !rstd) return MLPK_ENULL;
    if (rows <= 0 || nplanes <= 0 || group <= 0 || count <= 0 || plane_stride < rows * group) return MLPK_ESHAPE;
    if ((uintptr_t)part & 7) return MLPK_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (group == 1) {
        static const bool one_lane = !(getenv("MLPK_FINALIZE_LANES") && atoi(getenv("MLPK_FINALIZE_LANES")) == 4);
        if (one_lane)
            hipLaunchKernelGGL(stats_finalize_planar_kernel<1>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, rows, nplanes, plane_stride,
                               1.0f / (float)count, eps, mean, rstd);
        else
            hipLaunchKernelGGL(stats_finalize_planar_kernel<4>, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, s, part, rows, nplanes, plane_stride,
                               1.0f / (float)count, eps, mean, rstd);
    } else {
        hipLaunchKernelGGL(stats_finalize_group_kernel, dim3((unsigned)rows), dim3(256), 0, s, part, nplanes, plane_stride, group,
                           1.0 / (double)count, eps, mean, rstd);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_norm_apply(const mlpk_norm_desc* d, void* stream) {
    if (!d || !d->x) return MLPK_ENULL;
    if (d->dtype < 0 || d->dtype > 2) return MLPK_EDTYPE;
    if (d->rows <= 0 || d->C <= 0 || d->C % 8 || d->ldx % 8 || d->ldx < d->C) return MLPK_ESHAPE;
    if ((d->mean == nullptr) != (d->rstd == nullptr)) return MLPK_ENULL;
    if (d->act != MLPK_ACT_NONE && d->act != MLPK_ACT_GELU) return MLPK_EMODE;
    if ((uintptr_t)d->x & 15) return MLPK_EALIGN;
    const int sg = d->stat_group > 0 ? d->stat_group : 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->out_rm || d->out_tt) {
        NormArgs a;
        a.x = d->x; a.mean = d->mean; a.rstd = d->rstd; a.gamma = d->gamma; a.beta = d->beta;
        a.out_rm = d->out_rm; a.out_tt = d->out_tt; a.rows = d->rows; a.C = d->C; a.ldx = d->ldx;
        a.stat_group = sg; a.act = d->act; a.ld_rm = d->ld_rm; a.ld_tt = d->ld_tt;
        if (d->out_rm && (d->ld_rm % 8 || d->ld_rm < d->C || ((uintptr_t)d->out_rm & 15))) return MLPK_ESHAPE;
        int S = d->S, nimg;
        if (d->out_tt) {
            if (S <= 0 || d->rows % S || d->ld_tt % 8 || d->ld_tt < S || ((uintptr_t)d->out_tt & 15)) return MLPK_ESHAPE;
            nimg = (int)(d->rows / S);
            a.s_tiles = (d->ld_tt + 63) / 64;
        } else {
            // no transposed output: treat all rows as one "image" so tiles simply walk the rows
            if (d->rows > 0x7fffffffLL) return MLPK_ESHAPE;
            S = (int)d->rows;
            nimg = 1;
            a.s_tiles = (S + 63) / 64;
        }
        a.S = S;
        const dim3 grid((unsigned)(nimg * a.s_tiles), (unsigned)((d->C + 63) / 64));
        DISPATCH_DTYPE(d->dtype, hipLaunchKernelGGL((norm_apply_tile_kernel<T>), grid, dim3(256), 0, s, a));
        MLPK_LAUNCH_CHECK();
    }
    for (int which = 0; which < 2; ++which) {
        void* out = which == 0 ? d->out_ph : d->out_pw;
        if (!out) continue;
        if (d->H <= 0 || d->W <= 0 || d->seg <= 0 || d->C % d->seg) return MLPK_ESHAPE;
        if (d->rows % ((int64_t)d->H * d->W)) return MLPK_ESHAPE;
        const int L = which == 0 ? d->H : d->W;
        if (d->ld_p < L * d->seg) return MLPK_ESHAPE;
        if (sg != 1) return MLPK_EMODE;
        VipArgs a;
        a.x = d->x; a.mean = d->mean; a.rstd = d->rstd; a.gamma = d->gamma; a.beta = d->beta; a.out = out;
        a.B = (int)(d->rows / ((int64_t)d->H * d->W)); a.H = d->H; a.W = d->W; a.C = d->C; a.seg = d->seg;
        a.ldx = d->ldx; a.ld_p = d->ld_p; a.which = which; a.act = d->act;
        a.sums = which == 0 ? d->sum_ph : d->sum_pw;
        a.ld_sum = d->ld_sum;
        if (a.sums && d->ld_sum < (which == 0 ? d->W : d->H) * d->seg) return MLPK_ESHAPE;
        const unsigned grid = (unsigned)(a.B * (which == 0 ? d->W : d->H));
        const int es_ = esize(d->dtype);
        const size_t lds_fast = (size_t)(d->C / d->seg) * ((size_t)d->ld_p * es_ + 32) + (size_t)2 * d->C * 4;
        const bool fast = d->seg % 4 == 0 && d->C % 8 == 0 && d->ld_p % 8 == 0 && (d->ld_p * es_) % 16 == 0 &&
                          ((uintptr_t)out & 15) == 0 && lds_fast <= 160 * 1024;
        if (fast) {
            DISPATCH_DTYPE(d->dtype, {
                auto k = vip_permute_fast_kernel<T>;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_fast, s, a);
            });
            MLPK_LAUNCH_CHECK();
            continue;
        }
        if (a.sums) return MLPK_EMODE;                   // the by-product sums exist in the 16-byte fast path only
        const size_t lds = (size_t)L * d->C * esize(d->dtype);
        if (lds > 160 * 1024) return MLPK_ESHAPE;
        DISPATCH_DTYPE(d->dtype, {
            auto k = vip_permute_kernel<T>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
        });
        MLPK_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mlpk_layernorm_transpose(int dtype, const void* x, int64_t nimg, int S, int C, int ldx, const float* gamma, const float* beta,
                                        float eps, void* out_tt, int ld_tt, void* stream) {
    if (!x || !gamma || !beta || !out_tt) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (nimg <= 0 || S <= 0 || C <= 0 || C % 128 || C > 2048 || ldx < C || ldx % 8 || ld_tt < S || ld_tt % 8) return MLPK_ESHAPE;
    if (((uintptr_t)x | (uintptr_t)out_tt | (uintptr_t)gamma | (uintptr_t)beta) & 15) return MLPK_ESHAPE;
    LnTtArgs a;
    a.x = x; a.gamma = gamma; a.beta = beta; a.out = out_tt;
    a.S = S; a.C = C; a.ldx = ldx; a.ld_tt = ld_tt; a.s_tiles = (ld_tt + 31) / 32; a.eps = eps;
    if (nimg * a.s_tiles > 0x7fffffffLL) return MLPK_ESHAPE;
    const size_t lds = (size_t)(C / 2) * 34 * 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define LNTT_LAUNCH(TT, NV, LP)                                                                                          \
    {                                                                                                                    \
        auto k = layernorm_transpose_kernel<TT, NV, LP>;                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                              \
        hipLaunchKernelGGL(k, dim3((unsigned)(nimg * a.s_tiles)), dim3(32 * LP), lds, s, a);                             \
    }
    const bool wide32 = C > 1024 && C % 256 == 0;           // 32 lanes per token
    if (dtype == MLPK_BF16) {
        if (C <= 1024) LNTT_LAUNCH(bf16_t, 8, 16) else if (wide32) LNTT_LAUNCH(bf16_t, 8, 32) else LNTT_LAUNCH(bf16_t, 16, 16)
    } else {
        if (C <= 1024) LNTT_LAUNCH(f16_t, 8, 16) else if (wide32) LNTT_LAUNCH(f16_t, 8, 32) else LNTT_LAUNCH(f16_t, 16, 16)
    }
#undef LNTT_LAUNCH
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_vip_unpermute(int dtype, int which, const void* z, void* out, int B, int H, int W, int C,
                                  int seg, int ldz, void* stream) {
    if (!z || !out) return MLPK_ENULL;
    if (which != 0 && which != 1) return MLPK_EMODE;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || seg <= 0 || C % seg) return MLPK_ESHAPE;
    const int L = which == 0 ? H : W;
    if (ldz < L * seg) return MLPK_ESHAPE;
    const size_t lds = (size_t)L * C * esize(dtype);
    if (lds > 160 * 1024) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)(B * (which == 0 ? W : H));
    DISPATCH_DTYPE(dtype, {
        auto k = vip_unpermute_kernel<T>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const T*)z, (T*)out, B, H, W, C, seg, ldz, which);
    });
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_pool_mean(int dtype, const void* x, int B, int S, int C, int ldx, const float* mean,
                              const float* rstd, int stat_group, const float* gamma, const float* beta,
                              void* out, int ldo, void* stream) {
    if (!x || !out) return MLPK_ENULL;
    if (B <= 0 || S <= 0 || C <= 0 || C % 8 || ldx % 8 || ldx < C || ldo < C) return MLPK_ESHAPE;
    if ((mean == nullptr) != (rstd == nullptr)) return MLPK_ENULL;
    if ((uintptr_t)x & 15) return MLPK_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)B, (unsigned)((C + 63) / 64));
    const int sg = stat_group > 0 ? stat_group : 1;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pool_mean_kernel<T>), grid, dim3(256), 0, s, (const T*)x, B, S, C, ldx,
                                             mean, rstd, sg, gamma, beta, (T*)out, ldo));
    MLPK_LAUNCH_CHECK();
    return 0;
}

template <typename TS> static int convert_from(int dst_dtype, const void* src, void* dst, int64_t n, hipStream_t s) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    switch (dst_dtype) {
        case MLPK_F32: hipLaunchKernelGGL((convert_kernel<TS, float>), dim3(grid), dim3(256), 0, s, (const TS*)src, (float*)dst, n); break;
        case MLPK_F16: hipLaunchKernelGGL((convert_kernel<TS, f16_t>), dim3(grid), dim3(256), 0, s, (const TS*)src, (f16_t*)dst, n); break;
        case MLPK_BF16: hipLaunchKernelGGL((convert_kernel<TS, bf16_t>), dim3(grid), dim3(256), 0, s, (const TS*)src, (bf16_t*)dst, n); break;
        default: return MLPK_EDTYPE;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_convert(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst) return MLPK_ENULL;
    if (n <= 0) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (src_dtype) {
        case MLPK_F32: return convert_from<float>(dst_dtype, src, dst, n, s);
        case MLPK_F16: return convert_from<f16_t>(dst_dtype, src, dst, n, s);
        case MLPK_BF16: return convert_from<bf16_t>(dst_dtype, src, dst, n, s);
        default: return MLPK_EDTYPE;
    }
}

extern "C" int mlpk_abi_version(void) { return 12; }

extern "C" const char* mlpk_strerror(int code) {
    switch (code) {
        case MLPK_OK: return "ok";
        case MLPK_EDTYPE: return "mlpk: unknown dtype";
        case MLPK_ESHAPE: return "mlpk: size/stride violates a documented constraint";
        case MLPK_EALIGN: return "mlpk: pointer not 16-byte aligned";
        case MLPK_ENULL: return "mlpk: required pointer is NULL";
        case MLPK_EMODE: return "mlpk: unknown mode/flag";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "mlpk: unknown error";
    }
}
