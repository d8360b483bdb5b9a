// Sparse-MLP's sMLP block, first half (sparse_mlp.py:61-72), in ONE kernel for maps up to 32 x 32:
//
//     x^ = s x + h                                             (the BatchNorm in front of the block, eval mode: sparse_mlp.py:92, per channel)
//     x_h[b, h', w, c] = sum_h Wh[h', h] x^[b, h, w, c] + bh[h']    (proj_h on the permuted tensor, :68)
//     x_w[b, h, w', c] = sum_w Ww[w', w] x^[b, h, w, c] + bw[w']    (proj_w, :69)
//     out[(b, h, w), :] = [ x_h | x_w | x^ ]                    (torch.cat along the channels, :71: the operand of the 3C -> C fuse, one GEMM)
//
// Until round 4 this was four launches -- two normalise passes that wrote x^ transposed over W and over H, and two token products
// (mlpk_token_gemm with S = 14: 73 us each at 1.4 TB/s) -- moving 10 units of data for 4 (read x, write 3C).  Here a workgroup takes one
// image x 32 channels, loads its H W x 64-byte pieces once, writes x^ straight from the registers and stores the tile in LDS TWICE
// TRANSPOSED -- [h][c][w] and [w][c][h], 32-element rows -- so that a row is the K-contiguous MFMA operand of the mix along its last index.
// Per line (a fixed h, or a fixed w): D'[c][w'] = sum_w X[c][w] Ww[w'][w] as v_mfma_f32_16x16x32 with the 32 channels as two blocks of M whose
// rows are channels {8 q + 4 j + i}: a lane ends up with 8 CONSECUTIVE channels of one output position -- one 16-byte store, 64 contiguous
// bytes per 4 lanes.  The weights (zero-padded to 32 x 32) are four register fragments per mix for the whole kernel; the padding columns of
// the LDS rows are zeroed once (0 x garbage must not be NaN).  The next unit's pixels are requested before the MFMA phase of the current one.
// LDS: 2 x H x 32 x 64 B (H = 14: 57 KiB, H = 28: 115 KiB).
#include "mlpk_common.h"
#include <cstdlib>

namespace mlpk {

struct SmlpArgs {
    const void* x;          // (B*H*W, ldx) channel-last rows
    void* out;              // (B*H*W, ldo): [x_h | x_w | x^], 3 C columns
    const float* bn_s;      // per channel
    const float* bn_h;
    const void* wh;         // (32, 32) zero-padded: wh[h'][h]
    const void* ww;         // (32, 32): ww[w'][w]
    const float* bh;        // (32) zero-padded
    const float* bw;
    int B, H, W, C, ldx, ldo;
};

template <typename T> struct SmMma;
template <> struct SmMma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SmMma<f16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int SM_NT = 512;                                 // threads per workgroup
constexpr int SM_NL = 8;                                   // 16-byte pieces per thread: 32 x 32 pixels x 4 pieces / 512 threads

// element (line, channel c, position k) of a transposed copy: 64-byte rows, the four 16-byte chunks of a row XORed with the channel's octet
// (the writers of one instruction differ in the octet: without it they would all hit the same banks)
__device__ __forceinline__ unsigned sm_addr(int line, int c, int k) {
    return (unsigned)(((line * 32 + c) * 64) + ((((k >> 3) ^ (c >> 3)) & 3) << 4) + (k & 7) * 2);
}

template <typename T>
__global__ void __launch_bounds__(SM_NT, 4) smlp_mix_kernel(const SmlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, C = p.C;
    const int npx = H * W;
    char* const LA = smem;                                 // [h][c][w]
    char* const LB = smem + H * 32 * 64;                   // [w][c][h]
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    // zero both copies once: the positions >= W (>= H) of every row are never written again
    for (int i = tid; i < (H + W) * 32 * 4; i += SM_NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
    // weight fragments: B'[k][n] = Ww[w' = nb * 16 + n][w = k], lane n = lane & 15, k = 8 (lane >> 4) ..
    const int fn = lane & 15, fq = lane >> 4;
    u32x4 fw[2], fh[2];
    float bwv[2], bhv[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        fw[nb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.ww) + (nb * 16 + fn) * 32 + fq * 8);
        fh[nb] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.wh) + (nb * 16 + fn) * 32 + fq * 8);
        bwv[nb] = p.bw[nb * 16 + fn];
        bhv[nb] = p.bh[nb * 16 + fn];
    }
    const int cgroups = C / 32;
    const int units = p.B * cgroups;
    // per-piece geometry, the same for every unit: 32-bit byte offsets of the piece in x / of its x^ in out (relative to the unit's origin) and
    // the LDS addresses of its first channel in the two copies (channel e of the octet: + 64 e)
    unsigned xoff[SM_NL], ooff[SM_NL], la[SM_NL], lb[SM_NL];
#pragma unroll
    for (int i = 0; i < SM_NL; ++i) {
        int idx = tid + i * SM_NT;
        idx = idx < npx * 4 ? idx : npx * 4 - 1;             // (clamped, not branched: a load behind a branch is a serialised load)
        const int px = idx >> 2, o = idx & 3;
        const int hh = px / W, ww_ = px - hh * W;
        xoff[i] = (unsigned)(px * p.ldx + o * 8) * (unsigned)sizeof(T);
        ooff[i] = (unsigned)(px * p.ldo + o * 8) * (unsigned)sizeof(T);
        la[i] = sm_addr(hh, o * 8, ww_);
        lb[i] = (unsigned)(H * 32 * 64) + sm_addr(ww_, o * 8, hh);
    }
    u32x4 raw[SM_NL];
    auto request = [&](const int u) {
        const int b = u / cgroups, c0 = (u - b * cgroups) * 32;
        const char* base = reinterpret_cast<const char*>(x + (size_t)b * npx * p.ldx + c0);
#pragma unroll
        for (int i = 0; i < SM_NL; ++i) raw[i] = *reinterpret_cast<const u32x4*>(base + xoff[i]);
    };
    int u = blockIdx.x;
    if (u < units) request(u);
    __syncthreads();
    for (; u < units; u += gridDim.x) {
        const int b = u / cgroups, c0 = (u - b * cgroups) * 32;
        // ---- x^ = s x + h: out[:, 2C + c] from the registers, the two transposed copies into LDS
        {
            const int o = tid & 3;
            float sc[8], sh[8];
            {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.bn_s + c0 + o * 8), s1 = *reinterpret_cast<const f32x4*>(p.bn_s + c0 + o * 8 + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.bn_h + c0 + o * 8), h1 = *reinterpret_cast<const f32x4*>(p.bn_h + c0 + o * 8 + 4);
                sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
                sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
            }
            char* const ob = reinterpret_cast<char*>(out + (size_t)b * npx * p.ldo + 2 * C + c0);
#pragma unroll
            for (int i = 0; i < SM_NL; ++i) {
                if (tid + i * SM_NT < npx * 4) {
                    T e8[8];
                    __builtin_memcpy(e8, &raw[i], 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        e8[e] = from_f32<T>(__builtin_fmaf(to_f32(e8[e]), sc[e], sh[e]));
                        unsigned short bits;
                        __builtin_memcpy(&bits, &e8[e], 2);
                        *reinterpret_cast<unsigned short*>(smem + la[i] + e * 64) = bits;
                        *reinterpret_cast<unsigned short*>(smem + lb[i] + e * 64) = bits;
                    }
                    u32x4 v;
                    __builtin_memcpy(&v, e8, 16);
                    *reinterpret_cast<u32x4*>(ob + ooff[i]) = v;
                }
            }
        }
        if (u + (int)gridDim.x < units) request(u + gridDim.x);      // the next unit's pixels travel during the MFMA phase
        __syncthreads();
        // ---- the lines: H of the mix along w (outputs x_w), W of the mix along h (outputs x_h); a wave takes every eighth
        // A' fragment of block j: lane row m = lane & 15 is channel 8 (m >> 2) + 4 j + (m & 3), k = 8 (lane >> 4) ..
        const int am = lane & 15;
        for (int line = wave; line < H + W; line += SM_NT / 64) {
            const bool alongw = line < H;
            const int ln = alongw ? line : line - H;
            const char* src = alongw ? LA : LB;
            u32x4 a[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = 8 * (am >> 2) + 4 * j + (am & 3);
                a[j] = *reinterpret_cast<const u32x4*>(src + sm_addr(ln, c, fq * 8));
            }
            const int nout = alongw ? W : H;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (nb * 16 >= nout) break;                      // (wave-uniform)
                const u32x4 bfrag = alongw ? fw[nb] : fh[nb];
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 d0 = SmMma<T>::run(a[0], bfrag, z);
                const f32x4 d1 = SmMma<T>::run(a[1], bfrag, z);
                const int pos = nb * 16 + fn;                    // output position w' (h')
                if (pos < nout) {
                    const float bias = alongw ? bwv[nb] : bhv[nb];
                    T e8[8] = {from_f32<T>(d0.x + bias), from_f32<T>(d0.y + bias), from_f32<T>(d0.z + bias), from_f32<T>(d0.w + bias),
                               from_f32<T>(d1.x + bias), from_f32<T>(d1.y + bias), from_f32<T>(d1.z + bias), from_f32<T>(d1.w + bias)};
                    u32x4 v;
                    __builtin_memcpy(&v, e8, 16);
                    const int px = alongw ? ln * W + pos : pos * W + ln;
                    *reinterpret_cast<u32x4*>(out + ((size_t)b * npx + px) * p.ldo + (alongw ? C : 0) + c0 + fq * 8) = v;
                }
            }
        }
        __syncthreads();                                         // the tile is free for the next unit
    }
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_smlp_mix_supported(int dtype, int H, int W, int C) {
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && H >= 1 && W >= 1 && H <= 32 && W <= 32 && C >= 32 && C % 32 == 0 && (H + W) * 2048 <= 160 * 1024;
}

extern "C" int mlpk_smlp_mix(int dtype, const void* x, int ldx, int B, int H, int W, int C, const float* bn_scale, const float* bn_shift,
                             const void* wh, const float* bh, const void* ww, const float* bw, void* out, int ldo, void* stream) {
    if (!x || !out || !bn_scale || !bn_shift || !wh || !ww || !bh || !bw) return MLPK_ENULL;
    if (B <= 0 || !mlpk_smlp_mix_supported(dtype, H, W, C)) return MLPK_ESHAPE;
    if (ldx < C || ldx % 8 || ldo < 3 * C || ldo % 8) return MLPK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)wh & 15) || ((uintptr_t)ww & 15) || ((uintptr_t)bn_scale & 15) || ((uintptr_t)bn_shift & 15))
        return MLPK_EALIGN;
    SmlpArgs a;
    a.x = x; a.out = out; a.bn_s = bn_scale; a.bn_h = bn_shift; a.wh = wh; a.ww = ww; a.bh = bh; a.bw = bw;
    a.B = B; a.H = H; a.W = W; a.C = C; a.ldx = ldx; a.ldo = ldo;
    const int lds = (H + W) * 32 * 64;
    const int units = B * (C / 32);
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
    const int per_cu = lds <= 78 * 1024 ? 2 : 1;
    const int grid = units < cu * per_cu ? units : cu * per_cu;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (dtype == MLPK_BF16) {
        auto k = smlp_mix_kernel<bf16_t>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(SM_NT), lds, s, a);
    } else {
        auto k = smlp_mix_kernel<f16_t>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(SM_NT), lds, s, a);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}
