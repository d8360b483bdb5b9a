// Patch gather: the im2col half of a kernel == stride convolution (patch embedding, S2 stage
// embedding, AS-MLP PatchMerging).  The other half is mlpk_gemm_nt.  No im2col buffer beyond
// the GEMM's own A operand is ever materialised, and the dtype conversion of the input image
// (fp32 NCHW -> bf16/f16 tokens) happens here, in the same pass.
#include "mlpk_common.h"

namespace mlpk {

struct PatchArgs {
    const void* src;
    void* out;
    int B, Cin, H, W, ph, pw, sh, sw, pad, px_stride, ldo, order, layout;      // ph x pw window, sh x sw stride
    int Hp, Wp, K;
    int64_t rows;
};

template <typename TS, typename TD>
__global__ void __launch_bounds__(256) patchify_kernel(const PatchArgs p) {
    const TS* __restrict__ src = reinterpret_cast<const TS*>(p.src);
    TD* __restrict__ out = reinterpret_cast<TD*>(p.out);
    const int chunks = p.ldo / 8;
    const int64_t total = p.rows * chunks;
    const bool small = total <= 0x7fffffffll;             // 32-bit index arithmetic where it fits (64-bit divisions cost ~100 instructions each)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t row;
        int k0, wp, hp, b;
        if (small) {
            const unsigned i32 = (unsigned)idx, r32 = i32 / (unsigned)chunks, hw = r32 / (unsigned)p.Wp;
            row = r32;
            k0 = (int)(i32 - r32 * (unsigned)chunks) * 8;
            wp = (int)(r32 - hw * (unsigned)p.Wp);
            b = (int)(hw / (unsigned)p.Hp);
            hp = (int)(hw - (unsigned)b * (unsigned)p.Hp);
        } else {
            row = idx / chunks;
            k0 = (int)(idx - row * chunks) * 8;
            wp = (int)(row % p.Wp);
            hp = (int)((row / p.Wp) % p.Hp);
            b = (int)(row / ((int64_t)p.Wp * p.Hp));
        }
        TD e[8];
        if (p.layout == MLPK_LAYOUT_NHWC && (p.Cin & 7) == 0 && k0 < p.K) {
            // 8 consecutive k share the patch offset (i,j): one contiguous 8-channel run
            const int q = k0 / p.Cin;
            const int ci = k0 - q * p.Cin;
            const int i = p.order == 1 ? (q & 1) : q / p.pw;
            const int j = p.order == 1 ? (q >> 1) : q - (q / p.pw) * p.pw;
            const int y = hp * p.sh + i - p.pad, x = wp * p.sw + j - p.pad;
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
                const TS* s = src + (((int64_t)b * p.H + y) * p.W + x) * p.px_stride + ci;
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = from_f32<TD>(to_f32(s[t]));
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = from_f32<TD>(0.f);
            }
        } else if (p.layout == MLPK_LAYOUT_NCHW && (p.pw & 7) == 0 && p.sw == p.pw && p.sh == p.ph && p.pad == 0 && (p.W & 7) == 0 && k0 < p.K &&
                   (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            // 8 consecutive k = 8 consecutive pixels of one image row of one channel: 16-byte loads
            const int ci = k0 / (p.ph * p.pw);
            const int rem = k0 - ci * (p.ph * p.pw);
            const int i = rem / p.pw;
            const int j = rem - i * p.pw;
            const TS* s = src + (((int64_t)b * p.Cin + ci) * p.H + (hp * p.ph + i)) * p.W + wp * p.pw + j;
            if constexpr (sizeof(TS) == 4) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
                e[0] = from_f32<TD>(a0.x); e[1] = from_f32<TD>(a0.y); e[2] = from_f32<TD>(a0.z); e[3] = from_f32<TD>(a0.w);
                e[4] = from_f32<TD>(a1.x); e[5] = from_f32<TD>(a1.y); e[6] = from_f32<TD>(a1.z); e[7] = from_f32<TD>(a1.w);
            } else {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(s);
                TS t8[8];
                __builtin_memcpy(t8, &raw, 16);
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = from_f32<TD>(to_f32(t8[t]));
            }
        } else if (p.layout == MLPK_LAYOUT_NCHW) {
            // overlapping / padded windows on NCHW (the 7 x 7 stride-4 stems of Hire-MLP and CycleMLP): k -> (ci, i, j) decoded ONCE per
            // chunk and advanced with carries (round 6: two divisions per element made this path ALU-bound, 230 us for 77 MB in / 244 MB out)
            int ci = k0 / (p.ph * p.pw);
            const int rem = k0 - ci * (p.ph * p.pw);
            int i = rem / p.pw;
            int j = rem - i * p.pw;
            const int y0 = hp * p.sh - p.pad, x0 = wp * p.sw - p.pad;
            const TS* img = src + (int64_t)b * p.Cin * p.H * p.W;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int y = y0 + i, x = x0 + j;
                const bool ok = k0 + t < p.K && y >= 0 && y < p.H && x >= 0 && x < p.W;
                const TS raw = img[ok ? ((int64_t)ci * p.H + y) * p.W + x : 0];          // unconditional load: the eight are in flight together
                e[t] = ok ? from_f32<TD>(to_f32(raw)) : from_f32<TD>(0.f);
                if (++j == p.pw) { j = 0; if (++i == p.ph) { i = 0; ++ci; } }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = k0 + t;
                float v = 0.f;
                if (k < p.K) {
                    const int q = k / p.Cin;
                    const int ci = k - q * p.Cin;
                    const int i = p.order == 1 ? (q & 1) : q / p.pw;
                    const int j = p.order == 1 ? (q >> 1) : q - (q / p.pw) * p.pw;
                    const int y = hp * p.sh + i - p.pad, x = wp * p.sw + j - p.pad;
                    if (y >= 0 && y < p.H && x >= 0 && x < p.W) v = to_f32(src[(((int64_t)b * p.H + y) * p.W + x) * p.px_stride + ci]);
                }
                e[t] = from_f32<TD>(v);
            }
        }
        TD* o = out + row * p.ldo + k0;
        if constexpr (sizeof(TD) == 2) {
            u32x4 t;
            __builtin_memcpy(&t, e, 16);
            *reinterpret_cast<u32x4*>(o) = t;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = e[t];
        }
    }
}

template <typename TS> static int patchify_from(int dst, const PatchArgs& a, unsigned grid, hipStream_t s) {
    switch (dst) {
        case MLPK_F32: hipLaunchKernelGGL((patchify_kernel<TS, float>), dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_F16: hipLaunchKernelGGL((patchify_kernel<TS, f16_t>), dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_BF16: hipLaunchKernelGGL((patchify_kernel<TS, bf16_t>), dim3(grid), dim3(256), 0, s, a); break;
        default: return MLPK_EDTYPE;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

// ---- round 6: the 4 x 4 patch embedding of the hierarchical families in ONE kernel (mlpk_patch_embed4) --------------------------------------
// swin_mlp.py:324-333 / ms_mlp.py:255-262 / as_mlp.py:319,330 / sparse_mlp.py: Conv2d(3 -> C, k = stride = 4) on the NCHW image, flatten, transpose,
// (LayerNorm).  As separate launches this was the patch gather (77 MB in, 77 MB out), the GEMM (K = 48: 88 us for 0.07 GFLOP per image), a statistics
// pass and the normalise pass: 275 us per forward and five trips over the tokens.  Here a wave owns 32 consecutive patches and multiplies them where
// they are loaded: with the weight as the FIRST operand of v_mfma_f32_32x32x16 (M = 32 channels) a lane is ONE patch n = lane & 31 and the k-half
// kh = lane >> 5, and its second-operand fragment of k-step ks -- k = 16 ks + 8 kh .. + 7 in the Conv2d's (channel, row, column) order -- is rows 2 kh,
// 2 kh + 1 of input channel ks of its own patch: two 8-byte loads straight from the image (32 lanes = 256 contiguous bytes), no gather buffer, no LDS.
// The weight fragments (C / 32 x 3 x 16 bytes per lane) stay in registers.  A lane ends up with 16 channels per block of ONE token, so the LayerNorm
// is the token's own registers and one exchange with the other k-half: bias, rounding to the storage type (what the GEMM stored), two-pass
// statistics of the rounded values (what mlpk_row_stats computed), (v - mean) rstd gamma + beta, one more rounding.  The tokens leave through a
// per-wave LDS image as whole rows: 32 tokens x C x 2 contiguous bytes.
template <typename T> struct PeMma;
template <> struct PeMma<bf16_t> {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct PeMma<f16_t> {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

struct PatchEmbedArgs {
    const void* x;          // (B, 3, H, W) NCHW
    const void* w;          // (C, ldw) 16-bit, k = ci * 16 + i * 4 + j
    const float* bias;      // (C) or NULL
    const float* gamma;     // (C) LayerNorm affine, or NULL: no LayerNorm
    const float* beta;
    void* out;              // (B * Hp * Wp, ldo)
    int B, H, W, Hp, Wp, ldw, ldo;
    float eps;
    long long npatch;
};

template <typename TS, typename T, int NMB>
__global__ void __launch_bounds__(256) patch_embed4_kernel(const PatchEmbedArgs p) {
    typedef typename PeMma<T>::f32x16 f32x16;
    constexpr int C = NMB * 32;
    constexpr int ROWB = C * 2 + 16;                          // LDS row of a token (+ 16 bytes: the 8-byte writers of a phase on different banks)
    extern __shared__ __attribute__((aligned(16))) char pe_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, kh = lane >> 5;
    char* const img = pe_smem + wave * (32 * ROWB);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.w);
    const TS* __restrict__ x = reinterpret_cast<const TS*>(p.x);
    u32x4 wf[NMB][3];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) wf[mb][ks] = *reinterpret_cast<const u32x4*>(wgt + (size_t)(mb * 32 + n) * p.ldw + 16 * ks + 8 * kh);
    const long long first = ((long long)blockIdx.x * 4 + wave) * 32;
    if (first >= p.npatch) return;                            // (wave-uniform; no workgroup barrier below)
    long long pi = first + n;
    const bool live = pi < p.npatch;
    pi = live ? pi : p.npatch - 1;
    const int px = (int)(pi % p.Wp);
    const long long t = pi / p.Wp;
    const int py = (int)(t % p.Hp);
    const long long b = t / p.Hp;
    // ---- the token's 48 values: channel ks, rows 2 kh and 2 kh + 1 of the patch, 4 pixels each
    u32x4 bf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const TS* r0 = x + ((b * 3 + ks) * p.H + 4 * py + 2 * kh) * p.W + 4 * px;
        T e[8];
        if constexpr (sizeof(TS) == 4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0), a1 = *reinterpret_cast<const f32x4*>(r0 + p.W);
            e[0] = from_f32<T>(a0.x); e[1] = from_f32<T>(a0.y); e[2] = from_f32<T>(a0.z); e[3] = from_f32<T>(a0.w);
            e[4] = from_f32<T>(a1.x); e[5] = from_f32<T>(a1.y); e[6] = from_f32<T>(a1.z); e[7] = from_f32<T>(a1.w);
        } else {
            TS s8[8];
            *reinterpret_cast<u32x2*>(s8) = *reinterpret_cast<const u32x2*>(r0);
            *reinterpret_cast<u32x2*>(s8 + 4) = *reinterpret_cast<const u32x2*>(r0 + p.W);
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = from_f32<T>(to_f32(s8[q]));
        }
        __builtin_memcpy(&bf[ks], e, 16);
    }
    f32x16 acc[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc[mb] = PeMma<T>::run(wf[mb][ks], bf[ks], acc[mb]);
    }
    // lane (token n, kh): acc[mb][r] = channel 32 mb + 8 (r >> 2) + 4 kh + (r & 3)
    float v[NMB][16];
    float s = 0.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 32 * mb + 8 * q + 4 * kh;
            const f32x4 bz = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float b4[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[mb][4 * q + i] = to_f32(from_f32<T>(acc[mb][4 * q + i] + b4[i]));          // what the GEMM stored
                s += v[mb][4 * q + i];
            }
        }
    if (p.gamma) {
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = v[mb][r] - mean; ss = __builtin_fmaf(d, d, ss); }
        ss += __shfl_xor(ss, 32);
        const float rstd = 1.0f / __builtin_sqrtf(ss / (float)C + p.eps);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * mb + 8 * q + 4 * kh;
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c0), be = *reinterpret_cast<const f32x4*>(p.beta + c0);
                const float g4[4] = {g.x, g.y, g.z, g.w}, b4[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) v[mb][4 * q + i] = (v[mb][4 * q + i] - mean) * rstd * g4[i] + b4[i];      // (mlpk_norm_apply's expression)
            }
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            T e[4] = {from_f32<T>(v[mb][4 * q]), from_f32<T>(v[mb][4 * q + 1]), from_f32<T>(v[mb][4 * q + 2]), from_f32<T>(v[mb][4 * q + 3])};
            u32x2 pk;
            __builtin_memcpy(&pk, e, 8);
            *reinterpret_cast<u32x2*>(img + n * ROWB + (32 * mb + 8 * q + 4 * kh) * 2) = pk;
        }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- 32 tokens x C x 2 bytes, rows contiguous in memory where ldo == C
    constexpr int VPR = C / 8;                                // 16-byte vectors per token
    T* const orow = reinterpret_cast<T*>(p.out) + first * p.ldo;
    const long long left = p.npatch - first;
#pragma unroll
    for (int i = 0; i < (32 * VPR + 63) / 64; ++i) {
        const int idx = lane + 64 * i;
        const int tk = idx / VPR, vq = idx - tk * VPR;
        if (idx < 32 * VPR && tk < left) *reinterpret_cast<u32x4*>(orow + (size_t)tk * p.ldo + vq * 8) = *reinterpret_cast<const u32x4*>(img + tk * ROWB + vq * 16);
    }
}

// ---- round 6: the 7 x 7 stride-4 stems of Hire-MLP and CycleMLP as a direct convolution (mlpk_stem7) -----------------------------------------------
// hire_mlp.py:21 (pad 3), cycle_mlp.py:261 (pad 2): Conv2d(3 -> C, k = 7, stride = 4) on the NCHW image -> channel-last rows.  As window gather + GEMM
// this wrote and re-read a 244 MB operand (K = 147 padded to 152) for a 77 MB image: 177 + 130 us per forward.  Here a workgroup takes one image and TWO
// output rows: the 11 input rows x 3 channels it needs are staged in LDS once (zeros outside the image; tile column = x + pad, so the window of output
// column ox starts at column 4 ox: 8-byte aligned), and the product runs on v_mfma_f32_32x32x16 with the weight as the first operand -- a lane is one
// output pixel n = lane & 31 and the k-half kh; K is ordered (channel, window row, 8 columns): k-step ks, half kh is window row rr = 2 ks + kh (of 21),
// its 8 columns one 16-byte piece of the tile (two ds_read_b64); the eighth column and row 21 meet zero weights.  The weight (C x 176, packed by
// engine.pack_stem7) stays in registers; bias, one rounding, the pixels leave through a per-wave LDS image as whole rows.
struct Stem7Args {
    const void* x;          // (B, 3, H, W)
    const void* w;          // (C, 176): k = (ci * 7 + i) * 8 + j, zero for j = 7 and k >= 168
    const float* bias;
    void* out;              // (B * Ho * Wo, ldo)
    int B, H, W, Ho, Wo, pad, ldo, pitch;       // pitch: tile row in elements
    float* out_mean;        // or NULL: LayerNorm statistics (two-pass, of the rounded values) of the rows written, per pixel
    float* out_rstd;
    float eps;
};

template <typename TS, typename T, int NMB>
__global__ void __launch_bounds__(256) stem7_kernel(const Stem7Args p) {
    typedef typename PeMma<T>::f32x16 f32x16;
    constexpr int C = NMB * 32;
    constexpr int ROWB = C * 2 + 16;
    constexpr int NKS = 11;
    extern __shared__ __attribute__((aligned(16))) char st_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int pitch = p.pitch;
    T* const tile = reinterpret_cast<T*>(st_smem);                        // [3][11][pitch]
    char* const img = st_smem + 33 * pitch * 2 + wave * (32 * ROWB);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.w);
    const int opairs = (p.Ho + 1) >> 1;
    const int b = blockIdx.x / opairs, oy0 = (blockIdx.x - b * opairs) * 2;
    u32x4 wf[NMB][NKS];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) wf[mb][ks] = *reinterpret_cast<const u32x4*>(wgt + (size_t)(mb * 32 + n) * 176 + 16 * ks + 8 * kh);
    // ---- stage: zero the tile, then the image rows 4 oy0 - pad .. + 10 of the three channels at tile column x + pad
    for (int i = tid; i < 33 * pitch / 8; i += 256) reinterpret_cast<u32x4*>(tile)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    {
        const TS* __restrict__ x = reinterpret_cast<const TS*>(p.x) + (size_t)b * 3 * p.H * p.W;
        const int vpr = p.W / 8;                                          // 8-pixel pieces per image row
        const int y0 = 4 * oy0 - p.pad;
        const int total = 33 * vpr;
        // four pieces per thread and pass, their loads UNCONDITIONAL (a row outside the image reads row 0 and is not written) and in flight together:
        // behind `continue` every load waited for the one before it
        for (int base = tid; base < total; base += 4 * 256) {
            T e[4][8];
            int dsto[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 256;
                const int ic = i < total ? i : total - 1;
                const int row = ic / vpr, v = ic - row * vpr;             // row = ci * 11 + r
                const int ci = row / 11, r = row - ci * 11;
                const int y = y0 + r;
                const bool ok = i < total && y >= 0 && y < p.H;
                const TS* src = x + ((size_t)ci * p.H + (ok ? y : 0)) * p.W + v * 8;
                if constexpr (sizeof(TS) == 4) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
                    e[u][0] = from_f32<T>(a0.x); e[u][1] = from_f32<T>(a0.y); e[u][2] = from_f32<T>(a0.z); e[u][3] = from_f32<T>(a0.w);
                    e[u][4] = from_f32<T>(a1.x); e[u][5] = from_f32<T>(a1.y); e[u][6] = from_f32<T>(a1.z); e[u][7] = from_f32<T>(a1.w);
                } else {
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(src);
                    __builtin_memcpy(e[u], &raw, 16);
                }
                dsto[u] = ok ? row * pitch + v * 8 + p.pad : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (dsto[u] < 0) continue;
#pragma unroll
                for (int q = 0; q < 8; ++q) tile[dsto[u] + q] = e[u][q];
            }
        }
    }
    __syncthreads();
    // ---- this wave's 32 of the workgroup's 2 Wo pixels
    const int npx = (p.Ho - oy0 >= 2 ? 2 : 1) * p.Wo;
    const int first = wave * 32;
    if (first >= npx) return;                                             // (wave-uniform; no workgroup barrier below)
    int pi = first + n;
    pi = pi < npx ? pi : npx - 1;
    const int prow = pi >= p.Wo ? 1 : 0, ox = pi - prow * p.Wo;
    const T* const base = tile + (prow * 4) * pitch + 4 * ox;
    f32x16 acc[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int rr0 = 2 * ks, rr1 = 2 * ks + 1;                         // window rows of the two k-halves (compile-time)
        const int off0 = ((rr0 / 7) * 11 + rr0 % 7) * pitch;
        const int off1 = rr1 < 21 ? ((rr1 / 7) * 11 + rr1 % 7) * pitch : off0;
        const T* src = base + (kh ? off1 : off0);
        u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 4);
        if (rr1 >= 21 && kh) { lo = u32x2{0u, 0u}; hi = u32x2{0u, 0u}; }
        hi.y &= 0xffffu;                                                  // the eighth column is not part of the window (0 x inf must not be NaN)
        const u32x4 bfrag = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) acc[mb] = PeMma<T>::run(wf[mb][ks], bfrag, acc[mb]);
    }
    float ssum = 0.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 32 * mb + 8 * q + 4 * kh;
            const f32x4 bz = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
            T e[4] = {from_f32<T>(acc[mb][4 * q] + bz.x), from_f32<T>(acc[mb][4 * q + 1] + bz.y), from_f32<T>(acc[mb][4 * q + 2] + bz.z),
                      from_f32<T>(acc[mb][4 * q + 3] + bz.w)};
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[mb][4 * q + i] = to_f32(e[i]); ssum += acc[mb][4 * q + i]; }       // (the rounded values: what is stored)
            u32x2 pk;
            __builtin_memcpy(&pk, e, 8);
            *reinterpret_cast<u32x2*>(img + n * ROWB + c0 * 2) = pk;
        }
    const int left = npx - first;
    if (p.out_mean) {
        ssum += __shfl_xor(ssum, 32);
        const float mean = ssum / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[mb][r] - mean; ss = __builtin_fmaf(d, d, ss); }
        ss += __shfl_xor(ss, 32);
        if (kh == 0 && n < left) {
            const size_t row = ((size_t)b * p.Ho + oy0) * p.Wo + first + n;
            p.out_mean[row] = mean;
            p.out_rstd[row] = 1.0f / __builtin_sqrtf(ss / (float)C + p.eps);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int VPR = C / 8;
    T* const orow = reinterpret_cast<T*>(p.out) + (((size_t)b * p.Ho + oy0) * p.Wo + first) * p.ldo;
#pragma unroll
    for (int i = 0; i < (32 * VPR + 63) / 64; ++i) {
        const int idx = lane + 64 * i;
        const int tk = idx / VPR, vq = idx - tk * VPR;
        if (idx < 32 * VPR && tk < left) *reinterpret_cast<u32x4*>(orow + (size_t)tk * p.ldo + vq * 8) = *reinterpret_cast<const u32x4*>(img + tk * ROWB + vq * 16);
    }
}

// ---- round 6: LayerNorm statistics of 2 x 2 merged rows without the merged tensor (mlpk_merge2x2_row_stats) -----------------------------------------
// PatchMerging (swin_mlp.py:203-210, sparse_mlp.py:40-48) normalises the concatenation of a 2 x 2 window's four pixels (4 C values) in front of its
// reduction Linear.  With the reduction running as an implicit-convolution product (mlpk_conv_gemm_nhwc) the concatenated tensor is never stored; its row
// statistics come from here: one wave per window, the four pixels' 16-byte pieces spread over the lanes, two passes in registers (mean, then centred
// squares: mlpk_row_stats' definition), biased variance.
template <typename T>
__global__ void __launch_bounds__(256) merge2x2_row_stats_kernel(const T* __restrict__ x, int B, int H, int W, int C, float eps, float* __restrict__ mean,
                                                                 float* __restrict__ rstd) {
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int MAXV = 12;                                   // pieces per lane: 4 C / EPV / 64 <= 12 (C <= 1536 for 16-bit)
    const int lane = threadIdx.x & 63;
    const int H2 = H >> 1, W2 = W >> 1;
    const long long rows = (long long)B * H2 * W2;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ox = (int)(row % W2);
    const long long t = row / W2;
    const int oy = (int)(t % H2);
    const long long b = t / H2;
    const int cv = C / EPV, nv = 4 * cv;
    float v[MAXV][EPV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const int q = idx / cv, c = (idx - q * cv) * EPV;
            const T* src = x + (((b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * (long long)C + c);
            T e[EPV];
            *reinterpret_cast<u32x4*>(e) = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
            for (int k = 0; k < EPV; ++k) { v[i][k] = to_f32(e[k]); s += v[i][k]; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / (float)(4 * C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
#pragma unroll
            for (int k = 0; k < EPV; ++k) { const float d = v[i][k] - mu; ss = __builtin_fmaf(d, d, ss); }
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = 1.0f / __builtin_sqrtf(ss / (float)(4 * C) + eps);
    }
}

// The same statistics from the PER-PIXEL LayerNorm statistics the producing GEMM's epilogue already delivered (mlpk_merge2x2_stats_combine): with
// var_q = 1 / rstd_q^2 - eps_in of the window's four pixels, mean = avg(mean_q), var = avg(var_q + mean_q^2) - mean^2 -- combined in fp64 (a residual
// stream's mean can be large against its spread) -- so the merged rows' statistics cost 1.6 MB of traffic instead of a pass over the activations.
__global__ void __launch_bounds__(256) merge2x2_stats_combine_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, int B, int H, int W, float eps_in,
                                                                     float eps_out, float* __restrict__ out_mean, float* __restrict__ out_rstd) {
    const int H2 = H >> 1, W2 = W >> 1;
    const long long rows = (long long)B * H2 * W2;
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int ox = (int)(row % W2);
    const long long t = row / W2;
    const int oy = (int)(t % H2);
    const long long b = t / H2;
    double sm = 0.0, sv = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long px = (b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1);
        const double m = (double)mean[px], r = (double)rstd[px];
        double v = 1.0 / (r * r) - (double)eps_in;
        v = v > 0.0 ? v : 0.0;
        sm += m;
        sv += v + m * m;
    }
    const double mu = sm * 0.25;
    double var = sv * 0.25 - mu * mu;
    var = var > 0.0 ? var : 0.0;
    out_mean[row] = (float)mu;
    out_rstd[row] = (float)(1.0 / __builtin_sqrt(var + (double)eps_out));
}

}  // namespace mlpk

using namespace mlpk;

static int im2col_impl(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin, int H, int W,
                       int ph, int pw, int sh, int sw, int pad, int src_px_stride, int ldo, int order, void* stream) {
    if (!src || !out) return MLPK_ENULL;
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0 || pad < 0) return MLPK_ESHAPE;
    if (src_layout != MLPK_LAYOUT_NCHW && src_layout != MLPK_LAYOUT_NHWC) return MLPK_EMODE;
    if (order != 0 && order != 1) return MLPK_EMODE;
    if (order == 1 && (src_layout != MLPK_LAYOUT_NHWC || ph != 2 || pw != 2)) return MLPK_EMODE;
    PatchArgs a;
    a.src = src; a.out = out; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.ph = ph; a.pw = pw; a.sh = sh; a.sw = sw; a.pad = pad;
    a.px_stride = src_layout == MLPK_LAYOUT_NHWC ? src_px_stride : 0;
    a.ldo = ldo; a.order = order; a.layout = src_layout;
    if (H + 2 * pad < ph || W + 2 * pad < pw) return MLPK_ESHAPE;
    a.Hp = (H + 2 * pad - ph) / sh + 1;
    a.Wp = (W + 2 * pad - pw) / sw + 1;
    a.K = Cin * ph * pw;
    if (a.Hp <= 0 || a.Wp <= 0) return MLPK_ESHAPE;
    if (ldo % 8 || ldo < a.K) return MLPK_ESHAPE;
    if (src_layout == MLPK_LAYOUT_NHWC && src_px_stride < Cin) return MLPK_ESHAPE;
    if ((uintptr_t)out & 15) return MLPK_EALIGN;
    a.rows = (int64_t)B * a.Hp * a.Wp;
    const int64_t total = a.rows * (ldo / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (src_dtype) {
        case MLPK_F32: return patchify_from<float>(dst_dtype, a, grid, s);
        case MLPK_F16: return patchify_from<f16_t>(dst_dtype, a, grid, s);
        case MLPK_BF16: return patchify_from<bf16_t>(dst_dtype, a, grid, s);
        default: return MLPK_EDTYPE;
    }
}

extern "C" int mlpk_patchify(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin,
                             int H, int W, int ph, int pw, int pad, int src_px_stride, int ldo, int order,
                             void* stream) {
    return im2col_impl(src_dtype, dst_dtype, src_layout, src, out, B, Cin, H, W, ph, pw, ph, pw, pad, src_px_stride, ldo, order, stream);
}

extern "C" int mlpk_im2col(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin,
                           int H, int W, int kh, int kw, int stride_h, int stride_w, int pad, int src_px_stride, int ldo,
                           void* stream) {
    return im2col_impl(src_dtype, dst_dtype, src_layout, src, out, B, Cin, H, W, kh, kw, stride_h, stride_w, pad, src_px_stride, ldo, 0, stream);
}

extern "C" int mlpk_patch_embed4_supported(int src_dtype, int dst_dtype, int Cin, int H, int W, int C);
extern "C" int mlpk_patch_embed4(int src_dtype, int dst_dtype, const void* x, int B, int Cin, int H, int W, const void* w, int ldw, const float* bias,
                                 const float* gamma, const float* beta, float eps, void* out, int ldo, int C, void* stream);

extern "C" int mlpk_patch_embed4_supported(int src_dtype, int dst_dtype, int Cin, int H, int W, int C) {
    return (dst_dtype == MLPK_F16 || dst_dtype == MLPK_BF16) && (src_dtype == dst_dtype || src_dtype == MLPK_F32) && Cin == 3 && H >= 4 && W >= 4 &&
           H % 4 == 0 && W % 4 == 0 && C >= 32 && C <= 128 && C % 32 == 0;
}

extern "C" int mlpk_patch_embed4(int src_dtype, int dst_dtype, const void* x, int B, int Cin, int H, int W, const void* w, int ldw, const float* bias,
                                 const float* gamma, const float* beta, float eps, void* out, int ldo, int C, void* stream) {
    if (!x || !w || !out) return MLPK_ENULL;
    if ((gamma != nullptr) != (beta != nullptr)) return MLPK_ENULL;
    if (B <= 0 || !mlpk_patch_embed4_supported(src_dtype, dst_dtype, Cin, H, W, C)) return MLPK_ESHAPE;
    if (ldw < 48 || ldw % 8 || ldo < C || ldo % 8 || (gamma && !(eps > 0.f))) return MLPK_ESHAPE;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)gamma | (uintptr_t)beta) & 15) return MLPK_EALIGN;
    PatchEmbedArgs a;
    a.x = x; a.w = w; a.bias = bias; a.gamma = gamma; a.beta = beta; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Hp = H / 4; a.Wp = W / 4; a.ldw = ldw; a.ldo = ldo; a.eps = eps;
    a.npatch = (long long)B * a.Hp * a.Wp;
    const long long wgs = (a.npatch + 127) / 128;
    if (wgs > 0x7fffffffll) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nmb = C / 32;
    const int lds = 4 * 32 * (C * 2 + 16);
#define PE_GO(TS, TD, NMB) hipLaunchKernelGGL((patch_embed4_kernel<TS, TD, NMB>), dim3((unsigned)wgs), dim3(256), lds, s, a)
#define PE_NMB(TS, TD)                                                                 \
    switch (nmb) {                                                                     \
        case 1: PE_GO(TS, TD, 1); break;                                               \
        case 2: PE_GO(TS, TD, 2); break;                                               \
        case 3: PE_GO(TS, TD, 3); break;                                               \
        default: PE_GO(TS, TD, 4); break;                                              \
    }
    if (dst_dtype == MLPK_BF16) {
        if (src_dtype == MLPK_F32) { PE_NMB(float, bf16_t) } else { PE_NMB(bf16_t, bf16_t) }
    } else {
        if (src_dtype == MLPK_F32) { PE_NMB(float, f16_t) } else { PE_NMB(f16_t, f16_t) }
    }
#undef PE_NMB
#undef PE_GO
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_stem7_supported(int src_dtype, int dst_dtype, int Cin, int H, int W, int pad, int C);
extern "C" int mlpk_stem7(int src_dtype, int dst_dtype, const void* x, int B, int Cin, int H, int W, int pad, const void* w, const float* bias, void* out,
                          int ldo, int C, float* out_mean, float* out_rstd, float eps, void* stream);

extern "C" int mlpk_stem7_supported(int src_dtype, int dst_dtype, int Cin, int H, int W, int pad, int C) {
    if (!((dst_dtype == MLPK_F16 || dst_dtype == MLPK_BF16) && (src_dtype == dst_dtype || src_dtype == MLPK_F32) && Cin == 3)) return 0;
    if (pad < 0 || pad > 3 || H + 2 * pad < 7 || W + 2 * pad < 7 || W % 8 || C < 32 || C > 128 || C % 32) return 0;
    const int Wo = (W + 2 * pad - 7) / 4 + 1;
    return 2 * Wo <= 128;
}

extern "C" int mlpk_stem7(int src_dtype, int dst_dtype, const void* x, int B, int Cin, int H, int W, int pad, const void* w, const float* bias, void* out,
                          int ldo, int C, float* out_mean, float* out_rstd, float eps, void* stream) {
    if (!x || !w || !out) return MLPK_ENULL;
    if ((out_mean != nullptr) != (out_rstd != nullptr)) return MLPK_ENULL;
    if (out_mean && !(eps > 0.f)) return MLPK_ESHAPE;
    if (B <= 0 || !mlpk_stem7_supported(src_dtype, dst_dtype, Cin, H, W, pad, C)) return MLPK_ESHAPE;
    if (ldo < C || ldo % 8) return MLPK_ESHAPE;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15) return MLPK_EALIGN;
    Stem7Args a;
    a.x = x; a.w = w; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W; a.pad = pad; a.ldo = ldo;
    a.out_mean = out_mean; a.out_rstd = out_rstd; a.eps = eps;
    a.Ho = (H + 2 * pad - 7) / 4 + 1;
    a.Wo = (W + 2 * pad - 7) / 4 + 1;
    int pitch = 4 * a.Wo + 8;                                  // the last window: columns 4 (Wo - 1) .. + 7
    if (pitch < W + pad) pitch = W + pad;
    a.pitch = (pitch + 7) / 8 * 8;
    const long long wgs = (long long)B * ((a.Ho + 1) / 2);
    if (wgs > 0x7fffffffll) return MLPK_ESHAPE;
    const int lds = 33 * a.pitch * 2 + 4 * 32 * (C * 2 + 16);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nmb = C / 32;
#define S7_GO(TS, TD, NMB) hipLaunchKernelGGL((stem7_kernel<TS, TD, NMB>), dim3((unsigned)wgs), dim3(256), lds, s, a)
#define S7_NMB(TS, TD)                                                                 \
    switch (nmb) {                                                                     \
        case 1: S7_GO(TS, TD, 1); break;                                               \
        case 2: S7_GO(TS, TD, 2); break;                                               \
        case 3: S7_GO(TS, TD, 3); break;                                               \
        default: S7_GO(TS, TD, 4); break;                                              \
    }
    if (dst_dtype == MLPK_BF16) {
        if (src_dtype == MLPK_F32) { S7_NMB(float, bf16_t) } else { S7_NMB(bf16_t, bf16_t) }
    } else {
        if (src_dtype == MLPK_F32) { S7_NMB(float, f16_t) } else { S7_NMB(f16_t, f16_t) }
    }
#undef S7_NMB
#undef S7_GO
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_merge2x2_row_stats(int dtype, const void* x, int B, int H, int W, int C, float eps, float* mean, float* rstd, void* stream);

extern "C" int mlpk_merge2x2_row_stats(int dtype, const void* x, int B, int H, int W, int C, float eps, float* mean, float* rstd, void* stream) {
    if (!x || !mean || !rstd) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || C % 8 || 4 * (C / 8) > 12 * 64 || !(eps > 0.f)) return MLPK_ESHAPE;
    if ((uintptr_t)x & 15) return MLPK_EALIGN;
    const long long rows = (long long)B * (H / 2) * (W / 2);
    const long long wgs = (rows + 3) / 4;
    if (wgs > 0x7fffffffll) return MLPK_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MLPK_F16) hipLaunchKernelGGL(merge2x2_row_stats_kernel<f16_t>, dim3((unsigned)wgs), dim3(256), 0, s, (const f16_t*)x, B, H, W, C, eps, mean, rstd);
    else hipLaunchKernelGGL(merge2x2_row_stats_kernel<bf16_t>, dim3((unsigned)wgs), dim3(256), 0, s, (const bf16_t*)x, B, H, W, C, eps, mean, rstd);
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_merge2x2_stats_combine(const float* mean, const float* rstd, int B, int H, int W, float eps_in, float eps_out, float* out_mean, float* out_rstd,
                                           void* stream);

extern "C" int mlpk_merge2x2_stats_combine(const float* mean, const float* rstd, int B, int H, int W, float eps_in, float eps_out, float* out_mean, float* out_rstd,
                                           void* stream) {
    if (!mean || !rstd || !out_mean || !out_rstd) return MLPK_ENULL;
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || !(eps_in >= 0.f) || !(eps_out > 0.f)) return MLPK_ESHAPE;
    const long long rows = (long long)B * (H / 2) * (W / 2);
    const long long wgs = (rows + 255) / 256;
    if (wgs > 0x7fffffffll) return MLPK_ESHAPE;
    hipLaunchKernelGGL(merge2x2_stats_combine_kernel, dim3((unsigned)wgs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mean, rstd, B, H, W, eps_in, eps_out,
                       out_mean, out_rstd);
    MLPK_LAUNCH_CHECK();
    return 0;
}
