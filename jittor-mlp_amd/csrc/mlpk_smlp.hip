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
    // DW: the block's first sublayer in front (sparse_mlp.py:88-91: x + dwconv3x3(BatchNorm(x)) + bias, zero padding on the BatchNorm output):
    // x is that sublayer's INPUT; its output x' is written to xres and is what the BatchNorm above applies to
    void* xres;             // (B*H*W, ldxr)
    const float* dw_w;      // (9, C): tap t = 3 dy + dx, per channel
    const float* dw_b;      // (C)
    const float* dw_s;      // (C) scale / shift of the BatchNorm in front of the convolution
    const float* dw_h;
    int ldxr;
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
// (16-byte pieces per thread = template parameter SM_NL: 8 covers 32 x 32 pixels x 4 pieces / 512 threads, 2 covers maps up to 16 x 16)

// element (line, channel c, position k) of a transposed copy: 64-byte rows, the four 16-byte chunks of a row XORed with the channel's octet
// (the writers of one instruction differ in the octet: without it they would all hit the same banks)
// Round 6: a line (32 rows of 64 bytes) is SM_LINE = 2048 + 16 bytes long.  With 2048 the writers of the SECOND copy -- 16 pixels of one image row per wave,
// i.e. 16 different lines at the same position inside the line -- met on 4 banks (16-way: SQ_LDS_BANK_CONFLICT 74 % of the LDS cycles of the 28 x 28 stage);
// 4 dwords of skew per line spread them over 16 (2-byte writes reach a quarter of the banks at best).
constexpr int SM_LINE = 32 * 64 + 16;
__device__ __forceinline__ unsigned sm_addr(int line, int c, int k) {
    return (unsigned)((line * SM_LINE + c * 64) + ((((k >> 3) ^ (c >> 3)) & 3) << 4) + (k & 7) * 2);
}

// DW (maps whose raw tile fits beside the two transposed copies: 14 x 14, 7 x 7): the depthwise 3 x 3 sublayer of the block runs on the same tile
// first -- the loaded pieces go to a plain copy [pixel][32 channels] in LDS, every thread then reads the 9 neighbours of its pieces (clamped
// address + mask: what lies outside the map contributes zero, as the padding of the BatchNorm's OUTPUT does), applies the BatchNorm per tap in
// fp32 exactly like mlpk_dwconv_affine_nhwc, and adds the centre value: the result x' is stored (the residual of the fuse GEMM) and replaces
// the piece.  One pass over the activation fewer per block; the same bits as the two kernels.
template <typename T, bool DW, int SM_NL>
__global__ void __launch_bounds__(SM_NT, 4) smlp_mix_kernel(const SmlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, C = p.C;
    const int npx = H * W;
    char* const LA = smem;                                 // [h][c][w]
    char* const LB = smem + H * SM_LINE;                   // [w][c][h]
    char* const LP = smem + (H + W) * SM_LINE;             // DW: [pixel][32 channels] raw, then 12 x 32 floats: 9 taps, bias, scale, shift
    float* const LW = reinterpret_cast<float*>(LP + npx * 64);
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    // zero both copies once: the positions >= W (>= H) of every row are never written again
    for (int i = tid; i < (H + W) * (SM_LINE / 16); i += SM_NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
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
    unsigned nbm[DW ? SM_NL : 1];                            // DW: which of the 9 neighbours of the piece's pixel lie inside the map
#pragma unroll
    for (int i = 0; i < SM_NL; ++i) {
        int idx = tid + i * SM_NT;
        idx = idx < npx * 4 ? idx : npx * 4 - 1;             // (clamped, not branched: a load behind a branch is a serialised load)
        const int px = idx >> 2, o = idx & 3;
        const int hh = px / W, ww_ = px - hh * W;
        if constexpr (DW) {
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = hh + t / 3 - 1, xx = ww_ + t % 3 - 1;
                m |= ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W ? 1u : 0u) << t;
            }
            nbm[i] = m;
        }
        xoff[i] = (unsigned)(px * p.ldx + o * 8) * (unsigned)sizeof(T);
        ooff[i] = (unsigned)(px * p.ldo + o * 8) * (unsigned)sizeof(T);
        la[i] = sm_addr(hh, o * 8, ww_);
        lb[i] = (unsigned)(H * SM_LINE) + sm_addr(ww_, o * 8, hh);
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
        if constexpr (DW) {
            // ---- x' = x + dwconv3x3(BN(x)) + b on the tile
            for (int i = tid; i < 12 * 32; i += SM_NT) {
                const int r = i >> 5, c = i & 31;
                LW[i] = r < 9 ? p.dw_w[(size_t)r * C + c0 + c] : (r == 9 ? p.dw_b : r == 10 ? p.dw_s : p.dw_h)[c0 + c];
            }
#pragma unroll
            for (int i = 0; i < SM_NL; ++i)
                if (tid + i * SM_NT < npx * 4) *reinterpret_cast<u32x4*>(LP + (size_t)(tid + i * SM_NT) * 16) = raw[i];
            __syncthreads();
            const int o = tid & 3;
            float bs_[8], a_[8], h_[8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(LW + 9 * 32 + o * 8), b1 = *reinterpret_cast<const f32x4*>(LW + 9 * 32 + o * 8 + 4);
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(LW + 10 * 32 + o * 8), s1 = *reinterpret_cast<const f32x4*>(LW + 10 * 32 + o * 8 + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(LW + 11 * 32 + o * 8), h1 = *reinterpret_cast<const f32x4*>(LW + 11 * 32 + o * 8 + 4);
                bs_[0] = b0.x; bs_[1] = b0.y; bs_[2] = b0.z; bs_[3] = b0.w; bs_[4] = b1.x; bs_[5] = b1.y; bs_[6] = b1.z; bs_[7] = b1.w;
                a_[0] = s0.x; a_[1] = s0.y; a_[2] = s0.z; a_[3] = s0.w; a_[4] = s1.x; a_[5] = s1.y; a_[6] = s1.z; a_[7] = s1.w;
                h_[0] = h0.x; h_[1] = h0.y; h_[2] = h0.z; h_[3] = h0.w; h_[4] = h1.x; h_[5] = h1.y; h_[6] = h1.z; h_[7] = h1.w;
            }
            char* const rb = reinterpret_cast<char*>(reinterpret_cast<T*>(p.xres) + (size_t)b * npx * p.ldxr + c0);
#pragma unroll
            for (int i = 0; i < SM_NL; ++i) {
                const int idx = tid + i * SM_NT;
                if (idx < npx * 4) {
                    float acc[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = bs_[e];
#pragma unroll 1
                    for (int t = 0; t < 9; ++t) {                    // (rolled: unrolled, the 27 LDS reads of a piece went out at once and spilled)
                        const bool ok = (nbm[i] >> t) & 1u;
                        const int nidx = ok ? idx + ((t / 3 - 1) * W + (t % 3 - 1)) * 4 : idx;
                        const u32x4 nv = *reinterpret_cast<const u32x4*>(LP + (size_t)nidx * 16);
                        T e8[8];
                        __builtin_memcpy(e8, &nv, 16);
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(LW + t * 32 + o * 8), w1 = *reinterpret_cast<const f32x4*>(LW + t * 32 + o * 8 + 4);
                        const float wt[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float tap = ok ? __builtin_fmaf(a_[e], to_f32(e8[e]), h_[e]) : 0.f;
                            acc[e] = __builtin_fmaf(wt[e], tap, acc[e]);
                        }
                    }
                    T c8[8];
                    __builtin_memcpy(c8, &raw[i], 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) c8[e] = from_f32<T>(to_f32(c8[e]) + acc[e]);
                    __builtin_memcpy(&raw[i], c8, 16);
                    *reinterpret_cast<u32x4*>(rb + (size_t)(idx >> 2) * p.ldxr * sizeof(T) + (idx & 3) * 16) = raw[i];
                }
            }
        }
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
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && H >= 1 && W >= 1 && H <= 32 && W <= 32 && C >= 32 && C % 32 == 0 && (H + W) * mlpk::SM_LINE <= 160 * 1024;
}

static int smlp_dw_lds(int H, int W) { return (H + W) * mlpk::SM_LINE + H * W * 64 + 12 * 32 * 4; }
constexpr int SM_NL_SMALL = 2;

// the variant with the block's depthwise 3 x 3 sublayer in front (mlpk_smlp_mix_dw): the raw tile must fit beside the transposed copies,
// two workgroups per CU
extern "C" int mlpk_smlp_mix_dw_supported(int dtype, int H, int W, int C) {
    return mlpk_smlp_mix_supported(dtype, H, W, C) && smlp_dw_lds(H, W) <= 78 * 1024 && H * W * 4 <= SM_NL_SMALL * SM_NT;
}

static int smlp_launch(int dtype, SmlpArgs& a, bool dw, hipStream_t s);

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
    a.xres = nullptr; a.dw_w = a.dw_b = a.dw_s = a.dw_h = nullptr; a.ldxr = 0;
    return smlp_launch(dtype, a, false, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mlpk_smlp_mix_dw(int dtype, const void* x, int ldx, int B, int H, int W, int C, const float* dw_w, const float* dw_bias,
                                const float* dw_scale, const float* dw_shift, void* xres, int ldxr, const float* bn_scale, const float* bn_shift,
                                const void* wh, const float* bh, const void* ww, const float* bw, void* out, int ldo, void* stream) {
    if (!x || !out || !xres || !dw_w || !dw_bias || !dw_scale || !dw_shift || !bn_scale || !bn_shift || !wh || !ww || !bh || !bw) return MLPK_ENULL;
    if (B <= 0 || !mlpk_smlp_mix_dw_supported(dtype, H, W, C)) return MLPK_ESHAPE;
    if (ldx < C || ldx % 8 || ldxr < C || ldxr % 8 || ldo < 3 * C || ldo % 8 || x == xres) return MLPK_ESHAPE;   // (a stencil: not in place)
    if (((uintptr_t)x & 15) || ((uintptr_t)xres & 15) || ((uintptr_t)out & 15) || ((uintptr_t)wh & 15) || ((uintptr_t)ww & 15) || ((uintptr_t)bn_scale & 15) ||
        ((uintptr_t)bn_shift & 15))
        return MLPK_EALIGN;
    SmlpArgs a;
    a.x = x; a.out = out; a.bn_s = bn_scale; a.bn_h = bn_shift; a.wh = wh; a.ww = ww; a.bh = bh; a.bw = bw;
    a.B = B; a.H = H; a.W = W; a.C = C; a.ldx = ldx; a.ldo = ldo;
    a.xres = xres; a.dw_w = dw_w; a.dw_b = dw_bias; a.dw_s = dw_scale; a.dw_h = dw_shift; a.ldxr = ldxr;
    return smlp_launch(dtype, a, true, reinterpret_cast<hipStream_t>(stream));
}

static int smlp_launch(int dtype, SmlpArgs& a, bool dw, hipStream_t s) {
    const int H = a.H, W = a.W, C = a.C, B = a.B;
    const int lds = dw ? smlp_dw_lds(H, W) : (H + W) * mlpk::SM_LINE;
    const int units = B * (C / 32);
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
    const int per_cu = lds <= 78 * 1024 ? 2 : 1;
    const int grid = units < cu * per_cu ? units : cu * per_cu;
    const bool small = H * W * 4 <= SM_NL_SMALL * SM_NT;     // few pieces per thread: the short unrolling (no spills)
#define SM_LAUNCH(TT, DD)                                                                                               \
    if (small) SM_LAUNCH2(TT, DD, SM_NL_SMALL) else SM_LAUNCH2(TT, false, 8)
#define SM_LAUNCH2(TT, DD, NN)                                                                                          \
    {                                                                                                                   \
        auto k = smlp_mix_kernel<TT, DD, NN>;                                                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL(k, dim3(grid), dim3(SM_NT), lds, s, a);                                                      \
    }
    if (dtype == MLPK_BF16) {
        if (dw) SM_LAUNCH(bf16_t, true) else SM_LAUNCH(bf16_t, false)
    } else {
        if (dw) SM_LAUNCH(f16_t, true) else SM_LAUNCH(f16_t, false)
    }
#undef SM_LAUNCH
#undef SM_LAUNCH2
    MLPK_LAUNCH_CHECK();
    return 0;
}
