// Vision Permutator's h / w branch in ONE kernel (vip.py:66-76; round 5): LayerNorm + the einops rearrange + the branch Linear
//
//     z[(b, o, g), n] = sum_{l, j} W[n, l * s + j] * x^[b, pixel(o, l), g * s + j] + bias[n],     x^ = LayerNorm_C(x)
//     h branch (which = 0): o = w, l = h   ('b h w (c s) -> b w c (h s)', vip.py:69);   w branch (which = 1): o = h, l = w   (vip.py:74)
//
// Before: mlpk_norm_apply wrote the rearranged, normalised operand (a full tensor: 201 MB at ViP-S7 / 256 images) and the q4 GEMM read it
// back -- two launches and 402 MB of HBM traffic per branch that the operation does not need (north star: "permute done as LDS-staged
// index remaps instead of einops/rearrange copies").  Here the rearrange never leaves the chip: a persistent workgroup stages SG = 2 slabs
// (a slab = the L pixels of one (image, o) = one 32-row block of the product, G = C / s = 32 groups) in LDS ALREADY IN OPERAND ORDER
// -- row g, column l * s + j, exactly what vip_permute_fast_kernel staged before it stored -- and multiplies them where they lie:
//   * the weight fragments of a wave's output-channel chunks (3 x 24 x 16 bytes per lane for N = K = 384) are loaded ONCE per workgroup
//     and stay in registers (one wave per SIMD, 512 registers): the whole weight matrix is resident in the workgroup;
//   * a slab row is one 32-row operand of v_mfma_f32_32x32x16 (weights as the first operand, so a lane's accumulators are one row g
//     and 16 output channels); rows are padded by 16 bytes: the 16 lanes of a ds_read_b128 phase fall on 16 different bank quads;
//   * k ascending in steps of 16, fp32 accumulators, bias in the epilogue, one rounding: the SAME bits as the two-kernel path
//     (tests/test_gpu_ops.py holds the kernel bit-equal to mlpk_norm_apply + mlpk_gemm_nt);
//   * the by-product sums SplitAttention's linearity trick needs (mlpk.h mlpk_norm_desc.sum_ph / sum_pw) come out of the staged slab.
// HBM traffic per branch: read x once, write z once.
#include "mlpk_common.h"

namespace mlpk {

struct VipBranchArgs {
    const void* x;        // (B*H*W, ldx) channel-last
    const float* mean;    // per pixel
    const float* rstd;
    const float* gamma;   // (C)
    const float* beta;
    const void* w;        // (N, ldw) branch weight, K-contiguous, N = K = L * seg
    const float* bias;    // (N)
    void* out;            // rows (b, o, g), ldz
    float* sums;          // or NULL: sums[(b * G + g) * ld_sum + o * seg + j] = sum over l of the ROUNDED x^
    int B, H, W, C, seg, which, ldx, ldw, ldz, ld_sum, nslabs, SG;
};

template <typename T> struct VbMfma;
template <> struct VbMfma<bf16_t> {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct VbMfma<f16_t> {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int VB_STG_PITCH = 80;                  // per-wave output staging: 32 rows x (32 channels x 2 B + 16)
constexpr int VB_STG_BYTES = 32 * VB_STG_PITCH;
constexpr int VB_G = 32;
constexpr int VB_SGMAX = 2;                      // slabs per tile (two LDS buffers of that many)

// NKS = K / 16 k-steps; NCW = output-channel chunks of 32 per wave (chunk ids wave, wave + 4, wave + 8)
template <typename T, int NKS, int NCW>
__global__ void __launch_bounds__(256, 1) vip_branch_kernel(const VipBranchArgs p) {
    typedef typename VbMfma<T>::f32x16 f32x16;
    static_assert(sizeof(T) == 2, "16-bit storage types");
    constexpr int K = NKS * 16;
    constexpr int NCHUNK = K / 32;                  // N = K
    constexpr int ROWB = K * 2 + 16;                // LDS bytes per operand row
    constexpr int SLAB = VB_G * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int C = p.C, seg = p.seg, SG = p.SG;
    const int L = p.which == 0 ? p.H : p.W;
    const int O = p.which == 0 ? p.W : p.H;
    float* const gb = reinterpret_cast<float*>(smem + (size_t)2 * SG * SLAB);      // (behind the two slab buffers) gamma[C], beta[C], bias[K]
    char* const stg = reinterpret_cast<char*>(gb + 2 * C + K) + wave * VB_STG_BYTES;
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.w);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    for (int i = tid; i < C; i += 256) {
        gb[i] = p.gamma[i];
        gb[C + i] = p.beta[i];
    }
    for (int i = tid; i < K; i += 256) gb[2 * C + i] = p.bias[i];
    // ---- the wave's weight fragments: lane = output channel nc * 32 + l31, k-half hh; resident for the whole launch
    u32x4 wfr[NCW][NKS];
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
        const int nc = wave + 4 * c;
        const T* const wr = wgt + (size_t)((nc < NCHUNK ? nc : 0) * 32 + l31) * p.ldw + hh * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) wfr[c][ks] = *reinterpret_cast<const u32x4*>(wr + ks * 16);
    }
    // ---- staging geometry of this thread, the same for every slab: a slab is L * C / 8 = 4 K pieces of 16 bytes = NPT per thread
    //      (piece = pixel l, channels c .. c + 7; its two 4-channel halves go to rows g0 / g1 of the operand).  No division in the loops.
    constexpr int NPT = K / 64;
    const int cv = C / 8;
    const int pstride = p.which == 0 ? p.W * p.ldx : p.ldx;      // elements between the pixels l, l + 1 of a slab
    const int rstride = p.which == 0 ? p.W : 1;                  // ... between their statistics
    int lc[NPT], d0[NPT], d1[NPT];                               // (pixel l << 16 | channel c), the two LDS offsets
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int rem = tid + 256 * i;
        const int l = rem / cv, c = (rem - l * cv) * 8;
        lc[i] = (l << 16) | c;
        const int g0 = c / seg, j0 = c - g0 * seg, g1 = (c + 4) / seg, j1 = c + 4 - g1 * seg;
        d0[i] = g0 * ROWB + (l * seg + j0) * 2;
        d1[i] = g1 * ROWB + (l * seg + j1) * 2;
    }
    __syncthreads();
    // ---- software pipeline over the workgroup's tiles of SG slabs, two LDS buffers: while tile t is multiplied out of buffer t & 1 the raw
    //      pieces of tile t + 1 are in flight (requested BEFORE the MFMAs: a memory round trip hides behind them); they are normalised and
    //      written into the other buffer after the epilogue, one barrier per tile.  (First version: load -> wait -> multiply per tile, every
    //      slab's loads a memory latency of their own: 282 -> 175 us with the divisions gone, still 2 us of latency per slab.)
    u32x4 raw[VB_SGMAX][NPT];
    int64_t rbase[VB_SGMAX];
    auto request = [&](const int tile) {
        const int sl0 = tile * SG;
#pragma unroll
        for (int s = 0; s < VB_SGMAX; ++s) {
            if (s >= SG) break;
            int sl = sl0 + s;
            sl = sl < p.nslabs ? sl : p.nslabs - 1;                  // (a ragged last tile re-reads a valid slab; it is not multiplied)
            const int img = sl / O, o = sl - img * O;
            const int64_t r0 = p.which == 0 ? (int64_t)img * p.H * p.W + o : ((int64_t)img * p.H + o) * p.W;      // pixel l = 0 of the slab
            const T* const xs = x + r0 * p.ldx;
            rbase[s] = r0;
#pragma unroll
            for (int i = 0; i < NPT; ++i) raw[s][i] = *reinterpret_cast<const u32x4*>(xs + (lc[i] >> 16) * pstride + (lc[i] & 0xFFFF));
        }
    };
    // one piece of the requested tile: normalise, round, write its two halves into the operand rows of `slab`
    auto stage_item = [&](char* const slab, const u32x4 rv, const float m, const float r, const int i) {
        T v8[8], e[8];
        __builtin_memcpy(v8, &rv, 16);
        const int c = lc[i] & 0xFFFF;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + c), g1 = *reinterpret_cast<const f32x4*>(gb + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + C + c), b1 = *reinterpret_cast<const f32x4*>(gb + C + c + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float t = (to_f32(v8[q]) - m) * r * gm[q] + bt[q];                              // (the expression of vip_permute_fast_kernel)
            e[q] = from_f32<T>(t);
        }
        u32x2 t0, t1;
        __builtin_memcpy(&t0, e, 8);
        __builtin_memcpy(&t1, e + 4, 8);
        *reinterpret_cast<u32x2*>(slab + d0[i]) = t0;
        *reinterpret_cast<u32x2*>(slab + d1[i]) = t1;
    };
    auto stage = [&](const int buf) {                  // the whole requested tile at once (prologue)
#pragma unroll
        for (int s = 0; s < VB_SGMAX; ++s) {
            if (s >= SG) break;
            char* const slab = smem + (size_t)(buf * SG + s) * SLAB;
            float mu[NPT], rs[NPT];
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                mu[i] = p.mean[rbase[s] + (lc[i] >> 16) * rstride];
                rs[i] = p.rstd[rbase[s] + (lc[i] >> 16) * rstride];
            }
#pragma unroll
            for (int i = 0; i < NPT; ++i) stage_item(slab, raw[s][i], mu[i], rs[i], i);
        }
    };
    int tile = blockIdx.x;
    if (tile * SG < p.nslabs) {
        request(tile);
        stage(0);
    }
    __syncthreads();
    for (int it = 0; tile * SG < p.nslabs; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const int sl0 = tile * SG;
        const int ns = p.nslabs - sl0 < SG ? p.nslabs - sl0 : SG;
        const int next = tile + gridDim.x;
        const bool more = next * SG < p.nslabs;
        if (more) request(next);
        const char* const base = smem + (size_t)buf * SG * SLAB;
        // ---- by-product: sum over the walked axis of the ROUNDED values (what the product reads), per (group, j): a thread = four
        //      consecutive channels of a group (one 8-byte piece per pixel), pixels in ascending order as the kernel this replaces
        if (p.sums) {
            const int qpr = seg / 4;                               // 8-byte pieces per (group, pixel)
            const int nq = VB_G * qpr;                             // ... per slab
            for (int i = tid; i < ns * nq; i += 256) {
                const int s = i / nq, rem = i - s * nq;
                const int g = rem / qpr, j = (rem - g * qpr) * 4;
                const char* row = base + (size_t)s * SLAB + (size_t)g * ROWB + j * 2;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
                for (int l = 0; l < L; ++l) {
                    const u32x2 v = *reinterpret_cast<const u32x2*>(row + (size_t)l * seg * 2);
                    T e4[4];
                    __builtin_memcpy(e4, &v, 8);
                    a0 += to_f32(e4[0]); a1 += to_f32(e4[1]); a2 += to_f32(e4[2]); a3 += to_f32(e4[3]);
                }
                const int sl = sl0 + s;
                const int img = sl / O, o = sl - img * O;
                float* const dst = p.sums + ((int64_t)img * VB_G + g) * p.ld_sum + o * seg + j;
                dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
            }
        }
        // ---- multiply: per slab, every wave its chunks; lane = operand row l31 (group g), k-half hh; the operand fragment of the next
        //      k-step is requested before the MFMAs of this one
#pragma unroll
        for (int s = 0; s < VB_SGMAX; ++s) {
            if (s >= ns) break;
            // the next tile's slab s is normalised and written into the OTHER buffer between the MFMAs of this slab: one 16-byte piece
            // per four k-steps (NPT = NKS / 4 pieces per slab) -- ~11 VALU operations per k-step in the shadow of its three MFMAs
            // instead of a phase of its own (counters of the phase-by-phase version: 756 VALU instructions per slab and wave against
            // 72 MFMAs, SQ_WAIT_ANY 40 %)
            const bool fill = more && s < SG;
            char* const nslab = smem + (size_t)((buf ^ 1) * SG + s) * SLAB;
            float smu[NPT], srs[NPT];
            if (fill) {
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    smu[i] = p.mean[rbase[s] + (lc[i] >> 16) * rstride];
                    srs[i] = p.rstd[rbase[s] + (lc[i] >> 16) * rstride];
                }
            }
            f32x16 acc[NCW];
#pragma unroll
            for (int c = 0; c < NCW; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
            const char* const arow = base + (size_t)s * SLAB + (size_t)l31 * ROWB + hh * 16;
            // operand fragments in pairs, the next pair requested before the MFMAs of this one (NKS is even)
            u32x4 a0 = *reinterpret_cast<const u32x4*>(arow), a1 = *reinterpret_cast<const u32x4*>(arow + 32);
#pragma unroll
            for (int ks = 0; ks < NKS; ks += 2) {
                const u32x4 c0 = a0, c1 = a1;
                if (ks + 2 < NKS) {
                    a0 = *reinterpret_cast<const u32x4*>(arow + (ks + 2) * 32);
                    a1 = *reinterpret_cast<const u32x4*>(arow + (ks + 3) * 32);
                }
#pragma unroll
                for (int c = 0; c < NCW; ++c) acc[c] = VbMfma<T>::run(wfr[c][ks], c0, acc[c]);
#pragma unroll
                for (int c = 0; c < NCW; ++c) acc[c] = VbMfma<T>::run(wfr[c][ks + 1], c1, acc[c]);
                if ((ks & 3) == 2 && fill) stage_item(nslab, raw[s][ks >> 2], smu[ks >> 2], srs[ks >> 2], ks >> 2);
                __builtin_amdgcn_sched_barrier(0);     // (keeps the piece where it is written: between these MFMAs and the next)
            }
            // epilogue: lane = row g = l31, register r = output channel nc * 32 + 8 (r >> 2) + 4 hh + (r & 3)
            T* const orow = out + (size_t)(sl0 + s) * VB_G * p.ldz;
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                const int nc = wave + 4 * c;
                if (nc >= NCHUNK) break;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bz = *reinterpret_cast<const f32x4*>(gb + 2 * C + nc * 32 + 8 * q + 4 * hh);
                    T e4[4] = {from_f32<T>(acc[c][4 * q] + bz.x), from_f32<T>(acc[c][4 * q + 1] + bz.y), from_f32<T>(acc[c][4 * q + 2] + bz.z),
                               from_f32<T>(acc[c][4 * q + 3] + bz.w)};
                    u32x2 pk;
                    __builtin_memcpy(&pk, e4, 8);
                    *reinterpret_cast<u32x2*>(stg + l31 * VB_STG_PITCH + (8 * q + 4 * hh) * 2) = pk;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pr = (lane >> 2) + 16 * k, pc = lane & 3;
                    const u32x4 o4 = *reinterpret_cast<const u32x4*>(stg + pr * VB_STG_PITCH + pc * 16);
                    *reinterpret_cast<u32x4*>(orow + (size_t)pr * p.ldz + nc * 32 + pc * 8) = o4;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
}

static int vb_grid_cap() {
    static int cap = 0;
    if (!cap) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        cap = cu;
    }
    return cap;
}

template <typename T, int NKS>
static int vip_branch_launch(VipBranchArgs a, hipStream_t s) {
    constexpr int K = NKS * 16;
    constexpr int NCW = (K / 32 + 3) / 4;
    constexpr int SLAB = VB_G * (K * 2 + 16);
    const int fixed = (2 * a.C + K) * (int)sizeof(float) + 4 * VB_STG_BYTES;
    int sg = (160 * 1024 - fixed) / (2 * SLAB);
    if (sg < 1) return MLPK_ESHAPE;
    if (sg > VB_SGMAX) sg = VB_SGMAX;
    a.SG = sg;
    const int lds = 2 * sg * SLAB + fixed;
    const int tiles = (a.nslabs + sg - 1) / sg;
    const int grid = tiles < vb_grid_cap() ? tiles : vb_grid_cap();
    auto k = vip_branch_kernel<T, NKS, NCW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, a);
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk

using namespace mlpk;

// which = 0: the h branch (L = H), 1: the w branch (L = W).  K = L * seg must be one of the built sizes.
extern "C" int mlpk_vip_branch_supported(int dtype, int H, int W, int C, int seg, int which) {
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return 0;
    if (which != 0 && which != 1) return 0;
    if (H <= 0 || W <= 0 || C <= 0 || seg <= 0 || C % seg || C / seg != VB_G || seg % 4 || C % 8) return 0;
    const int K = (which == 0 ? H : W) * seg;
    return K == 128 || K == 256 || K == 384;
}

extern "C" int mlpk_vip_branch(int dtype, const void* x, int ldx, int B, int H, int W, int C, int seg, int which, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, const void* w, int ldw, const float* bias, void* out, int ldz, float* sums,
                               int ld_sum, void* stream) {
    if (!x || !mean || !rstd || !gamma || !beta || !w || !bias || !out) return MLPK_ENULL;
    if (B <= 0 || !mlpk_vip_branch_supported(dtype, H, W, C, seg, which)) return MLPK_ESHAPE;
    const int L = which == 0 ? H : W, O = which == 0 ? W : H;
    const int K = L * seg;
    if (ldx < C || ldx % 8 || ldw < K || ldw % 8 || ldz < K || ldz % 8 || (sums && ld_sum < O * seg)) return MLPK_ESHAPE;
    if ((int64_t)B * O > 0x7fffffff / VB_G) return MLPK_ESHAPE;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias) & 15) return MLPK_EALIGN;
    VipBranchArgs a;
    a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.w = w; a.bias = bias; a.out = out; a.sums = sums;
    a.B = B; a.H = H; a.W = W; a.C = C; a.seg = seg; a.which = which; a.ldx = ldx; a.ldw = ldw; a.ldz = ldz; a.ld_sum = ld_sum;
    a.nslabs = B * O; a.SG = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define VB_GO(TT)                                                      \
    switch (K) {                                                       \
        case 128: return vip_branch_launch<TT, 8>(a, s);               \
        case 256: return vip_branch_launch<TT, 16>(a, s);              \
        default: return vip_branch_launch<TT, 24>(a, s);               \
    }
    if (dtype == MLPK_BF16) { VB_GO(bf16_t) }
    VB_GO(f16_t)
#undef VB_GO
}
