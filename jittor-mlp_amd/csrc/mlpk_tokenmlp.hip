// Fused token-mixing MLP (MLP-Mixer, reference mlp_mixer.py:16-27 with dense = Conv1d(k=1), :34,:37):
//
//   x[b,s,c] += sum_t W2[s,t] * gelu( sum_s' W1[t,s'] * xn[b,s',c] + b1[t] ) + b2[s]
//
// computed on the token-transposed LayerNorm output xt[(b,c), s'] so both contractions are K-contiguous NT
// products.  Unfused, the hidden (B*C x 4S, 308 MB at Mixer-B/16 B=256) makes a round trip through HBM and
// the pair is memory/epilogue-bound; here it never leaves the CU.
//
// Workgroup = 128 rows (b,c) x ALL S output tokens, 4 waves (one per SIMD); every wave owns 32 rows end to
// end, so the hidden activations it produces are consumed by itself (no workgroup barrier for them).
//   * the wave's 32 x S_pad slice of xt lives in REGISTERS as MFMA operands for the whole kernel;
//   * the hidden axis is walked in groups of 32 (= one K-slab of the second product); per group g
//       fc1(g) : acc1 = X . W1[g]^T              2 x 2 blocks x (S_pad/32) MFMAs
//       gelu(g): H = gelu(acc1 + b1[g]) -> 16 bit -> wave-private LDS slab in MFMA A-operand order
//       fc2(g) : acc2 += H . W2[:, g]^T           2 x 13 blocks
//     software-pipelined over three iterations (fc1(t), gelu(t-1), fc2(t-2)) so that the VALU work of the
//     GELU and the LDS round trip of H sit beside independent MFMAs instead of between dependent ones;
//   * W1 groups and W2 slabs stream through two 4-stage LDS rings (64-byte rows, XOR-swizzled) filled by
//     global_load_lds three iterations ahead; every wave issues exactly 8 one-KiB pieces per iteration, so a
//     constant `s_waitcnt vmcnt(16)` + one s_barrier per iteration is the whole synchronisation;
//   * epilogue: acc2 + b2 + residual, stored through the per-image transpose (4 consecutive channels/lane).
// LDS: 64 (W1 ring) + 64 (W2 ring) + 16 (H, double-buffered) + 4 (b1) = 148 KiB, one workgroup per CU.
#include "mlpk_common.h"

namespace mlpk {

struct TokenMlpArgs {
    const void* xt;     // (M, ldxt) LayerNorm output, token-transposed, K-padded with zeros
    const void* w1;     // (G*32, 256): hidden rows and K zero-padded
    const void* w2;     // (S, ldw2), ldw2 >= G*32, zero-padded
    const float* b1;    // (G*32) zero-padded
    const float* b2;    // (S)
    void* x;            // (B*S, ldx) residual stream, updated in place
    int M, S, ks1, G;
    int ldxt, ldw2, ldx, t_rows;
    unsigned long long* dbg;   // tuning aid: s_memtime stamps of thread 0 (NULL in normal use)
};
#ifdef MLPK_TM_DEBUG   // stamps add branches that fence the instruction scheduler: tuning builds only
#define TM_STAMP(slot) if (p.dbg && threadIdx.x == 0 && (slot) < 256) p.dbg[(size_t)blockIdx.x * 256 + (slot)] = __builtin_readcyclecounter();
#else
#define TM_STAMP(slot)
#endif

__device__ __forceinline__ void tm_glds(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma2<f16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int TM_BM = 128;          // rows per workgroup
constexpr int TM_KMAX = 7;          // K-slabs of the first product kept in registers (S_pad <= 224)
constexpr int TM_FN2 = 13;          // 208 output tokens max
constexpr int TM_STAGE = 16384;     // W1 group: 8 planes x [32 rows x 64 B];  W2 slab: [256 rows x 64 B]
constexpr int TM_NST = 4;           // stages per ring
constexpr int TM_R1 = 0;
constexpr int TM_R2 = TM_NST * TM_STAGE;
constexpr int TM_HS = 2 * TM_NST * TM_STAGE;           // 4 waves x 2 buffers x [32 rows x 64 B]
constexpr int TM_B1 = TM_HS + 4 * 2 * 2048;
constexpr int TM_B1_FLOATS = 1024;                     // hidden (padded) <= 1024
constexpr int TM_LDS = TM_B1 + TM_B1_FLOATS * 4;

// ABL: tuning ablations, only 0 is instantiated (1 = identity instead of GELU, 2/3 = skip fc2/fc1 MFMAs, 4 = no LDS-DMA,
// 5 = no LDS operand reads, 6 = 5 + no barrier).  Measured on MI355X (profiles/r01_token_mlp_ablation.txt): each removes
// only 4-13 % -- no single phase dominates the 0.29 ms.
template <typename T, int ABL>
__global__ void __launch_bounds__(256, 1) token_mlp_kernel(const TokenMlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.x * TM_BM;
    const T* __restrict__ xt = reinterpret_cast<const T*>(p.xt);
    const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
    const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
    const int G = p.G;
    const int ks1 = p.ks1;

    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    char* const hs = smem + TM_HS + wave * 4096;          // two 2-KiB buffers
    float* const b1s = reinterpret_cast<float*>(smem + TM_B1);

    // ---- piece geometry: 16 rows x 64 B, lane -> (row lrow, physical chunk lane & 3), source-side swizzle ----
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((lrow & 8) >> 2);
    // W1 group = 8 planes (K-slabs) of [32 rows x 64 B]; piece pc = plane*2 + half; this wave issues pc = 4*wave + pi
    // W2 slab  = [256 rows x 64 B]; piece pc = rows pc*16..+16 (clamped to S-1); this wave issues pc = 4*wave + pi
    const T* w1src[4];
    const T* w2src[4];
    unsigned dst1[4], dst2[4];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
        const int pc = wave * 4 + pi;
        w1src[pi] = w1 + (size_t)((pc & 1) * 16 + lrow) * 256 + (pc >> 1) * 32 + lchunk * 8;
        int r2 = pc * 16 + lrow;
        r2 = r2 < p.S ? r2 : p.S - 1;
        w2src[pi] = w2 + (size_t)r2 * p.ldw2 + lchunk * 8;
        dst1[pi] = __builtin_amdgcn_readfirstlane(lds_base + TM_R1 + pc * 1024);
        dst2[pi] = __builtin_amdgcn_readfirstlane(lds_base + TM_R2 + pc * 1024);
    }
    // all 8 pieces of one iteration: W1 group g1 -> ring1 stage g1 % 4, W2 slab g2 -> ring2 stage g2 % 4
    // (group indices past the end are clamped: same count of pieces every iteration, harmless duplicates)
#define TM_ISSUE_W1(g1, pi)                                                                        \
    {                                                                                              \
        const int gg__ = (g1) < 0 ? 0 : ((g1) < G ? (g1) : G - 1);                                 \
        tm_glds(w1src[pi] + (size_t)gg__ * (32 * 256), dst1[pi] + ((g1) & 3) * TM_STAGE);          \
    }
#define TM_ISSUE_W2(g2, pi)                                                                        \
    {                                                                                              \
        const int gg__ = (g2) < 0 ? 0 : ((g2) < G ? (g2) : G - 1);                                 \
        tm_glds(w2src[pi] + gg__ * 32, dst2[pi] + ((g2) & 3) * TM_STAGE);                           \
    }
    // prologue = pseudo-iterations -3, -2, -1, each issuing what iteration t issues: W1(t+3) and W2(t+1).
    // The counted wait of iteration t ("everything older than the last 16 pieces has landed") then holds from
    // t = 0 on; the W2 slabs of negative index are clamped duplicates that the real ones overwrite in order.
#pragma unroll
    for (int it = -3; it < 0; ++it) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) { TM_ISSUE_W1(it + 3, pi); TM_ISSUE_W2(it + 1, pi); }
    }

    // ---- this wave's X operands straight into registers; b1 into LDS ----
    const int frow = lane & 15;
    const int fg = lane >> 4;
    u32x4 xa[2][TM_KMAX];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int gm = m0 + wave * 32 + i * 16 + frow;
        gm = gm < p.M ? gm : p.M - 1;
#pragma unroll
        for (int kk = 0; kk < TM_KMAX; ++kk)
            xa[i][kk] = kk < ks1 ? *reinterpret_cast<const u32x4*>(xt + (size_t)gm * p.ldxt + kk * 32 + fg * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < G * 32; i += 256) b1s[i] = p.b1[i];

    const int co = (fg ^ ((frow & 8) >> 2)) << 4;     // fragment chunk offset inside a 64-byte row
    const int f_rd = frow * 64 + co;                  // + block*1024 (+ plane*2048 in a W1 group)

    f32x4 acc2[2][TM_FN2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM_FN2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ar[2][2];                                       // fc1 result of the previous iteration (input of gelu)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ar[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                     // b1s visible
    // One pipeline iteration t:  fc1(t) -> aw,   gelu(t-1): ar -> H[(t-1)&1],   fc2(t-2): H[t&1] -> acc2.
    // (a single loop body: acc2 keeps its registers; only the 16 fc1 accumulators are handed over by copy)
    for (int t = 0; t < G + 2; ++t) {
        char* const hw = hs + ((t - 1) & 1) * 2048;
        const char* const hr = hs + (t & 1) * 2048;
        f32x4 aw[2][2];
        // pieces of the two previous iterations may still be in flight; older ones (W1(t), W2(t-2)) have landed
        TM_STAMP(5 * t);
        if constexpr (ABL == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        TM_STAMP(5 * t + 1);
        if constexpr (ABL != 6) __builtin_amdgcn_s_barrier();
        TM_STAMP(5 * t + 2);
        const char* r1 = smem + TM_R1 + (t & 3) * TM_STAGE;
        const char* r2 = smem + TM_R2 + ((t - 2) & 3) * TM_STAGE;
        // ---- (A) every LDS operand of this iteration is requested up front (branch-free: the W1 group always
        //      has 8 K-planes in LDS, planes >= ks1 hold the zero K-padding and meet zero X operands) ----
        u32x4 bw[TM_KMAX][2], af[2], bf2[TM_FN2];
#pragma unroll
        for (int kk = 0; kk < TM_KMAX; ++kk) {
            if constexpr (ABL == 5 || ABL == 6) { bw[kk][0] = xa[0][kk]; bw[kk][1] = xa[1][kk]; }
            else {
                bw[kk][0] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd);
                bw[kk][1] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd + 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const u32x4*>(hr + i * 1024 + f_rd);
#pragma unroll
        for (int j = 0; j < TM_FN2; ++j) {
            if constexpr (ABL == 5 || ABL == 6) bf2[j] = xa[j & 1][j % TM_KMAX];
            else bf2[j] = *reinterpret_cast<const u32x4*>(r2 + j * 1024 + f_rd);
        }
        if (t < 2) { af[0] = u32x4{0u, 0u, 0u, 0u}; af[1] = u32x4{0u, 0u, 0u, 0u}; }   // fc2(t-2) does not exist yet
        // bias of the group whose GELU runs now (t-1, clamped: the first / last iterations produce unused H)
        const int gb = t - 1 < 0 ? 0 : (t - 1 < G ? t - 1 : G - 1);
        f32x4 bb[2];
        bb[0] = *reinterpret_cast<const f32x4*>(b1s + gb * 32 + 4 * fg);
        bb[1] = *reinterpret_cast<const f32x4*>(b1s + gb * 32 + 16 + 4 * fg);
        TM_STAMP(5 * t + 3);
        // ---- (B) fc1(t) MFMAs (swapped operands: lane = row frow, 4 consecutive hidden columns 4*fg + r) with the
        //      GELU of group t-1 (pure VALU on last iteration's accumulators) and the W1 pieces slotted between ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) aw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < TM_KMAX; ++kk) {
            // (no LDS-DMA issue in this region: the asm statements would fence the instruction scheduler and
            //  expose every GELU dependency chain; all 8 pieces go out between the fc2 MFMAs below)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (ABL == 3) {
                    if (kk == 0) { aw[i][0] = Mma2<T>::run(bw[kk][0], xa[i][kk], aw[i][0]); aw[i][1] = Mma2<T>::run(bw[kk][1], xa[i][kk], aw[i][1]); }
                } else {
                    aw[i][0] = Mma2<T>::run(bw[kk][0], xa[i][kk], aw[i][0]);
                    aw[i][1] = Mma2<T>::run(bw[kk][1], xa[i][kk], aw[i][1]);
                }
            }
            if (kk < 4) {
                // block (i, j) = (kk >> 1, kk & 1) of gelu(t-1): bias, exact-erf GELU, round, store in A-operand order
                const int i = kk >> 1, j = kk & 1;
                const int row = i * 16 + frow;
                const int lc = j * 2 + (fg >> 1);
                T e[4];
                if constexpr (ABL == 1) {
                    e[0] = from_f32<T>(ar[i][j].x + bb[j].x); e[1] = from_f32<T>(ar[i][j].y + bb[j].y);
                    e[2] = from_f32<T>(ar[i][j].z + bb[j].z); e[3] = from_f32<T>(ar[i][j].w + bb[j].w);
                } else {
                    e[0] = from_f32<T>(gelu_f(ar[i][j].x + bb[j].x));
                    e[1] = from_f32<T>(gelu_f(ar[i][j].y + bb[j].y));
                    e[2] = from_f32<T>(gelu_f(ar[i][j].z + bb[j].z));
                    e[3] = from_f32<T>(gelu_f(ar[i][j].w + bb[j].w));
                }
                u32x2 pk;
                __builtin_memcpy(&pk, e, 8);
                *reinterpret_cast<u32x2*>(hw + row * 64 + ((lc ^ ((row & 8) >> 2)) << 4) + ((fg & 1) << 3)) = pk;
            }
        }
        TM_STAMP(5 * t + 4);
        // ---- (C) fc2(t-2): natural operands -> lane = token frow of block j, 4 consecutive rows 4*fg + r ----
#pragma unroll
        for (int j = 0; j < TM_FN2; ++j) {
            if constexpr (ABL != 4) {
                if (j < 4) { TM_ISSUE_W1(t + 3, j); }
                else if (j < 8) { TM_ISSUE_W2(t + 1, j - 4); }
            }
            if constexpr (ABL == 2) {
                if (j == 0) acc2[0][j] = Mma2<T>::run(af[0], bf2[j], acc2[0][j]);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc2[i][j] = Mma2<T>::run(af[i], bf2[j], acc2[i][j]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) ar[i][j] = aw[i][j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
#undef TM_ISSUE_W1
#undef TM_ISSUE_W2

    // ------------------------------ epilogue: + b2 + residual through the per-image transpose ------------------------------
    T* __restrict__ x = reinterpret_cast<T*>(p.x);
#pragma unroll
    for (int j = 0; j < TM_FN2; ++j) {
        const int n = j * 16 + frow;
        if (n >= p.S) continue;
        const float bn = p.b2[n];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wave * 32 + i * 16 + 4 * fg;
            if (mb >= p.M) continue;                       // M % 4 == 0
            const int img = mb / p.t_rows;
            const int c = mb - img * p.t_rows;
            T* px = x + ((size_t)img * p.S + n) * p.ldx + c;
            u32x2 rv = *reinterpret_cast<const u32x2*>(px);
            T e[4];
            __builtin_memcpy(e, &rv, 8);
            e[0] = from_f32<T>(acc2[i][j].x + bn + to_f32(e[0]));
            e[1] = from_f32<T>(acc2[i][j].y + bn + to_f32(e[1]));
            e[2] = from_f32<T>(acc2[i][j].z + bn + to_f32(e[2]));
            e[3] = from_f32<T>(acc2[i][j].w + bn + to_f32(e[3]));
            __builtin_memcpy(&rv, e, 8);
            *reinterpret_cast<u32x2*>(px) = rv;
        }
    }
}

static unsigned long long* g_tm_dbg = nullptr;

}  // namespace mlpk

using namespace mlpk;

extern "C" void mlpk_token_mlp_debug(void* buf) { g_tm_dbg = reinterpret_cast<unsigned long long*>(buf); }

extern "C" int mlpk_token_mlp_chunk(void) { return 32; }

extern "C" int mlpk_token_mlp(int dtype, const void* xt, int ldxt, int M, int S, const void* w1, int ldw1, const float* b1,
                              const void* w2, int ldw2, const float* b2, int nchunks, void* x, int ldx, int t_rows,
                              void* stream) {
    if (!xt || !w1 || !w2 || !b1 || !b2 || !x) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;   // 16-bit storage only (fp32 uses the two-GEMM path)
    if (M <= 0 || S <= 0 || nchunks <= 0 || t_rows <= 0) return MLPK_ESHAPE;
    if (S > 16 * TM_FN2 || nchunks * 32 > TM_B1_FLOATS) return MLPK_ESHAPE;   // up to 208 tokens, 1024 hidden
    if (ldxt % 32 || ldxt > 32 * TM_KMAX || ldxt < S) return MLPK_ESHAPE;       // K of fc1 = ldxt: whole 64-byte slabs, <= 7
    if (ldw1 != 256 || ldw2 < nchunks * 32 || ldw2 % 8) return MLPK_ESHAPE;
    if (M % 4 || t_rows % 4 || M % t_rows || ldx % 4 || ldx < t_rows) return MLPK_ESHAPE;
    if (((uintptr_t)xt & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)x & 7)) return MLPK_EALIGN;
    TokenMlpArgs a;
    a.xt = xt; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2; a.x = x;
    a.M = M; a.S = S; a.ks1 = ldxt / 32; a.G = nchunks;
    a.ldxt = ldxt; a.ldw2 = ldw2; a.ldx = ldx; a.t_rows = t_rows;
    a.dbg = g_tm_dbg;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((M + TM_BM - 1) / TM_BM);
    hipError_t e;
    if (dtype == MLPK_BF16) {
        auto k = token_mlp_kernel<bf16_t, 0>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), TM_LDS, s, a);
    } else {
        auto k = token_mlp_kernel<f16_t, 0>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), TM_LDS, s, a);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}
