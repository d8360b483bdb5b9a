// Fused channel MLP for NARROW channel-last tensors (the first stages of the hierarchical families: AS-MLP as_mlp.py:36-52 with
// C = 96 / 192, Swin-MLP, MS-MLP, Hire-MLP, CycleMLP, Sparse-MLP with C = 64 .. 192):
//
//     out[m, :] = R[m, :] + W2 . gelu( W1 . norm(x[m, :]) + b1 ) + b2          x, out, R: (M, C) rows; hidden = 4 C (<= 1024)
//
// As two GEMMs this is memory-bound on the HIDDEN tensor: at AS-MLP-T's first stage (802816 rows x 96 channels, 256 images) fc1
// writes and fc2 reads 617 MB for a 154 MB activation -- 259 + 219 us per block against 75 us for reading x and writing out once.
// Here the hidden never leaves the registers.  The machinery is the token-mixing kernel's (token_mlp_rr_kernel, mlpk_tokenmlp.hip):
//   * persistent workgroups of 8 waves walking 256-row tiles; a wave owns 32 rows, keeps their C channels in registers as MFMA
//     operands and runs fc1 -> GELU -> fc2 for them itself over the groups of 32 hidden units;
//   * fc1 with swapped operands (hidden x rows) leaves lane (row, fg) holding hidden units {4 fg + r} and {16 + 4 fg + r} of a group:
//     after bias + GELU + rounding those eight values ARE one 16x16x32 operand fragment of fc2, provided W2's columns are stored in
//     that order inside every group of 32 (layout 1 of mlpk_token_mlp_layout): nothing is exchanged between lanes or waves;
//   * fc2 ALSO with swapped operands (output channels x rows): a lane's accumulators are then consecutive output channels of ONE
//     row, and with W2's ROWS stored in the order [32 q + 8 fg + 4 (j & 1) + r  at  32 q + 16 (j & 1) + 4 fg + r] the two blocks
//     j = 2q, 2q + 1 give a lane 8 consecutive channels: the epilogue is one 16-byte store per (row, 8 channels), 64 contiguous
//     bytes per 4 lanes, straight from the accumulators -- no LDS staging, no barrier.  The accumulators START from R + b2;
//   * the normalisation in front (LayerNorm, or GroupNorm(1, C) with one statistic per ln_group rows) is folded as in mlpk_gemm_nt:
//     gamma in W1, beta in b1, v = (acc - mean * csum[h]) * rstd + b1[h] in the GELU stage;
//   * the two waves of a SIMD are skewed by half an iteration (one in its matrix phase while the other runs GELU); W1 groups and
//     W2 slabs stream through (D + 1)- and (D + 2)-stage LDS rings by LDS-DMA D = 2 iterations ahead, PPW one-KiB pieces per wave and iteration,
//     one barrier per iteration.
// LDS (D = 2): 3 x (C / 16) KiB + 4 x (C / 16) KiB + 4 (b1) + 4 (csum) + 1 (b2) KiB = 93 KiB at C = 192.
#include "mlpk_common.h"
#include <cstdlib>

namespace mlpk {

struct ChanMlpArgs {
    const void* x;          // (M, ldx) operand rows
    const void* w1;         // (G*32, 256): hidden rows, K zero-padded to 256 (pack_token_mlp's W1)
    const void* w2;         // (C, ldw2 >= G*32): rows and columns in the orders described above
    const float* b1;        // (G*32)
    const float* csum;      // (G*32) row sums of the folded W1, or NULL (no normalisation)
    const float* b2;        // (C) in W2's row order
    const float* ln_mean;   // statistic s = m / ln_group
    const float* ln_rstd;
    const void* R;          // (M, ldr) or NULL
    void* out;              // (M, ldo)
    float* row_part;        // optional by-product: (sum, sum of squares) of the C values written to row m at [2 m], [2 m + 1]
                            // (first product alone: planes of 32 columns, pair of row m in plane g at [2 (g M + m)])
    int M, G, ldx, ldw1, ldw2, ldr, ldo, ln_group;
};

static __device__ __forceinline__ void cm_glds(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <typename T> struct CmMma;
template <> struct CmMma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct CmMma<f16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <bool B> struct CmBool { static constexpr bool value = B; };

// A tile's rows are loaded by asm the compiler does not track: requested before the previous tile's epilogue, they are waited for by
// COUNT (everything but the stores behind them); a load the compiler tracks gets its own wait in front of the first use -- and with
// stores in conditional blocks behind it that wait is vmcnt(0): the epilogue's stores and the weight pieces just requested.
static __device__ __forceinline__ u32x4 cm_load16(const void* ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// v + (the same lane of the neighbouring lane row, lane ^ 16), then + (the other half of the wave, lane ^ 32): the (fg 0 + fg 1) +
// (fg 2 + fg 3) of a row's four lanes by the gfx950 lane-row swaps -- __shfl_xor needs the lane id, a register the kernel kept for the
// whole launch (spilled under the 128-register cap, and every reload of it waited for all memory operations in flight)
static __device__ __forceinline__ float cm_rows4_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float h = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned uh = __builtin_bit_cast(unsigned, h);
    const auto q = __builtin_amdgcn_permlane32_swap(uh, uh, false, false);
    return __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
}
static __device__ __forceinline__ float cm_load4(const void* ptr) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

constexpr int CM_BM = 256;
constexpr int CM_HID_MAX = 1024;                          // both products
constexpr int CM_N_MAX = 4096;                             // the first product alone

// D = how many iterations ahead the weight pieces are requested (an iteration here is 3-6x shorter than the token kernel's: 24-48
// MFMAs per wave; deeper rings were tried and change nothing, see cm_launch)
template <int KS1, int NB, int D, bool FC2 = true> struct CmGeo {
    static constexpr int HMAX = FC2 ? CM_HID_MAX : CM_N_MAX;
    static constexpr int N1 = 2 * KS1;                      // W1 pieces per group: plane kk, halves of 16 rows
    static constexpr int P = N1 + (FC2 ? NB : 0);           // + W2 pieces: 16 output channels each
    static constexpr int PPW = (P + 7) / 8;                 // pieces per wave and iteration (the last ones issued twice)
    static constexpr int ST1 = N1 * 1024, ST2 = FC2 ? NB * 1024 : 0;
    static constexpr int R1 = 0, R2 = (D + 1) * ST1;        // W1 ring: D + 1 stages; W2 ring: D + 2 (the late half reads slab t - 1)
    static constexpr int B1 = R2 + (D + 2) * ST2;
    static constexpr int CS = B1 + HMAX * 4;
    static constexpr int B2 = CS + HMAX * 4;
    static constexpr int LDS = B2 + 16 * NB * 4;
};

// FC2 = false: the FIRST product alone -- out[m, n] = gelu(norm-fold(x[m, :] . W1[n, :]) + b1[n]) for K = 32 KS1 <= 512 and any N = 32 G: the
// short-K GELU GEMM with its rows resident (gMLP's channel_proj1, the fc1 of the K = 384 channel MLPs).  The one-wave-per-SIMD q4 tile
// cannot hide a GELU epilogue behind 4-6 k-steps of MFMAs (gMLP proj1: 128 MFMAs against 2300 epilogue instructions per tile); here two
// waves per SIMD alternate between MFMAs and GELU, and a lane's 8 rounded values -- with W1's ROWS stored as [hidden 8 f + 4 j + r at row
// 16 j + 4 f + r] of every 32 -- are 8 consecutive output columns: one 16-byte store per (row, group), no epilogue at all.  STATS: the
// by-product planes of 32 columns in the canonical order (a lane's chunk by chunk_sums, then (c0 + c1) + (c2 + c3) across the 4 lanes).
// Round 6 (profiles/r06_chanmlp_variants.txt; -DCM_WGS2=0 / -DCM_PREFETCH=0 rebuild the round-5 kernel for A/B):
//   * the residual rows ARE the operand rows in every pre-norm residual block (R == x): read once (template SAME);
//   * the next tile's rows are requested straight after the tile's last fc1 -- their registers are free from there on -- and travel
//     under the last GELU, fc2 and the epilogue instead of after them;
//   * two workgroups per CU where their rings and registers fit twice (C <= 96): one's tile hand-over and barrier waits are filled
//     by the other's iterations.  802816 x 96: 253 -> 224 us, x 64: 151 -> 127, 200704 x 128: 114 -> 101, x 192: 197 -> 191.
#ifndef CM_WGS2
#define CM_WGS2 1
#endif
#ifndef CM_PREFETCH
#define CM_PREFETCH 1
#endif
// (f16 at C = 96: its longer GELU polynomial does not fit 128 registers without spilling -- one workgroup per CU there)
template <typename T, int KS1, int D, bool FC2, bool SAME> struct CmWgs {
    static constexpr int value = (CM_WGS2 && FC2 && SAME && D == 2 && (KS1 <= 2 || (KS1 == 3 && dtype_of<T>::value == MLPK_BF16))) ? 2 : 1;
};

// SAME: the residual rows ARE the operand rows (R == x, the pre-norm residual block of every family): read once
template <typename T, int KS1, int NB, int D, bool FC2 = true, bool STATS = false, bool SAME = false>
__global__ void __launch_bounds__(512, (2 * CmWgs<T, KS1, D, FC2, SAME>::value)) chan_mlp_kernel(const ChanMlpArgs p) {
    using Geo = CmGeo<KS1, NB, D, FC2>;
    constexpr int PPW = Geo::PPW, N1 = Geo::N1, P = Geo::P;
    constexpr int C = 16 * NB;
    constexpr int NSTORE = FC2 ? 0 : (STATS ? 4 : 2);       // vector-memory stores a wave issues per iteration (first product alone)
    // the next tile's rows requested straight after the tile's last fc1 (a residual that is not the operand doubles the registers
    // requested ahead: the wide widths would spill)
    constexpr bool PRE = CM_PREFETCH && FC2 && (SAME || KS1 <= 4);
    static_assert(!FC2 || NB == 2 * KS1, "C = 32 KS1 = 16 NB");
    static_assert(FC2 || D == 2, "the store-counting waits below are written for two iterations of look-ahead");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool lag = wave >= 4;                            // the half that runs gelu / fc2 one iteration late
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
    const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
    const T* __restrict__ R = reinterpret_cast<const T*>(p.R);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int G = p.G;
    const int ntiles = (p.M + CM_BM - 1) / CM_BM;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    float* const b1s = reinterpret_cast<float*>(smem + Geo::B1);
    float* const css = reinterpret_cast<float*>(smem + Geo::CS);
    float* const b2s = reinterpret_cast<float*>(smem + Geo::B2);
    const bool fold = p.ln_mean != nullptr;

    // (every per-lane quantity the iterations use is re-derived from an opaque copy of the lane id: see token_mlp_rr_kernel)
    auto lane_now = [&]() {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (int)l;
    };
    // ---- LDS-DMA pieces: wave w issues q = PPW w + pi (clamped to P - 1); q < N1: W1 piece (plane q >> 1, rows 16 (q & 1) ..),
    // else W2 piece q - N1 (output channels 16 (q - N1) ..) ----
    unsigned pdst[PPW];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi) {
        int q = wave * PPW + pi;
        q = q < P ? q : P - 1;
        pdst[pi] = __builtin_amdgcn_readfirstlane(q < N1 ? lds_base + Geo::R1 + q * 1024 : lds_base + Geo::R2 + (q - N1) * 1024);
    }
    auto piece_off = [&](const int pi, const int ln) {
        int q = wave * PPW + pi;
        q = q < P ? q : P - 1;
        const int lrow = ln >> 2;
        const int lchunk = (ln & 3) ^ ((lrow & 8) >> 2);
        unsigned o;
        if (q < N1) o = (unsigned)(((q & 1) * 16 + lrow) * p.ldw1 + (q >> 1) * 32 + lchunk * 8) * (unsigned)sizeof(T);
        else o = (unsigned)(((q - N1) * 16 + lrow) * p.ldw2 + lchunk * 8) * (unsigned)sizeof(T);
        return o;
    };
    const T* pb1 = w1;
    const T* pb2 = w2;
    unsigned so1 = 0, so2 = 0;
    auto piece_bases = [&](const int g) {
        pb1 = w1 + (size_t)g * 32 * p.ldw1;
        pb2 = w2 + g * 32;
    };
    auto issue = [&](const int pi, const int ln) {
        int q = wave * PPW + pi;
        q = q < P ? q : P - 1;
        const bool is1 = q < N1;
        cm_glds(piece_off(pi, ln), is1 ? pb1 : pb2, pdst[pi] + __builtin_amdgcn_readfirstlane(is1 ? so1 : so2));
    };
    unsigned s3 = 0, s4 = 0;                               // iteration counter modulo the ring sizes (W1: D + 1 stages, W2: D + 2)
#pragma unroll
    for (int it = 0; it < D; ++it) {
        int g0 = it;
        g0 = g0 < G ? g0 : g0 - G;
        g0 = g0 < G ? g0 : g0 - G;
        g0 = g0 < G ? g0 : g0 - G;                          // (G >= 2, D <= 6)
        piece_bases(g0);
        so1 = (unsigned)it * Geo::ST1;
        so2 = (unsigned)it * Geo::ST2;
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi) issue(pi, lane);
    }
    if (lag) __builtin_amdgcn_s_setprio(2);

    for (int i = tid; i < G * 32; i += 512) {
        b1s[i] = p.b1[i];
        css[i] = fold ? p.csum[i] : 0.f;
    }
    if (FC2 && tid < C) b2s[tid] = p.b2[tid];
    __syncthreads();

    u32x4 xa[2][KS1], rr[2][FC2 && !SAME ? KS1 : 1];
    float lmu[2], lrs[2], nmu[2], nrs[2];
    // RULE for the asm loads: the compiler believes their results are there when they are issued.  So no value they produce may meet
    // another definition of the same variable (a branch merge, a loop back edge: the register copies of such a merge would read the
    // registers early -- and free them for reuse while the data is still on its way) before `settle` has waited for them: the addresses are
    // selected, the loads are unconditional, and every path from a load_tile to the loop's back edge runs through settle.
    auto load_tile = [&](const int tile, const int ln) {     // operand rows, residual rows, row statistics of a tile
        const int frow = ln & 15, fg = ln >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gm = tile * CM_BM + wave * 32 + i * 16 + frow;
            gm = gm < p.M ? gm : p.M - 1;
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) xa[i][kk] = cm_load16(x + (size_t)gm * p.ldx + kk * 32 + fg * 8);
            if constexpr (FC2 && !SAME) {
                // (no residual: the operand rows once more, zeroed in settle)
                const T* rsrc = R ? R + (size_t)gm * p.ldr : x + (size_t)gm * p.ldx;
#pragma unroll
                for (int kk = 0; kk < KS1; ++kk) rr[i][kk] = cm_load16(rsrc + kk * 32 + fg * 8);
            }
            // the statistic of a row: m / ln_group.  LayerNorm: the row itself; a group that is a multiple of 16 rows (GroupNorm(1, C) over
            // a 56 x 56 or 28 x 28 map): the 16 rows of this half share it -- a scalar division, where the per-lane division kept its
            // reciprocal in a vector register across the whole kernel (spilled, and every reload of it waited for all memory operations
            // in flight); anything else: per lane, the divisor hidden from loop-invariant hoisting for the same reason.  No
            // normalisation: any readable address (the values are replaced in settle).
            int si = 0;
            if (fold) {
                if (p.ln_group == 1) {
                    si = gm;
                } else if ((p.ln_group & 15) == 0) {
                    int base = __builtin_amdgcn_readfirstlane(tile * CM_BM + wave * 32 + i * 16);
                    base = base < p.M ? base : p.M - 1;
                    si = base / p.ln_group;
                } else {
                    int g = p.ln_group;
                    asm volatile("" : "+s"(g));
                    si = gm / g;
                }
            }
            const float* const pm = fold ? p.ln_mean + si : p.b1;
            const float* const pr = fold ? p.ln_rstd + si : p.b1;
            nmu[i] = cm_load4(pm);
            nrs[i] = cm_load4(pr);
        }
    };
    // wait for a load_tile: `behind` = the loads were requested BEFORE an epilogue's stores (the 2 KS1 row stores of a wave; with the
    // by-product statistics two more, of which the first two stores are then waited for as well; a tile with fewer stores -- rows past M
    // -- is a workgroup's last): everything but those stores.  Then the loaded registers are handed to the compiler.
    auto settle = [&](const bool behind) {
        constexpr int NS = FC2 ? 2 * KS1 : 0;
        static_assert(NS <= 15, "vmcnt immediate");
        if (!behind) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk) asm volatile("" : "+v"(xa[i][kk]));
            if constexpr (FC2 && !SAME) {
#pragma unroll
                for (int kk = 0; kk < KS1; ++kk) {
                    asm volatile("" : "+v"(rr[i][kk]));
                    if (!R) rr[i][kk] = u32x4{0u, 0u, 0u, 0u};
                }
            }
            asm volatile("" : "+v"(nmu[i]), "+v"(nrs[i]));
            if (!fold) { nmu[i] = 0.f; nrs[i] = 1.f; }
        }
    };

    f32x4 acc2[2][FC2 ? NB : 1];
    f32x4 a1[2][2];
    auto frag_off = [&](const int ln) {
        const int fr = ln & 15;
        return fr * 64 + (((ln >> 4) ^ ((fr & 8) >> 2)) << 4);
    };
    auto fc1 = [&](const unsigned st3, const int ln) {
        const int f_rd = frag_off(ln);
        const char* r1 = smem + Geo::R1 + st3 * Geo::ST1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) a1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragment pairs read ahead of their MFMAs, pinned there (two workgroups per CU: 128 registers per wave, and twice the waves
        // to hide a fragment read behind)
        constexpr int BWD = CmWgs<T, KS1, D, FC2, SAME>::value > 1 ? 2 : (KS1 < 3 ? KS1 : 3);
        u32x4 bw[BWD + 1][2];
#pragma unroll
        for (int kk = 0; kk < BWD; ++kk) {
            bw[kk][0] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd);
            bw[kk][1] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd + 1024);
        }
#pragma unroll
        for (int kk = 0; kk < KS1; ++kk) {
            if (kk + BWD < KS1) {
                bw[(kk + BWD) % (BWD + 1)][0] = *reinterpret_cast<const u32x4*>(r1 + (kk + BWD) * 2048 + f_rd);
                bw[(kk + BWD) % (BWD + 1)][1] = *reinterpret_cast<const u32x4*>(r1 + (kk + BWD) * 2048 + f_rd + 1024);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a1[i][0] = CmMma<T>::run(bw[kk % (BWD + 1)][0], xa[i][kk], a1[i][0]);
                a1[i][1] = CmMma<T>::run(bw[kk % (BWD + 1)][1], xa[i][kk], a1[i][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    u32x4 hf[2];
    auto gelu = [&](const int g, const int ln) {           // (folded norm,) bias, GELU, rounding: acc1 -> the two operand fragments of fc2
        const int fg = ln >> 4;
        const f32x4 bb0 = *reinterpret_cast<const f32x4*>(b1s + g * 32 + 4 * fg);
        const f32x4 bb1 = *reinterpret_cast<const f32x4*>(b1s + g * 32 + 16 + 4 * fg);
        const f32x4 cs0 = *reinterpret_cast<const f32x4*>(css + g * 32 + 4 * fg);
        const f32x4 cs1 = *reinterpret_cast<const f32x4*>(css + g * 32 + 16 + 4 * fg);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float mu = lmu[i], rs = lrs[i];
            f32x2 v[4] = {f32x2{(a1[i][0].x - mu * cs0.x) * rs + bb0.x, (a1[i][0].y - mu * cs0.y) * rs + bb0.y},
                          f32x2{(a1[i][0].z - mu * cs0.z) * rs + bb0.z, (a1[i][0].w - mu * cs0.w) * rs + bb0.w},
                          f32x2{(a1[i][1].x - mu * cs1.x) * rs + bb1.x, (a1[i][1].y - mu * cs1.y) * rs + bb1.y},
                          f32x2{(a1[i][1].z - mu * cs1.z) * rs + bb1.z, (a1[i][1].w - mu * cs1.w) * rs + bb1.w}};
            gelu_pk_n<T, 4>(v);
            T e[8] = {from_f32<T>(v[0].x), from_f32<T>(v[0].y), from_f32<T>(v[1].x), from_f32<T>(v[1].y),
                      from_f32<T>(v[2].x), from_f32<T>(v[2].y), from_f32<T>(v[3].x), from_f32<T>(v[3].y)};
            __builtin_memcpy(&hf[i], e, 16);
        }
    };
    auto fc2 = [&](const unsigned st4, const int ln) {
      if constexpr (FC2) {
        const int f_rd = frag_off(ln);
        const char* r2 = smem + Geo::R2 + st4 * Geo::ST2;
        constexpr int BFD = CmWgs<T, KS1, D, FC2, SAME>::value > 1 ? 3 : (NB < 4 ? NB : 4);                // W2 fragments read ahead of their MFMAs
        u32x4 bf[BFD + 1];
#pragma unroll
        for (int j = 0; j < BFD; ++j) bf[j] = *reinterpret_cast<const u32x4*>(r2 + j * 1024 + f_rd);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j + BFD < NB) bf[(j + BFD) % (BFD + 1)] = *reinterpret_cast<const u32x4*>(r2 + (j + BFD) * 1024 + f_rd);
            // swapped: output channels x rows -- lane (row, fg) gets channels-in-W2-row-order 16 j + 4 fg + r of its row
            acc2[0][j] = CmMma<T>::run(bf[j % (BFD + 1)], hf[0], acc2[0][j]);
            acc2[1][j] = CmMma<T>::run(bf[j % (BFD + 1)], hf[1], acc2[1][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    // first product alone: the rounded hidden of group g IS the output -- 8 consecutive columns per lane
    auto store_h = [&](const int g, const int tile, const int ln) {
        const int frow = ln & 15, fg = ln >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t gm = (size_t)tile * CM_BM + wave * 32 + i * 16 + frow;
            *reinterpret_cast<u32x4*>(out + gm * p.ldo + g * 32 + fg * 8) = hf[i];
            if constexpr (STATS) {
                float ssum = 0.f, ssq = 0.f;
                chunk_sums<T>(hf[i], ssum, ssq);
                ssum = cm_rows4_sum(ssum);
                ssq = cm_rows4_sum(ssq);
                // (every lane stores: lanes fg != 0 to their own row's pair as well -- the same value four times, one store instruction
                // with a fixed count for the vmcnt arithmetic)
                *reinterpret_cast<f32x2*>(p.row_part + ((size_t)g * p.M + gm) * 2) = f32x2{ssum, ssq};
            }
        }
    };

    auto run = [&](auto lag_c) {
        constexpr bool LAG = decltype(lag_c)::value;
        load_tile(blockIdx.x, lane_now());
        settle(false);
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { lmu[i] = nmu[i]; lrs[i] = nrs[i]; }
            if constexpr (FC2) {
                // accumulators start from R + b2: lane (row, fg) holds channels 32 q + 8 fg + {0..3} in block 2q, + {4..7} in block 2q + 1
                const int fg = lane_now() >> 4;
#pragma unroll
                for (int q = 0; q < KS1; ++q) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(b2s + 32 * q + 8 * fg);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(b2s + 32 * q + 8 * fg + 4);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        T r8[8];
                        __builtin_memcpy(r8, SAME ? &xa[i][q] : &rr[i][q], 16);
                        acc2[i][2 * q] = f32x4{to_f32(r8[0]) + c0.x, to_f32(r8[1]) + c0.y, to_f32(r8[2]) + c0.z, to_f32(r8[3]) + c0.w};
                        acc2[i][2 * q + 1] = f32x4{to_f32(r8[4]) + c1.x, to_f32(r8[5]) + c1.y, to_f32(r8[6]) + c1.z, to_f32(r8[7]) + c1.w};
                    }
                }
            }
            auto iter = [&](auto first_c, auto second_c, const int t, auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
                constexpr bool FIRST = decltype(first_c)::value;
                // what may still be in flight at this point: the pieces requested last iteration -- and, first product alone, the stores
                // of last iteration (none in a tile's iteration 0 for the late half: the sync of iteration 1 does not count them)
                constexpr int INFLIGHT = (D - 1) * PPW + (decltype(second_c)::value ? 0 : NSTORE);
                // (both products: the pieces of a tile's first two groups were requested during the previous tile and are covered by the
                // wait at the top of the tile -- counting here would wait for the previous epilogue's stores, which are younger)
                if constexpr (FC2 && (FIRST || decltype(second_c)::value)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");
                asm volatile("s_barrier" ::: "memory");
                int g2 = t + D;                                // pieces of iteration t + D (the next tile's first groups at the end of this one)
                g2 = g2 < G ? g2 : g2 - G;
                g2 = g2 < G ? g2 : g2 - G;
                g2 = g2 < G ? g2 : g2 - G;
                g2 = g2 < G ? g2 : g2 - G;
                piece_bases(g2);
                so1 = (s3 == 0 ? (unsigned)D : s3 - 1) * Geo::ST1;                 // stage (gi + D) % (D + 1)
                so2 = (s4 >= 2 ? s4 - 2 : s4 + D) * Geo::ST2;                      // stage (gi + D) % (D + 2)
                const int ln = lane_now();
                constexpr int PH = PPW / 2;
#pragma unroll
                for (int pi = 0; pi < PH; ++pi) issue(pi, ln);
                if constexpr (!LAG) {
                    fc1(s3, ln);
#pragma unroll
                    for (int pi = PH; pi < PPW; ++pi) issue(pi, ln);
                    // the tile's last fc1 is done: its operand registers take the next tile's rows, which travel under the GELU, fc2 and
                    // the epilogue instead of after them
                    if constexpr (LAST) load_tile(tile + gridDim.x, ln);
                    gelu(t, ln);
                    if constexpr (FC2) fc2(s4, ln); else store_h(t, tile, ln);
                } else {
                    if constexpr (!FIRST) {
                        gelu(t - 1, ln);
                        if constexpr (FC2) fc2(s4 == 0 ? D + 1 : s4 - 1, ln);    // slab t - 1: stage (gi - 1) % (D + 2)
                        else store_h(t - 1, tile, ln);
                    }
#pragma unroll
                    for (int pi = PH; pi < PPW; ++pi) issue(pi, ln);
                    fc1(s3, ln);
                    if constexpr (LAST) load_tile(tile + gridDim.x, ln);
                }
                s3 = s3 == D ? 0 : s3 + 1;
                s4 = s4 == D + 1 ? 0 : s4 + 1;
            };
            iter(CmBool<true>{}, CmBool<false>{}, 0, CmBool<false>{});
            if (G > 1) iter(CmBool<false>{}, CmBool<true>{}, 1, CmBool<false>{});
            // (PRE: the host guarantees G >= 3, mlpk_channel_mlp_supported -- the tile's last iteration is then always the peeled one)
#pragma unroll 1
            for (int t = 2; t < (PRE ? G - 1 : G); ++t) iter(CmBool<false>{}, CmBool<false>{}, t, CmBool<false>{});
            if constexpr (PRE) iter(CmBool<false>{}, CmBool<false>{}, G - 1, CmBool<true>{});
            if constexpr (LAG) {
                const int ln = lane_now();
                gelu(G - 1, ln);
                if constexpr (FC2) fc2(s4 == 0 ? D + 1 : s4 - 1, ln); else store_h(G - 1, tile, ln);
            }
            // ---- tile epilogue: one rounding, 16-byte stores straight from the accumulators; then the next tile's rows ----
            const int le = lane_now();
            const int frow = le & 15, fg = le >> 4;
#pragma unroll
            for (int i = 0; i < (FC2 ? 2 : 0); ++i) {
                const int gm = tile * CM_BM + wave * 32 + i * 16 + frow;
                float ssum = 0.f, ssq = 0.f;
#pragma unroll
                for (int q = 0; q < KS1; ++q) {
                    const f32x4 v0 = acc2[i][2 * q], v1 = acc2[i][2 * q + 1];
                    T e[8] = {from_f32<T>(v0.x), from_f32<T>(v0.y), from_f32<T>(v0.z), from_f32<T>(v0.w),
                              from_f32<T>(v1.x), from_f32<T>(v1.y), from_f32<T>(v1.z), from_f32<T>(v1.w)};
                    u32x4 o;
                    __builtin_memcpy(&o, e, 16);
                    if (gm < p.M) *reinterpret_cast<u32x4*>(out + (size_t)gm * p.ldo + q * 32 + fg * 8) = o;
                    if (p.row_part) chunk_sums<T>(o, ssum, ssq);
                }
                if (p.row_part) {
                    // a row's C channels sit in the 4 lanes (row, fg = 0..3) of this wave: chunks q ascending inside a lane, then
                    // (fg 0 + fg 1) + (fg 2 + fg 3) -- one fixed order whatever the batch or the grid
                    ssum = cm_rows4_sum(ssum);
                    ssq = cm_rows4_sum(ssq);
                    if (fg == 0 && gm < p.M) *reinterpret_cast<f32x2*>(p.row_part + (size_t)gm * 2) = f32x2{ssum, ssq};
                }
            }
            // the next tile's rows (and every weight piece requested so far) have landed, and the compiler knows it -- settled on either
            // path BEFORE the two meet
            if constexpr (!PRE) {
                load_tile(tile + gridDim.x, le);              // (unconditionally: rows past M clamp to the last row)
                settle(false);
            } else {
                settle(true);
            }
        }
    };
    if (!lag) run(CmBool<false>{}); else run(CmBool<true>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup
}

static int cm_grid_cap() {
    static int cap = 0;
    if (!cap) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        cap = cu;
    }
    return cap;
}

template <typename T, int KS1, int D>
static int cm_launch_d(const ChanMlpArgs& a, hipStream_t s) {
    using Geo = CmGeo<KS1, 2 * KS1, D>;
    const bool same = a.R == a.x && a.ldr == a.ldx;
    auto k = same ? chan_mlp_kernel<T, KS1, 2 * KS1, D, true, false, true> : chan_mlp_kernel<T, KS1, 2 * KS1, D, true, false, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
    if (e != hipSuccess) return MLPK_ESHAPE;
    const int tiles = (a.M + CM_BM - 1) / CM_BM;
    const int cap = cm_grid_cap() * (same ? CmWgs<T, KS1, D, true, true>::value : 1);
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), Geo::LDS, s, a);
    MLPK_LAUNCH_CHECK();
    return 0;
}

template <typename T, int KS1>
static int cm_launch(const ChanMlpArgs& a, hipStream_t s) {
    // MLPK_CM_DEPTH = 2 | 4 | 6: A/B aid (6 only while its rings fit the LDS).  Measured (profiles/r04_chanmlp_depth_ab.txt): no
    // difference -- the kernel is bound by the GELU's VALU instructions (SQ_ACTIVE_INST_VALU 44 % of the time, MFMA busy 20 %), not by
    // the weight stream -- so the shallow rings of the token kernel stay.
    static const int want = getenv("MLPK_CM_DEPTH") ? atoi(getenv("MLPK_CM_DEPTH")) : 2;
    if constexpr (CmGeo<KS1, 2 * KS1, 6>::LDS <= 160 * 1024) {
        if (want >= 6) return cm_launch_d<T, KS1, 6>(a, s);
    }
    if (want >= 4) return cm_launch_d<T, KS1, 4>(a, s);
    return cm_launch_d<T, KS1, 2>(a, s);
}

template <typename T, int KS1>
static int cm_launch_fc1(const ChanMlpArgs& a, hipStream_t s) {
    using Geo = CmGeo<KS1, 2 * KS1, 2, false>;
    hipError_t e;
    const int tiles = a.M / CM_BM;
    const unsigned grid = (unsigned)(tiles < cm_grid_cap() ? tiles : cm_grid_cap());
    if (a.row_part) {
        auto k = chan_mlp_kernel<T, KS1, 2 * KS1, 2, false, true>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), Geo::LDS, s, a);
    } else {
        auto k = chan_mlp_kernel<T, KS1, 2 * KS1, 2, false, false>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), Geo::LDS, s, a);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_linear_gelu_supported(int dtype, int M, int K, int N) {
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && M > 0 && M % CM_BM == 0 && (K == 128 || K == 192 || K == 256 || K == 384 || K == 512) && N > 0 &&
           N % 32 == 0 && N <= CM_N_MAX;
}

extern "C" int mlpk_linear_gelu(int dtype, const void* x, int ldx, int M, int K, const float* ln_mean, const float* ln_rstd, int ln_group,
                                const float* csum, const void* w1, int ldw1, const float* b1, int nchunks, void* out, int ldo, float* row_part,
                                void* stream) {
    if (!x || !w1 || !b1 || !out) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (nchunks <= 0 || !mlpk_linear_gelu_supported(dtype, M, K, nchunks * 32)) return MLPK_ESHAPE;
    if ((ln_mean != nullptr) != (ln_rstd != nullptr) || (ln_mean != nullptr) != (csum != nullptr)) return MLPK_ENULL;
    if (ln_mean && ln_group <= 0) return MLPK_ESHAPE;
    if (ldw1 < K || ldw1 % 8 || ldx < K || ldx % 8 || ldo < nchunks * 32 || ldo % 8) return MLPK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)out & 15) || ((uintptr_t)row_part & 7)) return MLPK_EALIGN;
    ChanMlpArgs a;
    a.x = x; a.w1 = w1; a.w2 = nullptr; a.b1 = b1; a.csum = csum; a.b2 = nullptr; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.R = nullptr; a.out = out;
    a.row_part = row_part;
    a.M = M; a.G = nchunks; a.ldx = ldx; a.ldw1 = ldw1; a.ldw2 = 0; a.ldr = 0; a.ldo = ldo; a.ln_group = ln_mean ? ln_group : 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CM_CASE1(KS1) \
    case KS1: return dtype == MLPK_BF16 ? cm_launch_fc1<bf16_t, KS1>(a, s) : cm_launch_fc1<f16_t, KS1>(a, s);
    switch (K / 32) {
        CM_CASE1(4) CM_CASE1(6) CM_CASE1(8) CM_CASE1(12) CM_CASE1(16)
        default: return MLPK_ESHAPE;
    }
#undef CM_CASE1
}

extern "C" int mlpk_channel_mlp_supported(int dtype, int C, int hidden) {
    // (hidden >= 96: the kernel peels a tile's last group of 32 hidden units off its steady loop)
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && C % 32 == 0 && C >= 64 && C <= 192 && hidden > 64 && hidden <= CM_HID_MAX;
}

extern "C" int mlpk_channel_mlp(int dtype, const void* x, int ldx, int M, int C, const float* ln_mean, const float* ln_rstd, int ln_group,
                                const float* csum, const void* w1, int ldw1, const float* b1, const void* w2, int ldw2, const float* b2,
                                int nchunks, const void* R, int ldr, void* out, int ldo, float* row_part, void* stream) {
    if (!x || !w1 || !w2 || !b1 || !b2 || !out) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (M <= 0 || nchunks <= 0) return MLPK_ESHAPE;
    if (!mlpk_channel_mlp_supported(dtype, C, nchunks * 32)) return MLPK_ESHAPE;
    if ((ln_mean != nullptr) != (ln_rstd != nullptr) || (ln_mean != nullptr) != (csum != nullptr)) return MLPK_ENULL;
    if (ln_mean && ln_group <= 0) return MLPK_ESHAPE;
    if (ldw1 != 256 || ldw2 < nchunks * 32 || ldw2 % 8) return MLPK_ESHAPE;
    if (ldx < C || ldo < C || ldx % 8 || ldo % 8 || (R && (ldr < C || ldr % 8))) return MLPK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)out & 15) || ((uintptr_t)R & 15) ||
        ((uintptr_t)row_part & 7))
        return MLPK_EALIGN;
    ChanMlpArgs a;
    a.x = x; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.csum = csum; a.b2 = b2; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.R = R; a.out = out;
    a.row_part = row_part;
    a.M = M; a.G = nchunks; a.ldx = ldx; a.ldw1 = ldw1; a.ldw2 = ldw2; a.ldr = ldr; a.ldo = ldo; a.ln_group = ln_mean ? ln_group : 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CM_CASE(KS1) \
    case KS1: return dtype == MLPK_BF16 ? cm_launch<bf16_t, KS1>(a, s) : cm_launch<f16_t, KS1>(a, s);
    switch (C / 32) {
        CM_CASE(2) CM_CASE(3) CM_CASE(4) CM_CASE(5) CM_CASE(6)
        default: return MLPK_ESHAPE;
    }
#undef CM_CASE
}
