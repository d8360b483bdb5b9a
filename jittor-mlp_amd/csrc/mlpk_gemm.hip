// NT GEMM on MFMA with a fused epilogue -- the kernel every dense contraction of the path maps to.
//
//   acc[m,n] = sum_k A[m,k] * B[n,k]     A: M x K, B: N x K, both K-contiguous (nn.Linear layout)
//
// Design (gfx950 / CDNA4):
//   * workgroup tile BM x BN, K consumed in 128-byte slabs per row (64 bf16/f16 or 32 f32);
//     WM x WN wavefronts of 64 lanes, each owning a (BM/WM) x (BN/WN) sub-tile made of 16x16 MFMA
//     blocks (v_mfma_f32_16x16x32_{bf16,f16}; exact-f32 v_mfma_f32_16x16x4_f32 for float);
//   * LDS rows are 128 B = 8 x 16-byte chunks, chunk index XOR-swizzled with (row & 7) so the
//     ds_read_b128 fragment reads (16 rows x one chunk column per lane group) are conflict-free;
//   * global -> VGPR -> LDS staging, double-buffered, ONE barrier per K-slab: loads for slab t+1
//     are issued before the MFMAs of slab t and written to the other buffer after them;
//   * the accumulator is produced "transposed" (MFMA operands swapped) for row-major outputs so each
//     lane owns 4 consecutive n of one row -> 8/16-byte epilogue stores; for the token-mixing
//     (per-image transposed) output the natural orientation gives 4 consecutive channels instead;
//   * bias / exact GELU / per-column scale+shift (BatchNorm-eval, ResMLP gamma) / per-row scale /
//     residual-add or gate-multiply are applied on the fp32 accumulator before the single store;
//   * XCD-aware tile order: each of the 8 XCDs walks a contiguous range of tiles (n fastest) so the
//     A panel and the weight panels are re-used out of that XCD's private L2.
#include "mlpk_common.h"
#include "mlpk_gemm_q4.h"
#include "mlpk_gemm_skinny.h"
#include <stdio.h>
#include <stdlib.h>

namespace mlpk {

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    const void* R;
    const float* bias;
    const float* cscale;
    const float* cshift;
    const float* rscale;
    const float* ln_mean;   // folded LayerNorm / GroupNorm(1,C): v = (acc - ln_mean[m / ln_group]*ln_csum[n]) * ln_rstd[m / ln_group]
    const float* ln_rstd;
    const float* ln_csum;
    int ln_group;           // rows that share one statistic (1: LayerNorm; H*W: GroupNorm(1,C) on channel-last rows)
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int rperiod, act, res_mode;
    int t_rows, t_tokens;
    int vec_c, vec_r;   // vector (4-element) store / residual-load allowed
    // p8 launch (see gemm_nt_p8_kernel): row panels [m_base, m_base + panels * tile height), column groups
    int m_base, panels, cgroups;
    int rev;            // p8: walk the row panels from the last to the first (tuning: MLPK_P8_REVERSE)
    void* prof_buf;     // MLPK_P8_PROF builds: per-workgroup cycle sums (reserved & 8)
    int dbg_delay;      // de-phase sleep, units of 8128 cycles
    int dbg_q4;         // tuning bits of the generated q4 kernels (desc.reserved bits 16..23)
    int dbg;            // tuning ablations (desc.reserved): 1 = no main loop, 2 = no stores, 4 = no epilogue
    // by-product row statistics of the stored values (16-bit row-major outputs): pair (q, m) = (sum, sum of squares) of row m over
    // column block q; block width 128 (LDS-staged epilogue) or 32 (direct epilogue of the persistent tile).  PLANAR, one plane per
    // column block: a tile's pairs are one contiguous run (256 rows x 8 bytes), written as whole cache lines -- interleaved per
    // row ([m][q]) every pair was a lone 8-byte masked write next to pairs other workgroups write later (measured +27 us on a
    // 130 us GEMM)
    float* row_part;
    int row_part_ld;    // plane stride in pairs (>= M)
    // implicit-convolution A operand (gemm_nt_s3_conv_kernel, round 6): A = a channel-last (B, cv_H, cv_W, cv_Cin) tensor, row m = output pixel (b, oy, ox),
    // column k = (tap dy * cv_kw + dx) * cv_Cin + channel; a 64-byte slab of K = 32 channels of ONE tap (cv_Cin % 32 == 0), zeros outside the map
    int cv_H, cv_W, cv_Cin, cv_kw, cv_cpk, cv_inv_cpk, cv_inv_kw, cv_stride, cv_pad, cv_Ho, cv_Wo;
};

__device__ __attribute__((aligned(64))) unsigned g_zero_slab[16];      // 64 zero bytes: the slab of a tap outside the map

// tuning aid (dbg & 8): wave 0 / lane 0 of each workgroup logs s_memtime stamps into the buffer passed in R
#define MLPK_STAMP(slot)                                                                          \
    if ((p.dbg & 8) && threadIdx.x == 0 && (slot) < 64)                                            \
        reinterpret_cast<unsigned long long*>(const_cast<void*>(p.R))[(size_t)blockIdx.x * 64 + (slot)] = __builtin_readcyclecounter();

// One 1-KiB LDS-DMA piece: 64 lanes x 16 bytes, per-lane global source, LDS destination = M0 base + lane*16.
// Inline asm on purpose: hipcc tracks the builtin form as an LDS store and drains vmcnt(0) in front of any
// later ds_read it cannot disambiguate, which serialises the pipeline; here the counted waits are ours.
__device__ __forceinline__ void glds_piece(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ unsigned g_cu_ticket[2048];

template <typename T> struct Mma;
// `pinned` = the same instruction as volatile inline asm: it keeps its place between the barriers / waits /
// LDS-DMA pieces of a hand-scheduled loop (hipcc otherwise floats the side-effect-free builtin across them).
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void pinned(u32x4 a, u32x4 b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    static __device__ __forceinline__ void pinned(u32x4 a, u32x4 b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void pinned(u32x4 a, u32x4 b, f32x4& c) {
        float af[4], bf[4];
        __builtin_memcpy(af, &a, 16);
        __builtin_memcpy(bf, &b, 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(af[e]), "v"(bf[e]));
    }
    // one 16-byte chunk = 4 consecutive k per lane group; four K=4 steps, each taking element e of
    // every lane's chunk (any k <-> (lane group, e) bijection is valid as long as A and B agree).
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        // NB: __builtin_bit_cast(float, a.y) on an ext-vector ELEMENT miscompiles with hipcc 7.2 (every
        // element reads .x); go through memcpy of the whole vector instead.
        float af[4], bf[4];
        __builtin_memcpy(af, &a, 16);
        __builtin_memcpy(bf, &b, 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], c, 0, 0, 0);
        return c;
    }
};

// ---- 4-element vector access in the storage dtype ---------------------------------------------
template <typename T> __device__ __forceinline__ void load4(const T* p, bool vec, float (&v)[4]) {
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const u32x2 t = *reinterpret_cast<const u32x2*>(p);
            T e[4];
            __builtin_memcpy(e, &t, 8);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = to_f32(e[r]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = to_f32(p[r]);
    }
}

template <typename T> __device__ __forceinline__ void store4(T* p, bool vec, const float (&v)[4]) {
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            f32x4 t = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(p) = t;
        } else {
            T e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = from_f32<T>(v[r]);
            u32x2 t;
            __builtin_memcpy(&t, e, 8);
            *reinterpret_cast<u32x2*>(p) = t;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = from_f32<T>(v[r]);
    }
}

template <bool B> struct BoolK { static constexpr bool value = B; };

// Shared epilogue: consumes the fp32 accumulator blocks of one workgroup tile (see the kernels for
// the accumulator orientation) and writes C.  `smem` is the workgroup's LDS, free for staging once
// every wave has passed the barrier that ends the main loop.
// SPLIT = false: a wave owns one contiguous TM x TN sub-tile.  SPLIT = true (the "p8" kernel): it owns the
// 2 x 2 quadrants (TM/2) x (TN/2) at (half_m * BM/2 + wm * TM/2, half_n * BN/2 + wn * TN/2).
template <typename T, int BM, int BN, int WM, int WN, bool TRANS, bool SPLIT = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem,
                                              const int m0, const int n0, const int tid) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15;
    const int fg = lane >> 4;
    // first row / column (inside the workgroup tile) of accumulator block row i / block column j
    auto rbase = [&](int i) {
        if constexpr (SPLIT) return (i / (FM / 2)) * (BM / 2) + wm * (TM / 2) + (i % (FM / 2)) * 16;
        else return wm * TM + i * 16;
    };
    auto cbase = [&](int j) {
        if constexpr (SPLIT) return (j / (FN / 2)) * (BN / 2) + wn * (TN / 2) + (j % (FN / 2)) * 16;
        else return wn * TN + j * 16;
    };
    if (p.dbg & 4) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) sacc += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
        if (sacc == 123.456f) reinterpret_cast<float*>(p.C)[0] = sacc;
        return;
    }
    T* __restrict__ C = reinterpret_cast<T*>(p.C);
    const T* R = reinterpret_cast<const T*>(p.R);
    const bool gelu = p.act == MLPK_ACT_GELU;

    if constexpr (!TRANS && sizeof(T) == 2) {
        // ---- 2-byte row-major output: stage the finished tile through LDS (the pipeline buffers are
        // free now) so that global traffic is full 16-byte-per-lane, whole-row coalesced: the MFMA
        // accumulator layout only gives 8 bytes per lane in 32-byte runs, which measured 1.9 TB/s.
        // Phase 1: bias / GELU / column scale+shift / row scale on the fp32 accumulator, round to T,
        //          ds_write_b64 into a [BM][BN] tile whose 16-byte chunks are XOR-swizzled by row.
        // Phase 2: every thread moves 16-byte chunks LDS -> (optional residual add|mul) -> global.
        constexpr int CPR = BN / 8;                      // 16-byte chunks per tile row
        constexpr int XM = (CPR >= 16 ? 16 : CPR) - 1;   // swizzle mask
        T* tile = reinterpret_cast<T*>(smem);
        // phase-2 geometry, and the residual chunks of the first passes put in flight NOW so that their
        // latency hides under the phase-1 VALU work
        constexpr int RPASS = NT / CPR;                  // rows moved per pass
        constexpr int NP = BM / RPASS;                   // passes
        constexpr int PB0 = NP < 8 ? NP : 8;
        const int c16 = tid % CPR;
        const int rsub = tid / CPR;
        const int gn = n0 + c16 * 8;
        const bool has_res = p.res_mode != MLPK_RES_NONE;
        const bool fast = p.vec_c == 2 && (!has_res || p.vec_r == 2) && gn + 8 <= p.N;
        u32x4 rr0[PB0];
        if (fast && has_res) {
#pragma unroll
            for (int q = 0; q < PB0; ++q) {
                const int gm = m0 + q * RPASS + rsub;
                rr0[q] = gm < p.M ? *reinterpret_cast<const u32x4*>(R + (size_t)gm * p.ldr + gn) : u32x4{0u, 0u, 0u, 0u};
            }
        }
        // row parameters once per accumulator block row (not once per block): row scale, folded-LayerNorm mean / rstd
        float rs_[FM], lmu_[FM], lrs_[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            rs_[i] = 1.f; lmu_[i] = 0.f; lrs_[i] = 1.f;
            if (p.rscale || p.ln_mean) {
                int m = m0 + rbase(i) + frow;
                m = m < p.M ? m : p.M - 1;
                if (p.rscale) rs_[i] = p.rscale[m % p.rperiod];
                if (p.ln_mean) { lmu_[i] = p.ln_mean[m / p.ln_group]; lrs_[i] = p.ln_rstd[m / p.ln_group]; }
            }
        }
        // the element loop is instantiated with and without the activation (a per-element uniform branch around the
        // GELU code costs more than the GELU's arithmetic); GELU runs on float pairs (v_pk_*)
        auto phase1 = [&](auto gelu_c) {
            constexpr bool GELU = decltype(gelu_c)::value;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nl = cbase(j) + 4 * fg;            // column of this lane's 4-vector inside the tile
                float bz[4], cs[4], ch[4], lc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int n = n0 + nl + r;
                    n = n < p.N ? n : p.N - 1;
                    bz[r] = p.bias ? p.bias[n] : 0.f;
                    lc[r] = p.ln_mean ? p.ln_csum[n] : 0.f;
                    cs[r] = p.cscale ? p.cscale[n] : 1.f;
                    // (no shift = -0.0: t * 1 + (-0) keeps the sign of a zero t.  The logistic GELU returns x * 0 = -0 for x < -10.4,
                    //  and the generated q4 kernels store that: with +0.0 here the two tiles would differ in the sign bit of a zero)
                    ch[r] = p.cshift ? p.cshift[n] : -0.f;
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int rl = rbase(i) + frow;
                    const float rs = rs_[i], lmu = lmu_[i], lrs = lrs_[i];
                    float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
                    float t[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = (v[r] - lmu * lc[r]) * lrs + bz[r];
                    if constexpr (GELU) {
                        if constexpr (sizeof(T) == 2) {
                            f32x2 g2[2] = {f32x2{t[0], t[1]}, f32x2{t[2], t[3]}};
                            gelu_pk_n<T, 2>(g2);
                            t[0] = g2[0].x; t[1] = g2[0].y; t[2] = g2[1].x; t[3] = g2[1].y;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) t[r] = gelu_f(t[r]);
                        }
                    }
                    T e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = from_f32<T>((t[r] * cs[r] + ch[r]) * rs);
                    u32x2 pk;
                    __builtin_memcpy(&pk, e, 8);
                    const int c16w = nl >> 3;
                    char* dst = reinterpret_cast<char*>(tile) + rl * (BN * 2) + (((c16w ^ (rl & XM)) << 4) | ((nl & 4) << 1));
                    *reinterpret_cast<u32x2*>(dst) = pk;
                }
            }
        };
        if (gelu) phase1(BoolK<true>{});
        else phase1(BoolK<false>{});
        __syncthreads();
        // STATS = true (GemmArgs::row_part; the host has checked what `fast` checks, except that a lane's chunk may lie past N):
        // EVERY lane runs the passes, with its loads and stores predicated, because the DPP row reduction of the statistics
        // must not read lanes that a branch has switched off (a DPP read of an EXEC-disabled lane does not return 0)
        auto passes = [&](auto stats_c) {
            constexpr bool STATS = decltype(stats_c)::value;
            const bool live = !STATS || fast;
            // residual chunks of the remaining groups go in flight first, then each pass is LDS read ->
            // combine -> one 16-byte store; rows past M are skipped.
            u32x4 rr[NP];
            float ps1[STATS ? NP : 1], ps2[STATS ? NP : 1];
#pragma unroll
            for (int q = 0; q < PB0; ++q) rr[q] = (STATS && !(fast && has_res)) ? u32x4{0u, 0u, 0u, 0u} : rr0[q];
            if (has_res) {
#pragma unroll
                for (int q = PB0; q < NP; ++q) {
                    const int gm = m0 + q * RPASS + rsub;
                    rr[q] = (gm < p.M && live) ? *reinterpret_cast<const u32x4*>(R + (size_t)gm * p.ldr + gn) : u32x4{0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int rl = q * RPASS + rsub;
                const int gm = m0 + rl;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tile) + rl * (BN * 2) +
                                                                  ((c16 ^ (rl & XM)) << 4));
                u32x4 outv = raw;
                if (has_res) {
                    T a8[8], r8[8];
                    __builtin_memcpy(a8, &raw, 16);
                    __builtin_memcpy(r8, &rr[q], 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = to_f32(a8[e]), b = to_f32(r8[e]);
                        a8[e] = from_f32<T>(p.res_mode == MLPK_RES_ADD ? a + b : a * b);
                    }
                    __builtin_memcpy(&outv, a8, 16);
                }
                if (gm < p.M && live && (!(p.dbg & 2) || outv.x == 0x12345678u))
                    *reinterpret_cast<u32x4*>(C + (size_t)gm * p.ldc + gn) = outv;
                if constexpr (STATS && CPR >= 16) {
                    ps1[q] = 0.f; ps2[q] = 0.f;
                    chunk_sums<T>(outv, ps1[q], ps2[q]);
                }
            }
            if constexpr (STATS && CPR >= 16) {
                // by-product statistics: the row's 128 columns sit in 16 lanes (one DPP row), a plane's 32 columns in 4 consecutive
                // lanes.  All the passes' reductions in one block after the stores, so that their dependent DPP steps interleave
                // instead of each waiting out its own latency.  (Round 4: planes of 32 columns in the library-wide order -- quad_sum,
                // mlpk_common.h -- instead of 128: a row's statistics no longer depend on the tile the batch size selects.)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    ps1[q] = quad_sum(live ? ps1[q] : 0.f);
                    ps2[q] = quad_sum(live ? ps2[q] : 0.f);
                }
                if ((c16 & 3) == 0 && live && gn < p.N) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const int gm = m0 + q * RPASS + rsub;
                        if (gm < p.M) *reinterpret_cast<f32x2*>(p.row_part + ((size_t)(gn >> 5) * p.row_part_ld + gm) * 2) = f32x2{ps1[q], ps2[q]};
                    }
                }
            }
        };
        if (p.row_part) {
            passes(BoolK<true>{});
        } else if (fast) {
            passes(BoolK<false>{});
        } else if (gn < p.N) {
            for (int rp = 0; rp < NP; ++rp) {
                const int rl = rp * RPASS + rsub;
                const int gm = m0 + rl;
                if (gm >= p.M) break;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tile) + rl * (BN * 2) +
                                                                  ((c16 ^ (rl & XM)) << 4));
                T* cp = C + (size_t)gm * p.ldc + gn;
                T a8[8];
                __builtin_memcpy(a8, &raw, 16);
                for (int e = 0; e < 8 && gn + e < p.N; ++e) {
                    float a = to_f32(a8[e]);
                    if (has_res) {
                        const float b = to_f32(R[(size_t)gm * p.ldr + gn + e]);
                        a = p.res_mode == MLPK_RES_ADD ? a + b : a * b;
                    }
                    cp[e] = from_f32<T>(a);
                }
            }
        }
    } else if constexpr (!TRANS) {
        // lane owns row m = .. + (lane & 15) and 4 consecutive columns n = .. + 4*(lane >> 4) + r
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + rbase(i) + frow;
            if (m >= p.M) continue;
            const float rs = p.rscale ? p.rscale[m % p.rperiod] : 1.0f;
            const float lmu = p.ln_mean ? p.ln_mean[m / p.ln_group] : 0.f, lrs = p.ln_mean ? p.ln_rstd[m / p.ln_group] : 1.f;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nb = n0 + cbase(j) + 4 * fg;
                if (nb >= p.N) continue;
                const bool full = nb + 3 < p.N;
                float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = (nb + r < p.N) ? nb + r : p.N - 1;
                    float t = v[r];
                    if (p.ln_mean) t = (t - lmu * p.ln_csum[n]) * lrs;
                    if (p.bias) t += p.bias[n];
                    if (gelu) t = gelu_t<T>(t);
                    if (p.cscale) t *= p.cscale[n];
                    if (p.cshift) t += p.cshift[n];
                    v[r] = t * rs;
                }
                const size_t co = (size_t)m * p.ldc + nb;
                if (full) {
                    if (p.res_mode != MLPK_RES_NONE) {
                        float rv[4];
                        load4<T>(R + (size_t)m * p.ldr + nb, p.vec_r != 0, rv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (p.res_mode == MLPK_RES_ADD) ? v[r] + rv[r] : v[r] * rv[r];
                    }
                    if (!(p.dbg & 2) || v[0] == 123.456f) store4<T>(C + co, p.vec_c != 0, v);
                } else {
                    for (int r = 0; r < 4 && nb + r < p.N; ++r) {
                        float t = v[r];
                        if (p.res_mode != MLPK_RES_NONE) {
                            const float rv = to_f32(R[(size_t)m * p.ldr + nb + r]);
                            t = (p.res_mode == MLPK_RES_ADD) ? t + rv : t * rv;
                        }
                        C[co + r] = from_f32<T>(t);
                    }
                }
            }
        }
    } else {
        // token-transposed output: lane owns token n = .. + (lane & 15) and 4 consecutive rows
        // m = .. + 4*(lane >> 4) + r, i.e. 4 consecutive channels of one image (t_rows % 4 == 0).
        if constexpr (sizeof(T) == 2) {
            // Whole tile inside one image: stage the finished values through LDS as [token][channel] so that global
            // traffic (store, and the residual / gate operand) is 16 bytes per lane in whole BM-channel runs per token;
            // the accumulator layout itself only gives 8-byte pieces in 32-byte runs.
            const bool has_res = p.res_mode != MLPK_RES_NONE;
            if (p.t_rows % BM == 0 && m0 + BM <= p.M && p.vec_c == 2 && (!has_res || p.vec_r == 2) && !(p.dbg & 2)) {
                constexpr int CPT = BM / 8;                      // 16-byte chunks per token row
                constexpr int XT = (CPT >= 16 ? 16 : CPT) - 1;
                constexpr int TPP = NT / CPT;                    // tokens moved per pass
                char* tile = smem;
                const int img = m0 / p.t_rows;
                const int c0 = m0 - img * p.t_rows;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int nl = cbase(j) + frow;
                    int n = n0 + nl;
                    n = n < p.N ? n : p.N - 1;
                    const float bn = p.bias ? p.bias[n] : 0.0f;
                    const float cs = p.cscale ? p.cscale[n] : 1.0f;
                    const float ch = p.cshift ? p.cshift[n] : -0.0f;     // (-0: keeps the sign of a zero, see the row-major epilogue)
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int ml = rbase(i) + 4 * fg;
                        float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
                        T e[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float t = v[r] + bn;
                            if (gelu) t = gelu_t<T>(t);
                            t = t * cs + ch;
                            if (p.rscale) t *= p.rscale[(m0 + ml + r) % p.rperiod];        // (the GEMM row, as in the row-major epilogue: a period of t_rows gives the channel)
                            e[r] = from_f32<T>(t);
                        }
                        u32x2 pk;
                        __builtin_memcpy(&pk, e, 8);
                        *reinterpret_cast<u32x2*>(tile + nl * (BM * 2) + ((((ml >> 3) ^ (nl & XT)) << 4) | ((ml & 4) << 1))) = pk;
                    }
                }
                __syncthreads();
                const int cc = tid % CPT;
                const int tsub = tid / CPT;
#pragma unroll 4
                for (int ps = 0; ps < BN / TPP; ++ps) {
                    const int nl = ps * TPP + tsub;
                    const int n = n0 + nl;
                    if (n >= p.N) continue;
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(tile + nl * (BM * 2) + ((cc ^ (nl & XT)) << 4));
                    const size_t row = (size_t)img * p.t_tokens + n;
                    u32x4 outv = raw;
                    if (has_res) {
                        const u32x4 rr = *reinterpret_cast<const u32x4*>(R + row * p.ldr + c0 + cc * 8);
                        T a8[8], r8[8];
                        __builtin_memcpy(a8, &raw, 16);
                        __builtin_memcpy(r8, &rr, 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = to_f32(a8[e]), b = to_f32(r8[e]);
                            a8[e] = from_f32<T>(p.res_mode == MLPK_RES_ADD ? a + b : a * b);
                        }
                        __builtin_memcpy(&outv, a8, 16);
                    }
                    *reinterpret_cast<u32x4*>(C + row * p.ldc + c0 + cc * 8) = outv;
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + cbase(j) + frow;
            if (n >= p.N) continue;
            const float bn = p.bias ? p.bias[n] : 0.0f;
            const float cs = p.cscale ? p.cscale[n] : 1.0f;
            const float ch = p.cshift ? p.cshift[n] : -0.0f;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int mb = m0 + rbase(i) + 4 * fg;
                if (mb >= p.M) continue;     // M % 4 == 0 for TOKEN_T, so the 4 rows are all valid
                const int img = mb / p.t_rows;
                const int c = mb - img * p.t_rows;
                float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = v[r] + bn;
                    if (gelu) t = gelu_t<T>(t);
                    t = t * cs + ch;
                    if (p.rscale) t *= p.rscale[(mb + r) % p.rperiod];
                    v[r] = t;
                }
                const size_t row = (size_t)img * p.t_tokens + n;
                if (p.res_mode != MLPK_RES_NONE) {
                    float rv[4];
                    load4<T>(R + row * p.ldr + c, p.vec_r != 0, rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (p.res_mode == MLPK_RES_ADD) ? v[r] + rv[r] : v[r] * rv[r];
                }
                store4<T>(C + row * p.ldc + c, p.vec_c != 0, v);
            }
        }
    }
}

// GLDS = false: global -> VGPR -> LDS staging (handles every shape: predicated, zero-filled tails).
// GLDS = true : global_load_lds_dwordx4 straight into LDS (no staging VGPRs, no ds_write pass).  The
//               LDS image is lane-linear (wave base + lane*16), so the XOR swizzle is applied to the
//               per-lane SOURCE address; out-of-range rows are clamped (their results are never
//               stored) and K must be a multiple of half a slab (the ragged half is skipped).
template <typename T, int BM, int BN, int WM, int WN, bool TRANS, bool GLDS>
__global__ void __launch_bounds__(WM* WN * 64) gemm_nt_kernel(const GemmArgs p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                // elements per 128-byte K-slab
    constexpr int RPP = NT / 8;                // rows staged per pass (8 chunks per row)
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    constexpr int BUF = (BM + BN) * 128;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be made of 16x16 blocks");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * BM;
    const int n0 = (wg % tiles_n) * BN;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(p.B);

    // ---- staging geometry: thread -> (row r0 + i*RPP, chunk kc) ----
    const int kc = tid & 7;
    const int r0 = tid >> 3;
    const int st_off = (r0 * 128) + ((kc ^ (r0 & 7)) << 4);   // (row & 7) is the same for every pass (RPP % 8 == 0)
    u32x4 ra[A_IT], rb[B_IT];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    const int nk = (p.dbg & 1) ? 0 : (p.K + BK - 1) / BK;

#define MLPK_GLOAD(kt)                                                                           \
    {                                                                                            \
        const int k__ = (kt)*BK + kc * EPC;                                                      \
        const bool kok__ = k__ < p.K;                                                            \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                       \
            const int gm = m0 + r0 + i * RPP;                                                    \
            ra[i] = (kok__ && gm < p.M)                                                          \
                        ? *reinterpret_cast<const u32x4*>(A + (size_t)gm * p.lda + k__)          \
                        : zero4;                                                                 \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                       \
            const int gn = n0 + r0 + i * RPP;                                                    \
            rb[i] = (kok__ && gn < p.N)                                                          \
                        ? *reinterpret_cast<const u32x4*>(B + (size_t)gn * p.ldb + k__)          \
                        : zero4;                                                                 \
        }                                                                                        \
    }
#define MLPK_SSTORE(buf)                                                                         \
    {                                                                                            \
        char* base__ = smem + (buf)*BUF + st_off;                                                \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                         \
            *reinterpret_cast<u32x4*>(base__ + i * (RPP * 128)) = ra[i];                         \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                         \
            *reinterpret_cast<u32x4*>(base__ + BM * 128 + i * (RPP * 128)) = rb[i];              \
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = (wave sub-tile) + 16*block + (lane & 15); (row & 7) == (lane & 7)
    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int sw = lane & 7;
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = BM * 128 + (wn * TN + frow) * 128;
    const int c_off0 = ((fg ^ sw)) << 4;        // K sub-step 0: chunk = fg
    const int c_off1 = ((fg ^ sw) ^ 4) << 4;    // K sub-step 1: chunk = 4 + fg

#define MLPK_COMPUTE(kt)                                                                         \
    {                                                                                            \
        const char* buf = smem + ((kt)&1) * BUF;                                                 \
        const int krem = p.K - (kt)*BK; /* wave-uniform: skip the ragged half of the last slab */ \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                       \
            if (kk == 1 && krem <= BK / 2) break;                                                \
            const int co = kk ? c_off1 : c_off0;                                                 \
            u32x4 af[FM], bf[FN];                                                                \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                       \
                af[i] = *reinterpret_cast<const u32x4*>(buf + a_rd + i * 2048 + co);             \
            _Pragma("unroll") for (int j = 0; j < FN; ++j)                                       \
                bf[j] = *reinterpret_cast<const u32x4*>(buf + b_rd + j * 2048 + co);             \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                       \
                _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[i][j] =                       \
                    TRANS ? Mma<T>::run(af[i], bf[j], acc[i][j]) : Mma<T>::run(bf[j], af[i], acc[i][j]); \
        }                                                                                        \
    }

    if constexpr (!GLDS) {
        MLPK_GLOAD(0);
        MLPK_SSTORE(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1) < nk;
            if (more) MLPK_GLOAD(kt + 1);
            MLPK_COMPUTE(kt);
            if (more) MLPK_SSTORE((kt + 1) & 1);
            __syncthreads();
        }
    } else {
        constexpr int NW = WM * WN;
        constexpr int A_G = BM / 8 / NW, B_G = BN / 8 / NW;   // 1-KiB (8 rows x 128 B) pieces per wave
        static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row pieces per wave");
        const int lrow = lane >> 3;
        const int lchunk = (lane & 7) ^ lrow;               // logical chunk this lane fetches (source-side swizzle)
        const T* srcA[A_G];
        const T* srcB[B_G];
#pragma unroll
        for (int g = 0; g < A_G; ++g) {
            int gm = m0 + (wave * A_G + g) * 8 + lrow;
            gm = gm < p.M ? gm : p.M - 1;
            srcA[g] = A + (size_t)gm * p.lda;
        }
#pragma unroll
        for (int g = 0; g < B_G; ++g) {
            int gn = n0 + (wave * B_G + g) * 8 + lrow;
            gn = gn < p.N ? gn : p.N - 1;
            srcB[g] = B + (size_t)gn * p.ldb;
        }
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#define MLPK_STAGE(kt)                                                                           \
    {                                                                                            \
        int k__ = (kt)*BK + lchunk * EPC;                                                        \
        k__ = k__ < p.K ? k__ : p.K - EPC;                                                       \
        char* dst__ = smem + ((kt)&1) * BUF + wave * (A_G * 1024);                               \
        _Pragma("unroll") for (int g = 0; g < A_G; ++g)                                          \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(srcA[g] + k__), (lds_ptr_t)(dst__ + g * 1024), 16, 0, 0); \
        char* dstb__ = smem + ((kt)&1) * BUF + BM * 128 + wave * (B_G * 1024);                   \
        _Pragma("unroll") for (int g = 0; g < B_G; ++g)                                          \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(srcB[g] + k__), (lds_ptr_t)(dstb__ + g * 1024), 16, 0, 0); \
    }
        MLPK_STAMP(0);
        MLPK_STAGE(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MLPK_STAMP(1);
        for (int kt = 0; kt < nk; ++kt) {
            if ((kt + 1) < nk) MLPK_STAGE(kt + 1);
            MLPK_STAMP(2 + 4 * kt);
            MLPK_COMPUTE(kt);
            MLPK_STAMP(3 + 4 * kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MLPK_STAMP(4 + 4 * kt);
            __syncthreads();
            MLPK_STAMP(5 + 4 * kt);
        }
#undef MLPK_STAGE
    }
    MLPK_STAMP(60);
    if (p.dbg & 8) return;
#undef MLPK_COMPUTE
#undef MLPK_GLOAD
#undef MLPK_SSTORE

    gemm_epilogue<T, BM, BN, WM, WN, TRANS>(p, acc, smem, m0, n0, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// "s3" pipeline: 4 wavefronts per workgroup (one per SIMD), 64-byte LDS rows (K slabs of 32 bf16/f16 or
// 16 f32), THREE LDS stages filled by global_load_lds two slabs ahead, ONE raw s_barrier per slab and
// counted vmcnt waits (the loads of the next slab stay in flight across the barrier).  72 KiB of LDS
// and <= 256 VGPRs per wave let TWO independent workgroups share a CU: each SIMD then holds one wave
// of each, so one workgroup's MFMAs cover the other's LDS reads / barrier waits, and one workgroup's
// epilogue (VALU GELU + stores) overlaps the other's main loop -- the two costs the lockstep
// 8-wave / 1-workgroup-per-CU kernel above cannot hide (measured: 46 % MFMA busy in its main loop).
// 64-byte rows: chunk c of row r is stored at chunk c ^ ((r & 8) >> 2), conflict-free for ds_read_b128.
template <typename T, int BM, int BN, int WM, int WN, bool TRANS, bool CONV = false>
__device__ __forceinline__ void gemm_nt_s3_body(const GemmArgs& p, const int bid, const int nblocks) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = 4 * EPC;                           // elements per 64-byte slab
    constexpr int STAGE_B = (BM + BN) * 64;
    constexpr int A_G = BM / 16 / NW, B_G = BN / 16 / NW;  // 1-KiB pieces (16 rows x 64 B) per wave per slab
    constexpr int PIECES = A_G + B_G;
    static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile rows must split into 16-row pieces per wave");
    static_assert(3 * STAGE_B >= BM * BN * 2 || sizeof(T) != 2, "LDS too small for the staged epilogue");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int wg = xcd_remap(bid, nblocks);
    const int m0 = (wg / tiles_n) * BM;
    const int n0 = (wg % tiles_n) * BN;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(p.B);

    // staging: lane -> (row lrow of the 16-row piece, physical chunk lane & 3); source-side swizzle
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((lrow & 8) >> 2);
    const T* srcA[A_G];
    const T* srcB[B_G];
    int cvy[CONV ? A_G : 1], cvx[CONV ? A_G : 1];          // CONV: the window's first input row / column of the piece's output pixel
#pragma unroll
    for (int g = 0; g < A_G; ++g) {
        int gm = m0 + (wave * A_G + g) * 16 + lrow;
        gm = gm < p.M ? gm : p.M - 1;
        if constexpr (CONV) {
            const int img = gm / (p.cv_Ho * p.cv_Wo), rem = gm - img * (p.cv_Ho * p.cv_Wo);
            const int oy = rem / p.cv_Wo, ox = rem - oy * p.cv_Wo;
            cvy[g] = oy * p.cv_stride - p.cv_pad;
            cvx[g] = ox * p.cv_stride - p.cv_pad;
            // (may point in front of the tensor for a halo window: only dereferenced for taps inside the map)
            srcA[g] = A + (((ptrdiff_t)img * p.cv_H + cvy[g]) * p.cv_W + cvx[g]) * p.cv_Cin + lchunk * EPC;
        } else {
            srcA[g] = A + (size_t)gm * p.lda + lchunk * EPC;
        }
    }
#pragma unroll
    for (int g = 0; g < B_G; ++g) {
        int gn = n0 + (wave * B_G + g) * 16 + lrow;
        gn = gn < p.N ? gn : p.N - 1;
        srcB[g] = B + (size_t)gn * p.ldb + lchunk * EPC;
    }
    const T* const zslab = reinterpret_cast<const T*>(g_zero_slab) + lchunk * EPC;
    // CONV: the A piece q of slab kt = 32 channels of tap (dy, dx) of the piece's window, or the zero slab outside the map (kt is wave-uniform)
    auto conv_src = [&](const int q, const int kt) -> const T* {
        const int tap = (kt * p.cv_inv_cpk) >> 16, c32 = kt - tap * p.cv_cpk;
        const int dy = (tap * p.cv_inv_kw) >> 16, dx = tap - dy * p.cv_kw;
        const bool ok = (unsigned)(cvy[CONV ? q : 0] + dy) < (unsigned)p.cv_H && (unsigned)(cvx[CONV ? q : 0] + dx) < (unsigned)p.cv_W;
        return ok ? srcA[q] + ((ptrdiff_t)dy * p.cv_W + dx) * p.cv_Cin + c32 * BK : zslab;
    };
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int nk = (p.dbg & 1) ? 0 : p.K / BK;

    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned dst_a = __builtin_amdgcn_readfirstlane(lds_base + wave * (A_G * 1024));
    const unsigned dst_b = __builtin_amdgcn_readfirstlane(lds_base + BM * 64 + wave * (B_G * 1024));
#define S3_PIECE(kt, q)                                                                                \
    {                                                                                                  \
        const unsigned st__ = ((kt) % 3) * STAGE_B;                                                     \
        if ((q) < A_G) glds_piece(CONV ? conv_src((q) < A_G ? (q) : 0, (kt)) : srcA[(q) < A_G ? (q) : 0] + (size_t)(kt)*BK, dst_a + st__ + (q)*1024); \
        else glds_piece(srcB[(q) >= A_G ? (q)-A_G : 0] + (size_t)(kt)*BK, dst_b + st__ + ((q)-A_G) * 1024); \
    }
#define S3_STAGE(kt)                                                                                   \
    {                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < PIECES; ++q) S3_PIECE(kt, q);                            \
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int co = (fg ^ ((frow & 8) >> 2)) << 4;
    const int a_rd = (wm * TM + frow) * 64 + co;
    const int b_rd = BM * 64 + (wn * TN + frow) * 64 + co;

    // ---- de-phase the two workgroups that share a CU (tuning flag dbg & 16) ----
    // Both start together and would run main loop / epilogue in lockstep, leaving the matrix pipe idle
    // during both epilogues.  The second arriver on a CU (per-CU ticket, first launch round only) sleeps
    // for about one epilogue, so that from then on one workgroup's VALU/store phase overlaps the other's MFMAs.
    if ((p.dbg & 16) && bid < 512) {
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID[15:0]
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
            const unsigned key = ((xcc & 7) << 8) | ((hw >> 8) & 0xff);                   // cu / sh / se bits
            const unsigned ticket = atomicAdd(&g_cu_ticket[key], 1u);
            if (ticket & 1) {
                for (int i = 0; i < p.dbg_delay; ++i) __builtin_amdgcn_s_sleep(127);
            }
        }
        __syncthreads();
    }
    MLPK_STAMP(0);
    if (nk > 0) S3_STAGE(0);
    if (nk > 1) S3_STAGE(1);
    MLPK_STAMP(1);
    for (int kt = 0; kt < nk; ++kt) {
        MLPK_STAMP(2 + 2 * kt);
        // own pieces of slab kt have landed (slab kt+1, if issued, may still be in flight) ...
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and after the barrier everybody's have; every wave is also past its reads of slab kt-1,
        // whose stage is the one refilled next.
        __builtin_amdgcn_s_barrier();
        MLPK_STAMP(3 + 2 * kt);
        const char* buf = smem + (kt % 3) * STAGE_B;
        u32x4 af[FM], bf[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(buf + b_rd + j * 1024);
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const u32x4*>(buf + a_rd + i * 1024);
        // the fragment reads are issued first (they cannot sink below the asm pieces: "memory" clobber) ...
        // ... then the 1-KiB LDS-DMA pieces of slab kt+2 go out ONE AT A TIME between groups of MFMAs.
        // Issued back to back (all waves at once, right after the barrier) they saturate the CU's
        // texture-address path and every wave sits in VMEM issue for ~95 cycles per piece with the matrix
        // pipe idle (s_memtime stamps: ~760 of ~3660 cycles per 64-wide K step of the 8-wave kernel).
        constexpr int NB = FM * FN;                  // MFMA blocks per slab
        constexpr int GAP = NB / PIECES;             // MFMAs between two pieces
        const bool do_stage = kt + 2 < nk;           // wave-uniform
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const int i = blk / FN, j = blk % FN;
            if (blk % GAP == 0 && blk / GAP < PIECES) {
                if (do_stage) S3_PIECE(kt + 2, blk / GAP);
            }
            acc[i][j] = TRANS ? Mma<T>::run(af[i], bf[j], acc[i][j]) : Mma<T>::run(bf[j], af[i], acc[i][j]);
        }
    }
#undef S3_PIECE
#undef S3_STAGE
    MLPK_STAMP(60);
    __syncthreads();
    MLPK_STAMP(61);
    if (p.dbg & 8) return;
    gemm_epilogue<T, BM, BN, WM, WN, TRANS>(p, acc, smem, m0, n0, threadIdx.x);
}

template <typename T, int BM, int BN, int WM, int WN, bool TRANS>
__global__ void __launch_bounds__(WM* WN * 64, 2) gemm_nt_s3_kernel(const GemmArgs p) {
    gemm_nt_s3_body<T, BM, BN, WM, WN, TRANS>(p, (int)blockIdx.x, (int)gridDim.x);
}

// The A operand as an implicit convolution window (round 6, mlpk_conv_gemm_nhwc): the strided 3 x 3 transitions of Hire-MLP / CycleMLP (hire_mlp.py:161,
// cycle_mlp.py:220-231) without the gathered operand -- mlpk_im2col wrote B Ho Wo x 9 Cin values for the GEMM to read back.  Same tile, same K order, same
// epilogue: the bits of mlpk_im2col + mlpk_gemm_nt on this tile.
template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64, 2) gemm_nt_s3_conv_kernel(const GemmArgs p) {
    gemm_nt_s3_body<T, BM, BN, WM, WN, false, true>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Two independent products of the same tile family in ONE launch (round 6, mlpk_gemm_nt_pair): workgroups [0, tiles0) compute p0's tiles,
// the rest p1's -- Hire-MLP's h- and w-branch Linears (hire_mlp.py:139-143: 20-30 us each, launch- and latency-bound) as two launches per
// block instead of four and without a side stream.  Every tile is computed exactly as by the single launch: the same bits.
template <typename T, int BM, int BN, int WM, int WN, bool TRANS>
__global__ void __launch_bounds__(WM* WN * 64, 2) gemm_nt_s3_pair_kernel(const GemmArgs p0, const GemmArgs p1, const int tiles0) {
    if ((int)blockIdx.x < tiles0) gemm_nt_s3_body<T, BM, BN, WM, WN, TRANS>(p0, (int)blockIdx.x, tiles0);
    else gemm_nt_s3_body<T, BM, BN, WM, WN, TRANS>(p1, (int)blockIdx.x - tiles0, (int)gridDim.x - tiles0);
}

// ---------------------------------------------------------------------------------------------------
// "p8" pipeline: ONE 256 x 256 tile per CU, 8 wavefronts (2 per SIMD) in two groups of four that run half a
// window apart ("ping-pong"): while one group issues the 32 MFMAs of a window, the other issues the LDS
// fragment reads and the LDS-DMA pieces of its own window, so every SIMD always has one wave on the matrix
// pipe and one on the memory pipes.  K is consumed in 128-byte slabs (64 bf16/f16, 32 f32); LDS holds two
// slab buffers of four 16-KiB HALF-TILES each (A rows 0-127, A rows 128-255, B rows 0-127, B rows 128-255;
// 128-byte rows, 16-byte chunk c of row r at chunk c ^ (r & 7)).  A wave owns the 2 x 2 output quadrants
// (64 x 32 each) at (hm * 128 + g * 64, hn * 128 + wn * 32): quadrant operands A0/A1 come from the two A
// half-tiles and B0/B1 from the two B half-tiles, so three half-tiles are dead after the first window of a slab
// and are refilled for slab t + 2 while slab t is still being multiplied.  One slab = two windows:
//
//   window  LDS reads (this wave)   LDS-DMA issued (2 x 1 KiB per wave and half-tile)   wait before the barrier   MFMAs (32)
//   X       B0 (4), B1 (4), A0 (8)  A-hi of slab t+1                                    vmcnt(8) lgkmcnt(0)       A0 x B0, A0 x B1
//   Y       A1 (8)                  B-lo, A-lo, B-hi of slab t+2                        vmcnt(8) lgkmcnt(0)       A1 x B1, A1 x B0
//
// Every window is  [reads + DMA issue + waits] s_barrier [MFMAs] s_barrier ; the second group executes one extra
// barrier up front (and the first group one at the end), which is what puts the groups half a window apart.
// (Four windows of 16 MFMAs per slab, as first built, spent ~50 of every ~306 cycles on the barrier pair.)
// Hazards, with the groups staggered by one barrier (epochs between consecutive barriers: group 0 loads in
// epoch 2p and multiplies in 2p+1, group 1 loads in 2p+1 and multiplies in 2p+2):
//   * RAW: a half-tile is read in the window AFTER the one whose counted vmcnt retired its DMAs, i.e. at least
//     one epoch after BOTH groups' waits: Y's vmcnt(8) leaves A-hi(t+1) and the three half-tiles of slab t+2 in
//     flight, so B-lo / A-lo / B-hi of slab t+1 have landed before the next X reads them; X's vmcnt(8) leaves
//     those same three of slab t+1 (t+2 in steady state: the DMAs issued one slab ago) and the A-hi just issued,
//     so A-hi of slab t has landed before Y reads it;
//   * WAR: a half-tile is refilled one window after its last ds_read, and every read window ends with lgkmcnt(0)
//     before its barrier: B-lo / A-lo / B-hi are last read in X (epochs e, e+1 for the two groups) and refilled in
//     Y (epochs e+2, e+3); A-hi is last read in Y and refilled in the next X.
// The last two slabs are peeled (MODE 1 / 2) because their DMA counts differ.
template <typename T, int NI>
struct P8 {
    // NI x 2 blocks x 2 K-substeps = 4 NI MFMAs (NI = 4: the 16 of a quadrant pair)
    static __device__ __forceinline__ void mma16(f32x4 (&acc)[2 * NI][4], const u32x4 (&a)[NI][2], const u32x4 (&b)[2][2],
                                                 const int i0, const int j0) {
        asm volatile("s_setprio 1");
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    Mma<T>::pinned(b[j][s], a[i][s], acc[i0 + i][j0 + j]);     // operands swapped: a lane owns 4 consecutive columns
                }
        asm volatile("s_setprio 0");
    }
};

#define P8_BARRIER()                                  \
    do {                                              \
        asm volatile("s_barrier" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)

__device__ __forceinline__ void glds_piece_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// Fast epilogue of the p8 kernel for 2-byte row-major outputs on whole tiles: bias [+ folded LayerNorm]
// [+ GELU] [+ per-column scale/shift] [+ residual], the tile staged through 64 KiB of LDS in two 128-row halves so that the other 64 KiB
// slab buffer can already receive the next tile's first slab.  Column parameters are fetched as 16-byte
// vectors up front, the element math is branch-free and written on float pairs (v_pk_* issue).
template <typename T, bool GELU, bool LN, bool AFF>
__device__ __forceinline__ void p8_store_tile(const GemmArgs& p, f32x4 (&acc)[8][4], char* stg, const int m0, const int n0,
                                              const int tid, unsigned long long (&prof)[8]) {
#ifdef MLPK_P8_PROF   // fine-grained epilogue stamps cost registers: tuning builds only (-DMLPK_P8_PROF)
    const bool stamp = (p.dbg & 8) != 0;
    unsigned long long tprev = stamp ? __builtin_readcyclecounter() : 0;
#define P8_PROF(k)                                                           \
    if (stamp) {                                                             \
        const unsigned long long n__ = __builtin_readcyclecounter();          \
        prof[k] += n__ - tprev;                                               \
        tprev = n__;                                                          \
    }
#else
#define P8_PROF(k)
#endif
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = wave >> 2, wn = wave & 3;
    const int frow = lane & 15;
    const int fg = lane >> 4;
    T* __restrict__ C = reinterpret_cast<T*>(p.C);
    const T* R = reinterpret_cast<const T*>(p.R);
    const bool has_res = p.res_mode != MLPK_RES_NONE;
    const bool res_add = p.res_mode == MLPK_RES_ADD;
    f32x4 bz[4], lc[4], cs[4], ch[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + 4 * fg;
        bz[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (LN) lc[j] = *reinterpret_cast<const f32x4*>(p.ln_csum + n);
        if (AFF) {
            cs[j] = p.cscale ? *reinterpret_cast<const f32x4*>(p.cscale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
            ch[j] = p.cshift ? *reinterpret_cast<const f32x4*>(p.cshift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int c16 = tid & 31;
    const int rsub = tid >> 5;
    const int gn = n0 + c16 * 8;
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {
        u32x4 rr[8];
        if (has_res) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                rr[q] = *reinterpret_cast<const u32x4*>(R + (size_t)(m0 + hm * 128 + q * 16 + rsub) * p.ldr + gn);
        }
        float lmu[4], lrs[4];
        if (LN) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int m = m0 + hm * 128 + grp * 64 + i4 * 16 + frow;
                lmu[i4] = p.ln_mean[m / p.ln_group];
                lrs[i4] = p.ln_rstd[m / p.ln_group];
            }
        }
        // phase 1: element math on the fp32 accumulator, round, ds_write_b64 into the swizzled staging tile
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int sr = grp * 64 + i4 * 16 + frow;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                // two accumulator blocks = four float pairs at a time: their GELU chains are interleaved step by step
                // (gelu_pk_n<4>); one chain at a time leaves the wave waiting on its own previous instruction
                f32x2 v[4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jp * 2 + jj;
                    const f32x4 a = acc[hm * 4 + i4][j];
                    f32x2 lo = {a.x, a.y}, hi = {a.z, a.w};
                    const f32x2 blo = {bz[j].x, bz[j].y}, bhi = {bz[j].z, bz[j].w};
                    if (LN) {
                        const f32x2 nm = {-lmu[i4], -lmu[i4]}, rs = {lrs[i4], lrs[i4]};
                        lo = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, f32x2{lc[j].x, lc[j].y}, lo), rs, blo);
                        hi = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, f32x2{lc[j].z, lc[j].w}, hi), rs, bhi);
                    } else {
                        lo = lo + blo;
                        hi = hi + bhi;
                    }
                    v[2 * jj] = lo;
                    v[2 * jj + 1] = hi;
                }
                if (GELU) gelu_pk_n<T, 4>(v);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jp * 2 + jj;
                    const int nl = (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + 4 * fg;
                    f32x2 lo = v[2 * jj], hi = v[2 * jj + 1];
                    if (AFF) {
                        lo = __builtin_elementwise_fma(lo, f32x2{cs[j].x, cs[j].y}, f32x2{ch[j].x, ch[j].y});
                        hi = __builtin_elementwise_fma(hi, f32x2{cs[j].z, cs[j].w}, f32x2{ch[j].z, ch[j].w});
                    }
                    T e[4] = {from_f32<T>(lo.x), from_f32<T>(lo.y), from_f32<T>(hi.x), from_f32<T>(hi.y)};
                    u32x2 pk;
                    __builtin_memcpy(&pk, e, 8);
                    *reinterpret_cast<u32x2*>(stg + sr * 512 + ((((nl >> 3) ^ (sr & 15)) << 4) | ((nl & 4) << 1))) = pk;
                }
            }
        }
        P8_PROF(hm * 4 + 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        P8_PROF(hm * 4 + 1);
        // phase 2: 16-byte chunks LDS -> (residual) -> global, whole 512-byte rows per 32 lanes
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = q * 16 + rsub;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(stg + row * 512 + ((c16 ^ (row & 15)) << 4));
            u32x4 outv = raw;
            if (has_res) {
                T a8[8], r8[8];
                __builtin_memcpy(a8, &raw, 16);
                __builtin_memcpy(r8, &rr[q], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = to_f32(a8[e]), y = to_f32(r8[e]);
                    a8[e] = from_f32<T>(res_add ? x + y : x * y);
                }
                __builtin_memcpy(&outv, a8, 16);
            }
            if (!(p.dbg & 2) || outv.x == 0x12345678u) {
                u32x4* dst = reinterpret_cast<u32x4*>(C + (size_t)(m0 + hm * 128 + row) * p.ldc + gn);
                if (p.dbg & 32) __builtin_nontemporal_store(outv, dst);
                else *dst = outv;
            }
        }
        P8_PROF(hm * 4 + 2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        P8_PROF(hm * 4 + 3);
    }
#undef P8_PROF
}

// Direct epilogue of the p8 kernel (2-byte row-major outputs, whole tiles): NO LDS round trip and NO barrier.
// The accumulator gives a lane 4 consecutive columns (8 bytes) of one row per 16 x 16 block; the two blocks of a
// quadrant row that sit side by side (32 columns) are exchanged between lane rows with v_permlane16_swap_b32 (gfx950:
// odd 16-lane rows of the first operand <-> even rows of the second), after which every lane holds 8 consecutive
// columns = one 16-byte store, and a store instruction covers 16 rows x 64 contiguous bytes.  Each wave runs its own
// stream  [bias / folded LN / GELU / affine on four float pairs -> round -> 2 swaps -> (residual) -> store]  sixteen
// times, with nothing to wait for but its own residual loads: the waves of a workgroup drift apart, so one wave's VALU
// work overlaps another's store issue (the LDS-staged form alternated all-VALU and all-store phases between barriers:
// 20-23k cycles per fc1 tile of which ~10k GELU and ~8k store issue, strictly one after the other).
// What the direct epilogue needs before it can touch its first accumulator -- the LayerNorm statistics of the tile's rows and
// the bias / folded-LayerNorm vectors of its columns, 4 x 256 floats -- is brought into LDS behind the two slab buffers by
// p8_par_prefetch: TWO 256-byte LDS-DMA pieces per wave, issued a whole tile ahead (for the first tile before its first
// operand piece, for every later tile during the previous tile's epilogue, in front of that tile's slab-0 pieces), into one of
// two 4-KiB buffers that alternate from tile to tile.  Being OLDER than every operand piece of their tile they are retired
// by the loop-top wait with no count of their own, after a whole epilogue in flight.  Fetched with ordinary loads at the top
// of the epilogue these few scattered 64-byte reads were fully exposed (~2k cycles, once per half tile, behind the other
// CUs' store bursts: the folded LayerNorm cost fc1 22 us = 8 %, almost all of it waiting); issued in the second-to-last K
// slab the same wait merely moved into the last slab's vmcnt(0).  (Registers instead of LDS do not fit: 192 live + 48.)
// LDS image: [0, 1 KiB) ln_mean of logical tile rows 0..255, [1, 2) ln_rstd, [2, 3) bias of tile columns 0..255, [3, 4) ln_csum.
// 0: the statistics epilogue loads its residual chunks by untracked asm and waits for each by count (round 6 experiment: the
// compiler's own waits cannot count the stores issued since, so from the fifth chunk on every step waits for the stores of four steps
// earlier to be acknowledged).  Measured NEUTRAL on one box in alternation (profiles/r06_p8_counted_epilogue_ab.txt: Mixer-B/16 7.331 /
// 7.298 / 7.335 ms against 7.333 / 7.311 / 7.327; ResMLP-24, gMLP-S within 0.2 %): the 20 k cycles of a tile's epilogue are its ~1800
// VALU instructions per wave, not the write latency.  The tracked form stays the default (nothing to keep in step with the compiler).
#ifndef P8_RES_TRACKED
#define P8_RES_TRACKED 1
#endif
#define P8_PAR_OFF (2 * 4 * 128 * 128)
#define P8_LDS_BYTES (P8_PAR_OFF + 2 * 4096)

__device__ __forceinline__ void glds_dword_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <int BM>
__device__ __forceinline__ void p8_par_prefetch(const GemmArgs& p, const int m0, const int n0, const int wave, const int lane,
                                                const unsigned lds_par) {
    const int seg = wave & 3;                                  // 64 of the 256 entries
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_par + (unsigned)(seg * 256 + (wave >> 2) * 2048));
    // absent vectors (no LayerNorm / no bias) are "fetched" from the A operand: every wave issues exactly two pieces, which
    // is what the counted waits assume; the epilogue never reads what it does not use
    const char* dummy = reinterpret_cast<const char*>(p.A);
    if (wave < 4) {
        int r = seg * 64 + lane;
        r = r < BM ? r : BM - 1;                               // short tiles: stay inside this tile's rows
        // (the divisor is hidden from the compiler: it hoisted the reciprocal of a loop-invariant divisor out of the persistent loop into
        // vector registers that the 256-row kernel then spilled -- and the reload, a scratch load, waited for every memory operation in
        // flight, i.e. for the two pieces just issued, once per tile; no LayerNorm: no division)
        unsigned vo = (unsigned)r * 4u;
        if (p.ln_mean) {
            int g = p.ln_group;
            asm volatile("" : "+s"(g));
            vo = g == 1 ? (unsigned)(m0 + r) * 4u : (unsigned)((m0 + r) / g) * 4u;
        }
        glds_dword_s(vo, p.ln_mean ? reinterpret_cast<const char*>(p.ln_mean) : dummy, dst);
        glds_dword_s(vo, p.ln_mean ? reinterpret_cast<const char*>(p.ln_rstd) : dummy, dst + 1024);
    } else {
        const unsigned vo = (unsigned)(seg * 64 + lane) * 4u;
        glds_dword_s(vo, p.bias ? reinterpret_cast<const char*>(p.bias + n0) : dummy, dst);
        glds_dword_s(vo, p.ln_mean ? reinterpret_cast<const char*>(p.ln_csum + n0) : dummy, dst + 1024);
    }
}

// s_waitcnt vmcnt(n) for an n that is a constant once the epilogue's loops are unrolled
__device__ __forceinline__ void p8_wait_vm(const int n) {
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
        case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
        case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
        case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
        case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
        case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
        case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
        case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
        case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
        case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
        case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
        case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <typename T, bool GELU, bool LN, bool AFF, bool RES, int NI, bool STATS = false>
__device__ __forceinline__ void p8_store_direct(const GemmArgs& p, f32x4 (&acc)[2 * NI][4], const char* par_lds, const int m0, const int n0,
                                                const int tid) {
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = wave >> 2, wn = wave & 3;
    const int frow = lane & 15;
    const int fg = lane >> 4;
    T* __restrict__ C = reinterpret_cast<T*>(p.C);
    const T* R = reinterpret_cast<const T*>(p.R);
    constexpr bool has_res = RES;     // a template parameter: the 64 registers of the residual chunks must not weigh on the GELU classes
    const bool res_add = p.res_mode == MLPK_RES_ADD;
    f32x4 bz[4], lc[4], cs[4], ch[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + 4 * fg;              // column inside the tile
        bz[j] = p.bias ? *reinterpret_cast<const f32x4*>(par_lds + 2048 + nl * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (LN) lc[j] = *reinterpret_cast<const f32x4*>(par_lds + 3072 + nl * 4);
        if (AFF) {
            cs[j] = p.cscale ? *reinterpret_cast<const f32x4*>(p.cscale + n0 + nl) : f32x4{1.f, 1.f, 1.f, 1.f};
            ch[j] = p.cshift ? *reinterpret_cast<const f32x4*>(p.cshift + n0 + nl) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // after the swap: even lane rows hold columns 4*fg .. 4*fg+7 of the left block, odd ones 16 + 4*(fg-1) .. +7 (right block)
    const int ccol = (fg & 1) ? 16 + 4 * (fg - 1) : 4 * fg;
    // residual chunks of BOTH halves in flight before any math: the load latency (and the burst of every CU reading its
    // residual tile at once) is paid once per tile, the second half arrives under the first half's math
    u32x4 rr[RES ? 2 : 1][RES ? NI : 1][2];
    // COUNTED (round 6, the statistics epilogue = the residual-stream GEMMs): the residual chunks are loaded by asm the compiler does not
    // track and each is waited for by count.  Tracked, the compiler's wait in front of chunk s counted the 15 - s chunks behind it but
    // not the stores issued since (it cannot tell how many of them ran: they sit in conditional blocks) -- so from the fifth chunk on
    // every step waited for the STORES of four steps earlier to be acknowledged, and the last one for all of them: an epilogue paced by
    // the write latency (11 us per 256-row tile where its instructions take 5).  P8_RES_TRACKED=1 rebuilds the old form for A/B.
    constexpr bool COUNTED = RES && STATS && !P8_RES_TRACKED;
    if constexpr (has_res) {
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int i4 = 0; i4 < NI; ++i4)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn) {
                    const T* src = R + (size_t)(m0 + hm * (NI * 32) + grp * (NI * 16) + i4 * 16 + frow) * p.ldr + n0 + hn * 128 + wn * 32 + ccol;
                    if constexpr (COUNTED) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[hm][i4][hn]) : "v"(src) : "memory");
                    else rr[hm][i4][hn] = *reinterpret_cast<const u32x4*>(src);
                }
        // (tuning mode "no stores": the counts below assume the stores of the steps before)
        if (COUNTED && (p.dbg & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float lmu[2][NI], lrs[2][NI];
    if (LN) {
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int i4 = 0; i4 < NI; ++i4) {
                const int rl = hm * (NI * 32) + grp * (NI * 16) + i4 * 16 + frow;           // logical tile row
                lmu[hm][i4] = *reinterpret_cast<const float*>(par_lds + rl * 4);
                lrs[hm][i4] = *reinterpret_cast<const float*>(par_lds + 1024 + rl * 4);
            }
    }
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {
#pragma unroll
        for (int i4 = 0; i4 < NI; ++i4) {
            const size_t grow = (size_t)(m0 + hm * (NI * 32) + grp * (NI * 16) + i4 * 16 + frow);
#pragma unroll
            for (int hn = 0; hn < 2; ++hn) {
                f32x2 v[4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = hn * 2 + jj;
                    const f32x4 a = acc[hm * NI + i4][j];
                    f32x2 lo = {a.x, a.y}, hi = {a.z, a.w};
                    const f32x2 blo = {bz[j].x, bz[j].y}, bhi = {bz[j].z, bz[j].w};
                    if (LN) {
                        const f32x2 nm = {-lmu[hm][i4], -lmu[hm][i4]}, rs = {lrs[hm][i4], lrs[hm][i4]};
                        lo = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, f32x2{lc[j].x, lc[j].y}, lo), rs, blo);
                        hi = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, f32x2{lc[j].z, lc[j].w}, hi), rs, bhi);
                    } else {
                        lo = lo + blo;
                        hi = hi + bhi;
                    }
                    v[2 * jj] = lo;
                    v[2 * jj + 1] = hi;
                }
                if (GELU) gelu_pk_n<T, 4>(v);
                unsigned w[4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = hn * 2 + jj;
                    f32x2 lo = v[2 * jj], hi = v[2 * jj + 1];
                    if (AFF) {
                        lo = __builtin_elementwise_fma(lo, f32x2{cs[j].x, cs[j].y}, f32x2{ch[j].x, ch[j].y});
                        hi = __builtin_elementwise_fma(hi, f32x2{cs[j].z, cs[j].w}, f32x2{ch[j].z, ch[j].w});
                    }
                    T el[4] = {from_f32<T>(lo.x), from_f32<T>(lo.y), from_f32<T>(hi.x), from_f32<T>(hi.y)};
                    __builtin_memcpy(&w[2 * jj], el, 8);
                }
                // w[0..1] = this lane's 4 columns of the left block, w[2..3] = of the right block
                const auto s0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
                u32x4 outv = {s0[0], s1[0], s0[1], s1[1]};
                if constexpr (COUNTED) {
                    // chunk `step` has landed when at most the chunks behind it and the two stores (row, statistics pair) of every step
                    // before are still in flight; the empty statement hands the registers to the compiler only here
                    constexpr int NCH = 4 * NI;
                    const int step = (hm * NI + i4) * 2 + hn;
                    p8_wait_vm(NCH - 1 - step + 2 * step);
                    asm volatile("" : "+v"(rr[hm][i4][hn]));
                }
                if constexpr (has_res) {
                    T a8[8], r8[8];
                    __builtin_memcpy(a8, &outv, 16);
                    __builtin_memcpy(r8, &rr[hm][i4][hn], 16);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float x = to_f32(a8[k]), y = to_f32(r8[k]);
                        a8[k] = from_f32<T>(res_add ? x + y : x * y);
                    }
                    __builtin_memcpy(&outv, a8, 16);
                }
                if (!(p.dbg & 2) || outv.x == 0x12345678u)
                    *reinterpret_cast<u32x4*>(C + grow * p.ldc + n0 + hn * 128 + wn * 32 + ccol) = outv;
                if constexpr (STATS) {
                    // by-product statistics of the stored values: this wave's 32 columns of row `grow` sit in the four lanes
                    // frow, frow + 16, + 32, + 48, which hold the chunks 0, 2, 1, 3 of 8 columns (ccol).  Two swaps fold them in the
                    // library-wide order (c0 + c1) + (c2 + c3) (mlpk.h row_part): first the other half of the wave (lane ^ 32:
                    // chunks 0 + 1 and 2 + 3), then the neighbouring lane row (lane ^ 16)
                    float s1 = 0.f, s2 = 0.f;
                    chunk_sums<T>(outv, s1, s2);
                    float sv[2] = {s1, s2};
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const unsigned u = __builtin_bit_cast(unsigned, sv[k]);
                        const auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                        const float h = __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
                        const unsigned uh = __builtin_bit_cast(unsigned, h);
                        const auto r = __builtin_amdgcn_permlane16_swap(uh, uh, false, false);
                        sv[k] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
                    }
                    if (fg == 0)
                        *reinterpret_cast<f32x2*>(p.row_part + ((size_t)((n0 + hn * 128 + wn * 32) >> 5) * p.row_part_ld + grow) * 2) = f32x2{sv[0], sv[1]};
                }
            }
        }
    }
}

// Template parameters of the persistent kernel:
//   NI   16-row blocks per wave and half-tile: the tile is NI * 64 rows x 256 columns (NI = 4: the 256 x 256 tile described
//        above).  Shorter tiles use the SAME LDS image (128-row half-tiles, wave w stages rows 16w..16w+15 of each) and the
//        same windows with NI instead of 4 block rows per wave, so the two wave groups stay balanced for every NI; logical
//        tile row  h * 32NI + g * 16NI + i * 16 + r  sits at LDS row  h * 128 + g * 64 + i * 16 + r  (half h, group g,
//        block i < NI); the slots of blocks i >= NI are staged from the first rows of the tile and never multiplied.  A launch
//        covers the row panels of ONE height; the host mixes heights so that every launch is a whole number of rounds of one
//        tile per CU (p8_plan): Mixer-B fc2 = 588 tiles of 256 rows = 2.3 -> 3 rounds becomes 1 round of 256-row + 2 rounds of
//        192-row tiles (2.5 tile times), with the K order of every output element -- hence every result bit -- unchanged.
//   EPI  epilogue: 1 = direct (registers -> global, p8_store_direct), 0 = staged through LDS (p8_store_tile, NI = 4 only;
//        kept for A/B runs via reserved & 64), 2 = direct, bias + residual, with the by-product row statistics (GemmArgs::row_part).
// The kernel takes WHOLE tiles only (the host guarantees M % 64 == 0 with the panel heights adding up to M exactly,
// N % 256 == 0, K a multiple of the slab, 16-byte aligned rows and parameter vectors, no per-row scale): nothing is clamped
// or predicated, and the per-lane source offset of an LDS-DMA piece is ONE register per operand for the whole launch (lane ->
// row lane >> 3, chunk (lane & 7) ^ row of an 8-row piece); where a piece lies inside the tile is a scalar offset.
// Tile order: the launch's tiles are (panel, column tile) pairs.  The 8 XCDs are split into `cgroups` groups that each own a
// contiguous range of column tiles (so that a group's weight panels, cg * 256 * K elements, stay resident in the 4 MiB L2s of
// its XCDs instead of all N x K weights being re-fetched by every XCD every round), and a group's tiles are dealt to its XCDs
// in contiguous runs, column fastest: the column tiles of one row panel run on one XCD at the same time and share the panel.
// Register budget (256 per wave at two waves per SIMD): 32 NI accumulators + 24 NI operand fragments + ~20; hipcc must not
// spill inside the K loop -- a spill reload is a VMEM operation whose compiler-inserted wait drains the hand-counted LDS-DMA
// queue every slab; tools/isa_lint.py (run by tests/test_host_cpu.py) checks the generated loop for exactly that.
// (the body is a force-inlined function so that ONE launch can run the panels of two tile heights one after the other:
//  gemm_nt_p8_pair_kernel below)
template <typename T, int EPI, int NI>
__device__ __forceinline__ void p8_body(const GemmArgs& p, char* const smem) {
    static_assert(NI >= 1 && NI <= 4 && (EPI >= 1 || NI == 4), "tile height / epilogue combination");
    static_assert(sizeof(T) == 2, "16-bit operands");
    constexpr int BM = NI * 64, BN = 256;
    constexpr int EPC = 8;                                 // elements per 16-byte chunk
    constexpr int BK = 64;                                 // elements per 128-byte slab row
    constexpr int HALF_B = 128 * 128;                      // bytes of one half-tile
    constexpr int BUF_B = 4 * HALF_B;                      // A-lo, A-hi, B-lo, B-hi
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;

    // ---- work list: XCD xcd = blockIdx & 7 belongs to column group xcd / X and is member xj of it; it walks the tiles
    // u = xj * Q + l, l = blockIdx >> 3, + gridDim >> 3, ... of its group's U = panels * cg tiles ----
    const int tiles_n = p.N / BN;
    const int X = 8 / p.cgroups;
    const int cg = tiles_n / p.cgroups;
    const int U = p.panels * cg;
    const int Q = (U + X - 1) / X;
    const int xcd = (int)blockIdx.x & 7;
    const int cgrp = xcd / X;
    const int xj = xcd - cgrp * X;
    const int lstep = (int)gridDim.x >> 3;
    int lend = U - xj * Q;
    lend = __builtin_amdgcn_readfirstlane(lend < Q ? lend : Q);       // (integer divisions run on the VALU: pin the results to SGPRs)
    const int u0 = __builtin_amdgcn_readfirstlane(xj * Q);
    const int cg_s = __builtin_amdgcn_readfirstlane(cg);
    const int ncol0 = __builtin_amdgcn_readfirstlane(cgrp * cg);
    int l = (int)blockIdx.x >> 3;
    if (l >= lend) return;

    // ---- staging geometry: wave w fills rows 16w .. 16w+15 of every half-tile, two 1-KiB pieces of 8 rows ----
    const unsigned laneA = (unsigned)((lane >> 3) * p.lda + (((lane & 7) ^ (lane >> 3)) * EPC)) * 2u;
    const unsigned laneB = (unsigned)((lane >> 3) * p.ldb + (((lane & 7) ^ (lane >> 3)) * EPC)) * 2u;
    unsigned offA[2][2], offB[2][2];                       // scalar byte offsets of this wave's pieces inside a tile's panels
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // LDS row h * 128 + wave * 16 + q * 8 + r  =  half h, group wave >> 2, block wave & 3
            const int rowA = (wave & 3) < NI ? h * (NI * 32) + (wave >> 2) * (NI * 16) + (wave & 3) * 16 + q * 8 : q * 8;
            offA[h][q] = __builtin_amdgcn_readfirstlane((unsigned)(rowA * p.lda) * 2u);
            offB[h][q] = __builtin_amdgcn_readfirstlane((unsigned)((h * 128 + wave * 16 + q * 8) * p.ldb) * 2u);
        }
    // wave-uniform by construction, but a 64-bit product computed on the VALU lands in VGPRs and the "s" constraint of
    // the LDS-DMA asm then fails to assemble: pin both halves to SGPRs
    auto uniform64 = [](const size_t x) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
        return ((size_t)hi << 32) | lo;
    };
    int m0 = 0, n0 = 0;
    const char* tileA = nullptr;
    const char* tileB = nullptr;
    auto setup = [&](const int li) {
        const int u = u0 + li;
        const int panel = __builtin_amdgcn_readfirstlane(u / cg_s);
        m0 = p.m_base + (p.rev ? p.panels - 1 - panel : panel) * BM;
        n0 = (ncol0 + (u - panel * cg_s)) * BN;
        tileA = reinterpret_cast<const char*>(p.A) + uniform64((size_t)m0 * p.lda * 2u);
        tileB = reinterpret_cast<const char*>(p.B) + uniform64((size_t)n0 * p.ldb * 2u);
    };
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem + wave * 2048);
    const unsigned lds_par = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem + P8_PAR_OFF);
    // half ids: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi
    auto stage = [&](const int t, const int h) {
        // the slab counter may live in a VGPR under SGPR pressure: pin what the asm takes as scalars
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_w + (unsigned)((t & 1) * BUF_B + h * HALF_B));
        const unsigned tb = __builtin_amdgcn_readfirstlane((unsigned)(t * (BK * 2)));
        if (h < 2) {
            glds_piece_s(laneA, tileA + tb + offA[h][0], dst);
            glds_piece_s(laneA, tileA + tb + offA[h][1], dst + 1024);
        } else {
            glds_piece_s(laneB, tileB + tb + offB[h - 2][0], dst);
            glds_piece_s(laneB, tileB + tb + offB[h - 2][1], dst + 1024);
        }
    };

    const int nk = __builtin_amdgcn_readfirstlane(p.K / BK);   // >= 2 (host checked)

#ifdef MLPK_P8_PROF   // tuning builds only (tools/gemm_p8_timeline.py): per-workgroup sums of [first slabs | main loop | epilogue]
    unsigned long long tw = 0, tl = 0, te = 0, ts = 0;
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int ntiles = 0;
    const unsigned long long wall0 = wall_clock64(), cyc0 = __builtin_readcyclecounter();   // 100 MHz constant clock: start / end skew between CUs
#define P8_STAMP(acc_)                                                       \
    {                                                                        \
        const unsigned long long n__ = __builtin_readcyclecounter();          \
        acc_ += n__ - ts;                                                     \
        ts = n__;                                                             \
    }
#else
    unsigned long long prof[8];
#define P8_STAMP(acc_)
#endif

    // ---- first tile: slab 0 complete before the loop, three half-tiles of slab 1 in flight ----
    setup(l);
    int par_sel = 0;                                        // parameter buffer of the CURRENT tile
    if constexpr (EPI >= 1) p8_par_prefetch<BM>(p, m0, n0, wave, lane, lds_par);
    stage(0, 2); stage(0, 0); stage(0, 3); stage(0, 1);
    stage(1, 2); stage(1, 0); stage(1, 3);

    for (;;) {
#ifdef MLPK_P8_PROF
        ts = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        P8_BARRIER();
        if (grp == 1) P8_BARRIER();
        P8_STAMP(tw);

        // fragment read offsets, recomputed per tile from an opaque copy of the lane id: as launch-wide constants they
        // would be live across the register-hungry epilogue, and hipcc then spills them and reloads them INSIDE the K loop
        int kl = lane;
        asm volatile("" : "+v"(kl));
        const int frow = kl & 15;
        const int c0 = (((kl >> 4) ^ (frow & 7)) << 4);        // chunk of k-substep 0; substep 1 is c0 ^ 64
        const int a_rd = (grp * 64 + frow) * 128 + c0;          // + half * HALF_B + i * 2048
        const int b_rd = 2 * HALF_B + (wn * 32 + frow) * 128 + c0;

        f32x4 acc[2 * NI][4];
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 a0[NI][2], a1[NI][2], b0[2][2], b1[2][2];
#define P8_READ_A(dst, h)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                               \
        dst[i][0] = *reinterpret_cast<const u32x4*>(rb + (a_rd + (h) * HALF_B + i * 2048));                         \
        dst[i][1] = *reinterpret_cast<const u32x4*>(rb + ((a_rd + (h) * HALF_B + i * 2048) ^ 64));                  \
    }
#define P8_READ_B(dst, h)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                \
        dst[j][0] = *reinterpret_cast<const u32x4*>(rb + (b_rd + (h) * HALF_B + j * 2048));                         \
        dst[j][1] = *reinterpret_cast<const u32x4*>(rb + ((b_rd + (h) * HALF_B + j * 2048) ^ 64));                  \
    }
#define P8_SLAB(MODE, t)                                                                                          \
    {                                                                                                             \
        const char* rb = smem + ((t) & 1) * BUF_B;                                                                 \
        /* window X: operands of A0 x B0, A0 x B1 */                                                              \
        P8_READ_B(b0, 0);                                                                                          \
        P8_READ_B(b1, 1);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        P8_READ_A(a0, 0);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if (MODE <= 1) {                                                                                           \
            stage((t) + 1, 1);                                                                                     \
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                            \
        } else {                                                                                                   \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
        }                                                                                                          \
        P8_BARRIER();                                                                                              \
        P8<T, NI>::mma16(acc, a0, b0, 0, 0);                                                                        \
        P8<T, NI>::mma16(acc, a0, b1, 0, 2);                                                                        \
        P8_BARRIER();                                                                                              \
        /* window Y: operand of A1 x B1, A1 x B0 */                                                               \
        P8_READ_A(a1, 1);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if (MODE == 0) {                                                                                           \
            stage((t) + 2, 2);                                                                                     \
            stage((t) + 2, 0);                                                                                     \
            stage((t) + 2, 3);                                                                                     \
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                            \
        } else if (MODE == 1) {                                                                                    \
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                            \
        } else {                                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        }                                                                                                          \
        P8_BARRIER();                                                                                              \
        P8<T, NI>::mma16(acc, a1, b1, NI, 2);                                                                       \
        P8<T, NI>::mma16(acc, a1, b0, NI, 0);                                                                       \
        P8_BARRIER();                                                                                              \
    }
#pragma unroll 1
        for (int t = 0; t < nk - 2; ++t) P8_SLAB(0, t);
        P8_SLAB(1, nk - 2);
        P8_SLAB(2, nk - 1);
#undef P8_SLAB
#undef P8_READ_A
#undef P8_READ_B
        // the asm MFMAs are invisible to the hazard recogniser: cover the XDL-write -> VALU-read window by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if (grp == 0) P8_BARRIER();
        // every wave is past all its LDS reads of this tile; the groups are aligned again
        P8_STAMP(tl);

        const int cm0 = m0, cn0 = n0;
        l += lstep;
        const bool more = l < lend;                         // workgroup-uniform
        // opaque copy of the thread id: keeps the epilogue's address arithmetic inside this iteration (hoisted
        // out of the persistent loop it would sit in VGPRs through the main loop and spill)
        int etid = tid;
        asm volatile("" : "+v"(etid));
        // next tile's slab 0 streams into buffer 0 while this tile leaves (EPI 0: through the other buffer)
        const char* par = smem + P8_PAR_OFF + par_sel * 4096;
        if (more) {
            setup(l);
            par_sel ^= 1;
            if constexpr (EPI >= 1) p8_par_prefetch<BM>(p, m0, n0, wave, etid & 63, lds_par + (unsigned)(par_sel * 4096));
            stage(0, 2); stage(0, 0); stage(0, 3); stage(0, 1);
        }
        const int cls = (p.act == MLPK_ACT_GELU ? 1 : 0) | (p.ln_mean ? 2 : 0) | ((p.cscale || p.cshift) ? 4 : 0);
        if constexpr (EPI == 2) {
            // bias + residual with the by-product row statistics (its own kernel: the register allocation of the other classes
            // stays what it was tuned to be)
            p8_store_direct<T, false, false, false, true, NI, true>(p, acc, par, cm0, cn0, etid);
        } else if constexpr (EPI == 1) {
            // (the host sends residual + GELU / residual + LayerNorm combinations to the staged kernel)
            if (p.res_mode != MLPK_RES_NONE) {
                if (cls & 4) p8_store_direct<T, false, false, true, true, NI>(p, acc, par, cm0, cn0, etid);
                else p8_store_direct<T, false, false, false, true, NI>(p, acc, par, cm0, cn0, etid);
            } else {
                switch (cls) {
                    case 0: p8_store_direct<T, false, false, false, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    case 1: p8_store_direct<T, true, false, false, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    case 2: p8_store_direct<T, false, true, false, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    case 3: p8_store_direct<T, true, true, false, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    case 4: p8_store_direct<T, false, false, true, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    case 5: p8_store_direct<T, true, false, true, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    case 6: p8_store_direct<T, false, true, true, false, NI>(p, acc, par, cm0, cn0, etid); break;
                    default: p8_store_direct<T, true, true, true, false, NI>(p, acc, par, cm0, cn0, etid); break;
                }
            }
        } else {
            char* stg = smem + BUF_B;
            switch (cls) {
                case 0: p8_store_tile<T, false, false, false>(p, acc, stg, cm0, cn0, etid, prof); break;
                case 1: p8_store_tile<T, true, false, false>(p, acc, stg, cm0, cn0, etid, prof); break;
                case 2: p8_store_tile<T, false, true, false>(p, acc, stg, cm0, cn0, etid, prof); break;
                case 3: p8_store_tile<T, true, true, false>(p, acc, stg, cm0, cn0, etid, prof); break;
                case 4: p8_store_tile<T, false, false, true>(p, acc, stg, cm0, cn0, etid, prof); break;
                case 5: p8_store_tile<T, true, false, true>(p, acc, stg, cm0, cn0, etid, prof); break;
                case 6: p8_store_tile<T, false, true, true>(p, acc, stg, cm0, cn0, etid, prof); break;
                default: p8_store_tile<T, true, true, true>(p, acc, stg, cm0, cn0, etid, prof); break;
            }
        }
        // A wait hipcc can SEE (the asm ones are opaque to it): every load of the epilogue has landed.  Without it the
        // compiler carries "a VMEM load into these registers may be pending" around the persistent loop into the K loop and
        // protects the first fragment reads of every slab with s_waitcnt vmcnt(0..4) -- which drains the LDS-DMA prefetch
        // queue it knows nothing about (tools/isa_lint.py).  Here it costs nothing extra: the loop-top vmcnt(6) waits
        // for the same stores anyway, and only slab 1's pieces are issued behind it.
        // (round 6: the statistics epilogue loads its residual by untracked asm and waits by count -- nothing for the compiler to carry,
        // and slab 1's pieces leave straight behind the last store instead of after its acknowledgement)
        if constexpr (!(EPI == 2 && !P8_RES_TRACKED)) __builtin_amdgcn_s_waitcnt(0x0F70);
        if (more) { stage(1, 2); stage(1, 0); stage(1, 3); }
#ifdef MLPK_P8_PROF
        P8_STAMP(te);
        ++ntiles;
#endif
        if (!more) break;
    }
#ifdef MLPK_P8_PROF
    if ((p.dbg & 8) && tid == 0) {
        // (a pair launch runs two bodies: the second one, m_base != 0, reports 16 slots further on)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.prof_buf) + (size_t)blockIdx.x * 64 + (p.m_base ? 16 : 0);
        o[0] = tw; o[1] = tl; o[2] = te; o[3] = (unsigned long long)ntiles;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[4 + k] = prof[k];
        o[12] = wall0; o[13] = wall_clock64(); o[14] = __builtin_readcyclecounter() - cyc0;
    }
#endif
#undef P8_STAMP
}

template <typename T, int EPI, int NI>
__global__ void __launch_bounds__(512, 1) gemm_nt_p8_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    p8_body<T, EPI, NI>(p, smem);
}

// Two tile heights in ONE launch: a workgroup walks its tiles of the first height, then those of the second (the same tiles in
// the same K order as two launches -- every result bit unchanged -- without the drain, launch gap and ramp between them).
template <typename T, int EPI, int NIA, int NIB>
__global__ void __launch_bounds__(512, 1) gemm_nt_p8_pair_kernel(const GemmArgs pa, const GemmArgs pb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    p8_body<T, EPI, NIA>(pa, smem);
    __syncthreads();                                       // the LDS image of the last tile is dead only when every wave is done with it
    p8_body<T, EPI, NIB>(pb, smem);
}

// ------------------------------- host-side dispatch -------------------------------
struct TileCfg { int bm, bn, wm, wn, glds; };
static const TileCfg kTiles[] = {
    {256, 256, 2, 4, 0},   // algo 1: 8 waves, 128 KiB LDS, 1 workgroup / CU, register staging
    {256, 128, 4, 2, 0},   // algo 2: 8 waves,  96 KiB
    {128, 256, 2, 4, 0},   // algo 3: 8 waves,  96 KiB
    {128, 128, 2, 2, 0},   // algo 4: 4 waves,  64 KiB, 2 workgroups / CU
    {64, 64, 2, 2, 0},     // algo 5: 4 waves,  32 KiB, small / ragged problems
    {256, 256, 2, 4, 1},   // algo 6..9: the same tiles with direct-to-LDS loads (K % half-slab == 0)
    {256, 128, 4, 2, 1},
    {128, 256, 2, 4, 1},
    {128, 128, 2, 2, 1},
    {64, 64, 2, 2, 1},     // algo 10
    {256, 128, 2, 2, 2},   // algo 11..13: "s3" pipeline (4 waves, 3 LDS stages of 64-byte rows, 2 workgroups / CU)
    {128, 128, 2, 2, 2},
    {128, 256, 2, 2, 2},
    {256, 256, 2, 4, 3},   // algo 14: "p8" ping-pong pipeline (8 waves, 128 KiB LDS, 1 workgroup / CU; K % slab == 0, K >= 2 slabs)
    {256, 128, 2, 2, 4},   // algo 15: "q4" generated kernels (mlpk_gemm_q4.hip): 4 waves = one per SIMD, 144 KiB LDS, epilogue of tile
                           //          T - 1 issued behind the MFMAs of tile T
    {8, 64, 4, 1, 5},      // algo 16: skinny fp32 kernel (mlpk_gemm_skinny.hip): no MFMA, the whole chip on a product of a few hundred MFLOP
};
static const int kNumTiles = (int)(sizeof(kTiles) / sizeof(kTiles[0]));

template <typename T, int BM, int BN, int WM, int WN, bool GLDS>
static int launch_cfg(const GemmArgs& a, bool trans, hipStream_t stream) {
    const int lds = 2 * (BM + BN) * 128;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipError_t e;
    if (trans) {
        auto k = gemm_nt_kernel<T, BM, BN, WM, WN, true, GLDS>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), lds, stream, a);
    } else {
        auto k = gemm_nt_kernel<T, BM, BN, WM, WN, false, GLDS>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), lds, stream, a);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_s3(const GemmArgs& a, bool trans, hipStream_t stream) {
    const int lds = 3 * (BM + BN) * 64;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipError_t e;
    if (trans) {
        auto k = gemm_nt_s3_kernel<T, BM, BN, WM, WN, true>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), lds, stream, a);
    } else {
        auto k = gemm_nt_s3_kernel<T, BM, BN, WM, WN, false>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), lds, stream, a);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_s3_pair(const GemmArgs& a0, const GemmArgs& a1, hipStream_t stream) {
    const int lds = 3 * (BM + BN) * 64;
    const int tiles0 = ((a0.M + BM - 1) / BM) * ((a0.N + BN - 1) / BN), tiles1 = ((a1.M + BM - 1) / BM) * ((a1.N + BN - 1) / BN);
    auto k = gemm_nt_s3_pair_kernel<T, BM, BN, WM, WN, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(tiles0 + tiles1), dim3(WM * WN * 64), lds, stream, a0, a1, tiles0);
    MLPK_LAUNCH_CHECK();
    return 0;
}

// one persistent workgroup per compute unit (a multiple of 8: whole XCDs)
static int p8_grid_cap() {
    static int cap = 0;
    if (!cap) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
        cap = cu & ~7;
    }
    return cap;
}

// Height plan of a persistent launch sequence.  A problem of tiles_m x tiles_n tiles of 256 rows leaves the last round of one
// tile per CU partly empty (Mixer-B fc2: 588 tiles on 256 CUs = 2.3 -> 3 rounds).  Instead, the M = 64 m rows are cut into
// panels of 256, 192, 128 or 64 rows (NI = 4 .. 1) that add up to M exactly, and every height gets its own launch of r_h whole
// rounds; cost model per round: w_h = (365 + 512 NI) / 2413 of a 256-row round (fixed per-slab cost of the barriers and the
// B operand + MFMA time), plus a fixed cost per extra launch.  Exhaustive search over (r4, r3, r2, r1).
struct P8Plan { int n; int ni[4]; int panels[4]; };
static const double* p8_weights() {
    // time of one round of NI * 64-row tiles relative to a 256-row round; MLPK_P8_W="w1,w2,w3" overrides (calibration runs)
    static double w[5] = {0.0, 0.37, 0.58, 0.79, 1.0};
    static bool init = false;
    if (!init) {
        init = true;
        if (const char* e = getenv("MLPK_P8_W")) {
            double a, b, c;
            if (sscanf(e, "%lf,%lf,%lf", &a, &b, &c) == 3) { w[1] = a; w[2] = b; w[3] = c; }
        }
    }
    return w;
}

// 0 = mixed tile heights (the shortest single launch), 1 = whole 256-row tiles only (the least total CU time): mlpk_gemm_set_plan
static int g_p8_plan_mode = 0;

static P8Plan p8_plan(int M, int tiles_n, int nk, int G, bool mixed, double* cost_out = nullptr) {
    const double* w = p8_weights();
    // whole tiles where that still fills a round of CUs (a launch with fewer tiles than CUs keeps its mixed heights: gMLP's N = 256 product) and the
    // tiles are long (K >= 1024: a short tile is mostly fixed cost, and more, lower tiles then pack better -- ResMLP's K = 384 fc1 loses 7 %)
    if (g_p8_plan_mode == 1 && (long long)(M / 256) * tiles_n >= G && nk >= 16) mixed = false;
    const int m = M / 64;
    const double launch_pen = 6000.0 / (nk * 2413.0 + 8000.0);
    auto rounds = [&](long long panels) { return (int)((panels * tiles_n + G - 1) / G); };
    // default: 256-row panels, and one short panel for the M % 256 rows
    P8Plan best;
    best.n = 0;
    if (m / 4) { best.ni[best.n] = 4; best.panels[best.n] = m / 4; ++best.n; }
    if (m % 4) { best.ni[best.n] = m % 4; best.panels[best.n] = 1; ++best.n; }
    double best_cost = rounds(m / 4) + (m % 4 ? w[m % 4] * rounds(1) : 0.0) + (best.n - 1) * launch_pen;
    if (cost_out) *cost_out = best_cost;
    if (!mixed) return best;
    static const int force_ni = getenv("MLPK_P8_FORCE_NI") ? atoi(getenv("MLPK_P8_FORCE_NI")) : 0;     // calibration runs: one height
    if (force_ni >= 1 && force_ni <= 4 && m % force_ni == 0) {
        best.n = 1; best.ni[0] = force_ni; best.panels[0] = m / force_ni;
        return best;
    }
    const int rmax = rounds(m / 4) + 1;
    auto cap = [&](int r) { return (int)((long long)r * G / tiles_n); };      // panels that fit r rounds
    for (int r4 = 0; r4 <= rmax; ++r4)
        for (int r3 = 0; r3 * 3 <= (rmax + 1 - r4) * 4; ++r3)
            for (int r2 = 0; r2 <= 2; ++r2)
                for (int r1 = 0; r1 <= 1; ++r1) {
                    const int r[5] = {0, r1, r2, r3, r4};
                    // greedy exact cover, tallest first
                    int left = m, take[5] = {0, 0, 0, 0, 0}, launches = 0;
                    double cost = 0.0;
                    for (int h = 4; h >= 1; --h) {
                        if (!r[h]) continue;
                        const int c = cap(r[h]);
                        take[h] = left / h < c ? left / h : c;
                        left -= take[h] * h;
                        if (take[h]) { cost += rounds(take[h]) * w[h]; ++launches; }
                    }
                    if (left != 0 || launches == 0) continue;
                    cost += (launches - 1) * launch_pen;
                    if (cost < best_cost - 1e-9) {
                        best_cost = cost;
                        best.n = 0;
                        for (int h = 4; h >= 1; --h)
                            if (take[h]) { best.ni[best.n] = h; best.panels[best.n] = take[h]; ++best.n; }
                    }
                }
    if (cost_out) *cost_out = best_cost;
    return best;
}

// column groups: the smallest power of two G | tiles_n, G <= 8, whose group of tiles_n / G weight panels fits ~2.5 MiB of
// an XCD's 4 MiB L2 -- when all the weights do not (else 1: no panel is ever re-fetched anyway)
static int p8_cgroups(int tiles_n, int K, int es) {
    const double panel = 256.0 * K * es;
    if (panel * tiles_n <= 3.0 * 1048576.0) return 1;
    for (int g = 2; g <= 8; g *= 2)
        if (tiles_n % g == 0 && panel * (tiles_n / g) <= 2.5 * 1048576.0) return g;
    for (int g = 8; g >= 2; g /= 2)
        if (tiles_n % g == 0) return g;
    return 1;
}

// what the persistent tile accepts (everything else goes to the other pipelines): 16-bit row-major output in whole tiles
static bool p8_eligible(const GemmArgs& a, int es, bool trans) {
    if (es != 2 || trans || a.rscale || a.vec_c != 2 || (a.res_mode != MLPK_RES_NONE && a.vec_r != 2)) return false;
    if (a.M % 64 || a.N % 256 || a.K % 64 || a.K < 128) return false;
    // residual + GELU / residual + LayerNorm go through the LDS-staged epilogue, which exists for 256-row tiles only
    if (a.res_mode != MLPK_RES_NONE && (a.act == MLPK_ACT_GELU || a.ln_mean) && a.M % 256) return false;
    const uintptr_t par = reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.ln_csum) | reinterpret_cast<uintptr_t>(a.cscale) |
                          reinterpret_cast<uintptr_t>(a.cshift);
    return (par & 15) == 0;
}

template <typename T> static int launch_p8(const GemmArgs& a0, bool trans, hipStream_t stream) {
    if constexpr (sizeof(T) != 2) {
        return MLPK_EDTYPE;                  // the persistent tile is a 16-bit kernel (fp32 uses the exact-f32 MFMA tiles)
    } else {
        GemmArgs a = a0;
        if (!p8_eligible(a, 2, trans)) return MLPK_ESHAPE;
        const int lds = P8_LDS_BYTES;
        const int tiles_n = a.N / 256;
        const int cap = p8_grid_cap();
        a.cgroups = (a.dbg & 128) ? 1 : p8_cgroups(tiles_n, a.K, 2);           // reserved & 128: one column group (A/B runs)
        // reserved & 64: LDS-staged epilogue (A/B runs); it also serves the rare residual + GELU / residual + LayerNorm
        // combinations, which the direct epilogue does not instantiate
        const bool staged = (a.dbg & 64) != 0 || (a.res_mode != MLPK_RES_NONE && (a.act == MLPK_ACT_GELU || a.ln_mean));
        if (staged && a.M % 256) return MLPK_ESHAPE;                            // the staged epilogue is built for 256-row tiles only
        const P8Plan plan = p8_plan(a.M, tiles_n, a.K / 64, cap, !staged && !(a.dbg & 16));   // reserved & 16: 256-row tiles only
        int m_base = 0;
        auto grid_of = [&](const int panels) {
            const int X = 8 / a.cgroups;
            const int U = panels * (tiles_n / a.cgroups);
            const int Q = (U + X - 1) / X;
            return 8 * (Q < cap / 8 ? Q : cap / 8);
        };
        const char* pe = getenv("MLPK_P8_PAIR");
        const bool pair_on = !(pe && pe[0] == '0');                                                   // A/B aid
        for (int s = 0; s < plan.n; ++s) {
            const int ni = plan.ni[s];
            a.m_base = m_base;
            a.panels = plan.panels[s];
            m_base += a.panels * ni * 64;
            hipError_t e = hipSuccess;
            // the 256-row panels of a plan and the shorter ones that follow them (Mixer-B fc2: one round + two rounds of 192-row tiles;
            // gMLP proj1: nine rounds + a tail of 64-row tiles) go out as one launch
            const int nj = s + 1 < plan.n ? plan.ni[s + 1] : 0;
            if (pair_on && !staged && nj && ((ni == 4 && nj < 4) || (nj == 4 && ni < 4))) {
                GemmArgs b = a;
                b.m_base = m_base;
                b.panels = plan.panels[s + 1];
                m_base += b.panels * nj * 64;
                const GemmArgs& a4 = ni == 4 ? a : b;
                const GemmArgs& ax = ni == 4 ? b : a;
                const int nx = ni == 4 ? nj : ni;
                const int g4 = grid_of(a4.panels), gx = grid_of(ax.panels);
                const int gridp = g4 > gx ? g4 : gx;
#define P8_PAIR(EP, NX)                                                                                                 \
    {                                                                                                                   \
        auto k = gemm_nt_p8_pair_kernel<T, EP, 4, NX>;                                                                  \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);     \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL(k, dim3(gridp), dim3(512), lds, stream, a4, ax);                                             \
    }
                if (a.row_part) {
                    if (nx == 3) P8_PAIR(2, 3) else if (nx == 2) P8_PAIR(2, 2) else P8_PAIR(2, 1)
                } else {
                    if (nx == 3) P8_PAIR(1, 3) else if (nx == 2) P8_PAIR(1, 2) else P8_PAIR(1, 1)
                }
#undef P8_PAIR
                MLPK_LAUNCH_CHECK();
                ++s;
                continue;
            }
            const int grid = grid_of(a.panels);
#define P8_LAUNCH(EP, NIv)                                                                                              \
    {                                                                                                                   \
        auto k = gemm_nt_p8_kernel<T, EP, NIv>;                                                                         \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);     \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, stream, a);                                                   \
    }
            if (staged) P8_LAUNCH(0, 4)
            else if (a.row_part && ni == 4) P8_LAUNCH(2, 4)
            else if (a.row_part && ni == 3) P8_LAUNCH(2, 3)
            else if (a.row_part && ni == 2) P8_LAUNCH(2, 2)
            else if (a.row_part) P8_LAUNCH(2, 1)
            else if (ni == 4) P8_LAUNCH(1, 4)
            else if (ni == 3) P8_LAUNCH(1, 3)
            else if (ni == 2) P8_LAUNCH(1, 2)
            else P8_LAUNCH(1, 1)
#undef P8_LAUNCH
            MLPK_LAUNCH_CHECK();
        }
        return 0;
    }
}

// the call as the skinny fp32 kernel takes it (bias, GELU, row-major, nothing else)
static bool skinny_call_of(const GemmArgs& a, int dtype, bool trans, SkinnyCall& c) {
    if (dtype != MLPK_F32 || trans || a.rscale || a.cscale || a.cshift || a.ln_mean || a.res_mode != MLPK_RES_NONE || a.row_part) return false;
    c.M = a.M; c.N = a.N; c.K = a.K; c.lda = a.lda; c.ldb = a.ldb; c.ldc = a.ldc;
    c.A = reinterpret_cast<const float*>(a.A); c.B = reinterpret_cast<const float*>(a.B); c.C = reinterpret_cast<float*>(a.C);
    c.bias = a.bias; c.gelu = a.act == MLPK_ACT_GELU;
    return skinny_supported(c);
}

// the call as the generated q4 kernels take it; false when they do not implement it (other tiles do)
static bool q4_call_of(const GemmArgs& a, int dtype, bool trans, Q4Call& c) {
    if (dtype == MLPK_F32 || trans || a.rscale || a.cscale || a.cshift) return false;
    if (a.res_mode != MLPK_RES_NONE && a.res_mode != MLPK_RES_ADD) return false;
    if (a.ln_mean && a.ln_group != 1) return false;
    c.dtype = dtype;
    c.M = a.M; c.N = a.N; c.K = a.K;
    c.lda = a.lda; c.ldb = a.ldb; c.ldc = a.ldc; c.ldr = a.ldr;
    c.A = a.A; c.B = a.B; c.C = a.C; c.R = a.res_mode != MLPK_RES_NONE ? a.R : nullptr;
    c.bias = a.bias; c.ln_mean = a.ln_mean; c.ln_rstd = a.ln_rstd; c.ln_csum = a.ln_csum;
    c.gelu = a.act == MLPK_ACT_GELU; c.ln = a.ln_mean != nullptr; c.res = a.res_mode != MLPK_RES_NONE;
    c.row_part = a.row_part; c.row_part_ld = a.row_part_ld;
    c.one_group = (a.dbg & 128) != 0;
    c.dbg = (a.dbg & (31 | 64)) | (a.dbg_q4 << 8);
    c.prof = (a.dbg & 32) ? a.prof_buf : nullptr;      // reserved & 32: cycle counts into desc.workspace
    return q4_supported(c);
}

template <typename T> static int launch_algo(int algo, const GemmArgs& a, bool trans, hipStream_t s, void* ws, long long ws_bytes) {
    switch (algo) {
        case 1: return launch_cfg<T, 256, 256, 2, 4, false>(a, trans, s);
        case 2: return launch_cfg<T, 256, 128, 4, 2, false>(a, trans, s);
        case 3: return launch_cfg<T, 128, 256, 2, 4, false>(a, trans, s);
        case 4: return launch_cfg<T, 128, 128, 2, 2, false>(a, trans, s);
        case 5: return launch_cfg<T, 64, 64, 2, 2, false>(a, trans, s);
        case 6: return launch_cfg<T, 256, 256, 2, 4, true>(a, trans, s);
        case 7: return launch_cfg<T, 256, 128, 4, 2, true>(a, trans, s);
        case 8: return launch_cfg<T, 128, 256, 2, 4, true>(a, trans, s);
        case 9: return launch_cfg<T, 128, 128, 2, 2, true>(a, trans, s);
        case 10: return launch_cfg<T, 64, 64, 2, 2, true>(a, trans, s);
        case 11: return launch_s3<T, 256, 128, 2, 2>(a, trans, s);
        case 12: return launch_s3<T, 128, 128, 2, 2>(a, trans, s);
        case 13: return launch_s3<T, 128, 256, 2, 2>(a, trans, s);
        case 14: return launch_p8<T>(a, trans, s);
        case 15: {
            Q4Call c;
            if (!q4_call_of(a, dtype_of<T>::value, trans, c)) return MLPK_ESHAPE;
            return q4_launch(c, s);
        }
        case 16: {
            SkinnyCall c;
            if (!skinny_call_of(a, dtype_of<T>::value, trans, c)) return MLPK_ESHAPE;
            return skinny_launch(c, s);
        }
        default: return MLPK_EMODE;
    }
}

// Pick the tile that minimises (padded MFMA work / tile efficiency) x (a soft tail penalty for grids that
// do not fill the CUs many times over).  Efficiencies follow the MI355X sweeps in profiles/: with K a
// multiple of 32 elements the 2-workgroup-per-CU "s3" tiles win on every shape of the path (their
// epilogue overlaps the other workgroup's main loop); ragged K falls back to the register-staged tiles.
static int auto_algo(int M, int N, int K, int epc, bool glds_ok, bool p8_ok, bool stats = false) {
    double best = 1e300;
    int best_algo = 4;
    for (int i = 0; i < kNumTiles; ++i) {
        const TileCfg& t = kTiles[i];
        const int area = t.bm * t.bn;
        if (stats && t.bn < 128) continue;        // the by-product row statistics reduce over whole 16-lane rows = 128 columns
        if (t.glds >= 4) continue;                // the generated tile / the skinny kernel are chosen in gemm_prepare, not by this cost model
        if (t.glds == 3) {
            // persistent ping-pong tile: whole launch rounds of one tile per CU, tile heights mixed to fill them (p8_plan);
            // its per-tile fixed cost (first slabs + epilogue, not overlapped with another workgroup) weighs more the
            // shorter K is.  Calibrated on the MI355X sweeps and model benches: 0.76-0.82 x the best s3 time at
            // K = 768 / 3072, break-even around K = 384, behind below that.
            if (!p8_ok) continue;
            const double cap = (double)p8_grid_cap();
            double rounds = 0.0;
            p8_plan(M, N / 256, K / 64, (int)cap, true, &rounds);
            const double kb = (double)K / epc * 16.0;          // bytes of K per row
            // (round 2: with the direct epilogue and the parameter prefetch the break-even against the s3 tiles moved from
            //  K = 384 down to K = 192: measured 0.164 vs 0.192 ms at K = 256, 0.078 vs 0.078 at 192, 0.054 vs 0.049 at 128,
            //  tools/gemm_sweep.py on the synthetic short-K shapes)
            const double eff = 1.6 * kb / (kb + 260.0);
            const double cost = rounds * cap * area / eff;
            if (cost < best) { best = cost; best_algo = i + 1; }
            continue;
        }
        if (glds_ok) {
            if (!(t.glds == 2 || (t.glds == 1 && area <= 64 * 64))) continue;
        } else if (t.glds != 0) {
            continue;
        }
        const double tiles = (double)((M + t.bm - 1) / t.bm) * (double)((N + t.bn - 1) / t.bn);
        const double slots = area >= 256 * 256 ? 256.0 : area >= 128 * 128 ? 512.0 : 1024.0;
        // 128 x 256 (wave tile 64 x 128) measured 2-8 % ahead of 256 x 128 on the channel-MLP shapes
        // ... for K > 512; at short K the 256 x 128 tile is the better of the two (K = 128 .. 320: 5-20 % ahead)
        const double wide = K > 512 * (epc / 8.0) ? 1.0 : 0.92, tall = K > 512 * (epc / 8.0) ? 0.97 : 1.0;
        const double eff = area >= 256 * 128 ? (t.bn > t.bm ? wide : tall) : area >= 128 * 128 ? 0.9 : 0.45;
        // (tiles + 256): a soft tail for grids that do not fill the CUs many times over; the same constant for every
        // tile size, so that a problem of a few tiles (the M = batch GEMMs of SplitAttention) is costed by the time of
        // ONE tile, area / eff, and gets the small tile (128 x 128 fp32 tiles took 38 us for 75 MFLOP)
        (void)slots;
        const double cost = area / eff * (tiles + 256.0);
        if (cost < best) { best = cost; best_algo = i + 1; }
    }
    return best_algo;
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_gemm_algo_count(void) { return kNumTiles; }

extern "C" int mlpk_gemm_algo_info(int algo, int* bm, int* bn, int* threads, int* lds_bytes) {
    if (algo < 1 || algo > kNumTiles) return MLPK_EMODE;
    const TileCfg& t = kTiles[algo - 1];
    if (bm) *bm = t.bm;
    if (bn) *bn = t.bn;
    if (threads) *threads = t.wm * t.wn * 64;
    if (lds_bytes) *lds_bytes = t.glds == 5 ? 6144 : t.glds == 4 ? Q4_LDS_BYTES : t.glds == 3 ? P8_LDS_BYTES : t.glds == 2 ? 3 * (t.bm + t.bn) * 64 : 2 * (t.bm + t.bn) * 128;
    return 0;
}

extern "C" long long mlpk_gemm_workspace_bytes(void) {
    // no kernel needs scratch any more (the split-K hand-over of a partial last round was replaced by mixed tile heights);
    // the descriptor's workspace fields stay in the ABI and are ignored
    return 0;
}

// validation + tile choice shared by mlpk_gemm_nt and mlpk_gemm_row_parts
static int gemm_prepare(const mlpk_gemm_desc* d, GemmArgs& a, int& algo, bool& trans) {
    if (!d) return MLPK_ENULL;
    if (!d->A || !d->B || !d->C) return MLPK_ENULL;
    if (d->dtype != MLPK_F32 && d->dtype != MLPK_F16 && d->dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return MLPK_ESHAPE;
    const int es = d->dtype == MLPK_F32 ? 4 : 2;
    const int epc = 16 / es;
    if (d->K % epc || d->lda % epc || d->ldb % epc || d->lda < d->K || d->ldb < d->K) return MLPK_ESHAPE;
    if (((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15) || ((uintptr_t)d->workspace & 15)) return MLPK_EALIGN;
    if (d->workspace_bytes < 0 || (d->workspace_bytes > 0 && !d->workspace)) return MLPK_ENULL;
    if (d->act != MLPK_ACT_NONE && d->act != MLPK_ACT_GELU) return MLPK_EMODE;
    if (d->res_mode < MLPK_RES_NONE || d->res_mode > MLPK_RES_MUL) return MLPK_EMODE;
    if (d->res_mode != MLPK_RES_NONE && !d->R) return MLPK_ENULL;
    if (d->out_mode != MLPK_OUT_ROWMAJOR && d->out_mode != MLPK_OUT_TOKEN_T) return MLPK_EMODE;
    if (d->rscale && d->rperiod <= 0) return MLPK_ESHAPE;
    trans = d->out_mode == MLPK_OUT_TOKEN_T;
    if ((d->ln_mean != nullptr) != (d->ln_rstd != nullptr) || (d->ln_mean != nullptr) != (d->ln_csum != nullptr)) return MLPK_ENULL;
    if (d->ln_mean && trans) return MLPK_EMODE;     // the fold is per GEMM row; token-transposed GEMMs normalise along K
    if (trans) {
        if (d->t_rows <= 0 || d->t_tokens <= 0 || d->t_rows % 4 || d->M % d->t_rows || d->N > d->t_tokens) return MLPK_ESHAPE;
        if (d->ldc < d->t_rows) return MLPK_ESHAPE;
    } else if (d->ldc < d->N) {
        return MLPK_ESHAPE;
    }
    a.A = d->A; a.B = d->B; a.C = d->C; a.R = d->R;
    a.bias = d->bias; a.cscale = d->cscale; a.cshift = d->cshift; a.rscale = d->rscale;
    a.ln_mean = d->ln_mean; a.ln_rstd = d->ln_rstd; a.ln_csum = d->ln_csum;
    a.ln_group = d->ln_group > 0 ? d->ln_group : 1;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr;
    a.rperiod = d->rperiod > 0 ? d->rperiod : 1;
    a.act = d->act; a.res_mode = d->res_mode;
    a.t_rows = d->t_rows; a.t_tokens = d->t_tokens;
    a.dbg = d->reserved & 0xff;
    a.dbg_delay = (d->reserved >> 8) & 0xff;
    a.dbg_q4 = (d->reserved >> 16) & 0xff;
    a.m_base = 0; a.panels = 0; a.cgroups = 1; a.prof_buf = d->workspace;
    {
        // tuning (round 4): the persistent tile walks its row panels backwards when its A operand is larger than the Infinity Cache and was
        // just written front to back by the previous kernel (channel-MLP fc2 reading the 308 MB hidden): the rows written last are read first
        static const int rev_mode = getenv("MLPK_P8_REVERSE") ? atoi(getenv("MLPK_P8_REVERSE")) : 0;
        a.rev = rev_mode == 2 || (rev_mode == 1 && (long long)d->M * d->K * es > (200ll << 20)) ? 1 : 0;
    }
    a.row_part = d->row_part; a.row_part_ld = d->row_part_ld;
    const int vb = 4 * es;   // bytes of a 4-element vector
    a.vec_c = (d->ldc % 4 == 0) && (((uintptr_t)d->C % vb) == 0);
    a.vec_r = d->R ? ((d->ldr % 4 == 0) && (((uintptr_t)d->R % vb) == 0)) : 0;
    if (a.vec_c && d->ldc % 8 == 0 && ((uintptr_t)d->C % 16) == 0) a.vec_c = 2;     // 8-element (16-byte) vectors
    if (a.vec_r && d->ldr % 8 == 0 && ((uintptr_t)d->R % 16) == 0) a.vec_r = 2;
    const bool stats = d->row_part != nullptr;
    if (stats) {
        // by-product statistics come out of the 16-byte-chunk store passes of the 16-bit row-major epilogues only
        if (es != 2 || trans) return MLPK_EMODE;
        if (d->N % 8 || a.vec_c != 2 || (d->res_mode != MLPK_RES_NONE && a.vec_r != 2)) return MLPK_ESHAPE;
        if ((uintptr_t)d->row_part & 7) return MLPK_EALIGN;
    }
    algo = d->algo;
    const bool glds_ok = d->K % (4 * epc) == 0;      // K a multiple of half a 128-byte slab
    // the persistent tile is auto-selected where its overlapped epilogue applies (16-bit row-major, no row scale)
    static const bool no_p8 = getenv("MLPK_GEMM_NO_P8") != nullptr;      // tuning hook: A/B the tile choice in one run
    bool p8_ok = !no_p8 && p8_eligible(a, es, trans);
    // ... and with statistics, where the direct epilogue instantiates them: bias + residual (the GEMMs that produce a residual stream)
    const bool p8_stats_ok = d->res_mode != MLPK_RES_NONE && d->act == MLPK_ACT_NONE && !d->ln_mean && !d->cscale && !d->cshift && !(a.dbg & 64);
    if (stats && !p8_stats_ok) p8_ok = false;
    if (algo == 0) {
        // round 3: the generated one-wave-per-SIMD tile where it is ahead (MLPK_GEMM_Q4=0 switches it off for A/B runs; 2 = wherever
        // it applies).  Rule from the per-shape A/B of every GEMM call of the bs=256 models (profiles/r03_gemm_shapes_q4_ab_v1.txt,
        // r03_q4_probe_v3.txt):
        //  * where the persistent 256 x 256 tile cannot run (N % 256 != 0: the N = 384 / 1152 / 640 / 128 shapes of ViP, S2-MLP, Swin-,
        //    Hire-, CycleMLP, ResMLP) it replaces the s3 tile: 0.69-0.94 of its time on every measured shape;
        //  * against the persistent tile it is ahead on the epilogue-heavy fc1 shapes (GELU, K = 768 / 1024, N >= 3072: 0.93-0.95) and
        //    at N = 768 with short K (0.80-0.87: three column tiles fill the persistent tile's rounds badly), behind elsewhere (short K
        //    with many column tiles 1.05-1.10; K >= 3072 1.10: its LDS-DMA runs two 48-KiB slabs ahead, HBM-latency bound on long K);
        //  * its pipeline spends one extra (draining) block per workgroup: only grids of several tiles per CU.
        static const int q4_mode = getenv("MLPK_GEMM_Q4") ? atoi(getenv("MLPK_GEMM_Q4")) : 1;
        Q4Call qc;
        // (... and only when a generated kernel exists for the call's class AND its tuning bits: reserved bits meant for the persistent
        //  tile, or a forced MLPK_Q4_NKF, must fall through to the other tiles instead of failing the call)
        if (q4_mode && q4_call_of(a, d->dtype, trans, qc) && q4_variant_name(qc)) {
            const long long tiles = (long long)(d->M / 256) * (d->N / 128);
            bool take = false;
            if (!p8_ok) take = tiles >= 512;
            else take = (d->act == MLPK_ACT_GELU && d->K >= 640 && d->K <= 1280 && d->N >= 3072 && tiles >= 2048) ||
                        (d->N == 768 && d->K <= 512 && tiles >= 1024);
            if (q4_mode >= 2 || take) algo = 15;
        }
    }
    if (algo == 0 && d->dtype == MLPK_F32) {
        // round 4: the small fp32 products of the SplitAttention / re-weighting MLPs (M = batch rows, K <= 1536) CAN run on the skinny
        // kernel (algo 16) -- built because the MFMA tiles run them on <= 72 workgroups with a serial K loop (15-24 us each).  Measured
        // (profiles/r04_skinny_ab.txt): S2-MLPv2 8.76 -> 8.74 ms, CycleMLP-B1 unchanged (these products are launch-latency bound either
        // way), ViP-S7 30.1 -> 30.6 ms WORSE: its chain runs on a side stream beside a persistent GEMM, and a grid of 1024 small
        // workgroups sits on every CU the persistent kernel's workgroups (all of a CU's LDS and registers each) are waiting for, where
        // the 64 x 64 tile's 24-128 workgroups hold up only that many.  So it is opt-in (MLPK_GEMM_SKINNY=1), not the default.  The rule
        // uses N and K only: which kernel computes a row must not depend on the batch
        static const bool sk_on = getenv("MLPK_GEMM_SKINNY") && atoi(getenv("MLPK_GEMM_SKINNY")) == 1;
        SkinnyCall sc;
        if (sk_on && (long long)d->N * d->K <= (1ll << 21) && d->K <= 2048 && d->M <= 16384 && skinny_call_of(a, d->dtype, trans, sc)) algo = 16;
    }
    if (algo == 0) algo = auto_algo(d->M, d->N, d->K, epc, glds_ok, p8_ok, stats);
    if (algo < 1 || algo > kNumTiles) return MLPK_EMODE;
    if (kTiles[algo - 1].glds && !glds_ok) return MLPK_ESHAPE;
    if (kTiles[algo - 1].glds == 4) {
        Q4Call qc;
        if (!q4_call_of(a, d->dtype, trans, qc)) return MLPK_ESHAPE;
    }
    if (stats) {
        if (kTiles[algo - 1].bn < 128 || (kTiles[algo - 1].glds == 3 && !p8_stats_ok)) return MLPK_EMODE;
        if (d->row_part_ld < d->M) return MLPK_ESHAPE;
    }
    return 0;
}

extern "C" int mlpk_gemm_kernel_name(const mlpk_gemm_desc* d, char* buf, int len) {
    if (!d || !buf || len <= 0) return MLPK_ENULL;
    GemmArgs a;
    int algo = 0;
    bool trans = false;
    const int rc = gemm_prepare(d, a, algo, trans);
    if (rc) return rc;
    const TileCfg& t = kTiles[algo - 1];
    if (t.glds == 4) {
        Q4Call qc;
        const char* nm = q4_call_of(a, d->dtype, trans, qc) ? q4_variant_name(qc) : nullptr;
        snprintf(buf, (size_t)len, "%s", nm ? nm : "q4 (no variant)");
    } else if (t.glds == 3) {
        // the persistent tile: which template runs follows from the plan of tile heights (launch_p8)
        const bool staged = (a.dbg & 64) != 0 || (a.res_mode != MLPK_RES_NONE && (a.act == MLPK_ACT_GELU || a.ln_mean));
        const P8Plan plan = p8_plan(a.M, a.N / 256, a.K / 64, p8_grid_cap(), !staged && !(a.dbg & 16));
        const char* pe = getenv("MLPK_P8_PAIR");
        const bool pair_on = !(pe && pe[0] == '0');
        const bool pair = pair_on && !staged && plan.n >= 2 && ((plan.ni[0] == 4 && plan.ni[1] < 4) || (plan.ni[1] == 4 && plan.ni[0] < 4));
        char hs[32] = "";
        for (int s = 0, o = 0; s < plan.n && o < 28; ++s) o += snprintf(hs + o, sizeof(hs) - (size_t)o, "%s%d", s ? "+" : "", plan.ni[s] * 64);
        snprintf(buf, (size_t)len, "%s<EPI=%d> rows %s", pair ? "gemm_nt_p8_pair_kernel" : "gemm_nt_p8_kernel", staged ? 0 : a.row_part ? 2 : 1, hs);
    } else if (t.glds == 5) {
        snprintf(buf, (size_t)len, "gemm_skinny_f32_kernel");
    } else {
        snprintf(buf, (size_t)len, "%s %dx%d", t.glds == 2 ? "gemm_nt_s3_kernel" : t.glds == 1 ? "gemm_nt_glds_kernel" : "gemm_nt_kernel", t.bm, t.bn);
    }
    return 0;
}

extern "C" int mlpk_gemm_row_parts(const mlpk_gemm_desc* d, int* nparts) {
    if (!d || !nparts) return MLPK_ENULL;
    mlpk_gemm_desc q = *d;
    alignas(8) float dummy_pair[2];
    if (!q.row_part) q.row_part = dummy_pair;     // a question about the tile choice, nothing is written
    q.row_part_ld = 0x7fffffff;
    GemmArgs a;
    int algo = 0;
    bool trans = false;
    const int rc = gemm_prepare(&q, a, algo, trans);
    if (rc) return rc;
    // (round 4: every tile writes planes of 32 columns, reduced in one order -- the answer no longer depends on the tile)
    *nparts = (d->N + 31) / 32;
    return 0;
}

extern "C" int mlpk_gemm_nt(const mlpk_gemm_desc* d, void* stream) {
    GemmArgs a;
    int algo = 0;
    bool trans = false;
    const int rc = gemm_prepare(d, a, algo, trans);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (d->dtype) {
        case MLPK_F32: return launch_algo<float>(algo, a, trans, s, d->workspace, d->workspace_bytes);
        case MLPK_F16: return launch_algo<f16_t>(algo, a, trans, s, d->workspace, d->workspace_bytes);
        default: return launch_algo<bf16_t>(algo, a, trans, s, d->workspace, d->workspace_bytes);
    }
}

extern "C" int mlpk_conv_gemm_nhwc_supported(int dtype, int Cin, int kh, int kw, int stride, int pad);
extern "C" int mlpk_conv_gemm_nhwc(const mlpk_gemm_desc* d, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, void* stream);

extern "C" int mlpk_conv_gemm_nhwc_supported(int dtype, int Cin, int kh, int kw, int stride, int pad) {
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && Cin >= 32 && Cin % 32 == 0 && kh >= 1 && kw >= 1 && kh * kw <= 49 && stride >= 1 && pad >= 0 && pad < kh &&
           pad < kw && (long long)kh * kw * (Cin / 32) < 65536;
}

template <typename T>
static int launch_s3_conv(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = 128, BN = 128;
    const int lds = 3 * (BM + BN) * 64;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    auto k = gemm_nt_s3_conv_kernel<T, BM, BN, 2, 2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, stream, a);
    MLPK_LAUNCH_CHECK();
    return 0;
}

// d: the product's descriptor with A = the channel-last (B, H, W, Cin) input (dense pixels), M = B Ho Wo, K = kh kw Cin, lda = K (unused), row-major output
extern "C" int mlpk_conv_gemm_nhwc(const mlpk_gemm_desc* d, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, void* stream) {
    GemmArgs a;
    int algo = 0;
    bool trans = false;
    const int rc = gemm_prepare(d, a, algo, trans);
    if (rc) return rc;
    if (B <= 0 || H <= 0 || W <= 0 || !mlpk_conv_gemm_nhwc_supported(d->dtype, Cin, kh, kw, stride, pad) || trans) return MLPK_ESHAPE;
    if (H + 2 * pad < kh || W + 2 * pad < kw) return MLPK_ESHAPE;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if ((long long)B * Ho * Wo != d->M || (long long)kh * kw * Cin != d->K || d->N % 8) return MLPK_ESHAPE;
    a.cv_H = H; a.cv_W = W; a.cv_Cin = Cin; a.cv_kw = kw; a.cv_cpk = Cin / 32; a.cv_stride = stride; a.cv_pad = pad; a.cv_Ho = Ho; a.cv_Wo = Wo;
    a.cv_inv_cpk = (65536 + a.cv_cpk - 1) / a.cv_cpk;      // kt / cpk for kt < 65536 / cpk by one multiply
    a.cv_inv_kw = (65536 + kw - 1) / kw;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == MLPK_F16 ? launch_s3_conv<f16_t>(a, s) : launch_s3_conv<bf16_t>(a, s);
}

extern "C" int mlpk_gemm_nt_pair(const mlpk_gemm_desc* d0, const mlpk_gemm_desc* d1, void* stream);

// two products in one launch where the dispatch gives both the same "s3" tile (row-major outputs, 16-bit); anything else: one after the other
extern "C" int mlpk_gemm_nt_pair(const mlpk_gemm_desc* d0, const mlpk_gemm_desc* d1, void* stream) {
    GemmArgs a0, a1;
    int algo0 = 0, algo1 = 0;
    bool t0 = false, t1 = false;
    int rc = gemm_prepare(d0, a0, algo0, t0);
    if (rc) return rc;
    rc = gemm_prepare(d1, a1, algo1, t1);
    if (rc) return rc;
    static const bool off = getenv("MLPK_GEMM_PAIR") && atoi(getenv("MLPK_GEMM_PAIR")) == 0;      // A/B aid: two launches
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!off && algo0 == algo1 && algo0 >= 11 && algo0 <= 13 && !t0 && !t1 && d0->dtype == d1->dtype && d0->dtype != MLPK_F32 &&
        !(a0.dbg | a1.dbg)) {
#define MLPK_PAIR(TT)                                                                  \
    switch (algo0) {                                                                   \
        case 11: return launch_s3_pair<TT, 256, 128, 2, 2>(a0, a1, s);                 \
        case 12: return launch_s3_pair<TT, 128, 128, 2, 2>(a0, a1, s);                 \
        default: return launch_s3_pair<TT, 128, 256, 2, 2>(a0, a1, s);                 \
    }
        if (d0->dtype == MLPK_F16) { MLPK_PAIR(f16_t) } else { MLPK_PAIR(bf16_t) }
#undef MLPK_PAIR
    }
    rc = mlpk_gemm_nt(d0, stream);
    if (rc) return rc;
    return mlpk_gemm_nt(d1, stream);
}

extern "C" int mlpk_gemm_set_plan(int mode) {
    if (mode != 0 && mode != 1) return MLPK_EMODE;
    g_p8_plan_mode = mode;
    return 0;
}
