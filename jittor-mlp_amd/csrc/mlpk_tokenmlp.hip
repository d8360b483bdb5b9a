// Fused token-mixing MLP (MLP-Mixer, reference mlp_mixer.py:16-27 with dense = Conv1d(k=1), :34,:37):
//
//   x[b,s,c] += sum_t W2[s,t] * gelu( sum_s' W1[t,s'] * xn[b,s',c] + b1[t] ) + b2[s]
//
// computed on the token-transposed LayerNorm output xt[(b,c), s'] so both contractions are K-contiguous NT
// products.  Unfused, the hidden (B*C x 4S, 308 MB at Mixer-B/16 B=256) makes a round trip through HBM and
// the pair is memory/epilogue-bound; here it never leaves the CU.
//
// Persistent workgroup (one per CU) of 8 waves walking 128-row tiles of (b,c) rows x ALL S output tokens.
// A slice is 32 rows; its two waves sit on the SAME SIMD (waves w and w + 4 share one) and are specialised:
//   * the MATRIX wave (w < 4) holds the slice's 32 x S_pad block of xt in registers as MFMA operands and runs
//       fc1(t): acc1 = X . W1[t]^T (28 MFMAs) -> fp32 to LDS,  fc2(t-2) for token blocks 0..7 (16 MFMAs);
//   * the ACTIVATION wave (w >= 4) runs gelu(t-1): acc1 + b1 -> exact-erf GELU -> 16 bit -> LDS in A-operand
//       order (the VALU work, which costs about what the matrix wave's MFMAs cost), and fc2(t-2) for token
//       blocks 8..13 (10-12 MFMAs);
//   so on every SIMD the matrix pipe and the VALU are busy at the same time, from different waves.
//   The hidden axis is walked in groups of 32 (= one K-slab of the second product), one group per iteration,
//   three groups in flight (fc1 / gelu / fc2).
//   * W1 groups and W2 slabs stream through two 3-stage LDS rings (64-byte rows, XOR-swizzled) filled by
//     global_load_lds two iterations ahead; every matrix wave issues exactly 7 one-KiB pieces per iteration, so
//     a constant `s_waitcnt vmcnt(7)` + one s_barrier per iteration is the whole synchronisation (it also hands
//     acc1 and H between the waves).  The weight streams are the same for every tile, so across tiles the
//     rings simply keep turning (period G + 2 iterations).
//   * Tile epilogue: acc2 + b2 staged through LDS as fp32 in 8 passes of 32 tokens; the reader side adds the
//     residual (one rounding) with whole 256-byte token rows per 16 lanes, for both its load and its store.  The
//     residual tile is requested during the last (drain) iteration and awaited once; the next tile's X operands are
//     requested before the passes, which then issue nothing but stores.
// LDS: 48 (W1 ring) + 48 (W2 ring) + 16 (H, double-buffered) + 32 (acc1 exchange / epilogue staging) + 4 (b1) + 0.9 (b2) = 149 KiB.
#include "mlpk_common.h"
#include "mlpk_tokenmlp_t4.h"
#include <cstdlib>

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
    float* stats;       // optional: per (128-channel tile, token row) partial (sum, sum of squares) of the values written to x, planar
    unsigned long long* dbg;   // tuning aid: per-workgroup [loop, epilogue] shader-clock sums (NULL in normal use)
};

// one 1-KiB LDS-DMA piece: uniform base + 32-bit per-lane offset, LDS destination = M0 base + lane * 16
__device__ __forceinline__ void tm_glds(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// (builtin MFMAs here, not volatile asm as in the p8 GEMM: as operands of an opaque asm the accumulators were shuffled
//  through v_mov copies around the GELU code, and such a copy reads an MFMA result inside the XDL-write -> VALU-read
//  hazard window that the hazard recogniser cannot see -- measured as wrong results.  There is only one barrier per
//  iteration to stay clear of, so the compiler's own placement is acceptable.)
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

constexpr int TM_BM = 128;          // rows per tile
constexpr int TM_KMAX = 7;          // K-slabs of the first product kept in registers (S_pad <= 224)
constexpr int TM_NB0 = 8;           // token blocks (of 16) of the matrix wave
constexpr int TM_NB1 = 6;           // ... of the activation wave: up to 224 output tokens together
constexpr int TM_STAGE = 16384;     // W1 group: 8 planes x [32 rows x 64 B];  W2 slab: [256 rows x 64 B]
constexpr int TM_NST = 3;           // stages per ring
constexpr int TM_R1 = 0;
constexpr int TM_R2 = TM_NST * TM_STAGE;
constexpr int TM_HS = 2 * TM_NST * TM_STAGE;           // 4 slices x 2 buffers x [32 rows x 64 B]
constexpr int TM_AX = TM_HS + 4 * 2 * 2048;            // 4 slices x 2 buffers x 4 KiB of fp32 acc1; epilogue: 2 x 8 KiB
constexpr int TM_B1 = TM_AX + 4 * 2 * 4096;
constexpr int TM_B1_FLOATS = 1024;                     // hidden (padded) <= 1024
constexpr int TM_B2 = TM_B1 + TM_B1_FLOATS * 4;    // b2, zero-padded to 16 * (NB0 + NB1) tokens: a global load inside an epilogue
                                                   // pass made hipcc wait vmcnt(0) there -- on the previous pass's stores and on the next tile's X
constexpr int TM_LDS = TM_B2 + 16 * (TM_NB0 + TM_NB1) * 4;

#define TM_BARRIER() asm volatile("s_barrier" ::: "memory")
// (lgkmcnt(0): this wave's LDS writes of the previous iteration must have reached LDS before the barrier)
#define TM_ITER_SYNC() do { asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); TM_BARRIER(); } while (0)

// iteration flavours: RAMP (t < 2, run-time predicates), STEADY (branch-free), DRAIN (t >= G, t >= 2: no fc1 code at all,
// which leaves the registers of its operands to the residual tile that is requested during the last iteration)
#ifndef TM_ABL
#define TM_ABL 0      // tuning aid (tools/build_variant.sh): 1 no in-loop LDS-DMA, 2 identity GELU, 4 no fc1 MFMAs, 8 no fc2 MFMAs -- wrong results, timing only
#endif
enum { TM_RAMP = 0, TM_STEADY = 1, TM_DRAIN = 2, TM_LAST = 3 };   // LAST = the DRAIN iteration that requests the residual tile
template <int M> struct ModeC { static constexpr int value = M; };
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int I> struct IntC { static constexpr int value = I; };

// A wave issues at most one instruction per ~4 cycles, so the loop bodies below are written for instruction
// count: the steady-state iterations are branch-free instantiations (STEADY), the two ramp-up and two drain
// iterations of a tile go through the same code with run-time predicates.
template <typename T>
__global__ void __launch_bounds__(512, 1) token_mlp_kernel(const TokenMlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, slice = wave & 3;
    const T* __restrict__ xt = reinterpret_cast<const T*>(p.xt);
    const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
    const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
    T* __restrict__ x = reinterpret_cast<T*>(p.x);
    const int G = p.G;
    const int ks1 = p.ks1;
    const int ntiles = (p.M + TM_BM - 1) / TM_BM;
    const int nblk = (p.S + 15) >> 4;                      // token blocks that exist

    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    char* const hs = smem + TM_HS + slice * 4096;          // the slice's two 2-KiB H buffers
    char* const axs = smem + TM_AX + slice * 8192;         // the slice's two 4-KiB acc1 buffers (lane-linear)
    float* const b1s = reinterpret_cast<float*>(smem + TM_B1);
    float* const b2s = reinterpret_cast<float*>(smem + TM_B2);

    // (All workgroups walk the hidden groups in the same order: rotating it per CU to spread the L2 accesses was
    //  measured neutral, and a fixed order keeps every row's result independent of the batch it is computed in.)
    // ---- piece geometry: 16 rows x 64 B, lane -> (row lrow, physical chunk lane & 3), source-side swizzle ----
    // W1 group = 8 planes (K-slabs) of [32 rows x 64 B]; piece pc = plane*2 + half;  W2 slab = [256 rows x 64 B];
    // piece pc = rows pc*16..+16 (clamped to S-1).  Only planes 0-6 and rows 0-223 can be referenced (S_pad <= 224), so
    // an iteration needs 14 + 14 pieces.  They are issued by the four MATRIX waves, 7 each, between their MFMAs (the
    // LDS-DMA issue rate, ~100+ cycles per piece when every wave issues at once, must stay off the activation waves'
    // critical path).  Matrix wave w issues combined pieces q = 7w .. 7w+6 (q < 14: W1, else W2).
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((lrow & 8) >> 2);
    unsigned poff[7];                                      // per-lane byte offset from the group's base
    unsigned pdst[7];
#pragma unroll
    for (int pi = 0; pi < 7; ++pi) {
        const int q = slice * 7 + pi;
        if (q < 14) {
            poff[pi] = (unsigned)(((q & 1) * 16 + lrow) * 256 + (q >> 1) * 32 + lchunk * 8) * (unsigned)sizeof(T);
            pdst[pi] = __builtin_amdgcn_readfirstlane(lds_base + TM_R1 + q * 1024);
        } else {
            int r2 = (q - 14) * 16 + lrow;
            r2 = r2 < p.S ? r2 : p.S - 1;
            poff[pi] = (unsigned)(r2 * p.ldw2 + lchunk * 8) * (unsigned)sizeof(T);
            pdst[pi] = __builtin_amdgcn_readfirstlane(lds_base + TM_R2 + (q - 14) * 1024);
        }
    }
    // The schedule is periodic with period G + 2 (tile-local iteration t: fc1 group t, fc2 group t - 2).  The pieces
    // an iteration consumes are issued two iterations earlier into ring stage (iteration mod 3).  Groups that do not
    // exist (fc1 at t >= G, fc2 at t < 2) are clamped duplicates: the same piece count every iteration.
    const T* pb1 = w1;                                     // group bases of the pieces being issued (uniform)
    const T* pb2 = w2;
    auto piece_bases = [&](const int tl) {                 // tl = tile-local iteration the pieces are for
        int t3 = tl;
        if (t3 >= G + 2) t3 -= G + 2;
        const int g1 = t3 < G ? t3 : G - 1;
        const int g2 = t3 >= 2 ? t3 - 2 : 0;
        pb1 = w1 + (size_t)g1 * (32 * 256);
        pb2 = w2 + g2 * 32;
    };
    auto issue = [&](const unsigned stoff, const int pi) {
        tm_glds(poff[pi], slice * 7 + pi < 14 ? pb1 : pb2, pdst[pi] + stoff);
    };
    auto issue_loop = [&](const unsigned stoff, const int pi) {
        if (!(TM_ABL & 1) || p.ldxt == 12345) issue(stoff, pi);
    };
    if (role == 0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            piece_bases(it);
#pragma unroll
            for (int pi = 0; pi < 7; ++pi) issue((unsigned)it * TM_STAGE, pi);
        }
    } else {
        __builtin_amdgcn_s_setprio(2);    // waves 4-7 are the younger half and would lose every VALU arbitration
    }

    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int co = (fg ^ ((frow & 8) >> 2)) << 4;     // fragment chunk offset inside a 64-byte row
    const int f_rd = frow * 64 + co;                  // + block*1024 (+ plane*2048 in a W1 group)

    for (int i = tid; i < G * 32; i += 512) b1s[i] = p.b1[i];
    if (tid < 16 * (TM_NB0 + TM_NB1)) b2s[tid] = tid < p.S ? p.b2[tid] : 0.f;
    __syncthreads();

    unsigned long long t_loop = 0, t_epi = 0, ts = 0;
    const bool stamp = p.dbg != nullptr;
    // Epilogue geometry.  A pass moves 32 token slots (0-15: the matrix wave's block j, 16-31: the activation wave's
    // block 8 + j) x 128 channels of fp32 (acc2 + b2) through LDS; reader thread = (slot tid >> 4, 8 channels tid & 15)
    // adds the residual in fp32, rounds ONCE and writes 16 bytes, i.e. whole 256-byte token rows per 16 lanes.
    const int rt = tid >> 4, rc = tid & 15;
    char* const stg = smem + TM_AX;
    const int wc4 = slice * 8 + fg;                        // writer's 16-byte chunk (4 channels) for i = 0; + 4 for i = 1
    // Statistics for the LayerNorm that follows (the channel-mixing PreNormResidual, mlp_mixer.py:38): the 16 lanes rc = 0..15 of a
    // token hold the tile's 128 channels of it, AFTER rounding -- exactly what that LayerNorm will read -- so a DPP row reduction
    // gives the tile's partial (sum, sum of squares) per token; mlpk_stats_finalize_planar turns the t_rows / 128 partials of a row into
    // mean / rstd.  One separate statistics pass over x per block (77 MB at Mixer-B/16, 256 images) disappears.
    // (PLANAR, one plane of B*S pairs per 128-channel tile: a workgroup's S pairs are one contiguous run; interleaved per token row
    //  every pair was a lone 8-byte masked write between pairs that other workgroups write later.)
    const size_t srows = (size_t)(p.M / p.t_rows) * p.S;             // plane stride in pairs (t_rows % 128 == 0, host checked)
    auto epilogue_reader = [&](const int j, const char* sb, const u32x4 res, const int rimg, const int rcc, const bool row_ok) {
        const int rn = (rt < 16 ? j : TM_NB0 + j) * 16 + (rt & 15);   // token this thread moves in pass j
        const bool live = (rt < 16 || j < TM_NB1) && rn < p.S && row_ok;
        float ssum = 0.f, ssq = 0.f;
        if (live) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sb + rt * 512 + (((2 * rc) ^ (rt & 15)) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + rt * 512 + (((2 * rc + 1) ^ (rt & 15)) << 4));
            T r8[8];
            __builtin_memcpy(r8, &res, 16);
            T e[8] = {from_f32<T>(v0.x + to_f32(r8[0])), from_f32<T>(v0.y + to_f32(r8[1])), from_f32<T>(v0.z + to_f32(r8[2])),
                      from_f32<T>(v0.w + to_f32(r8[3])), from_f32<T>(v1.x + to_f32(r8[4])), from_f32<T>(v1.y + to_f32(r8[5])),
                      from_f32<T>(v1.z + to_f32(r8[6])), from_f32<T>(v1.w + to_f32(r8[7]))};
            u32x4 o;
            __builtin_memcpy(&o, e, 16);
            *reinterpret_cast<u32x4*>(x + ((size_t)rimg * p.S + rn) * p.ldx + rcc) = o;
            if (p.stats) chunk_sums<T>(o, ssum, ssq);
        }
        if (p.stats) {                                                // (workgroup-uniform: every lane takes part in the DPP rows)
            ssum = row16_sum(ssum);
            ssq = row16_sum(ssq);
            if (live && rc == 0)
                *reinterpret_cast<f32x2*>(p.stats + ((size_t)(rcc >> 7) * srows + (size_t)rimg * p.S + rn) * 2) = f32x2{ssum, ssq};
        }
    };
    auto residual_load = [&](const int j, const int rimg, const int rcc, const bool row_ok) {
        const int rn = (rt < 16 ? j : TM_NB0 + j) * 16 + (rt & 15);
        u32x4 r = {0u, 0u, 0u, 0u};
        if ((rt < 16 || j < TM_NB1) && rn < p.S && row_ok) r = *reinterpret_cast<const u32x4*>(x + ((size_t)rimg * p.S + rn) * p.ldx + rcc);
        return r;
    };

    auto request_residual = [&](const int tile, u32x4 (&res)[TM_NB0], int& rimg, int& rcc, bool& row_ok) {
        const int mr = tile * TM_BM + rc * 8;
        rimg = mr / p.t_rows;
        rcc = mr - rimg * p.t_rows;
        row_ok = mr < p.M;
#pragma unroll
        for (int j = 0; j < TM_NB0; ++j) res[j] = residual_load(j, rimg, rcc, row_ok);
    };

    if (role == 0) {
        // =============================== matrix wave ===============================
        // X operands; K-slabs past ks1 repeat the last one: they meet the zero K-padding of W1 (planes >= ks1 are zero)
        u32x4 xa[2][TM_KMAX];
        auto load_x = [&](const int tile) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int gm = tile * TM_BM + slice * 32 + i * 16 + frow;
                gm = gm < p.M ? gm : p.M - 1;
#pragma unroll
                for (int kk = 0; kk < TM_KMAX; ++kk)
                    xa[i][kk] = *reinterpret_cast<const u32x4*>(xt + (size_t)gm * p.ldxt + (kk < ks1 ? kk : ks1 - 1) * 32 + fg * 8);
            }
        };
        load_x(blockIdx.x);
        unsigned st = 0;                                   // ring stage of the current iteration (global iteration mod 3)
        f32x4 acc2[2][TM_NB0];
        u32x4 res[TM_NB0];
        int rimg = 0, rcc = 0;
        bool row_ok = false;
        auto iter = [&](auto mode_c, const int t, const int tile) {
            constexpr int MODE = decltype(mode_c)::value;
            const bool fc1 = MODE == TM_STEADY || (MODE == TM_RAMP && t < G);
            const bool fc2 = MODE != TM_RAMP || t >= 2;
            TM_ITER_SYNC();
            if constexpr (MODE == TM_LAST) request_residual(tile, res, rimg, rcc, row_ok);      // after this iteration's vmcnt(7)
            const char* r1 = smem + TM_R1 + st * TM_STAGE;
            const char* r2 = smem + TM_R2 + st * TM_STAGE;
            const unsigned stoff2 = (st == 0 ? 2 : st - 1) * TM_STAGE;   // stage (st + 2) mod 3
            piece_bases(t + 2);
            // fc1(t) first (its operands requested up front, one LDS-DMA piece per four MFMAs), then the fc2 operand
            // requests go out while the last fc1 MFMAs drain, acc1 is handed over, and fc2(t-2) runs
            const char* hr = hs + (t & 1) * 2048;
            u32x4 af0, af1, bf[TM_NB0];
            f32x4 a1[2][2];
            if (fc2) {
                af0 = *reinterpret_cast<const u32x4*>(hr + f_rd);
                af1 = *reinterpret_cast<const u32x4*>(hr + 1024 + f_rd);
            }
            if (fc1) {
                // swapped operands -> lane = row frow, 4 consecutive hidden columns 4*fg + r
                u32x4 bw[TM_KMAX][2];
#pragma unroll
                for (int kk = 0; kk < TM_KMAX; ++kk) {
                    bw[kk][0] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd);
                    bw[kk][1] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd + 1024);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) a1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < TM_KMAX; ++kk) {
                    issue_loop(stoff2, kk);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if constexpr (TM_ABL & 4) {
                            a1[i][0] += __builtin_bit_cast(f32x4, bw[kk][0]) + __builtin_bit_cast(f32x4, xa[i][kk]);
                            continue;
                        }
                        a1[i][0] = Mma2<T>::run(bw[kk][0], xa[i][kk], a1[i][0]);
                        a1[i][1] = Mma2<T>::run(bw[kk][1], xa[i][kk], a1[i][1]);
                    }
                }
            } else {
#pragma unroll
                for (int pi = 0; pi < 7; ++pi) issue_loop(stoff2, pi);
            }
            if (fc2) {
#pragma unroll
                for (int j = 0; j < TM_NB0; ++j) bf[j] = *reinterpret_cast<const u32x4*>(r2 + j * 1024 + f_rd);
            }
            if (fc1) {
                char* const aw = axs + (t & 1) * 4096 + lane * 16;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(aw + (i * 2 + j) * 1024) = a1[i][j];
            }
            if (fc2) {
                // natural operands -> lane = token frow of block j, 4 consecutive rows 4*fg + r
#pragma unroll
                for (int j = 0; j < TM_NB0; ++j) {
                    if constexpr (TM_ABL & 8) {
                        acc2[0][j] += __builtin_bit_cast(f32x4, af0) + __builtin_bit_cast(f32x4, bf[j]);
                        continue;
                    }
                    acc2[0][j] = Mma2<T>::run(af0, bf[j], acc2[0][j]);
                    acc2[1][j] = Mma2<T>::run(af1, bf[j], acc2[1][j]);
                }
            }
            st = st == 2 ? 0 : st + 1;
        };
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            if (stamp) ts = __builtin_readcyclecounter();
            // The X operands must have LANDED before the iteration loop, and the compiler must know it (a builtin, not
            // asm): otherwise its own "s_waitcnt vmcnt(n)" for them sits in front of their first use in EVERY
            // iteration, where it also drains the LDS-DMA pieces issued one iteration earlier.
            __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TM_NB0; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            int t = 0;
            for (; t < 2; ++t) iter(ModeC<TM_RAMP>{}, t, tile);
#pragma unroll 1
            for (; t < G; ++t) iter(ModeC<TM_STEADY>{}, t, tile);
            if (t == G) { iter(ModeC<TM_DRAIN>{}, t, tile); ++t; }       // (G == 1: the ramp already ran iteration G)
            iter(ModeC<TM_LAST>{}, t, tile);
            if (stamp) { const unsigned long long n = __builtin_readcyclecounter(); t_loop += n - ts; ts = n; }
            // ---- tile epilogue ----
            // The residual tile (requested one iteration ago) must have landed HERE, with a wait the compiler knows about:
            // left to itself it waits vmcnt(0) in front of every pass's first use -- loads and stores share the counter and
            // count as unordered once mixed -- which serialised the eight passes on each other's store round trips and on
            // the next tile's X (8 x ~2k cycles per tile).  After this point the passes only issue stores.
            __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
            if (tile + (int)gridDim.x < ntiles) load_x(tile + gridDim.x);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TM_BARRIER();                                                   // every wave is done with the exchange buffers
#pragma unroll
            for (int j = 0; j < TM_NB0; ++j) {
                char* const sb = stg + (j & 1) * 16384;
                const float bn = b2s[j * 16 + frow];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 v = {acc2[i][j].x + bn, acc2[i][j].y + bn, acc2[i][j].z + bn, acc2[i][j].w + bn};
                    *reinterpret_cast<f32x4*>(sb + frow * 512 + (((wc4 + i * 4) ^ frow) << 4)) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                epilogue_reader(j, sb, res[j], rimg, rcc, row_ok);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (stamp) { const unsigned long long n = __builtin_readcyclecounter(); t_epi += n - ts; ts = n; }
        }
    } else {
        // =============================== activation wave ===============================
        unsigned st = 0;
        f32x4 acc2[2][TM_NB1];
        const bool blk14 = TM_NB0 + TM_NB1 - 1 < nblk;     // the 14th token block usually does not exist
        u32x4 res[TM_NB0];
        int rimg = 0, rcc = 0;
        bool row_ok = false;
        auto iter = [&](auto mode_c, const int t, const int tile) {
            constexpr int MODE = decltype(mode_c)::value;
            const bool gelu = MODE == TM_STEADY || (t >= 1 && t <= G);
            const bool fc2 = MODE != TM_RAMP || t >= 2;
            TM_ITER_SYNC();
            if constexpr (MODE == TM_LAST) request_residual(tile, res, rimg, rcc, row_ok);
            const char* r2 = smem + TM_R2 + st * TM_STAGE;
            u32x4 af0, af1, bf[TM_NB1];
            if (fc2) {
                const char* hr = hs + (t & 1) * 2048;
                af0 = *reinterpret_cast<const u32x4*>(hr + f_rd);
                af1 = *reinterpret_cast<const u32x4*>(hr + 1024 + f_rd);
#pragma unroll
                for (int j = 0; j < TM_NB1; ++j) bf[j] = *reinterpret_cast<const u32x4*>(r2 + (TM_NB0 + j) * 1024 + f_rd);
            }
            if (gelu) {
                // gelu(t-1): bias, exact-erf GELU, round, store in A-operand order
                const char* ar = axs + ((t - 1) & 1) * 4096 + lane * 16;
                char* const hw = hs + ((t - 1) & 1) * 2048;
                const int gb = t - 1;
                const f32x4 bb0 = *reinterpret_cast<const f32x4*>(b1s + gb * 32 + 4 * fg);
                const f32x4 bb1 = *reinterpret_cast<const f32x4*>(b1s + gb * 32 + 16 + 4 * fg);
                f32x4 a[4];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) a[blk] = *reinterpret_cast<const f32x4*>(ar + blk * 1024);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    // two blocks = four float pairs, their GELU chains interleaved (a lone chain is latency-bound)
                    f32x2 v[4];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 bb = q ? bb1 : bb0;                 // block (i = half, j = q)
                        v[2 * q] = f32x2{a[half * 2 + q].x + bb.x, a[half * 2 + q].y + bb.y};
                        v[2 * q + 1] = f32x2{a[half * 2 + q].z + bb.z, a[half * 2 + q].w + bb.w};
                    }
#if TM_ABL & 2
#elif defined(TM_GELU_SCALAR)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = f32x2{gelu16_f<T>(v[c].x), gelu16_f<T>(v[c].y)};
#else
                    gelu_pk_n<T, 4>(v);
#endif
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int row = half * 16 + frow;
                        const int lc = q * 2 + (fg >> 1);
                        T e[4] = {from_f32<T>(v[2 * q].x), from_f32<T>(v[2 * q].y), from_f32<T>(v[2 * q + 1].x), from_f32<T>(v[2 * q + 1].y)};
                        u32x2 pk;
                        __builtin_memcpy(&pk, e, 8);
                        *reinterpret_cast<u32x2*>(hw + row * 64 + ((lc ^ ((row & 8) >> 2)) << 4) + ((fg & 1) << 3)) = pk;
                    }
                }
            }
            if (fc2) {
#pragma unroll
                for (int j = 0; j < TM_NB1; ++j) {
                    if constexpr (TM_ABL & 8) {
                        acc2[0][j] += __builtin_bit_cast(f32x4, af0) + __builtin_bit_cast(f32x4, bf[j]);
                        continue;
                    }
                    if (j < TM_NB1 - 1 || blk14) {
                        acc2[0][j] = Mma2<T>::run(af0, bf[j], acc2[0][j]);
                        acc2[1][j] = Mma2<T>::run(af1, bf[j], acc2[1][j]);
                    }
                }
            }
            st = st == 2 ? 0 : st + 1;
        };
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TM_NB1; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            int t = 0;
            for (; t < 2; ++t) iter(ModeC<TM_RAMP>{}, t, tile);
#pragma unroll 1
            for (; t < G; ++t) iter(ModeC<TM_STEADY>{}, t, tile);
            if (t == G) { iter(ModeC<TM_DRAIN>{}, t, tile); ++t; }       // (G == 1: the ramp already ran iteration G)
            iter(ModeC<TM_LAST>{}, t, tile);
            // ---- tile epilogue ----
            __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): the residual tile has landed (see the matrix wave)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TM_BARRIER();
#pragma unroll
            for (int j = 0; j < TM_NB0; ++j) {
                char* const sb = stg + (j & 1) * 16384;
                if (j < TM_NB1) {
                    const float bn = b2s[(TM_NB0 + j) * 16 + frow];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 a = acc2[i][j < TM_NB1 ? j : 0];
                        const f32x4 v = {a.x + bn, a.y + bn, a.z + bn, a.w + bn};
                        *reinterpret_cast<f32x4*>(sb + (16 + frow) * 512 + (((wc4 + i * 4) ^ frow) << 4)) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                epilogue_reader(j, sb, res[j], rimg, rcc, row_ok);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
    if (stamp && tid == 0) {
        p.dbg[(size_t)blockIdx.x * 4 + 0] = t_loop;
        p.dbg[(size_t)blockIdx.x * 4 + 1] = t_epi;
        p.dbg[(size_t)blockIdx.x * 4 + 2] = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    }
}

// ====================================================================================================================
// Second form of the same operation: 256-row tiles, the hidden NEVER leaves the registers.
//
// The kernel above hands acc1 (fp32) and H (16 bit) between two specialised waves through LDS; its iteration is bound by
// that chain of LDS round trips, LDS-DMA issue and the barrier that couples eight waves doing different things (removing
// the MFMAs, the GELU or the DMA alone changes its time by 0-9 %, profiles/r02_token_mlp_ablation.txt).  Here every wave
// owns 32 rows and runs fc1 -> GELU -> fc2 for them itself:
//   * fc1 with swapped operands leaves lane (row frow, fg) holding hidden columns {4 fg + r} and {16 + 4 fg + r} of its row:
//     after bias + GELU + rounding those eight values ARE one 16x16x32 A fragment, provided k slot 8 fg + e of the second
//     product means hidden column (e < 4 ? 4 fg + e : 16 + 4 fg + e - 4).  The host packs W2 with that permutation inside
//     every group of 32 hidden columns (layout 1, mlpk_token_mlp_layout), so nothing is exchanged at all.
//   * 8 waves x 32 rows = 256-row tiles: a W1 group / W2 slab staged in LDS serves twice the rows, one barrier per
//     iteration (the ring hand-over) is all the synchronisation, G iterations per tile instead of G + 2.
//   * The two waves of a SIMD are skewed by half an iteration so that one is in its matrix phase while the other runs
//     GELU: waves 0-3 run [fc1(t), gelu(t), fc2(t)], waves 4-7 run [gelu(t-1), fc2(t-1), fc1(t)] (acc1 stays in their
//     registers across the barrier) and drain gelu/fc2(G-1) after the loop.  W2 slab t-1 is therefore still read during
//     iteration t: the W2 ring has 4 stages, the W1 ring 3; both are filled two iterations ahead, 4 one-KiB pieces per
//     wave per iteration (14 + 14 + 4 duplicates), constant `s_waitcnt vmcnt(4)`.
//   * Epilogue: 13 passes of one token block x 256 channels of fp32 through LDS; reader = (token tid >> 5, 8 channels
//     tid & 31): 512-byte runs of one token row per 32 lanes for the residual load and the store.
//   * All 256 workgroups reach their epilogue together, so its 100 KB residual tile + 100 KB of stores + the 115 KB X tile of
//     the next iteration arrive as one chip-wide burst (81 MB at ~8 TB/s = the measured 19k cycles per tile) while HBM idles
//     during the 90k-cycle loops.  Touching those lines from inside the loop (LDS-DMA dword loads into a sink, all at once or
//     one wave per iteration) moved the same wait into the loop's vmcnt and measured slower (0.207-0.218 vs 0.199 ms); left out.
// LDS: 48 (W1 ring) + 64 (W2 ring) + 32 (epilogue staging) + 4 (b1) + 0.9 (b2) = 149 KiB.
#ifndef T2_BWD
#define T2_BWD 3         // W1 fragment pairs read ahead of their MFMAs
#endif
#ifndef T2_BFD
#define T2_BFD 4         // W2 fragments read ahead of their MFMAs
#endif
constexpr int T2_BM = 256;
constexpr int T2_NB = 13;                              // token blocks of 16: S <= 208
constexpr int T2_R1 = 0;
constexpr int T2_R2 = 3 * TM_STAGE;
constexpr int T2_STG = T2_R2 + 4 * TM_STAGE;
constexpr int T2_B1 = T2_STG + 2 * 16384;
constexpr int T2_B2 = T2_B1 + TM_B1_FLOATS * 4;
constexpr int T2_LDS = T2_B2 + 16 * T2_NB * 4;
#define T2_ITER_SYNC() do { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); TM_BARRIER(); } while (0)

template <typename T>
__global__ void __launch_bounds__(512, 1) token_mlp_rr_kernel(const TokenMlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool lag = wave >= 4;                            // the half that runs gelu/fc2 one iteration late
    const T* __restrict__ xt = reinterpret_cast<const T*>(p.xt);
    const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
    const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
    T* __restrict__ x = reinterpret_cast<T*>(p.x);
    const int G = p.G;
    const int ks1 = p.ks1;
    const int ntiles = (p.M + T2_BM - 1) / T2_BM;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    float* const b1s = reinterpret_cast<float*>(smem + T2_B1);
    float* const b2s = reinterpret_cast<float*>(smem + T2_B2);

    // ---- LDS-DMA pieces (geometry of the kernel above): wave w issues q = 4 w + pi; q < 14: W1 piece q (plane q >> 1, half
    // q & 1), 14 <= q < 28: W2 piece q - 14 (rows 16 (q - 14) .., clamped to S - 1), q >= 28: W2 piece 13 again ----
    // Every per-lane quantity the iterations use (piece offsets, fragment offsets, bias offsets) is re-derived inside the
    // iteration from an opaque copy of the lane id: values that live across the loop get spilled around it, their reloads in
    // the loop preheader leave "VMEM pending" at the loop header, and hipcc then waits vmcnt(3..0) in front of each LDS-DMA
    // piece in EVERY iteration -- draining the prefetch ring (seen in the ISA: 3.9k cycles per iteration).
    auto lane_now = [&]() {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (int)l;
    };
    unsigned pdst[4];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
        int q = wave * 4 + pi;
        q = q < 28 ? q : 27;
        pdst[pi] = __builtin_amdgcn_readfirstlane(q < 14 ? lds_base + T2_R1 + q * 1024 : lds_base + T2_R2 + (q - 14) * 1024);
    }
    auto piece_off = [&](const int pi, const int ln) {       // per-lane byte offset of piece pi from its group's base
        int q = wave * 4 + pi;
        q = q < 28 ? q : 27;
        const int lrow = ln >> 2;
        const int lchunk = (ln & 3) ^ ((lrow & 8) >> 2);
        unsigned o;
        if (q < 14) {
            o = (unsigned)(((q & 1) * 16 + lrow) * 256 + (q >> 1) * 32 + lchunk * 8) * (unsigned)sizeof(T);
        } else {
            int r2 = (q - 14) * 16 + lrow;
            r2 = r2 < p.S ? r2 : p.S - 1;
            o = (unsigned)(r2 * p.ldw2 + lchunk * 8) * (unsigned)sizeof(T);
        }
        return o;
    };
    const T* pb1 = w1;
    const T* pb2 = w2;
    unsigned so1 = 0, so2 = 0;                             // ring stage byte offsets of the pieces being issued
    auto piece_bases = [&](const int g) {                  // hidden group the pieces are for (already wrapped to 0..G-1)
        pb1 = w1 + (size_t)g * (32 * 256);
        pb2 = w2 + g * 32;
    };
    auto issue = [&](const int pi, const int ln) {
        const bool is1 = wave * 4 + pi < 14;
        tm_glds(piece_off(pi, ln), is1 ? pb1 : pb2, pdst[pi] + __builtin_amdgcn_readfirstlane(is1 ? so1 : so2));
    };
    // global iteration counter modulo the ring sizes; pieces of iteration gi live in W1 stage gi % 3, W2 stage gi % 4
    unsigned s3 = 0, s4 = 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        piece_bases(it < G ? it : it - G);                 // (G == 1: iteration 1 is the next tile's group 0)
        so1 = so2 = (unsigned)it * TM_STAGE;
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) issue(pi, lane);
    }
    if (lag) __builtin_amdgcn_s_setprio(2);                // the younger half would lose every VALU arbitration

    for (int i = tid; i < G * 32; i += 512) b1s[i] = p.b1[i];
    if (tid < 16 * T2_NB) b2s[tid] = tid < p.S ? p.b2[tid] : 0.f;
    __syncthreads();

    unsigned long long t_loop = 0, t_epi = 0, ts = 0;
    const bool stamp = p.dbg != nullptr;
    char* const stg = smem + T2_STG;
    const size_t srows = (size_t)(p.M / p.t_rows) * p.S;      // plane stride of the statistics pairs

    u32x4 xa[2][TM_KMAX];
    auto load_x = [&](const int tile, const int ln) {
        const int frow = ln & 15, fg = ln >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gm = tile * T2_BM + wave * 32 + i * 16 + frow;
            gm = gm < p.M ? gm : p.M - 1;
#pragma unroll
            for (int kk = 0; kk < TM_KMAX; ++kk)
                xa[i][kk] = *reinterpret_cast<const u32x4*>(xt + (size_t)gm * p.ldxt + (kk < ks1 ? kk : ks1 - 1) * 32 + fg * 8);
        }
    };

    f32x4 acc2[2][T2_NB];
    f32x4 a1[2][2];
    auto frag_off = [&](const int ln) {                    // fragment chunk of a 64-byte LDS row: row ln & 15, swizzled chunk
        const int fr = ln & 15;
        return fr * 64 + (((ln >> 4) ^ ((fr & 8) >> 2)) << 4);
    };
    auto fc1 = [&](const unsigned st3, const int ln) {
        const int f_rd = frag_off(ln);
        const char* r1 = smem + T2_R1 + st3 * TM_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) a1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragment reads two K-slabs ahead of their MFMAs, pinned there (left alone, hipcc hoists all 14 reads: 56 registers
        // on top of the 184 that acc2 / X / acc1 / H hold, and spills)
        u32x4 bw[T2_BWD + 1][2];
#pragma unroll
        for (int kk = 0; kk < T2_BWD; ++kk) {
            bw[kk][0] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd);
            bw[kk][1] = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd + 1024);
        }
#pragma unroll
        for (int kk = 0; kk < TM_KMAX; ++kk) {
            if (kk + T2_BWD < TM_KMAX) {
                bw[(kk + T2_BWD) % (T2_BWD + 1)][0] = *reinterpret_cast<const u32x4*>(r1 + (kk + T2_BWD) * 2048 + f_rd);
                bw[(kk + T2_BWD) % (T2_BWD + 1)][1] = *reinterpret_cast<const u32x4*>(r1 + (kk + T2_BWD) * 2048 + f_rd + 1024);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a1[i][0] = Mma2<T>::run(bw[kk % (T2_BWD + 1)][0], xa[i][kk], a1[i][0]);
                a1[i][1] = Mma2<T>::run(bw[kk % (T2_BWD + 1)][1], xa[i][kk], a1[i][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    u32x4 hf[2];
    auto gelu = [&](const int g, const int ln) {           // bias + exact-erf GELU + rounding: acc1 -> the two A fragments of fc2
        const int fg = ln >> 4;
        const f32x4 bb0 = *reinterpret_cast<const f32x4*>(b1s + g * 32 + 4 * fg);
        const f32x4 bb1 = *reinterpret_cast<const f32x4*>(b1s + g * 32 + 16 + 4 * fg);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x2 v[4] = {f32x2{a1[i][0].x + bb0.x, a1[i][0].y + bb0.y}, f32x2{a1[i][0].z + bb0.z, a1[i][0].w + bb0.w},
                          f32x2{a1[i][1].x + bb1.x, a1[i][1].y + bb1.y}, f32x2{a1[i][1].z + bb1.z, a1[i][1].w + bb1.w}};
#if !(TM_ABL & 2)
            gelu_pk_n<T, 4>(v);
#endif
            T e[8] = {from_f32<T>(v[0].x), from_f32<T>(v[0].y), from_f32<T>(v[1].x), from_f32<T>(v[1].y),
                      from_f32<T>(v[2].x), from_f32<T>(v[2].y), from_f32<T>(v[3].x), from_f32<T>(v[3].y)};
            __builtin_memcpy(&hf[i], e, 16);
        }
    };
    auto fc2 = [&](const unsigned st4, const int ln) {
        const int f_rd = frag_off(ln);
        const char* r2 = smem + T2_R2 + st4 * TM_STAGE;
        u32x4 bf[T2_BFD + 1];
#pragma unroll
        for (int j = 0; j < T2_BFD; ++j) bf[j] = *reinterpret_cast<const u32x4*>(r2 + j * 1024 + f_rd);
#pragma unroll
        for (int j = 0; j < T2_NB; ++j) {
            if (j + T2_BFD < T2_NB) bf[(j + T2_BFD) % (T2_BFD + 1)] = *reinterpret_cast<const u32x4*>(r2 + (j + T2_BFD) * 1024 + f_rd);
            acc2[0][j] = Mma2<T>::run(hf[0], bf[j % (T2_BFD + 1)], acc2[0][j]);
            acc2[1][j] = Mma2<T>::run(hf[1], bf[j % (T2_BFD + 1)], acc2[1][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto run = [&](auto lag_c) {
        constexpr bool LAG = decltype(lag_c)::value;
        load_x(blockIdx.x, lane_now());                    // (inside each half's own code: one live range per path)
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            if (stamp) ts = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the X operands have landed, and the compiler knows it
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < T2_NB; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto iter = [&](auto first_c, const int t) {
                constexpr bool FIRST = decltype(first_c)::value;
                T2_ITER_SYNC();
                // pieces of iteration t + 2 (the next tile's first groups at the end of this one: the weights are the same)
                int g2 = t + 2;
                g2 = g2 < G ? g2 : g2 - G;
                g2 = g2 < G ? g2 : g2 - G;                    // (G == 1)
                piece_bases(g2);
                so1 = (s3 == 0 ? 2u : s3 - 1) * TM_STAGE;     // stage (gi + 2) % 3
                so2 = ((s4 + 2) & 3) * TM_STAGE;              // stage (gi + 2) % 4
                const int ln = lane_now();
                issue(0, ln); issue(1, ln);
                if constexpr (!LAG) {
                    fc1(s3, ln);
                    issue(2, ln); issue(3, ln);
                    gelu(t, ln);
                    fc2(s4, ln);
                } else {
                    if constexpr (!FIRST) {
                        gelu(t - 1, ln);
                        fc2((s4 + 3) & 3, ln);                // slab t - 1: stage (gi - 1) % 4
                    }
                    issue(2, ln); issue(3, ln);
                    fc1(s3, ln);
                }
                s3 = s3 == 2 ? 0 : s3 + 1;
                s4 = (s4 + 1) & 3;
            };
            iter(BoolC<true>{}, 0);
#pragma unroll 1
            for (int t = 1; t < G; ++t) iter(BoolC<false>{}, t);
            if constexpr (LAG) {
                const int ln = lane_now();
                gelu(G - 1, ln);
                fc2((s4 + 3) & 3, ln);
            }
            if (stamp) { const unsigned long long n = __builtin_readcyclecounter(); t_loop += n - ts; ts = n; }
            // ---- tile epilogue ----
            // (per-lane epilogue geometry from a fresh lane id as well: nothing lane-derived lives across the iterations)
            const int le = lane_now();
            const int frow = le & 15, fg = le >> 4;
            const int rt = wave * 2 + (le >> 5), rc = le & 31;    // reader: token slot tid >> 5, 8-channel chunk tid & 31
            const int mr = tile * T2_BM + rc * 8;
            const int rimg = mr / p.t_rows;
            const int rcc = mr - rimg * p.t_rows;
            const bool row_ok = mr < p.M;
            // token row j * 16 + rt of the image: one 32-bit per-lane offset for all passes + a uniform per-pass base (thirteen
            // hoisted 64-bit addresses were spilled, and every reload in a pass is a vmcnt(0) on the previous pass's stores)
            const unsigned voff = (unsigned)((((size_t)rimg * p.S + rt) * p.ldx + rcc) * sizeof(T));
            const size_t pass_stride = (size_t)16 * p.ldx * sizeof(T);
            u32x4 res[T2_NB];
#pragma unroll
            for (int j = 0; j < T2_NB; ++j) {
                const int rn = j * 16 + rt;
                res[j] = u32x4{0u, 0u, 0u, 0u};
                if (rn < p.S && row_ok) res[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(x) + j * pass_stride + voff);
            }
            // one wait for the whole residual tile; the passes issue nothing but stores (see above)
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int j = 0; j < T2_NB; ++j) {
                // the next tile's X (56 registers) is requested once seven passes have released as many accumulators
                // (unconditionally -- rows past M clamp to the last row: under a condition the OLD X stays live through all passes)
                if (j == 7) load_x(tile + gridDim.x, le);
                char* const sb = stg + (j & 1) * 16384;
                const float bn = b2s[j * 16 + frow];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 v = {acc2[i][j].x + bn, acc2[i][j].y + bn, acc2[i][j].z + bn, acc2[i][j].w + bn};
                    *reinterpret_cast<f32x4*>(sb + frow * 1024 + (((wave * 8 + i * 4 + fg) ^ frow) << 4)) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const int rn = j * 16 + rt;
                const bool live = rn < p.S && row_ok;
                float ssum = 0.f, ssq = 0.f;
                if (live) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sb + rt * 1024 + (((2 * rc) ^ rt) << 4));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + rt * 1024 + (((2 * rc + 1) ^ rt) << 4));
                    T r8[8];
                    __builtin_memcpy(r8, &res[j], 16);
                    T e[8] = {from_f32<T>(v0.x + to_f32(r8[0])), from_f32<T>(v0.y + to_f32(r8[1])), from_f32<T>(v0.z + to_f32(r8[2])),
                              from_f32<T>(v0.w + to_f32(r8[3])), from_f32<T>(v1.x + to_f32(r8[4])), from_f32<T>(v1.y + to_f32(r8[5])),
                              from_f32<T>(v1.z + to_f32(r8[6])), from_f32<T>(v1.w + to_f32(r8[7]))};
                    u32x4 o;
                    __builtin_memcpy(&o, e, 16);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(x) + j * pass_stride + voff) = o;
                    if (p.stats) chunk_sums<T>(o, ssum, ssq);
                }
                if (p.stats) {                                 // (workgroup-uniform: every lane takes part in the DPP rows)
                    ssum = row16_sum(ssum);
                    ssq = row16_sum(ssq);
                    if (live && (rc & 15) == 0)
                        *reinterpret_cast<f32x2*>(p.stats + ((size_t)(rcc >> 7) * srows + (size_t)rimg * p.S + rn) * 2) = f32x2{ssum, ssq};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (stamp) { const unsigned long long n = __builtin_readcyclecounter(); t_epi += n - ts; ts = n; }
        }
    };
    if (!lag) run(BoolC<false>{}); else run(BoolC<true>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
    if (stamp && tid == 0) {
        p.dbg[(size_t)blockIdx.x * 4 + 0] = t_loop;
        p.dbg[(size_t)blockIdx.x * 4 + 1] = t_epi;
        p.dbg[(size_t)blockIdx.x * 4 + 2] = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    }
}

// ====================================================================================================================
// Single token-mixing product with the per-image transpose in the epilogue (gMLP's spatial gating unit, g_mlp.py:17-22; ResMLP's
// cross-patch sublayer, res_mlp.py:52-55):
//     out[b,t,c] = R[b,t,c] (+ | *) rscale[c] * ( sum_s W[t,s] * xt[b*C + c, s] + bias[t] )
// The general NT GEMM with the token-transposed epilogue runs this at 2-2.5x its memory floor (N = S = 196 is one and a half
// 128-column tiles, K = 224 is three and a half slabs: 202 us for gMLP-S at 256 images against 80 us of traffic).  Here the
// machinery of the kernels above is reused: persistent workgroups, 256-row tiles, every wave keeps its 32 x S_pad block of xt in
// registers, W streams through a two-stage LDS ring in groups of 32 output tokens (one-KiB LDS-DMA pieces, 2 per wave and
// iteration).  Iteration g: wait + barrier; request group g + 1 and the R values of group g - 1; the 28 MFMAs of group g (natural
// operands: lane = token, 4 consecutive channels); group g - 1 leaves -- fp32 staging tile [32 tokens][256 channels] in LDS ->
// (token, 8 channels) items, 16-byte R loads and stores in 512-byte runs; then group g's accumulators (+ bias) are staged in the
// other buffer.  One barrier per iteration; loads and stores share the iterations evenly, so there is no chip-wide memory burst.
// LDS: 32 (W ring) + 64 (staging) + 1 (bias) = 97 KiB.
constexpr int T3_BM = 256;
constexpr int T3_R1 = 0;                               // W ring: 2 stages x 16 KiB
constexpr int T3_STG = 2 * TM_STAGE;                   // staging: 2 x 32 KiB
constexpr int T3_B = T3_STG + 2 * 32768;
constexpr int T3_LDS = T3_B + 256 * 4;
// Round 4 -- the LayerNorm / affine in front of the product as this kernel's OPERAND LOADER (LNL = 1; mlpk_token_gemm_ln): instead of
// a pass that writes LN(x) token-transposed (gMLP's SGU: mlpk_layernorm_transpose, 72 us and 308 MB per layer at 256 images, 17 % of
// the model; ResMLP's Aff: the transposed half of mlpk_norm_apply) and this kernel reading it back, every wave reads ITS 32 channels
// of the token-major rows of x itself (64-byte pieces of full lines), normalises them in fp32 -- (x rstd - mean rstd) gamma + beta,
// the token kernel's expression (csrc/gen/t4gen.py) -- rounds once, and transposes through a per-wave LDS tile
// [32 channels][16 token pairs] (80-byte pitch) into the A fragments it keeps in registers for the whole tile.
constexpr int T3_XT = T3_LDS;                          // 8 waves x 32 channels x 80 B
constexpr int T3_XT_PITCH = 80;
constexpr int T3_LDS_LN = T3_XT + 8 * 32 * T3_XT_PITCH;

struct TokenGemmArgs {
    const void* xt;     // (M, ldxt) token-transposed operand, K-padded with zeros
    const void* w;      // (G*32, 256): output-token rows and K zero-padded
    const float* bias;  // (G*32) zero-padded, or NULL
    const float* rscale;// per channel (index = row % rperiod), or NULL
    const void* R;      // residual / gate, (B*S, ldr), or NULL
    void* out;          // (B*S, ldo)
    int M, S, ks1, G;
    int ldxt, ldr, ldo, t_rows, rperiod, res_mode;
    // LNL: the operand is LN(x) / Aff(x) of the token-major x (B*S, ldx), channels [0, t_rows) of the rows the pointer addresses
    const void* x;
    const float* ln_mean;   // per token row (B*S), or NULL (affine only: mean 0, rstd 1)
    const float* ln_rstd;
    const float* gamma;     // per channel (t_rows)
    const float* beta;
    int ldx;
    // round 5 (mlpk_token_gemm_ln_post, pipelined kernel only): out = post_scale[c] * round(result) + post_shift[c], rounded again -- the Aff that
    // FOLLOWS the cross-patch sublayer of ResMLP (res_mlp.py:56: x = self.post_affine(x)) applied where the sublayer's result is stored
    const float* post_scale;   // per channel (t_rows), or NULL
    const float* post_shift;
};

template <typename T, int RES, int LNL = 0>
__global__ void __launch_bounds__(512, 1) token_gemm_kernel(const TokenGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const T* __restrict__ xt = reinterpret_cast<const T*>(p.xt);
    const T* __restrict__ w = reinterpret_cast<const T*>(p.w);
    const T* __restrict__ Rp = reinterpret_cast<const T*>(p.R);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int G = p.G;
    const int ks1 = p.ks1;
    const int ntiles = (p.M + T3_BM - 1) / T3_BM;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    float* const bs = reinterpret_cast<float*>(smem + T3_B);
    for (int i = tid; i < 256; i += 512) bs[i] = (p.bias && i < G * 32) ? p.bias[i] : 0.f;

    auto lane_now = [&]() {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (int)l;
    };
    // W group = 8 planes (K-slabs) x [32 rows x 64 B], pieces pc = plane * 2 + half, planes 0-6 only: 14 pieces, wave w issues 2w, 2w + 1
    // (waves 7: pieces 14, 15 -> duplicates of 13)
    auto issue_w = [&](const int g, const unsigned stage, const int ln) {
        const T* base = w + (size_t)g * (32 * 256);
        const int lrow = ln >> 2;
        const int lchunk = (ln & 3) ^ ((lrow & 8) >> 2);
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            int q = wave * 2 + pi;
            q = q < 14 ? q : 13;
            const unsigned off = (unsigned)(((q & 1) * 16 + lrow) * 256 + (q >> 1) * 32 + lchunk * 8) * (unsigned)sizeof(T);
            tm_glds(off, base, __builtin_amdgcn_readfirstlane(lds_base + T3_R1 + stage * TM_STAGE + q * 1024));
        }
    };

    u32x4 xa[2][TM_KMAX];
    auto load_x = [&](const int tile, const int ln) {
        const int frow = ln & 15, fg = ln >> 4;
        if constexpr (LNL) {
            // this wave's 32 rows of the transposed operand = channels c0 .. c0 + 31 of image b (32 | t_rows: one image per wave)
            int m0w = tile * T3_BM + wave * 32;
            m0w = m0w < p.M ? m0w : p.M - 32;
            const int b = m0w / p.t_rows;
            const int c0 = m0w - b * p.t_rows;
            const int pr = ln >> 2, q = ln & 3;                 // token pair of the k-step, octet of channels
            const T* xr = reinterpret_cast<const T*>(p.x) + (size_t)b * p.S * p.ldx + c0 + q * 8;
            float ga[8], be[8];
            {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + q * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + q * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + c0 + q * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + c0 + q * 8 + 4);
                ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
                be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
            }
            char* const xt_w = smem + T3_XT + wave * (32 * T3_XT_PITCH);
            // loads one k-step ahead of the arithmetic
            u32x4 raw[2][2];
            float mu[2][2], rs[2][2];
            auto request = [&](const int kk, const int buf) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int sidx = kk * 32 + pr * 2 + h;
                    sidx = sidx < p.S ? sidx : p.S - 1;           // (tokens past S: valid memory, zeroed below)
                    raw[buf][h] = *reinterpret_cast<const u32x4*>(xr + (size_t)sidx * p.ldx);
                    mu[buf][h] = p.ln_mean ? p.ln_mean[(size_t)b * p.S + sidx] : 0.f;
                    rs[buf][h] = p.ln_mean ? p.ln_rstd[(size_t)b * p.S + sidx] : 1.f;
                }
            };
            request(0, 0);
#pragma unroll
            for (int kk = 0; kk < TM_KMAX; ++kk) {
                if (kk + 1 < TM_KMAX) request(kk + 1 < ks1 ? kk + 1 : ks1 - 1, (kk + 1) & 1);
                const int buf = kk & 1;
                float y[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    T e8[8];
                    __builtin_memcpy(e8, &raw[buf][h], 16);
                    const bool live = kk < ks1 && kk * 32 + pr * 2 + h < p.S;
                    const float nm = -mu[buf][h] * rs[buf][h];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = __builtin_fmaf(__builtin_fmaf(to_f32(e8[e]), rs[buf][h], nm), ga[e], be[e]);
                        y[h][e] = live ? v : 0.f;
                    }
                }
                // word (channel 8 q + e, token pair pr) = (token 2 pr, token 2 pr + 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    T pk2[2] = {from_f32<T>(y[0][e]), from_f32<T>(y[1][e])};
                    unsigned wv;
                    __builtin_memcpy(&wv, pk2, 4);
                    *reinterpret_cast<unsigned*>(xt_w + (q * 8 + e) * T3_XT_PITCH + pr * 4) = wv;
                }
                __builtin_amdgcn_wave_barrier();                // (scheduling only: the reads below stay behind the writes)
                // (LDS executes a wave's operations in order: no wait between the writes and the reads, nor before the next k-step's writes)
#pragma unroll
                for (int i = 0; i < 2; ++i) xa[i][kk] = *reinterpret_cast<const u32x4*>(xt_w + (i * 16 + frow) * T3_XT_PITCH + fg * 16);
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int gm = tile * T3_BM + wave * 32 + i * 16 + frow;
                gm = gm < p.M ? gm : p.M - 1;
#pragma unroll
                for (int kk = 0; kk < TM_KMAX; ++kk)
                    xa[i][kk] = *reinterpret_cast<const u32x4*>(xt + (size_t)gm * p.ldxt + (kk < ks1 ? kk : ks1 - 1) * 32 + fg * 8);
            }
        }
    };
    load_x(blockIdx.x, lane_now());
    issue_w(0, 0, lane);                                   // group 0 of the first tile
    __syncthreads();
    // W ring stage of the group multiplied in the current iteration.  It must run ON across tiles: the last MFMA iteration of a
    // tile prefetches group 0 of the next tile into the OTHER stage, and with an odd number of groups (S = 196 -> 7) that is stage 1
    // -- indexing the ring by the group number read stage 0 (still the previous tile's last group) for the first 32 output tokens of
    // every tile after a workgroup's first (round-3 finding of test_batch_256_rows_match_small_batch: gMLP-S at 256 images 5e-2 off;
    // ResMLP's layer scale of 1e-4 hid the same fault)
    unsigned wst = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // reader geometry of this tile: item = (token slot tid >> 5 (+16), channels (tid & 31) * 8 .. + 7 of the tile's 256)
        const int le = lane_now();
        const int rt = wave * 2 + (le >> 5), rc = le & 31;
        const int mr = tile * T3_BM + rc * 8;
        const int rimg = mr / p.t_rows;
        const int rcc = mr - rimg * p.t_rows;
        const bool row_ok = mr < p.M;
        float rsc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rsc[e] = 1.0f;
        if (p.rscale && row_ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) rsc[e] = p.rscale[(mr + e) % p.rperiod];
        }
        // LNL == 2: the residual is the affine output itself, R = round(gamma x + beta) (res_mlp.py:53-55: x + gamma_1 token_mix(x) on the
        // POST-affine tensor), rebuilt from x for the reader item's 8 channels -- no Aff pass, no x1 tensor
        float rga[LNL == 2 ? 8 : 1], rbe[LNL == 2 ? 8 : 1];
        if constexpr (LNL == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { rga[e] = row_ok ? p.gamma[rcc + e] : 0.f; rbe[e] = row_ok ? p.beta[rcc + e] : 0.f; }
        }
        f32x4 acc[2][2];
        u32x4 rv[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
        auto request_r = [&](const int g) {                // R values of group g's reader items (used one iteration later)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = g * 32 + rt + 16 * k;
                rv[k] = u32x4{0u, 0u, 0u, 0u};
                if constexpr (LNL == 2) {
                    if (t < p.S && row_ok) rv[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.x) + ((size_t)rimg * p.S + t) * p.ldx + rcc);
                } else {
                    if (RES != MLPK_RES_NONE && t < p.S && row_ok) rv[k] = *reinterpret_cast<const u32x4*>(Rp + ((size_t)rimg * p.S + t) * p.ldr + rcc);
                }
            }
        };
        // Order inside an iteration: (1) group g - 1 leaves (its R values were requested an iteration ago), (2) W group g + 1 and
        // the R values of group g are requested, (3) the MFMAs of group g, (4) its accumulators are staged.  The single
        // vmcnt(0) at the top then only meets operations that have had (almost) a whole iteration to complete -- the stores
        // first of all (with the stores issued late in the iteration the same wait exposed their round trip).
        for (int g = 0; g <= G; ++g) {
            __builtin_amdgcn_s_waitcnt(0x0070);           // vmcnt(0) lgkmcnt(0) expcnt(7)
            TM_BARRIER();
            const int ln = lane_now();
            const int frow = ln & 15, fg = ln >> 4;
            if (g > 0) {
                const char* sb = smem + T3_STG + ((g - 1) & 1) * 32768;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int slot = rt + 16 * k;
                    const int t = (g - 1) * 32 + slot;
                    if (t < p.S && row_ok) {
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sb + slot * 1024 + (((2 * rc) ^ (slot & 15)) << 4));
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + slot * 1024 + (((2 * rc + 1) ^ (slot & 15)) << 4));
                        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        T r8[8], e[8];
                        __builtin_memcpy(r8, &rv[k], 16);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            float y = v[q] * rsc[q];
                            if constexpr (LNL == 2) y += to_f32(from_f32<T>(__builtin_fmaf(to_f32(r8[q]), rga[q], rbe[q])));
                            else if constexpr (RES == MLPK_RES_ADD) y += to_f32(r8[q]);
                            else if constexpr (RES == MLPK_RES_MUL) y *= to_f32(r8[q]);
                            e[q] = from_f32<T>(y);
                        }
                        u32x4 o;
                        __builtin_memcpy(&o, e, 16);
                        *reinterpret_cast<u32x4*>(out + ((size_t)rimg * p.S + t) * p.ldo + rcc) = o;
                    }
                }
            }
            if (g < G) {
                // next W group: g + 1 of this tile, or group 0 again for the next tile (the weights are the same for every tile)
                issue_w(g + 1 < G ? g + 1 : 0, wst ^ 1u, ln);
                request_r(g);
                const char* r1 = smem + T3_R1 + wst * TM_STAGE;
                wst ^= 1u;
                const int f_rd = frow * 64 + ((fg ^ ((frow & 8) >> 2)) << 4);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < TM_KMAX; ++kk) {
                    const u32x4 bw0 = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd);
                    const u32x4 bw1 = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd + 1024);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        acc[i][0] = Mma2<T>::run(xa[i][kk], bw0, acc[i][0]);      // natural operands: lane = token frow, channels 4 fg + r
                        acc[i][1] = Mma2<T>::run(xa[i][kk], bw1, acc[i][1]);
                    }
                }
                if (g == G - 1) load_x(tile + gridDim.x, ln);                  // X is dead now: the next tile's (rows past M clamp)
                char* const sw = smem + T3_STG + (g & 1) * 32768;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float bn = bs[g * 32 + j * 16 + frow];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 v = {acc[i][j].x + bn, acc[i][j].y + bn, acc[i][j].z + bn, acc[i][j].w + bn};
                        *reinterpret_cast<f32x4*>(sw + (j * 16 + frow) * 1024 + (((wave * 8 + i * 4 + fg) ^ frow) << 4)) = v;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
}

// ====================================================================================================================
// Round 5 -- mlpk_token_gemm_ln as a software pipeline across tiles (G >= 3 groups, even S: gMLP / ResMLP at 196 tokens).
// The kernel above waits `vmcnt(0)` at the top of every iteration, builds a tile's operand in seven dependent load -> normalise ->
// transpose steps, and spends one iteration per tile on draining the last group.  Here
//   * the iterations of a workgroup's tiles form ONE sequence i = 0 .. n (n = tiles x G): iteration i sends group i - 1 out (possibly the
//     previous tile's last one), requests the gate / residual values of group i (used one iteration later) and the weights of group
//     i + 2 (three-stage ring), and multiplies group i -- no per-tile drain iteration;
//   * the NEXT tile's x rows (+ their statistics) are requested at the first iteration of a tile, all 14 + 14 loads at once, and turned into
//     MFMA operands at the first iteration of their own tile, G iterations later;
//   * every load of the loop is issued by `asm volatile` and waited for by a COUNTED `s_waitcnt vmcnt(N)`, N = the loads issued AFTER the
//     ones the iteration needs (vector-memory operations retire in order, so "at most N outstanding" = everything older has landed;
//     the stores, which the compiler may skip for a wave without live items, are older than the loads of their iteration and never
//     counted).  An asm load's destination is not touched until the wait that covers it (each use is behind a "+v" no-op asm after
//     the wait); tools/isa_lint.py `inflight` (tests/test_host_cpu.py) scans the built ISA for a compiler instruction on such a
//     register in between -- two R register sets selected by the iteration's parity failed exactly that check.
// gamma / beta / rscale come from LDS tables (filled once per workgroup), so the loop has no compiler-visible load at all.
// Measured (profiles/r05_token_gemm_pipe_ab.txt): ResMLP-24 5.40 -> 5.20 ms, gMLP-S 9.38 -> 9.32 ms same-box; gMLP's call moves 462 MB
// (u, v: 308 MB in, 154 MB out at 256 images) in 119 us = 3.9 TB/s either way -- that one is bound by the bytes, not by the waits.
// LDS: 48 (W ring) + 64 (staging) + 1 (bias) + 20 (transposes) + 24 (tables) = 157 KiB.
constexpr int T5_TMAX = 2048;                          // channels per image / rscale period the tables hold
constexpr int T5_R1 = 0;
constexpr int T5_STG = 3 * TM_STAGE;
constexpr int T5_B = T5_STG + 2 * 32768;
constexpr int T5_XT = T5_B + 256 * 4;
constexpr int T5_TAB = T5_XT + 8 * 32 * T3_XT_PITCH;
constexpr int T5_LDS = T5_TAB + 3 * T5_TMAX * 4;
static_assert(T5_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ u32x4 tg_load16(const void* ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ f32x2 tg_load8(const void* ptr) {
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// n = number of asm loads issued after the ones that must have landed (wave-uniform; one of the sums the loop can produce)
__device__ __forceinline__ void tg_wait(const int n) {
    switch (n) {
        case 30: asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28) lgkmcnt(0)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
}

template <typename T, int RES, int LNL>
__global__ void __launch_bounds__(512, 1) token_gemm_pipe_kernel(const TokenGemmArgs p) {
    static_assert(LNL == 1 || LNL == 2, "operand-loader variants only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const T* __restrict__ w = reinterpret_cast<const T*>(p.w);
    const T* __restrict__ Rp = reinterpret_cast<const T*>(LNL == 2 ? p.x : p.R);
    const int ldr = LNL == 2 ? p.ldx : p.ldr;
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int G = p.G;
    const int ks1 = p.ks1;
    const int ntiles = (p.M + T3_BM - 1) / T3_BM;
    const int mine = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // (grid <= tiles: >= 1)
    const int n = mine * G;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    float* const bs = reinterpret_cast<float*>(smem + T5_B);
    float* const tgam = reinterpret_cast<float*>(smem + T5_TAB);
    float* const tbet = tgam + T5_TMAX;
    float* const trs = tbet + T5_TMAX;
    // (the post-affine tables share the second half of the gamma / beta tables: the launcher admits t_rows <= T5_TMAX / 2 with them)
    float* const tpa = tgam + T5_TMAX / 2;
    float* const tpb = tbet + T5_TMAX / 2;
    const bool post = p.post_scale != nullptr;
    for (int i = tid; i < 256; i += 512) bs[i] = (p.bias && i < G * 32) ? p.bias[i] : 0.f;
    for (int i = tid; i < p.t_rows; i += 512) { tgam[i] = p.gamma[i]; tbet[i] = p.beta[i]; }
    if (post)
        for (int i = tid; i < p.t_rows; i += 512) { tpa[i] = p.post_scale[i]; tpb[i] = p.post_shift[i]; }
    // (no rscale: rperiod = 1 and items index trs[0 .. 7])
    for (int i = tid; i < (p.rscale ? p.rperiod : 8); i += 512) trs[i] = p.rscale ? p.rscale[i] : 1.f;
    const bool has_ln = p.ln_mean != nullptr;
    const int nx = has_ln ? 28 : 14;                       // asm loads of one x request

    auto lane_now = [&]() {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (int)l;
    };
    auto issue_w = [&](const int g, const unsigned stage, const int ln) {
        const T* base = w + (size_t)__builtin_amdgcn_readfirstlane(g) * (32 * 256);      // (the asm's scalar operand)
        const int lrow = ln >> 2;
        const int lchunk = (ln & 3) ^ ((lrow & 8) >> 2);
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            int q = wave * 2 + pi;
            q = q < 14 ? q : 13;
            const unsigned off = (unsigned)(((q & 1) * 16 + lrow) * 256 + (q >> 1) * 32 + lchunk * 8) * (unsigned)sizeof(T);
            tm_glds(off, base, __builtin_amdgcn_readfirstlane(lds_base + T5_R1 + stage * TM_STAGE + q * 1024));
        }
    };

    // ---- the operand: this wave's 32 channels of one image, all tokens
    u32x4 xa[2][TM_KMAX];
    u32x4 raw[TM_KMAX][2];
    f32x2 mu[TM_KMAX], rs[TM_KMAX];                        // (S is even: the statistics of a lane's token pair are one 8-byte load)
    auto x_origin = [&](const int tile, int& b, int& c0) {
        int m0w = tile * T3_BM + wave * 32;
        m0w = m0w < p.M ? m0w : p.M - 32;
        b = m0w / p.t_rows;
        c0 = m0w - b * p.t_rows;
    };
    auto request_x = [&](const int tile, const int ln) {       // nx asm loads
        int b, c0;
        x_origin(tile, b, c0);
        const int pr = ln >> 2, q = ln & 3;
        const T* xr = reinterpret_cast<const T*>(p.x) + (size_t)b * p.S * p.ldx + c0 + q * 8;
        const float* mr_ = p.ln_mean + (size_t)b * p.S;
        const float* rr_ = p.ln_rstd + (size_t)b * p.S;
#pragma unroll
        for (int kk = 0; kk < TM_KMAX; ++kk) {
            int s0 = (kk < ks1 ? kk : ks1 - 1) * 32 + pr * 2;
            s0 = s0 < p.S ? s0 : p.S - 2;                       // (pairs past S: valid memory, zeroed in finish_x)
            raw[kk][0] = tg_load16(xr + (size_t)s0 * p.ldx);
            raw[kk][1] = tg_load16(xr + (size_t)(s0 + 1) * p.ldx);
            if (has_ln) {
                mu[kk] = tg_load8(mr_ + s0);
                rs[kk] = tg_load8(rr_ + s0);
            }
        }
    };
    auto finish_x = [&](const int tile, const int ln) {         // (after the wait that covers the request)
        int b, c0;
        x_origin(tile, b, c0);
        const int frow = ln & 15, fg = ln >> 4;
        const int pr = ln >> 2, q = ln & 3;
        float ga[8], be[8];
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(tgam + c0 + q * 8), g1 = *reinterpret_cast<const f32x4*>(tgam + c0 + q * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(tbet + c0 + q * 8), b1 = *reinterpret_cast<const f32x4*>(tbet + c0 + q * 8 + 4);
            ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
            be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
        }
        char* const xt_w = smem + T5_XT + wave * (32 * T3_XT_PITCH);
#pragma unroll
        for (int kk = 0; kk < TM_KMAX; ++kk) {
            float y[2][8];
            asm volatile("" : "+v"(raw[kk][0]), "+v"(raw[kk][1]));
            f32x2 m2 = {0.f, 0.f}, r2 = {1.f, 1.f};
            if (has_ln) {
                asm volatile("" : "+v"(mu[kk]), "+v"(rs[kk]));
                m2 = mu[kk]; r2 = rs[kk];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float m_ = h ? m2.y : m2.x, r_ = h ? r2.y : r2.x;
                T e8[8];
                __builtin_memcpy(e8, &raw[kk][h], 16);
                const bool live = kk < ks1 && kk * 32 + pr * 2 + h < p.S;
                const float nm = -m_ * r_;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = __builtin_fmaf(__builtin_fmaf(to_f32(e8[e]), r_, nm), ga[e], be[e]);
                    y[h][e] = live ? v : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                T pk2[2] = {from_f32<T>(y[0][e]), from_f32<T>(y[1][e])};
                unsigned wv;
                __builtin_memcpy(&wv, pk2, 4);
                *reinterpret_cast<unsigned*>(xt_w + (q * 8 + e) * T3_XT_PITCH + pr * 4) = wv;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) xa[i][kk] = *reinterpret_cast<const u32x4*>(xt_w + (i * 16 + frow) * T3_XT_PITCH + fg * 16);
            __builtin_amdgcn_wave_barrier();
        }
    };

    // ---- reader items: (token slot wave * 2 + (lane >> 5) (+16), channels (lane & 31) * 8 .. + 7 of the tile's 256)
    struct RG { int rimg, rcc, rso; bool ok; };
    auto rgeo = [&](const int tile, const int rc) {
        RG r;
        const int mr = tile * T3_BM + rc * 8;
        r.ok = mr < p.M;
        const int mc = r.ok ? mr : 0;
        r.rimg = mc / p.t_rows;
        r.rcc = mc - r.rimg * p.t_rows;
        r.rso = p.rscale ? mc % p.rperiod : 0;
        return r;
    };
    // R values of ONE group: requested right after the previous group's values were used, one iteration ahead of their own use (two
    // register sets selected by the iteration's parity -- a branch, or a loop unrolled by two -- made hipcc copy the set in flight around
    // the branch, resp. run out of scalar registers: tools/isa_lint.py inflight)
    u32x4 rv[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    // R values of group g of a tile (2 asm loads; dead items read a valid address and are not used)
    auto request_r = [&](const RG& r, const int g, const int rt, u32x4 (&dst)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int t = g * 32 + rt + 16 * k;
            t = t < p.S ? t : p.S - 1;
            dst[k] = tg_load16(Rp + ((size_t)r.rimg * p.S + t) * ldr + r.rcc);
        }
    };
    auto leave = [&](const RG& r, const int g, const int rt, const int rc, u32x4 (&src)[2], const char* sb) {
        asm volatile("" : "+v"(src[0]), "+v"(src[1]));
        float rsc[8];
        {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(trs + r.rso), s1 = *reinterpret_cast<const f32x4*>(trs + r.rso + 4);
            rsc[0] = s0.x; rsc[1] = s0.y; rsc[2] = s0.z; rsc[3] = s0.w; rsc[4] = s1.x; rsc[5] = s1.y; rsc[6] = s1.z; rsc[7] = s1.w;
        }
        float rga[LNL == 2 ? 8 : 1], rbe[LNL == 2 ? 8 : 1];
        if constexpr (LNL == 2) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(tgam + r.rcc), g1 = *reinterpret_cast<const f32x4*>(tgam + r.rcc + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(tbet + r.rcc), b1 = *reinterpret_cast<const f32x4*>(tbet + r.rcc + 4);
            rga[0] = g0.x; rga[1] = g0.y; rga[2] = g0.z; rga[3] = g0.w; rga[4] = g1.x; rga[5] = g1.y; rga[6] = g1.z; rga[7] = g1.w;
            rbe[0] = b0.x; rbe[1] = b0.y; rbe[2] = b0.z; rbe[3] = b0.w; rbe[4] = b1.x; rbe[5] = b1.y; rbe[6] = b1.z; rbe[7] = b1.w;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int slot = rt + 16 * k;
            const int t = g * 32 + slot;
            if (t < p.S && r.ok) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sb + slot * 1024 + (((2 * rc) ^ (slot & 15)) << 4));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + slot * 1024 + (((2 * rc + 1) ^ (slot & 15)) << 4));
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                T r8[8], e[8];
                __builtin_memcpy(r8, &src[k], 16);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float y = v[q] * rsc[q];
                    if constexpr (LNL == 2) y += to_f32(from_f32<T>(__builtin_fmaf(to_f32(r8[q]), rga[q], rbe[q])));
                    else if constexpr (RES == MLPK_RES_ADD) y += to_f32(r8[q]);
                    else if constexpr (RES == MLPK_RES_MUL) y *= to_f32(r8[q]);
                    e[q] = from_f32<T>(y);
                }
                if (post) {                                    // (workgroup-uniform)
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(tpa + r.rcc), a1 = *reinterpret_cast<const f32x4*>(tpa + r.rcc + 4);
                    const f32x4 c0_ = *reinterpret_cast<const f32x4*>(tpb + r.rcc), c1_ = *reinterpret_cast<const f32x4*>(tpb + r.rcc + 4);
                    const float pa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, pb[8] = {c0_.x, c0_.y, c0_.z, c0_.w, c1_.x, c1_.y, c1_.z, c1_.w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) e[q] = from_f32<T>(__builtin_fmaf(to_f32(e[q]), pa[q], pb[q]));   // (mlpk_norm_apply's form without statistics)
                }
                u32x4 o;
                __builtin_memcpy(&o, e, 16);
                *reinterpret_cast<u32x4*>(out + ((size_t)r.rimg * p.S + t) * p.ldo + r.rcc) = o;
            }
        }
    };
    constexpr bool NEEDS_R = LNL == 2 || RES != MLPK_RES_NONE;

    __syncthreads();                                       // tables
    int tile = blockIdx.x, g = 0;                          // of iteration i
    RG gp, gc;                                             // reader geometry of the previous / current tile
    {
        const int ln = lane_now();
        gc = rgeo(tile, ln & 31);
        gp = gc;
        issue_w(0, 0, ln);
        issue_w(1, 1, ln);
    }
    int nprev = -1;                                        // asm loads issued after the previous iteration's R request (-1: wait for everything)
    // Iteration -1 requests the first tile's x; iteration i >= 0 as described above.  (ONE copy of the request / operand code in the kernel:
    // with a prologue copy and an unrolled pair of iterations the kernel ran out of scalar registers.)
    for (int i = -1; i <= n; ++i) {
        const unsigned par = (unsigned)i & 1u;
        if (i >= 0) {
            tg_wait(nprev);
            TM_BARRIER();
        }
        const int ln = lane_now();
        const int frow = ln & 15, fg = ln >> 4;
        const int rt = wave * 2 + (ln >> 5), rc = ln & 31;
        const bool more = i >= 0 && i < n;
        const bool first = g == 0 && more;
        // this tile's x (requested G iterations ago) -> operands; the last MFMA of the previous tile was issued before the barrier
        if (first) finish_x(tile, ln);
        // (1) group i - 1 leaves
        if (i > 0) leave(g == 0 ? gp : gc, g == 0 ? G - 1 : g - 1, rt, rc, rv, smem + T5_STG + (par ^ 1u) * 32768);
        // (2) requests: the R values of group i (the wait at the top of iteration i + 1 allows what follows them to stay in flight), the next
        // tile's x, the weights of group i + 2
        if (more && NEEDS_R) request_r(gc, g, rt, rv);
        int after = 0;
        const int xt_tile = i < 0 ? tile : tile + (int)gridDim.x;
        if ((i < 0 || first) && xt_tile < ntiles) { request_x(xt_tile, ln); after += nx; }
        if (i >= 0 && i + 2 < n) { issue_w(g + 2 >= G ? g + 2 - G : g + 2, (unsigned)((i + 2) % 3), ln); after += 2; }
        nprev = i < 0 ? -1 : after;
        // (3) the MFMAs of group i, (4) its accumulators (+ bias) staged
        if (more) {
            const char* r1 = smem + T5_R1 + (i % 3) * TM_STAGE;
            const int f_rd = frow * 64 + ((fg ^ ((frow & 8) >> 2)) << 4);
            f32x4 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < TM_KMAX; ++kk) {
                const u32x4 bw0 = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd);
                const u32x4 bw1 = *reinterpret_cast<const u32x4*>(r1 + kk * 2048 + f_rd + 1024);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    acc[a][0] = Mma2<T>::run(xa[a][kk], bw0, acc[a][0]);
                    acc[a][1] = Mma2<T>::run(xa[a][kk], bw1, acc[a][1]);
                }
            }
            char* const sw = smem + T5_STG + par * 32768;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bn = bs[g * 32 + j * 16 + frow];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const f32x4 v = {acc[a][j].x + bn, acc[a][j].y + bn, acc[a][j].z + bn, acc[a][j].w + bn};
                    *reinterpret_cast<f32x4*>(sw + (j * 16 + frow) * 1024 + (((wave * 8 + a * 4 + fg) ^ frow) << 4)) = v;
                }
            }
            if (++g == G) { g = 0; tile += gridDim.x; gp = gc; gc = rgeo(tile, rc); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static unsigned long long* g_tm_dbg = nullptr;

static int tm_grid_cap() {
    static int cap = 0;
    if (!cap) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        cap = cu;
    }
    return cap;
}

}  // namespace mlpk

using namespace mlpk;

extern "C" void mlpk_token_mlp_debug(void* buf) { g_tm_dbg = reinterpret_cast<unsigned long long*>(buf); }

extern "C" int mlpk_token_mlp_chunk(void) { return 32; }

// 1: W2 packed with the hidden columns of every group of 32 permuted (k slot 8 f + e <- column (e < 4 ? 4 f + e : 16 + 4 f + e - 4))
// for the 256-row kernel that keeps the hidden in registers; 0: natural order (128-row kernel).
extern "C" int mlpk_token_mlp_layout(int S, int nchunks) {
    const char* e = getenv("MLPK_TOKEN_MLP_LAYOUT");
    if (e && e[0] == '0') return 0;
    return (S <= 16 * T2_NB && nchunks * 32 <= TM_B1_FLOATS) ? 1 : 0;
}

// 2 when the generated one-wave-per-SIMD kernel (mlpk_tokenmlp_t4.hip) takes the shape, else mlpk_token_mlp_layout's answer.
extern "C" int mlpk_token_mlp_layout_for(int dtype, int S, int nchunks, int t_rows) {
    const char* e = getenv("MLPK_TOKEN_MLP_LAYOUT");
    if (e && (e[0] == '0' || e[0] == '1')) return mlpk_token_mlp_layout(S, nchunks);
    if (t4_supported(dtype, S, nchunks, ((S + 31) / 32) * 32, t_rows, t_rows, t_rows)) {
        // 3 = layout 2 with the hidden kept in f16 (bf16 storage only; MLPK_T4_H2=0 keeps the all-bf16 kernels: A/B aid)
        const char* h = getenv("MLPK_T4_H2");
        return (dtype == MLPK_BF16 && !(h && h[0] == '0')) ? 3 : 2;
    }
    return mlpk_token_mlp_layout(S, nchunks);
}

extern "C" int mlpk_token_mlp(int dtype, const void* xt, int ldxt, int M, int S, const void* w1, int ldw1, const float* b1,
                              const void* w2, int ldw2, const float* b2, int nchunks, void* x, int ldx, int t_rows,
                              float* stats, int layout, void* stream) {
    if (!xt || !w1 || !w2 || !b1 || !b2 || !x) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;   // 16-bit storage only (fp32 uses the two-GEMM path)
    if (M <= 0 || S <= 0 || nchunks <= 0 || t_rows <= 0) return MLPK_ESHAPE;
    if (layout < 0 || layout > 3 || (layout == 3 && dtype != MLPK_BF16)) return MLPK_EMODE;
    if (layout == 2 || layout == 3) {
        // generated kernel: W2 group-major ((nchunks + 1) * 224 rows of 32 k slots), b1 / b2 as padded tables (mlpk.h)
        if (ldw1 != 256 || ldw2 != 32) return MLPK_ESHAPE;
        if (!t4_supported(dtype, S, nchunks, ldxt, M, t_rows, ldx)) return MLPK_ESHAPE;
        if (stats && ((uintptr_t)stats & 7)) return MLPK_ESHAPE;
        if (((uintptr_t)xt & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)x & 15) || ((uintptr_t)b1 & 15)) return MLPK_EALIGN;
        T4Call c;
        c.dtype = dtype; c.M = M; c.S = S; c.G = nchunks; c.ldxt = ldxt; c.ldx = ldx; c.t_rows = t_rows;
        c.xt = xt; c.w1 = w1; c.w2 = w2; c.b1 = b1; c.b2 = b2; c.x = x; c.stats = stats; c.prof = g_tm_dbg;
        c.ln_mean = c.ln_rstd = c.gamma = c.beta = nullptr;
        const char* d = getenv("MLPK_T4_DBG");
        c.dbg = d ? atoi(d) : 0;
        c.h2 = layout == 3;
        return t4_launch(c, reinterpret_cast<hipStream_t>(stream));
    }
    if (S > 16 * (TM_NB0 + TM_NB1) || nchunks * 32 > TM_B1_FLOATS) return MLPK_ESHAPE;  // up to 224 tokens, 1024 hidden
    if (ldxt % 32 || ldxt > 32 * TM_KMAX || ldxt < S) return MLPK_ESHAPE;       // K of fc1 = ldxt: whole 64-byte slabs, <= 7
    if (ldw1 != 256 || ldw2 < nchunks * 32 || ldw2 % 8) return MLPK_ESHAPE;
    // a 16-byte chunk of the output = 8 consecutive rows (channels) of one image
    if (M % 8 || t_rows % 8 || M % t_rows || ldx % 8 || ldx < t_rows) return MLPK_ESHAPE;
    if (stats && (t_rows % 128 || ((uintptr_t)stats & 7))) return MLPK_ESHAPE;     // statistics partials: whole 128-channel tiles per image
    if (((uintptr_t)xt & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)x & 15)) return MLPK_EALIGN;
    TokenMlpArgs a;
    a.xt = xt; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2; a.x = x;
    a.M = M; a.S = S; a.ks1 = ldxt / 32; a.G = nchunks;
    a.ldxt = ldxt; a.ldw2 = ldw2; a.ldx = ldx; a.t_rows = t_rows;
    a.stats = stats;
    a.dbg = g_tm_dbg;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (layout == 1) {
        if (S > 16 * T2_NB) return MLPK_ESHAPE;
        if ((unsigned long long)(M / t_rows) * S * ldx * 2ull >= (1ull << 32)) return MLPK_ESHAPE;   // 32-bit per-lane offsets into x
        if ((unsigned long long)M * ldxt * 2ull >= (1ull << 32)) return MLPK_ESHAPE;                // ... and into xt (prefetch)
        const int tiles2 = (M + T2_BM - 1) / T2_BM;
        const unsigned grid2 = (unsigned)(tiles2 < tm_grid_cap() ? tiles2 : tm_grid_cap());
        hipError_t e2;
        if (dtype == MLPK_BF16) {
            auto k = token_mlp_rr_kernel<bf16_t>;
            e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            if (e2 != hipSuccess) return (int)e2;
            hipLaunchKernelGGL(k, dim3(grid2), dim3(512), T2_LDS, s, a);
        } else {
            auto k = token_mlp_rr_kernel<f16_t>;
            e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
            if (e2 != hipSuccess) return (int)e2;
            hipLaunchKernelGGL(k, dim3(grid2), dim3(512), T2_LDS, s, a);
        }
        MLPK_LAUNCH_CHECK();
        return 0;
    }
    const int tiles = (M + TM_BM - 1) / TM_BM;
    const unsigned grid = (unsigned)(tiles < tm_grid_cap() ? tiles : tm_grid_cap());
    hipError_t e;
    if (dtype == MLPK_BF16) {
        auto k = token_mlp_kernel<bf16_t>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), TM_LDS, s, a);
    } else {
        auto k = token_mlp_kernel<f16_t>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, TM_LDS);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), TM_LDS, s, a);
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

// The token-mixing PreNormResidual of MLP-Mixer in ONE kernel (mlp_mixer.py:34 with :6-13, :16-27): the LayerNorm + per-image transpose
// is the generated kernel's operand loader (no xt tensor).  Weights packed for layout 2 or 3 (`layout`: what mlpk_token_mlp_layout_for answered).
extern "C" int mlpk_token_mlp_ln(int dtype, void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                                 const float* beta, const void* w1, int ldw1, const float* b1, const void* w2, int ldw2, const float* b2,
                                 int nchunks, int t_rows, float* stats, int layout, void* stream) {
    if (!x || !ln_mean || !ln_rstd || !gamma || !beta || !w1 || !w2 || !b1 || !b2) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if ((layout != 2 && layout != 3) || (layout == 3 && dtype != MLPK_BF16)) return MLPK_EMODE;
    if (ldw1 != 256 || ldw2 != 32 || nchunks < 2) return MLPK_ESHAPE;
    if (!t4_supported(dtype, S, nchunks, 224, M, t_rows, ldx)) return MLPK_ESHAPE;
    if (stats && ((uintptr_t)stats & 7)) return MLPK_ESHAPE;
    if (((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)x & 15) || ((uintptr_t)b1 & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) ||
        ((uintptr_t)ln_mean & 7) || ((uintptr_t)ln_rstd & 7))
        return MLPK_EALIGN;
    T4Call c;
    c.dtype = dtype; c.M = M; c.S = S; c.G = nchunks; c.ldxt = 224; c.ldx = ldx; c.t_rows = t_rows;
    c.xt = nullptr; c.w1 = w1; c.w2 = w2; c.b1 = b1; c.b2 = b2; c.x = x; c.stats = stats; c.prof = g_tm_dbg;
    c.ln_mean = ln_mean; c.ln_rstd = ln_rstd; c.gamma = gamma; c.beta = beta;
    c.dbg = 0;
    c.h2 = layout == 3;
    return t4_launch(c, reinterpret_cast<hipStream_t>(stream));
}

static int token_gemm_launch(int dtype, TokenGemmArgs& a, int res_mode, int lnl, hipStream_t s) {
    const int tiles = (a.M + T3_BM - 1) / T3_BM;
    const unsigned grid = (unsigned)(tiles < tm_grid_cap() ? tiles : tm_grid_cap());
#define TG_LAUNCH(TT, RR, LL)                                                                                          \
    {                                                                                                                   \
        auto k = token_gemm_kernel<TT, RR, LL>;                                                                         \
        const int lds = LL ? T3_LDS_LN : T3_LDS;                                                                        \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);                                                        \
    }
#define TG_RES(TT, LL)                                                                                                 \
    if (res_mode == MLPK_RES_ADD) TG_LAUNCH(TT, MLPK_RES_ADD, LL) else if (res_mode == MLPK_RES_MUL) TG_LAUNCH(TT, MLPK_RES_MUL, LL) else TG_LAUNCH(TT, MLPK_RES_NONE, LL)
    // round 5: the operand-loader variants with >= 3 groups run as the two-iterations-deep pipeline (MLPK_TOKEN_GEMM_PIPE=0: the kernel above, A/B aid)
    static const bool pipe_on = !(getenv("MLPK_TOKEN_GEMM_PIPE") && atoi(getenv("MLPK_TOKEN_GEMM_PIPE")) == 0);
    const bool pipe_ok = lnl && pipe_on && a.G >= 3 && a.S % 2 == 0 && a.t_rows <= T5_TMAX && (!a.rscale || (a.rperiod % 8 == 0 && a.rperiod <= T5_TMAX)) &&
                         !(((uintptr_t)a.ln_mean | (uintptr_t)a.ln_rstd) & 7);
    if (a.post_scale && !(pipe_ok && a.t_rows <= T5_TMAX / 2)) return MLPK_ESHAPE;   // (the caller applies the Aff itself)
    if (pipe_ok) {      // (the statistics of a token pair are one 8-byte load)
#define TP_LAUNCH(TT, RR, LL)                                                                                          \
    {                                                                                                                   \
        auto k = token_gemm_pipe_kernel<TT, RR, LL>;                                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS); \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), T5_LDS, s, a);                                                     \
    }
#define TP_RES(TT)                                                                                                     \
    if (res_mode == MLPK_RES_ADD) TP_LAUNCH(TT, MLPK_RES_ADD, 1) else if (res_mode == MLPK_RES_MUL) TP_LAUNCH(TT, MLPK_RES_MUL, 1) else TP_LAUNCH(TT, MLPK_RES_NONE, 1)
        if (dtype == MLPK_BF16) {
            if (lnl == 2) TP_LAUNCH(bf16_t, MLPK_RES_ADD, 2) else { TP_RES(bf16_t) }
        } else {
            if (lnl == 2) TP_LAUNCH(f16_t, MLPK_RES_ADD, 2) else { TP_RES(f16_t) }
        }
#undef TP_RES
#undef TP_LAUNCH
        MLPK_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == MLPK_BF16) {
        if (lnl == 2) TG_LAUNCH(bf16_t, MLPK_RES_ADD, 2) else if (lnl) { TG_RES(bf16_t, 1) } else { TG_RES(bf16_t, 0) }
    } else {
        if (lnl == 2) TG_LAUNCH(f16_t, MLPK_RES_ADD, 2) else if (lnl) { TG_RES(f16_t, 1) } else { TG_RES(f16_t, 0) }
    }
#undef TG_RES
#undef TG_LAUNCH
    MLPK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mlpk_token_gemm(int dtype, const void* xt, int ldxt, int M, int S, const void* w, int ldw, const float* bias, int ngroups,
                               const float* rscale, int rperiod, const void* R, int ldr, int res_mode, void* out, int ldo, int t_rows,
                               void* stream) {
    if (!xt || !w || !out) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (M <= 0 || S <= 0 || ngroups <= 0 || t_rows <= 0) return MLPK_ESHAPE;
    if (res_mode < MLPK_RES_NONE || res_mode > MLPK_RES_MUL) return MLPK_EMODE;
    if (res_mode != MLPK_RES_NONE && !R) return MLPK_ENULL;
    if (rscale && rperiod <= 0) return MLPK_ESHAPE;
    if (ngroups * 32 < S || ngroups > 8) return MLPK_ESHAPE;                      // output tokens: whole groups of 32, <= 256
    if (ldxt % 32 || ldxt > 32 * TM_KMAX || ldxt < S) return MLPK_ESHAPE;           // K = ldxt: whole 64-byte slabs, <= 7
    if (ldw != 256) return MLPK_ESHAPE;
    if (M % 8 || t_rows % 8 || M % t_rows || ldo % 8 || ldo < t_rows || (R && (ldr % 8 || ldr < t_rows))) return MLPK_ESHAPE;
    if (((uintptr_t)xt & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15) || ((uintptr_t)R & 15)) return MLPK_EALIGN;
    TokenGemmArgs a;
    a.xt = xt; a.w = w; a.bias = bias; a.rscale = rscale; a.R = R; a.out = out;
    a.M = M; a.S = S; a.ks1 = ldxt / 32; a.G = ngroups;
    a.ldxt = ldxt; a.ldr = ldr; a.ldo = ldo; a.t_rows = t_rows; a.rperiod = rperiod > 0 ? rperiod : 1; a.res_mode = res_mode;
    a.x = nullptr; a.ln_mean = a.ln_rstd = a.gamma = a.beta = nullptr; a.ldx = 0;
    a.post_scale = a.post_shift = nullptr;
    return token_gemm_launch(dtype, a, res_mode, 0, reinterpret_cast<hipStream_t>(stream));
}

// mlpk_token_gemm with the LayerNorm / affine of its operand inside (mlpk.h): x token-major, no xt tensor
extern "C" int mlpk_token_gemm_ln_post(int dtype, const void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                                       const float* beta, const void* w, int ldw, const float* bias, int ngroups, const float* rscale, int rperiod,
                                       const void* R, int ldr, int res_mode, const float* post_scale, const float* post_shift, void* out, int ldo,
                                       int t_rows, void* stream);

extern "C" int mlpk_token_gemm_ln(int dtype, const void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                                  const float* beta, const void* w, int ldw, const float* bias, int ngroups, const float* rscale, int rperiod,
                                  const void* R, int ldr, int res_mode, void* out, int ldo, int t_rows, void* stream) {
    return mlpk_token_gemm_ln_post(dtype, x, ldx, M, S, ln_mean, ln_rstd, gamma, beta, w, ldw, bias, ngroups, rscale, rperiod, R, ldr, res_mode, nullptr, nullptr,
                                   out, ldo, t_rows, stream);
}

// ... with the per-channel affine that FOLLOWS the sublayer applied to what is stored (ResMLP's post_affine, res_mlp.py:56); MLPK_ESHAPE when the shape
// does not run on the pipelined kernel (the caller then applies the affine with mlpk_norm_apply, as before)
extern "C" int mlpk_token_gemm_ln_post(int dtype, const void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                                       const float* beta, const void* w, int ldw, const float* bias, int ngroups, const float* rscale, int rperiod,
                                       const void* R, int ldr, int res_mode, const float* post_scale, const float* post_shift, void* out, int ldo,
                                       int t_rows, void* stream) {
    if (!x || !w || !out || !gamma || !beta) return MLPK_ENULL;
    if ((post_scale != nullptr) != (post_shift != nullptr)) return MLPK_ENULL;
    if ((ln_mean != nullptr) != (ln_rstd != nullptr)) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;
    if (M <= 0 || S <= 0 || ngroups <= 0 || t_rows <= 0) return MLPK_ESHAPE;
    if (res_mode < MLPK_RES_NONE || res_mode > MLPK_RES_ADD_AFFINE) return MLPK_EMODE;
    // ADD_AFFINE: the residual is the affine output itself, rebuilt in the kernel from x (no statistics: ResMLP's Aff); an explicit mode
    // since ABI 9 -- plain ADD with R aliasing x is refused instead of silently meaning this (advisor, round 4)
    const bool raff = res_mode == MLPK_RES_ADD_AFFINE;
    if (raff) {
        if (ln_mean || (R && R != x)) return MLPK_EMODE;
        R = x; ldr = ldx; res_mode = MLPK_RES_ADD;
    } else if (res_mode == MLPK_RES_ADD && R == x) {
        return MLPK_EMODE;
    }
    if (res_mode != MLPK_RES_NONE && !R) return MLPK_ENULL;
    if (rscale && rperiod <= 0) return MLPK_ESHAPE;
    if (ngroups * 32 < S || ngroups > 8 || S > 32 * TM_KMAX) return MLPK_ESHAPE;
    if (ldw != 256) return MLPK_ESHAPE;
    // a wave owns 32 channels of ONE image; x rows in 16-byte pieces
    if (t_rows % 32 || M % t_rows || ldx % 8 || ldx < t_rows || ldo % 8 || ldo < t_rows || (R && (ldr % 8 || ldr < t_rows))) return MLPK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15) || ((uintptr_t)R & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return MLPK_EALIGN;
    TokenGemmArgs a;
    a.xt = nullptr; a.w = w; a.bias = bias; a.rscale = rscale; a.R = R; a.out = out;
    a.M = M; a.S = S; a.ks1 = (S + 31) / 32; a.G = ngroups;
    a.ldxt = 0; a.ldr = ldr; a.ldo = ldo; a.t_rows = t_rows; a.rperiod = rperiod > 0 ? rperiod : 1; a.res_mode = res_mode;
    a.x = x; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.gamma = gamma; a.beta = beta; a.ldx = ldx;
    a.post_scale = post_scale; a.post_shift = post_shift;
    return token_gemm_launch(dtype, a, res_mode, raff ? 2 : 1, reinterpret_cast<hipStream_t>(stream));
}
