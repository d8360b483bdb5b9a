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
//     residual (one rounding) with whole 256-byte token rows per 16 lanes, for both its load and its store;
//     the next tile's X operands are requested before it.
// LDS: 48 (W1 ring) + 48 (W2 ring) + 16 (H, double-buffered) + 32 (acc1 exchange / epilogue staging) + 4 (b1) = 148 KiB.
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
constexpr int TM_LDS = TM_B1 + TM_B1_FLOATS * 4;

#define TM_BARRIER() asm volatile("s_barrier" ::: "memory")
// (lgkmcnt(0): this wave's LDS writes of the previous iteration must have reached LDS before the barrier)
#define TM_ITER_SYNC() do { asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); TM_BARRIER(); } while (0)

template <bool B> struct BoolC { static constexpr bool value = B; };

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
    __syncthreads();

    unsigned long long t_loop = 0, t_epi = 0, ts = 0;
    const bool stamp = p.dbg != nullptr;
    // Epilogue geometry.  A pass moves 32 token slots (0-15: the matrix wave's block j, 16-31: the activation wave's
    // block 8 + j) x 128 channels of fp32 (acc2 + b2) through LDS; reader thread = (slot tid >> 4, 8 channels tid & 15)
    // adds the residual in fp32, rounds ONCE and writes 16 bytes, i.e. whole 256-byte token rows per 16 lanes.
    const int rt = tid >> 4, rc = tid & 15;
    char* const stg = smem + TM_AX;
    const int wc4 = slice * 8 + fg;                        // writer's 16-byte chunk (4 channels) for i = 0; + 4 for i = 1
    auto epilogue_reader = [&](const int j, const char* sb, const u32x4 res, const int rimg, const int rcc, const bool row_ok) {
        const int rn = (rt < 16 ? j : TM_NB0 + j) * 16 + (rt & 15);   // token this thread moves in pass j
        if ((rt < 16 || j < TM_NB1) && rn < p.S && row_ok) {
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
        }
    };
    auto residual_load = [&](const int j, const int rimg, const int rcc, const bool row_ok) {
        const int rn = (rt < 16 ? j : TM_NB0 + j) * 16 + (rt & 15);
        u32x4 r = {0u, 0u, 0u, 0u};
        if ((rt < 16 || j < TM_NB1) && rn < p.S && row_ok) r = *reinterpret_cast<const u32x4*>(x + ((size_t)rimg * p.S + rn) * p.ldx + rcc);
        return r;
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
        auto iter = [&](auto steady_c, const int t) {
            constexpr bool STEADY = decltype(steady_c)::value;
            const bool fc1 = STEADY || t < G;
            const bool fc2 = STEADY || t >= 2;
            TM_ITER_SYNC();
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
                    issue(stoff2, kk);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a1[i][0] = Mma2<T>::run(bw[kk][0], xa[i][kk], a1[i][0]);
                        a1[i][1] = Mma2<T>::run(bw[kk][1], xa[i][kk], a1[i][1]);
                    }
                }
            } else {
#pragma unroll
                for (int pi = 0; pi < 7; ++pi) issue(stoff2, pi);
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
            for (; t < 2; ++t) iter(BoolC<false>{}, t);
#pragma unroll 1
            for (; t < G; ++t) iter(BoolC<true>{}, t);
            for (; t < G + 2; ++t) iter(BoolC<false>{}, t);
            if (stamp) { const unsigned long long n = __builtin_readcyclecounter(); t_loop += n - ts; ts = n; }
            // ---- tile epilogue ----
            const int m0 = tile * TM_BM;
            const int mr = m0 + rc * 8;
            const int rimg = mr / p.t_rows;
            const int rcc = mr - rimg * p.t_rows;
            const bool row_ok = mr < p.M;
            u32x4 res[TM_NB0];
#pragma unroll
            for (int j = 0; j < TM_NB0; ++j) res[j] = residual_load(j, rimg, rcc, row_ok);
            if (tile + (int)gridDim.x < ntiles) load_x(tile + gridDim.x);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TM_BARRIER();                                                   // every wave is done with the exchange buffers
#pragma unroll
            for (int j = 0; j < TM_NB0; ++j) {
                char* const sb = stg + (j & 1) * 16384;
                const int n = j * 16 + frow;
                const float bn = n < p.S ? p.b2[n] : 0.f;
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
        auto iter = [&](auto steady_c, const int t) {
            constexpr bool STEADY = decltype(steady_c)::value;
            const bool gelu = STEADY || (t >= 1 && t <= G);
            const bool fc2 = STEADY || t >= 2;
            TM_ITER_SYNC();
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
                    gelu_pk_n<4>(v);
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
            for (; t < 2; ++t) iter(BoolC<false>{}, t);
#pragma unroll 1
            for (; t < G; ++t) iter(BoolC<true>{}, t);
            for (; t < G + 2; ++t) iter(BoolC<false>{}, t);
            // ---- tile epilogue ----
            const int m0 = tile * TM_BM;
            const int mr = m0 + rc * 8;
            const int rimg = mr / p.t_rows;
            const int rcc = mr - rimg * p.t_rows;
            const bool row_ok = mr < p.M;
            u32x4 res[TM_NB0];
#pragma unroll
            for (int j = 0; j < TM_NB0; ++j) res[j] = residual_load(j, rimg, rcc, row_ok);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TM_BARRIER();
#pragma unroll
            for (int j = 0; j < TM_NB0; ++j) {
                char* const sb = stg + (j & 1) * 16384;
                if (j < TM_NB1) {
                    const int n = (TM_NB0 + j) * 16 + frow;
                    const float bn = n < p.S ? p.b2[n] : 0.f;
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

extern "C" int mlpk_token_mlp(int dtype, const void* xt, int ldxt, int M, int S, const void* w1, int ldw1, const float* b1,
                              const void* w2, int ldw2, const float* b2, int nchunks, void* x, int ldx, int t_rows,
                              void* stream) {
    if (!xt || !w1 || !w2 || !b1 || !b2 || !x) return MLPK_ENULL;
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return MLPK_EDTYPE;   // 16-bit storage only (fp32 uses the two-GEMM path)
    if (M <= 0 || S <= 0 || nchunks <= 0 || t_rows <= 0) return MLPK_ESHAPE;
    if (S > 16 * (TM_NB0 + TM_NB1) || nchunks * 32 > TM_B1_FLOATS) return MLPK_ESHAPE;  // up to 224 tokens, 1024 hidden
    if (ldxt % 32 || ldxt > 32 * TM_KMAX || ldxt < S) return MLPK_ESHAPE;       // K of fc1 = ldxt: whole 64-byte slabs, <= 7
    if (ldw1 != 256 || ldw2 < nchunks * 32 || ldw2 % 8) return MLPK_ESHAPE;
    // a 16-byte chunk of the output = 8 consecutive rows (channels) of one image
    if (M % 8 || t_rows % 8 || M % t_rows || ldx % 8 || ldx < t_rows) return MLPK_ESHAPE;
    if (((uintptr_t)xt & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)x & 15)) return MLPK_EALIGN;
    TokenMlpArgs a;
    a.xt = xt; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2; a.x = x;
    a.M = M; a.S = S; a.ks1 = ldxt / 32; a.G = nchunks;
    a.ldxt = ldxt; a.ldw2 = ldw2; a.ldx = ldx; a.t_rows = t_rows;
    a.dbg = g_tm_dbg;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
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
