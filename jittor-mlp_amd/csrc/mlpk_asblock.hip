// AS-MLP: the whole axial-shift half of a block for ONE SAMPLE per workgroup (round 6; as_mlp.py:55-95,149-159), for the stage whose sample
// fits a CU's LDS -- AS-MLP-T / -S / -B stage 3: a 14 x 14 map of 384 channels = 147 KiB in 16 bits:
//
//     x <- x + conv3( norm2( gelu(conv2_1(sh_W u)) + gelu(conv2_2(sh_H u)) ) ),     u = gelu( norm1a( conv1( norm1(x) ) ) )
//
// norm1 / norm1a / norm2 are GroupNorm(1, C): ONE statistic per sample over all H W C values.  As separate launches that statistic is a
// grid-wide dependency between every two 1x1 convolutions: per block 4 statistics passes + 4 GEMMs of 50176 x 384 x 384 (2.3 rounds of
// tiles, 37 us each for 14.8 GFLOP) + the normalise-and-shift pass = 9 launches, 203 us, for 59 GFLOP.  With the sample resident in one
// workgroup the statistics are workgroup reductions, the GEMMs are 196 x 384 x 384 products of the workgroup (256 samples = one workgroup
// per CU, one round, no tail), and the only HBM traffic is x in, x out (the two shifted operands go through a scratch tensor that stays in
// the L2: the shifts are addresses of the staging loads).
//
// Workgroup = 8 waves; LDS = the operand tile [196 rows][C channels] (row pitch 2 C + 16 bytes: 16 consecutive rows on distinct banks).
// A product: wave w owns output channels [48 w, 48 w + 48) for all 13 row blocks of 16 (156 accumulator registers); the weight fragments
// come straight from global memory (295 KB per matrix: L2), the activation fragments from the LDS; operands swapped (channels x rows) so
// that a lane's four accumulators of a block are four consecutive channels of one pixel (8-byte pieces).
//   P0  x -> LDS, statistics of x                                                 (norm1, folded into conv1: gamma in the weights)
//   P1  conv1: t = (acc - mu csum) rstd + b' -> LDS (over the dead operand), statistics of the rounded t         (norm1a)
//   P2  u = gelu(t scale[c] + shift[c]) -> scratch U
//   P3  LDS <- U shifted along W; y1 = gelu(conv2_1 + b) -> scratch V
//   P4  LDS <- U shifted along H; s = round(gelu(conv2_2 + b)) + y1 -> LDS, statistics of the rounded s          (norm2, folded into conv3)
//   P5  conv3: y = (acc - mu csum) rstd + b' + x -> x, statistics of the rounded y -> mean_out / rstd_out        (the block's norm2)
// Every reduction runs in a fixed order inside the workgroup: a sample's result does not depend on the batch it is in.
#include "mlpk_common.h"
#include <cstdio>
#include <cstdlib>

namespace mlpk {

struct AsBlockArgs {
    void* x;                    // (B * HW, C) channel-last, updated in place
    void* u;                    // scratch (B * HW, C)
    void* v;                    // scratch (B * HW, C)
    const void* w1;             // (C, ldw) conv1, block.norm1's gamma folded in
    const void* w21;            // conv2_1 (the W-shifted operand)
    const void* w22;            // conv2_2 (the H-shifted operand)
    const void* w3;             // conv3, AxialShift.norm2's gamma folded in
    const float* b1;            // conv1 bias + W1 beta
    const float* cs1;           // column sums of the folded W1
    const float* ag;            // AxialShift.norm1 gamma / beta
    const float* ab;
    const float* b21;
    const float* b22;
    const float* b3;
    const float* cs3;
    float* mean_out;            // statistics of the new x per sample (the block's norm2), or NULL
    float* rstd_out;
    int B, H, W, ldw, ks;
    float eps;
    long long* prof;            // MLPK_AS_BLOCK_PROF: cycle stamps of workgroup 0's first sample (tuning aid)
};

template <typename T> struct AbMma;
template <> struct AbMma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct AbMma<f16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int AB_MB = 13;                         // row blocks of 16: maps of 193 .. 208 pixels (14 x 14)
constexpr int AB_NB = 3;                          // column blocks of 16 per wave: 8 waves x 48 = 384 channels

template <typename T, int C>
__global__ void __launch_bounds__(512, 1) as_block_kernel(const AsBlockArgs p) {
    static_assert(C == 8 * 16 * AB_NB, "8 waves x 3 blocks of 16 channels");
    constexpr int PITCH = 2 * C + 16;             // bytes per staged row
    constexpr int NOCT = C / 8;                   // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int HW = p.H * p.W;
    float* const red = reinterpret_cast<float*>(smem + (size_t)HW * PITCH);     // [8 waves][2] + (mu, rstd)
    float* const tab = red + 32;                                                 // [2][C]: scale / shift of norm1a
    const int GS = (C + p.ks - 1) / p.ks;         // channels per shift group
    const int n0 = wave * 16 * AB_NB;
    const float inv_cnt = 1.0f / ((float)HW * (float)C);

    // workgroup statistics of what the threads accumulated: lanes by shuffles, the eight waves in order by one thread (fp64)
    auto wg_stats = [&](float s1, float s2, float& mu, float& rs) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        __syncthreads();                                         // (red[] of the previous reduction has been read by everyone)
        if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) { a += (double)red[2 * w8]; b += (double)red[2 * w8 + 1]; }
            const double m = a * (double)inv_cnt;
            double var = b * (double)inv_cnt - m * m;
            var = var > 0.0 ? var : 0.0;
            red[16] = (float)m;
            red[17] = 1.0f / __builtin_sqrtf((float)var + p.eps);
        }
        __syncthreads();
        mu = red[16];
        rs = red[17];
    };

    // one product of the workgroup: acc[mb][nb] = (W rows n0 + 16 nb ..) x (LDS rows 16 mb ..), K = C.
    // hipcc left to itself reads one activation fragment, waits for it, multiplies, reads the next (and re-derives the "next step's" weight
    // fragments into loads at the top of the step that uses them): every LDS and L2 latency in line.  So the weight fragments are asm loads
    // waited for by COUNT (two register sets, one step ahead; vector-memory operations retire in order, so a compiler-inserted one in between
    // only makes the wait conservative), and the activation fragments go through a ring of four reads issued three blocks ahead.
    f32x4 acc[AB_MB][AB_NB];
    auto wload = [&](const T* ptr) {
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    auto product = [&](const T* __restrict__ Wm) {
#pragma unroll
        for (int mb = 0; mb < AB_MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < AB_NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* wrow[AB_NB];
#pragma unroll
        for (int nb = 0; nb < AB_NB; ++nb) wrow[nb] = Wm + (size_t)(n0 + nb * 16 + l15) * p.ldw + kq * 8;
        unsigned xoff[AB_MB];
#pragma unroll
        for (int mb = 0; mb < AB_MB; ++mb) {
            int r = mb * 16 + l15;
            r = r < HW ? r : HW - 1;                             // (the rows past the map: any readable row, their results are dropped)
            xoff[mb] = (unsigned)r * PITCH + kq * 16;
        }
        auto step = [&](const u32x4 (&wf)[AB_NB], const int k0) {
            u32x4 xr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[i] = *reinterpret_cast<const u32x4*>(smem + xoff[i] + k0 * 2);
#pragma unroll
            for (int mb = 0; mb < AB_MB; ++mb) {
#pragma unroll
                for (int nb = 0; nb < AB_NB; ++nb) acc[mb][nb] = AbMma<T>::run(wf[nb], xr[mb & 3], acc[mb][nb]);
                if (mb + 4 < AB_MB) xr[mb & 3] = *reinterpret_cast<const u32x4*>(smem + xoff[mb + 4] + k0 * 2);
            }
        };
        u32x4 wa[AB_NB], wb[AB_NB];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (nothing of the compiler's is in flight when the counting starts)
#pragma unroll
        for (int nb = 0; nb < AB_NB; ++nb) wa[nb] = wload(wrow[nb]);
#pragma unroll 1
        for (int k0 = 0; k0 < C; k0 += 64) {
#pragma unroll
            for (int nb = 0; nb < AB_NB; ++nb) wb[nb] = wload(wrow[nb] + k0 + 32);
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]) : : "memory");
            step(wa, k0);
            const int kn = k0 + 64 < C ? k0 + 64 : k0;           // (the last pair re-reads its own fragments: unused)
#pragma unroll
            for (int nb = 0; nb < AB_NB; ++nb) wa[nb] = wload(wrow[nb] + kn);
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]) : : "memory");
            step(wb, k0 + 32);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]) : : "memory");
    };
    // a lane's piece of block (mb, nb): pixel m = 16 mb + l15, channels n = n0 + 16 nb + 4 kq .. + 3
    auto pack4 = [&](const float (&v)[4]) {
        T e[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
        u32x2 o;
        __builtin_memcpy(&o, e, 8);
        return o;
    };
    auto sums4 = [&](const u32x2 o, float& s1, float& s2) {
        T e[4];
        __builtin_memcpy(e, &o, 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float f = to_f32(e[r]);
            s1 += f;
            s2 = __builtin_fmaf(f, f, s2);
        }
    };
    // LDS <- scratch U read through the axial shift along `dim` (2: rows, 3: columns): out[p, c] = U[p + s(c)], s(c) = ks / 2 - c / GS, zero outside
    // The staging loops: a thread's chunks (<= NCH of 16 bytes) are ALL requested before the first is used -- one memory latency per phase
    // (left as one load per iteration hipcc waits for every load on the spot: 19 round trips per thread and phase).
    constexpr int NCH = (16 * AB_MB * NOCT + 511) / 512;
    const int total = HW * NOCT;
    auto chunk_of = [&](const int j, int& px, int& c0) {
        int idx = tid + j * 512;
        const bool live = idx < total;
        idx = live ? idx : total - 1;
        px = idx / NOCT;
        c0 = (idx - px * NOCT) * 8;
        return live;
    };
    // LDS tile <- global rows (HW x C), optionally with the statistics of what is staged
    auto global_to_lds = [&](const T* __restrict__ src, float* s1, float* s2) {
        u32x4 v[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int px, c0;
            chunk_of(j, px, c0);
            v[j] = *reinterpret_cast<const u32x4*>(src + (size_t)px * C + c0);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int px, c0;
            if (chunk_of(j, px, c0)) {
                *reinterpret_cast<u32x4*>(smem + (size_t)px * PITCH + c0 * 2) = v[j];
                if (s1) chunk_sums<T>(v[j], *s1, *s2);
            }
        }
    };
    // global rows <- LDS tile
    auto lds_to_global = [&](T* __restrict__ dst) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int px, c0;
            if (chunk_of(j, px, c0)) *reinterpret_cast<u32x4*>(dst + (size_t)px * C + c0) = *reinterpret_cast<const u32x4*>(smem + (size_t)px * PITCH + c0 * 2);
        }
    };
    // LDS <- scratch U read through the axial shift along `dim` (2: rows, 3: columns): out[p, c] = U[p + s(c)], s(c) = ks / 2 - c / GS, zero outside;
    // a chunk of 8 channels that straddles two shift groups (GS = 77: 4 of the 48 chunk columns) is assembled from two loads
    auto stage_shifted = [&](const T* __restrict__ ub, const int dim) {
        u32x4 v[NCH], v1[NCH];
        const int lim = dim == 2 ? p.H : p.W;
        const int step = dim == 2 ? p.W * C : C;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int px, c0;
            chunk_of(j, px, c0);
            const int y = px / p.W, xx = px - y * p.W;
            const int pos = dim == 2 ? y : xx;
            const T* src = ub + (size_t)px * C + c0;
            const int g0 = c0 / GS, g1 = (c0 + 7) / GS;
            const int s0 = p.ks / 2 - g0, s1 = p.ks / 2 - g1;
            const bool in0 = pos + s0 >= 0 && pos + s0 < lim, in1 = pos + s1 >= 0 && pos + s1 < lim;
            v[j] = *reinterpret_cast<const u32x4*>(src + (in0 ? (ptrdiff_t)s0 * step : 0));       // (a readable address + a select below: no branch)
            v1[j] = v[j];
            if (g1 != g0) v1[j] = *reinterpret_cast<const u32x4*>(src + (in1 ? (ptrdiff_t)s1 * step : 0));
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int px, c0;
            if (chunk_of(j, px, c0)) {
                const int y = px / p.W, xx = px - y * p.W;
                const int pos = dim == 2 ? y : xx;
                const int g0 = c0 / GS, g1 = (c0 + 7) / GS;
                const int s0 = p.ks / 2 - g0, s1 = p.ks / 2 - g1;
                const bool in0 = pos + s0 >= 0 && pos + s0 < lim, in1 = pos + s1 >= 0 && pos + s1 < lim;
                const int nb0 = g1 != g0 ? g1 * GS - c0 : 8;     // elements of the chunk that belong to group g0
                unsigned short a[8], bq[8];
                __builtin_memcpy(a, &v[j], 16);
                __builtin_memcpy(bq, &v1[j], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = e < nb0 ? (in0 ? a[e] : (unsigned short)0) : (in1 ? bq[e] : (unsigned short)0);
                u32x4 o;
                __builtin_memcpy(&o, a, 16);
                *reinterpret_cast<u32x4*>(smem + (size_t)px * PITCH + c0 * 2) = o;
            }
        }
    };

    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
        T* const xb = reinterpret_cast<T*>(p.x) + (size_t)b * HW * C;
        T* const ub = reinterpret_cast<T*>(p.u) + (size_t)b * HW * C;
        T* const vb = reinterpret_cast<T*>(p.v) + (size_t)b * HW * C;
        float mu, rs;
        __syncthreads();                                         // (the previous sample's last product is done with the LDS)
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[0] = (long long)__builtin_readcyclecounter();
        // ---- P0: x -> LDS, statistics of x
        {
            float s1 = 0.f, s2 = 0.f;
            global_to_lds(xb, &s1, &s2);
            wg_stats(s1, s2, mu, rs);
        }
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[1] = (long long)__builtin_readcyclecounter();
        // ---- P1: conv1 with norm1 folded; t -> LDS; statistics of t
        product(reinterpret_cast<const T*>(p.w1));
        __syncthreads();                                         // every wave is done reading the operand
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[2] = (long long)__builtin_readcyclecounter();
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < AB_NB; ++nb) {
                const int n = n0 + nb * 16 + 4 * kq;
                const f32x4 cs = *reinterpret_cast<const f32x4*>(p.cs1 + n), bb = *reinterpret_cast<const f32x4*>(p.b1 + n);
#pragma unroll
                for (int mb = 0; mb < AB_MB; ++mb) {
                    const int m = mb * 16 + l15;
                    const f32x4 a = acc[mb][nb];
                    const float v[4] = {(a.x - mu * cs.x) * rs + bb.x, (a.y - mu * cs.y) * rs + bb.y, (a.z - mu * cs.z) * rs + bb.z,
                                        (a.w - mu * cs.w) * rs + bb.w};
                    const u32x2 o = pack4(v);
                    if (mb + 1 < AB_MB || m < HW) {
                        sums4(o, s1, s2);
                        *reinterpret_cast<u32x2*>(smem + (size_t)m * PITCH + n * 2) = o;
                    }
                }
            }
            wg_stats(s1, s2, mu, rs);
        }
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[3] = (long long)__builtin_readcyclecounter();
        // ---- P2: u = gelu(GroupNorm(t)) -> scratch U   (scale / shift per channel through an LDS table: no global load in the loop)
        if (tid < C) {
            const float sc = rs * p.ag[tid];
            tab[tid] = sc;
            tab[C + tid] = p.ab[tid] - mu * sc;
        }
        __syncthreads();
        for (int idx = tid; idx < HW * NOCT; idx += 512) {
            const int px = idx / NOCT, c0 = (idx - px * NOCT) * 8;
            const u32x4 tv = *reinterpret_cast<const u32x4*>(smem + (size_t)px * PITCH + c0 * 2);
            T e[8];
            __builtin_memcpy(e, &tv, 16);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(tab + c0), g1 = *reinterpret_cast<const f32x4*>(tab + c0 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(tab + C + c0), h1 = *reinterpret_cast<const f32x4*>(tab + C + c0 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            f32x2 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = f32x2{__builtin_fmaf(to_f32(e[2 * k]), gg[2 * k], hh[2 * k]), __builtin_fmaf(to_f32(e[2 * k + 1]), gg[2 * k + 1], hh[2 * k + 1])};
            gelu_pk_n<T, 4>(q);
            T o[8] = {from_f32<T>(q[0].x), from_f32<T>(q[0].y), from_f32<T>(q[1].x), from_f32<T>(q[1].y),
                      from_f32<T>(q[2].x), from_f32<T>(q[2].y), from_f32<T>(q[3].x), from_f32<T>(q[3].y)};
            u32x4 ov;
            __builtin_memcpy(&ov, o, 16);
            *reinterpret_cast<u32x4*>(ub + (size_t)px * C + c0) = ov;
        }
        __syncthreads();                                         // U is written (workgroup scope: one CU, one L1) and the LDS tile is free
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[4] = (long long)__builtin_readcyclecounter();
        // ---- P3: conv2_1 on the W-shifted u; y1 -> scratch V
        stage_shifted(ub, 3);
        __syncthreads();
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[5] = (long long)__builtin_readcyclecounter();
        product(reinterpret_cast<const T*>(p.w21));
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[6] = (long long)__builtin_readcyclecounter();
        __syncthreads();                                         // the operand is dead: y1 goes through the tile into whole-line stores
#pragma unroll
        for (int nb = 0; nb < AB_NB; ++nb) {
            const int n = n0 + nb * 16 + 4 * kq;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b21 + n);
#pragma unroll
            for (int mb = 0; mb < AB_MB; ++mb) {
                const int m = mb * 16 + l15;
                const f32x4 a = acc[mb][nb];
                f32x2 q[2] = {f32x2{a.x + bb.x, a.y + bb.y}, f32x2{a.z + bb.z, a.w + bb.w}};
                gelu_pk_n<T, 2>(q);
                const float v[4] = {q[0].x, q[0].y, q[1].x, q[1].y};
                if (mb + 1 < AB_MB || m < HW) *reinterpret_cast<u32x2*>(smem + (size_t)m * PITCH + n * 2) = pack4(v);
            }
        }
        __syncthreads();
        lds_to_global(vb);
        __syncthreads();                                         // the tile is free again (and V is written: one CU, one L1)
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[7] = (long long)__builtin_readcyclecounter();
        // ---- P4: conv2_2 on the H-shifted u; s = round(gelu(.)) + y1 -> LDS; statistics of s
        stage_shifted(ub, 2);
        __syncthreads();
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[8] = (long long)__builtin_readcyclecounter();
        product(reinterpret_cast<const T*>(p.w22));
        __syncthreads();
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[9] = (long long)__builtin_readcyclecounter();
        global_to_lds(vb, nullptr, nullptr);                      // y1 back into the (dead) tile: a lane then finds its own pieces in the LDS
        __syncthreads();
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < AB_NB; ++nb) {
                const int n = n0 + nb * 16 + 4 * kq;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b22 + n);
#pragma unroll
                for (int mb = 0; mb < AB_MB; ++mb) {
                    const int m = mb * 16 + l15;
                    const f32x4 a = acc[mb][nb];
                    f32x2 q[2] = {f32x2{a.x + bb.x, a.y + bb.y}, f32x2{a.z + bb.z, a.w + bb.w}};
                    gelu_pk_n<T, 2>(q);
                    if (mb + 1 < AB_MB || m < HW) {
                        const float y2[4] = {q[0].x, q[0].y, q[1].x, q[1].y};
                        const u32x2 r2 = pack4(y2);                                              // y2 = round(gelu(.)), as the GEMM it replaces stores it
                        u32x2* const cell = reinterpret_cast<u32x2*>(smem + (size_t)m * PITCH + n * 2);
                        const u32x2 r1 = *cell;
                        T e1[4], e2[4];
                        __builtin_memcpy(e1, &r1, 8);
                        __builtin_memcpy(e2, &r2, 8);
                        const float sv[4] = {to_f32(e2[0]) + to_f32(e1[0]), to_f32(e2[1]) + to_f32(e1[1]), to_f32(e2[2]) + to_f32(e1[2]),
                                             to_f32(e2[3]) + to_f32(e1[3])};
                        const u32x2 o = pack4(sv);
                        sums4(o, s1, s2);
                        *cell = o;
                    }
                }
            }
            wg_stats(s1, s2, mu, rs);
        }
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[10] = (long long)__builtin_readcyclecounter();
        // ---- P5: conv3 with norm2 folded + bias + residual -> x; statistics of the new x
        product(reinterpret_cast<const T*>(p.w3));
        if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[11] = (long long)__builtin_readcyclecounter();
        __syncthreads();                                         // the operand is dead: the residual comes in, the result leaves, through the tile
        global_to_lds(xb, nullptr, nullptr);
        __syncthreads();
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < AB_NB; ++nb) {
                const int n = n0 + nb * 16 + 4 * kq;
                const f32x4 cs = *reinterpret_cast<const f32x4*>(p.cs3 + n), bb = *reinterpret_cast<const f32x4*>(p.b3 + n);
#pragma unroll
                for (int mb = 0; mb < AB_MB; ++mb) {
                    const int m = mb * 16 + l15;
                    if (mb + 1 < AB_MB || m < HW) {
                        const f32x4 a = acc[mb][nb];
                        u32x2* const cell = reinterpret_cast<u32x2*>(smem + (size_t)m * PITCH + n * 2);
                        const u32x2 xr = *cell;
                        T ex[4];
                        __builtin_memcpy(ex, &xr, 8);
                        const float v[4] = {(a.x - mu * cs.x) * rs + bb.x + to_f32(ex[0]), (a.y - mu * cs.y) * rs + bb.y + to_f32(ex[1]),
                                            (a.z - mu * cs.z) * rs + bb.z + to_f32(ex[2]), (a.w - mu * cs.w) * rs + bb.w + to_f32(ex[3])};
                        const u32x2 o = pack4(v);
                        sums4(o, s1, s2);
                        *cell = o;
                    }
                }
            }
            __syncthreads();
            lds_to_global(xb);
            if (p.mean_out) {
                wg_stats(s1, s2, mu, rs);
                if (tid == 0) { p.mean_out[b] = mu; p.rstd_out[b] = rs; }
            }
            if (p.prof && blockIdx.x == 0 && tid == 0 && b == (int)blockIdx.x) p.prof[12] = (long long)__builtin_readcyclecounter();
        }
    }
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_as_block_supported(int dtype, int H, int W, int C, int kernel_size) {
    const int HW = H * W;
    return (dtype == MLPK_F16 || dtype == MLPK_BF16) && C == 384 && HW > 16 * (AB_MB - 1) && HW <= 16 * AB_MB && kernel_size >= 3 && (kernel_size & 1) &&
           (size_t)HW * (2 * C + 16) + 128 + 8 * C <= 160 * 1024;
}

extern "C" int mlpk_as_block(int dtype, void* x, void* u, void* v, int B, int H, int W, int C, int kernel_size, const void* w1, const float* b1,
                             const float* cs1, const float* ag, const float* ab, const void* w21, const float* b21, const void* w22, const float* b22,
                             const void* w3, const float* b3, const float* cs3, int ldw, float eps, float* mean_out, float* rstd_out, void* stream) {
    if (!x || !u || !v || !w1 || !b1 || !cs1 || !ag || !ab || !w21 || !b21 || !w22 || !b22 || !w3 || !b3 || !cs3) return MLPK_ENULL;
    if ((mean_out == nullptr) != (rstd_out == nullptr)) return MLPK_ENULL;
    if (B <= 0 || !mlpk_as_block_supported(dtype, H, W, C, kernel_size) || ldw < C || ldw % 8) return MLPK_ESHAPE;
    if (((uintptr_t)x | (uintptr_t)u | (uintptr_t)v | (uintptr_t)w1 | (uintptr_t)w21 | (uintptr_t)w22 | (uintptr_t)w3 | (uintptr_t)b1 | (uintptr_t)cs1 |
         (uintptr_t)ag | (uintptr_t)ab | (uintptr_t)b21 | (uintptr_t)b22 | (uintptr_t)b3 | (uintptr_t)cs3) & 15)
        return MLPK_EALIGN;
    AsBlockArgs a{x, u, v, w1, w21, w22, w3, b1, cs1, ag, ab, b21, b22, b3, cs3, mean_out, rstd_out, B, H, W, ldw, kernel_size, eps, nullptr};
    static const bool prof_on = getenv("MLPK_AS_BLOCK_PROF") != nullptr;
    static long long* prof_dev = nullptr;
    if (prof_on) {
        if (!prof_dev && hipMalloc(&prof_dev, 16 * sizeof(long long)) != hipSuccess) prof_dev = nullptr;
        a.prof = prof_dev;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int lds = H * W * (2 * C + 16) + 128 + 8 * C;
    int dev = 0, cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)(B < cu ? B : cu);
    hipError_t e;
    if (dtype == MLPK_BF16) {
        auto k = as_block_kernel<bf16_t, 384>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
    } else {
        auto k = as_block_kernel<f16_t, 384>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
    }
    MLPK_LAUNCH_CHECK();
    if (prof_on && prof_dev) {
        long long h[16];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, prof_dev, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            static const char* names[12] = {"P0 x->LDS+stats", "P1 product", "P1 epilogue+stats", "(-)", "P2 u->U", "P3 stage", "P3 product", "P3 epilogue",
                                            "P4 stage", "P4 product", "P4 epilogue+stats", "P5 product"};
            fprintf(stderr, "as_block cycles:");
            for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %lld |", names[i], h[i + 1] - h[i]);
            fprintf(stderr, " total %lld\n", h[12] - h[0]);
        }
    }
    return 0;
}
