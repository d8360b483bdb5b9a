// AS-MLP's axial-shift core in ONE kernel (as_mlp.py:84-93; utils/shift_cuda.py:49-69), round 4:
//
//     y[b,h,w,:] = gelu(W21 . sh_W(u)[b,h,w,:] + b21) + gelu(W22 . sh_H(u)[b,h,w,:] + b22),     u = gelu(GroupNorm(1,C)(t))
//     sh_W(u)[b,h,w,c] = u[b, h, w + s(c), c],  sh_H(u)[b,h,w,c] = u[b, h + s(c), w, c],  s(c) = k/2 - c / ceil(C/k),  zero outside the map
//
// Before: mlpk_norm_shift_nhwc read t and WROTE both shifted copies of u (2 x the tensor), then two GEMM launches read one copy each,
// the second also re-reading the first one's output as its residual: 9 tensor passes over HBM per block where the operation needs 2
// (read t, write y) -- at AS-MLP-T's first two stages (C = 96 / 192, 56^2 / 28^2 maps: K = N = C GEMMs of pure traffic) that was
// norm_shift_vec 15.8 % + most of the s3-tile time of the model.  Here a workgroup owns a band of TH image rows with its halo of
// k/2 rows / columns:
//   1. stage: every thread reads 16-byte pieces of t, applies the GroupNorm affine (per-sample mean / rstd given) and the GELU ONCE per
//      element, rounds, and writes u into an LDS band [TH + 4][W + 4] pixels x (2 C + 16) bytes -- zeros outside the map, so the
//      shifts need no bounds logic afterwards; the 16 bytes of padding per pixel put 16 consecutive pixels on distinct banks;
//   2. multiply: a task = 32 pixels x 32 output channels of BOTH convolutions on v_mfma_f32_32x32x16.  The shift is applied when the
//      activation fragment is READ from LDS: a lane's 8 consecutive channels of a k-step come from the pixel s(c) columns (rows)
//      away; where the 8 channels straddle two shift groups (ceil(C/k) is 20 / 39: not a multiple of 8) the fragment is assembled
//      from two reads with a constant bit mask.  Weight fragments come straight from global memory (18 / 73 KiB per matrix: L1 / L2);
//   3. epilogue in the order of the kernels it replaces, so the result is BIT-EQUAL to them: y1 = round(gelu(acc1 + b21)),
//      y2 = round(gelu(acc2 + b22)), y = round(y2 + y1); staged through a per-wave LDS tile into 16-byte stores.
// HBM traffic per block: read t once (halo rows again, from L2), write y once.
#include "mlpk_common.h"

namespace mlpk {

struct AsConvArgs {
    const void* t;        // (B*H*W, C) conv1 output, channel-last
    void* y;              // (B*H*W, C)
    const float* mean;    // per sample
    const float* rstd;
    const float* gamma;   // per channel (C)
    const float* beta;
    const void* w1;       // (C, ldw) conv2_1 weight (out, in), K-contiguous  -> the W-shifted operand
    const void* w2;       // conv2_2 -> the H-shifted operand
    const float* b1;
    const float* b2;
    int B, H, W, ldw, TH, bands, seg_rows;
    float* part;          // optional by-product: (sum, sum of squares) of the values stored by step s of image b at [2 (b * steps + s)], or NULL
    int steps;            // steps of TH rows per image
    float* mean_out;      // with part: GroupNorm(1, C) statistics of y per image, finished inside the kernel
    float* rstd_out;
    int* counter;         // with part and bands > 1: one zeroed counter per image (left zeroed)
    float eps;
};

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32<f16_t> {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int ASC_STG_PITCH = 80;                 // per-wave output staging: 32 pixels x (32 channels x 2 B + 16)
constexpr int ASC_STG_BYTES = 32 * ASC_STG_PITCH;

// Round 6: the workgroup is PERSISTENT over (image, row segment) units and walks a unit top to bottom in steps of TH rows through an LDS
// RING of TH + 4 staged rows (row gy of the unit at slot (gy - ra + 2) mod R): a step stages only its TH NEW rows -- the four halo
// rows it shares with its neighbours are already there (round 4 staged a band's halo again: 11 rows read and 11 rows of GELU for 7
// rows of output) -- and the rows of the NEXT step (or of the next unit) are requested into registers BEFORE the matrix phase of the
// current one and written to the ring after it, so their memory latency runs under the MFMAs instead of in front of them (round 4: load,
// barrier, compute, strictly one after the other, one workgroup per CU: 202 us at stage 1 for 308 MB).  The zero columns left and
// right of the map are written once per launch; a row outside the map is staged as zeros.  Same arithmetic per element, same bits.
template <typename T, int C, int KS, int NPRE>
__global__ void __launch_bounds__(512, 1) as_conv2_kernel(const AsConvArgs p) {
    typedef typename Mfma32<T>::f32x16 f32x16;
    constexpr int GS = (C + KS - 1) / KS;          // channels per shift group
    constexpr int P2 = KS / 2;                     // halo
    constexpr int PITCH = 2 * C + 16;              // bytes per staged pixel
    constexpr int NOCT = C / 8;
    constexpr int NC = C / 32;                     // output-channel chunks of a task
    constexpr int NKS = C / 16;
    constexpr int PL = 512 / NOCT;                 // pixels per staging sweep
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, Wp = W + 2 * P2, TH = p.TH;
    const int R = TH + 2 * P2;                                   // ring rows
    const int rowb = Wp * PITCH;                                 // bytes of a staged row
    char* const stg = smem + (size_t)R * rowb + wave * ASC_STG_BYTES;
    const int nunits = p.B * p.bands;                            // bands = row segments per image here
    const int seg_rows = p.seg_rows;                             // rows of a segment (a multiple of TH; the last one may be shorter)

    // zero the ring once: the left / right halo columns are never written again
    for (int i = tid; i < (R * rowb) / 16; i += 512) *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- staging geometry: thread = (octet of channels, pixel lane); a stage set is `nrows` whole rows of W pixels, swept PL pixels at a time
    const int oct = tid % NOCT, pl = tid / NOCT;
    const bool stager = pl < PL;
    const int pl_r = pl / W, pl_x = pl - pl_r * W;               // pixel pl of a stage set, and the step of a sweep
    const int pl_dr = PL / W, pl_dx = PL - pl_dr * W;
    // per-channel scale / shift of the image being staged, as tables in LDS behind the staging tiles (registers are what this kernel is short of)
    float* const scs = reinterpret_cast<float*>(smem + (size_t)R * rowb + 8 * ASC_STG_BYTES);
    int tab_img = -1;
    // ... and the two bias vectors, once per launch: read from global memory inside the epilogue (round 4) every one of a task's eight
    // 16-byte bias loads was waited for on the spot -- eight memory round trips per task, most of the kernel's time
    float* const btab = scs + 2 * C;
    for (int c = tid; c < 2 * C; c += 512) btab[c] = c < C ? p.b1[c] : p.b2[c - C];

    // descriptor of a stage set (workgroup-uniform): image, first row gy0 (may be negative), number of rows, ring slot of the first row
    struct Set { int img, gy0, nrows, slot0; };
    auto unit_rows = [&](const int unit, int& img, int& ra, int& rb) {
        img = unit / p.bands;
        const int seg = unit - img * p.bands;
        ra = seg * seg_rows;
        rb = ra + seg_rows < H ? ra + seg_rows : H;
    };
    u32x4 raw[NPRE > 0 ? NPRE : 1];
    // request the first NPRE sweeps of a set into raw[] (no wait); rows outside the map are not read
    auto request = [&](const Set& st) {
        if (!stager) return;
        const T* __restrict__ tin = reinterpret_cast<const T*>(p.t) + (size_t)st.img * H * W * C;
        const int npix = st.nrows * W;
        int r = pl_r, gx = pl_x;                     // (row, column) of pixel pl + k PL, advanced by (PL / W, PL % W) per sweep: no division
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int pp = pl + k * PL;
            const int gy = st.gy0 + r;
            const bool in = pp < npix && gy >= 0 && gy < H;
            // (clamped address, not a predicated load: nothing between the loads that could make hipcc wait for one before the next)
            const size_t off = in ? ((size_t)gy * W + gx) * C + oct * 8 : (size_t)oct * 8;
            raw[k] = *reinterpret_cast<const u32x4*>(tin + off);
            gx += pl_dx; r += pl_dr;
            if (gx >= W) { gx -= W; ++r; }
        }
    };
    // write a set into the ring: u = round(gelu(t * sc + sh)); sweeps beyond NPRE (a unit's first set is TH + 4 rows) are loaded here
    auto stage = [&](const Set& st) {
        if (st.img != tab_img) {                     // (workgroup-uniform)
            const float mu = p.mean[st.img], rs = p.rstd[st.img];
            for (int c = tid; c < C; c += 512) {
                const float g = p.gamma[c];
                scs[c] = rs * g;
                scs[C + c] = p.beta[c] - mu * rs * g;
            }
            tab_img = st.img;
            __syncthreads();
        }
        if (!stager) return;
        const T* __restrict__ tin = reinterpret_cast<const T*>(p.t) + (size_t)st.img * H * W * C;
        const int npix = st.nrows * W;
        auto put = [&](const int r, const int gx, const u32x4 rawv) {
            const int gy = st.gy0 + r;
            int slot = st.slot0 + r;
            slot = slot >= R ? slot - R : slot;
            u32x4 o = {0u, 0u, 0u, 0u};
            if (gy >= 0 && gy < H) {
                T v8[8], e8[8];
                __builtin_memcpy(v8, &rawv, 16);
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scs + oct * 8), sc1 = *reinterpret_cast<const f32x4*>(scs + oct * 8 + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(scs + C + oct * 8), sh1 = *reinterpret_cast<const f32x4*>(scs + C + oct * 8 + 4);
                const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
                const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
                f32x2 g2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) g2[e] = f32x2{__builtin_fmaf(to_f32(v8[2 * e]), sc[2 * e], sh[2 * e]), __builtin_fmaf(to_f32(v8[2 * e + 1]), sc[2 * e + 1], sh[2 * e + 1])};
                gelu_pk_n<T, 4>(g2);                 // (four pairs abreast: the same bits as gelu16_f per element)
#pragma unroll
                for (int e = 0; e < 4; ++e) { e8[2 * e] = from_f32<T>(g2[e].x); e8[2 * e + 1] = from_f32<T>(g2[e].y); }
                __builtin_memcpy(&o, e8, 16);
            }
            *reinterpret_cast<u32x4*>(smem + (size_t)slot * rowb + (size_t)(gx + P2) * PITCH + oct * 16) = o;
        };
        int r = pl_r, gx = pl_x;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int pp = pl + k * PL;
            if (pp < npix) put(r, gx, raw[k]);
            gx += pl_dx; r += pl_dr;
            if (gx >= W) { gx -= W; ++r; }
        }
        // the rest of a long set, four sweeps at a time with their loads abreast
        for (int pp0 = pl + NPRE * PL; pp0 < npix; pp0 += 4 * PL) {
            u32x4 rr[4];
            int rk[4], xk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pp = pp0 + k * PL;
                rk[k] = r; xk[k] = gx;
                const int gy = st.gy0 + r;
                const bool in = pp < npix && gy >= 0 && gy < H;
                rr[k] = *reinterpret_cast<const u32x4*>(tin + (in ? ((size_t)gy * W + gx) * C + oct * 8 : (size_t)oct * 8));
                gx += pl_dx; r += pl_dr;
                if (gx >= W) { gx -= W; ++r; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (pp0 + k * PL < npix) put(rk[k], xk[k], rr[k]);
        }
    };

    const int l31 = lane & 31, hh = lane >> 5;
    const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
    const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
    u32x4 wfr[2][NKS];
    int nc_have = -1;

    int unit = blockIdx.x;
    if (unit >= nunits) return;
    int img, ra, rb;
    unit_rows(unit, img, ra, rb);
    Set cur = {img, ra - P2, R, 0};                  // a unit's first set: the whole ring, row ra - 2 at slot 0
    __syncthreads();                                 // (the zero fill)
    request(cur);
    stage(cur);
    int r0 = ra;                                     // first output row of the step
    int base = 0;                                    // ring slot of row r0 - 2
    double ds1 = 0.0, ds2 = 0.0;                     // (thread 0) the image's step pairs so far
    for (;;) {
        __syncthreads();                             // the rows of this step are staged
        // ---- what comes next: the TH new rows of the following step, or the first set of the next unit
        bool more = true, same_unit = r0 + TH < rb;
        Set nxt;
        int n_img = img, n_ra = ra, n_rb = rb;
        if (same_unit) {
            // rows r0 - 2 .. r0 + TH - 3 (slots base .. base + TH - 1) are dead after this step: rows r0 + TH + 2 .. r0 + 2 TH + 1 take them
            nxt = Set{img, r0 + TH + P2, TH, base};
        } else {
            const int nu = unit + (int)gridDim.x;
            more = nu < nunits;
            if (more) {
                unit_rows(nu, n_img, n_ra, n_rb);
                nxt = Set{n_img, n_ra - P2, R, 0};
            } else {
                nxt = Set{img, 0, 0, 0};
            }
        }
        if (more) request(nxt);

        // ---- the matrix phase of rows r0 .. r0 + th - 1
        float st1 = 0.f, st2 = 0.f;
        {
            const int th = rb - r0 < TH ? rb - r0 : TH;
            const int M = th * W;
            T* __restrict__ yout = reinterpret_cast<T*>(p.y) + ((size_t)img * H + r0) * W * C;
            const int npb = (M + 31) / 32;
            const int ntask = npb * NC;
            const int t_lo = (ntask * wave) / 8, t_hi = (ntask * (wave + 1)) / 8;
            for (int task = t_lo; task < t_hi; ++task) {
                const int nc = task / npb, pb = task - nc * npb;
                if (nc != nc_have) {
                    const T* const wr1 = w1 + (size_t)(nc * 32 + l31) * p.ldw + hh * 8;
                    const T* const wr2 = w2 + (size_t)(nc * 32 + l31) * p.ldw + hh * 8;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        wfr[0][ks] = *reinterpret_cast<const u32x4*>(wr1 + ks * 16);
                        wfr[1][ks] = *reinterpret_cast<const u32x4*>(wr2 + ks * 16);
                    }
                    nc_have = nc;
                }
                int m = pb * 32 + l31;
                m = m < M ? m : M - 1;                                  // (pixels past the step: computed on a valid pixel, not stored)
                const int py = m / W, px = m - py * W;
                // ring rows py - 2 .. py + 2 of this lane's pixel (row r0 + py + d at slot (base + py + 2 + d) mod R)
                int rof[2 * P2 + 1];
#pragma unroll
                for (int d = 0; d <= 2 * P2; ++d) {
                    int sl = base + py + d;
                    sl = sl >= R ? sl - R : sl;
                    sl = sl >= R ? sl - R : sl;
                    rof[d] = sl * rowb + (px + P2) * PITCH + hh * 16;
                }
                float yv[16];
#pragma unroll
                for (int conv = 0; conv < 2; ++conv) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const u32x4 wf = wfr[conv][ks];
                        // channels 16 ks + 8 hh .. + 7: shift groups of the first and the last one, for both lane halves (constants once unrolled)
                        const int c0 = 16 * ks;
                        const int ga0 = c0 / GS, gb0 = (c0 + 7) / GS;                    // hh = 0
                        const int ga1 = (c0 + 8) / GS, gb1 = (c0 + 15) / GS;             // hh = 1
                        // conv2_1: shift along W = a pixel of the centre row; conv2_2: along H = the same pixel of another ring row
                        const int oa = conv == 0 ? rof[P2] + (P2 - (hh ? ga1 : ga0)) * PITCH : (hh ? rof[2 * P2 - ga1] : rof[2 * P2 - ga0]);
                        u32x4 af = *reinterpret_cast<const u32x4*>(smem + oa + ks * 32);
                        if (ga0 != gb0 || ga1 != gb1) {
                            // a straddling octet (in at least one half): elements from index `first` on belong to the next group
                            const int ga = hh ? ga1 : ga0, gb = hh ? gb1 : gb0;
                            const int ob = conv == 0 ? rof[P2] + (P2 - gb) * PITCH : (hh ? rof[2 * P2 - gb1] : rof[2 * P2 - gb0]);
                            const u32x4 bf = *reinterpret_cast<const u32x4*>(smem + ob + ks * 32);
                            const int first = ga != gb ? gb * GS - (c0 + 8 * hh) : 8;     // 1 .. 7, or 8 = nothing from the second read
                            unsigned av[4] = {af.x, af.y, af.z, af.w};
                            const unsigned bv[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const unsigned mask = 2 * w + 1 < first ? 0xFFFFFFFFu : (2 * w < first ? 0x0000FFFFu : 0u);   // 1 bits: keep the first read
                                av[w] = (av[w] & mask) | (bv[w] & ~mask);
                            }
                            af = u32x4{av[0], av[1], av[2], av[3]};
                        }
                        acc = Mfma32<T>::run(wf, af, acc);
                    }
                    // epilogue of this convolution: lane = pixel l31, register r = output channel nc * 32 + 8 (r >> 2) + 4 hh + (r & 3)
                    const float* const bias = btab + conv * C;
                    // (the epilogue's LDS reads stay behind this convolution's MFMAs: hoisted to the top of the task they cost C = 192 its registers)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g2i = 0; g2i < 2; ++g2i) {
                        // two channel groups = four float pairs abreast through the GELU (the packed-f16 steps of two pairs alone left a
                        // wait state after every instruction: 90 s_nop per task); per element the same operation sequence, the same bits
                        f32x2 g2[4];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int g = 2 * g2i + q;
                            const f32x4 bz = *reinterpret_cast<const f32x4*>(bias + nc * 32 + 8 * g + 4 * hh);
                            g2[2 * q] = f32x2{acc[4 * g] + bz.x, acc[4 * g + 1] + bz.y};
                            g2[2 * q + 1] = f32x2{acc[4 * g + 2] + bz.z, acc[4 * g + 3] + bz.w};
                        }
                        gelu_pk_n<T, 4>(g2);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int g = 2 * g2i + q;
                            const float gv[4] = {g2[2 * q].x, g2[2 * q].y, g2[2 * q + 1].x, g2[2 * q + 1].y};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = to_f32(from_f32<T>(gv[r]));
                                yv[4 * g + r] = conv == 0 ? v : to_f32(from_f32<T>(v + yv[4 * g + r]));
                            }
                        }
                    }
                }
                if (p.part && pb * 32 + l31 < M) {              // by-product statistics of what is stored (the rounded values), this lane's pixel
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        st1 += yv[r];
                        st2 = __builtin_fmaf(yv[r], yv[r], st2);
                    }
                }
                // ---- store: [32 pixels][32 channels] through the wave's staging tile, then 16-byte pieces (pixel lane >> 2, + 16; piece lane & 3)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    T e4[4] = {from_f32<T>(yv[4 * g]), from_f32<T>(yv[4 * g + 1]), from_f32<T>(yv[4 * g + 2]), from_f32<T>(yv[4 * g + 3])};
                    u32x2 pk;
                    __builtin_memcpy(&pk, e4, 8);
                    *reinterpret_cast<u32x2*>(stg + l31 * ASC_STG_PITCH + (8 * g + 4 * hh) * 2) = pk;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pr = (lane >> 2) + 16 * k, pc = lane & 3;
                    const u32x4 o = *reinterpret_cast<const u32x4*>(stg + pr * ASC_STG_PITCH + pc * 16);
                    const int mm = pb * 32 + pr;
                    if (mm < M) *reinterpret_cast<u32x4*>(yout + (size_t)mm * C + nc * 32 + pc * 8) = o;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (p.part) {
            // GroupNorm(1, C) statistics of the sample as a by-product: one (sum, sum of squares) pair per STEP -- the step grid depends on the
            // map only, not on the batch or on how images are cut into units, so a sample's statistics do not depend on the batch it is in --
            // lanes by xor-shuffles, the eight waves in order by one thread; mlpk_stats_finalize_planar(group = steps) adds the steps up
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                st1 += __shfl_xor(st1, o);
                st2 += __shfl_xor(st2, o);
            }
            float* const red = btab + 2 * C;
            if (lane == 0) { red[2 * wave] = st1; red[2 * wave + 1] = st2; }
            __syncthreads();
            if (tid == 0) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) { a1 += red[2 * w8]; a2 += red[2 * w8 + 1]; }
                float* const dst = p.part + 2 * ((size_t)img * p.steps + r0 / TH);
                dst[0] = a1;
                dst[1] = a2;
                // ... and the statistics themselves, without a launch of their own: the step pairs of an image added in step order in
                // fp64 by ONE thread -- the workgroup that owns the whole image as it goes, else (small batches: an image cut into
                // segments) whichever workgroup finishes the image last, reading the others' pairs behind a device-scope fence
                if (p.mean_out) {
                    bool fin = false;
                    if (p.bands == 1) {
                        ds1 += (double)a1;
                        ds2 += (double)a2;
                        fin = !same_unit;
                    } else if (!same_unit) {
                        __threadfence();
                        fin = atomicAdd(p.counter + img, 1) == p.bands - 1;
                        if (fin) {
                            __threadfence();
                            ds1 = ds2 = 0.0;
                            const volatile float* pp = p.part + 2 * (size_t)img * p.steps;
                            for (int q = 0; q < p.steps; ++q) { ds1 += (double)pp[2 * q]; ds2 += (double)pp[2 * q + 1]; }
                            p.counter[img] = 0;
                        }
                    }
                    if (fin) {
                        const double inv = 1.0 / ((double)H * W * C);
                        const double mu = ds1 * inv;
                        double var = ds2 * inv - mu * mu;
                        var = var > 0.0 ? var : 0.0;
                        p.mean_out[img] = (float)mu;
                        p.rstd_out[img] = 1.0f / __builtin_sqrtf((float)var + p.eps);
                        ds1 = ds2 = 0.0;
                    }
                }
            }
        }
        if (!more) break;
        __syncthreads();                             // every wave is done reading the rows the next set overwrites
        stage(nxt);
        if (same_unit) {
            r0 += TH;
            base += TH;
            base = base >= R ? base - R : base;
        } else {
            unit += (int)gridDim.x;
            img = n_img; ra = n_ra; rb = n_rb;
            r0 = ra;
            base = 0;
        }
    }
}

// sweeps of a stage set held in registers across the matrix phase: what the register file leaves beside the weight fragments of both
// convolutions (2 x C / 16 x 4 registers; the f16 GELU polynomial needs a few more temporaries than the bf16 form) without a spill
template <typename T, int C> struct AsPre {
    static constexpr int value = C == 96 ? (dtype_of<T>::value == MLPK_F16 ? 6 : 10) : 0;       // (C = 192: the weight fragments leave no room -- hipcc parks the prefetch in scratch, i.e. waits for it)
};

// the step grid of a map: rows per step and steps per image -- a function of (dtype, C, H, W) alone, never of the batch (the by-product
// statistics are one pair per step)
template <typename T, int C>
static int as_conv2_steps(const int H, const int W, int& th_out) {
    constexpr int PITCH = 2 * C + 16;
    constexpr int NPRE = AsPre<T, C>::value;
    constexpr int PL = 512 / (C / 8);
    const int Wp = W + 4;
    const int budget = 160 * 1024 - 8 * ASC_STG_BYTES - C * 16 - 64;     // ring + 8 staging tiles + the scale / shift and bias tables + the waves' statistics
    int th = budget / (Wp * PITCH) - 4;              // ring = th + 4 rows
    if (th < 1) return 0;
    if (th > H) th = H;
    // a step's new rows should fit the prefetch registers where that leaves a step tall enough to feed eight waves (longer sets still
    // work: the sweeps beyond NPRE are loaded at staging time, four abreast)
    const int th_pre = (NPRE * PL) / W;
    if (th_pre >= 5 && th > th_pre) th = th_pre;
    // steps of equal height where possible
    const int steps = (H + th - 1) / th;
    th_out = (H + steps - 1) / steps;
    return steps;
}

template <typename T, int C>
static int as_conv2_launch(const AsConvArgs& a0, hipStream_t s) {
    AsConvArgs a = a0;
    constexpr int PITCH = 2 * C + 16;
    constexpr int NPRE = AsPre<T, C>::value;
    const int Wp = a.W + 4;
    int th = 0;
    const int steps = as_conv2_steps<T, C>(a.H, a.W, th);
    if (steps < 1) return MLPK_ESHAPE;
    a.TH = th;
    a.steps = steps;
    // row segments per image: one unit per CU and launch when the batch allows, more (each with its own halo) when it is small
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    int segs = 1;
    while (a.B * segs < cus && segs * 2 <= steps) segs *= 2;
    const int steps_per_seg = (steps + segs - 1) / segs;
    segs = (steps + steps_per_seg - 1) / steps_per_seg;
    a.bands = segs;
    a.seg_rows = steps_per_seg * th;
    const int units = a.B * segs;
    const int lds = (th + 4) * Wp * PITCH + 8 * ASC_STG_BYTES + C * 16 + 64;
    auto k = as_conv2_kernel<T, C, 5, NPRE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return MLPK_ESHAPE;
    hipLaunchKernelGGL(k, dim3((unsigned)(units < cus ? units : cus)), dim3(512), lds, s, a);
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_as_conv2_supported(int dtype, int H, int W, int C, int kernel_size) {
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return 0;
    if (kernel_size != 5 || (C != 96 && C != 192)) return 0;
    const int budget = 160 * 1024 - 8 * ASC_STG_BYTES - C * 16 - 64;
    return H >= 1 && W >= 1 && budget / ((W + 4) * (2 * C + 16)) - 4 >= 1;
}

extern "C" int mlpk_as_conv2_steps(int dtype, int H, int W, int C, int kernel_size) {
    if (!mlpk_as_conv2_supported(dtype, H, W, C, kernel_size)) return 0;
    int th = 0;
    if (dtype == MLPK_BF16) return C == 96 ? as_conv2_steps<bf16_t, 96>(H, W, th) : as_conv2_steps<bf16_t, 192>(H, W, th);
    return C == 96 ? as_conv2_steps<f16_t, 96>(H, W, th) : as_conv2_steps<f16_t, 192>(H, W, th);
}

static int as_conv2_run(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                        float* part, float* mean_out, float* rstd_out, int* counter, float eps, void* stream);

extern "C" int mlpk_as_conv2(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                             void* stream) {
    return as_conv2_run(dtype, t, y, B, H, W, C, kernel_size, mean, rstd, gamma, beta, w1, b1, w2, b2, ldw, nullptr, nullptr, nullptr, nullptr, 0.f, stream);
}

extern "C" int mlpk_as_conv2_stats(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                                   float* part, float* mean_out, float* rstd_out, int* counter, float eps, void* stream) {
    if (!part || !mean_out || !rstd_out || !counter) return MLPK_ENULL;
    if (mean_out == mean || rstd_out == rstd) return MLPK_ESHAPE;            // other workgroups still read the input's statistics
    if (((uintptr_t)part & 7) || ((uintptr_t)counter & 3)) return MLPK_EALIGN;
    return as_conv2_run(dtype, t, y, B, H, W, C, kernel_size, mean, rstd, gamma, beta, w1, b1, w2, b2, ldw, part, mean_out, rstd_out, counter, eps, stream);
}

static int as_conv2_run(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                        float* part, float* mean_out, float* rstd_out, int* counter, float eps, void* stream) {
    if (!t || !y || !mean || !rstd || !gamma || !beta || !w1 || !w2 || !b1 || !b2) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || ldw < C || ldw % 8) return MLPK_ESHAPE;
    if (!mlpk_as_conv2_supported(dtype, H, W, C, kernel_size)) return MLPK_ESHAPE;
    if (t == y) return MLPK_ESHAPE;                                          // a band reads its neighbours' rows
    if (((uintptr_t)t | (uintptr_t)y | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)gamma | (uintptr_t)beta) & 15) return MLPK_EALIGN;
    AsConvArgs a;
    a.t = t; a.y = y; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2;
    a.B = B; a.H = H; a.W = W; a.ldw = ldw; a.TH = 0; a.bands = 0; a.seg_rows = 0;
    a.part = part; a.steps = 0; a.mean_out = mean_out; a.rstd_out = rstd_out; a.counter = counter; a.eps = eps;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MLPK_BF16) return C == 96 ? as_conv2_launch<bf16_t, 96>(a, s) : as_conv2_launch<bf16_t, 192>(a, s);
    return C == 96 ? as_conv2_launch<f16_t, 96>(a, s) : as_conv2_launch<f16_t, 192>(a, s);
}
