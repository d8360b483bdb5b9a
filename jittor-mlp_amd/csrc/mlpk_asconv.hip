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
    int B, H, W, ldw, TH, bands;
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

template <typename T, int C, int KS>
__global__ void __launch_bounds__(512, 1) as_conv2_kernel(const AsConvArgs p) {
    typedef typename Mfma32<T>::f32x16 f32x16;
    constexpr int GS = (C + KS - 1) / KS;          // channels per shift group
    constexpr int P2 = KS / 2;                     // halo
    constexpr int PITCH = 2 * C + 16;              // bytes per staged pixel
    constexpr int NOCT = C / 8;
    constexpr int NC = C / 32;                     // output-channel chunks of a task
    constexpr int NKS = C / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.bands;
    const int band = blockIdx.x - b * p.bands;
    const int h0 = band * p.TH;
    const int th = p.H - h0 < p.TH ? p.H - h0 : p.TH;          // rows of this band
    const int W = p.W, Wp = W + 2 * P2, Hp = th + 2 * P2;
    const int M = th * W;                                        // output pixels of the band
    const T* __restrict__ tin = reinterpret_cast<const T*>(p.t) + (size_t)b * p.H * W * C;
    T* __restrict__ yout = reinterpret_cast<T*>(p.y) + ((size_t)b * p.H + h0) * W * C;
    char* const stg = smem + (size_t)(p.TH + 2 * P2) * Wp * PITCH + wave * ASC_STG_BYTES;

    // ---- 1. stage u = round(gelu(t * sc + sh)) with its zero halo.  Thread = (octet of channels, pixel lane): its 8 scale / shift pairs
    //         stay in registers, consecutive threads read consecutive 16-byte pieces of a pixel
    {
        constexpr int PL = 512 / NOCT;             // pixels in flight per sweep
        const int oct = tid % NOCT, pl = tid / NOCT;
        if (pl < PL) {
            const float mu = p.mean[b], rs = p.rstd[b];
            float sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = p.gamma[oct * 8 + e];
                sc[e] = rs * g;
                sh[e] = p.beta[oct * 8 + e] - mu * rs * g;
            }
            // eight pixels per sweep: all eight loads go out before the first value is used (one load per trip left the band's
            // ~14 trips waiting out a memory round trip each: 2/3 of the kernel's time in its first version)
            constexpr int UN = 8;
            for (int pp0 = pl; pp0 < Hp * Wp; pp0 += PL * UN) {
                u32x4 raw[UN];
                bool in[UN];
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const int pp = pp0 + k * PL;
                    const int py = pp / Wp, px = pp - py * Wp;
                    const int gy = h0 - P2 + py, gx = px - P2;
                    in[k] = pp < Hp * Wp && gy >= 0 && gy < p.H && gx >= 0 && gx < W;
                    raw[k] = u32x4{0u, 0u, 0u, 0u};
                    if (in[k]) raw[k] = *reinterpret_cast<const u32x4*>(tin + ((size_t)gy * W + gx) * C + oct * 8);
                }
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const int pp = pp0 + k * PL;
                    if (pp >= Hp * Wp) break;
                    u32x4 o = {0u, 0u, 0u, 0u};
                    if (in[k]) {
                        T v8[8], e8[8];
                        __builtin_memcpy(v8, &raw[k], 16);
                        f32x2 g2[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) g2[e] = f32x2{__builtin_fmaf(to_f32(v8[2 * e]), sc[2 * e], sh[2 * e]), __builtin_fmaf(to_f32(v8[2 * e + 1]), sc[2 * e + 1], sh[2 * e + 1])};
                        gelu_pk_n<T, 4>(g2);                 // (four pairs abreast: the same bits as gelu16_f per element)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { e8[2 * e] = from_f32<T>(g2[e].x); e8[2 * e + 1] = from_f32<T>(g2[e].y); }
                        __builtin_memcpy(&o, e8, 16);
                    }
                    *reinterpret_cast<u32x4*>(smem + (size_t)pp * PITCH + oct * 16) = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- 2. + 3. tasks (channel chunk nc, pixel block pb), nc-major, a contiguous share per wave: a wave's consecutive tasks share their
    //              channel chunk, whose weight fragments of BOTH convolutions stay in registers (2 x C/16 x 4) and are re-read only
    //              when the chunk changes -- loaded per task they were the other third of the first version's time (every MFMA behind
    //              its own L2 round trip)
    const int l31 = lane & 31, hh = lane >> 5;
    const int npb = (M + 31) / 32;
    const int ntask = npb * NC;
    const int t_lo = (ntask * wave) / 8, t_hi = (ntask * (wave + 1)) / 8;
    const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w1);
    const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
    u32x4 wfr[2][NKS];
    int nc_have = -1;
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
        m = m < M ? m : M - 1;                                  // (pixels past the band: computed on a valid pixel, not stored)
        const int py = m / W, px = m - py * W;
        const char* const pix = smem + ((size_t)(py + P2) * Wp + px + P2) * PITCH + hh * 16;     // this lane's pixel, its k-half
        float yv[16];
#pragma unroll
        for (int conv = 0; conv < 2; ++conv) {
            const int step = conv == 0 ? PITCH : Wp * PITCH;      // conv2_1: shift along W (a pixel), conv2_2: along H (a staged row)
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
                const int ga = hh ? ga1 : ga0, gb = hh ? gb1 : gb0;
                u32x4 af = *reinterpret_cast<const u32x4*>(pix + ks * 32 + (P2 - ga) * step);
                if (ga0 != gb0 || ga1 != gb1) {
                    // a straddling octet (in at least one half): elements from index `first` on belong to the next group
                    const u32x4 bf = *reinterpret_cast<const u32x4*>(pix + ks * 32 + (P2 - gb) * step);
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
            const float* const bias = conv == 0 ? p.b1 : p.b2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bz = *reinterpret_cast<const f32x4*>(bias + nc * 32 + 8 * g + 4 * hh);
                const float bb[4] = {bz.x, bz.y, bz.z, bz.w};
                f32x2 g2[2] = {f32x2{acc[4 * g] + bb[0], acc[4 * g + 1] + bb[1]}, f32x2{acc[4 * g + 2] + bb[2], acc[4 * g + 3] + bb[3]}};
                gelu_pk_n<T, 2>(g2);
                const float gv[4] = {g2[0].x, g2[0].y, g2[1].x, g2[1].y};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = to_f32(from_f32<T>(gv[r]));
                    yv[4 * g + r] = conv == 0 ? v : to_f32(from_f32<T>(v + yv[4 * g + r]));
                }
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

template <typename T, int C>
static int as_conv2_launch(const AsConvArgs& a0, hipStream_t s) {
    AsConvArgs a = a0;
    constexpr int PITCH = 2 * C + 16;
    const int Wp = a.W + 4;
    const int budget = 160 * 1024 - 8 * ASC_STG_BYTES;
    int th = budget / (Wp * PITCH) - 4;
    if (th < 1) return MLPK_ESHAPE;
    if (th > a.H) th = a.H;
    // bands of equal height where possible (a short last band costs a whole workgroup its halo)
    const int bands = (a.H + th - 1) / th;
    th = (a.H + bands - 1) / bands;
    a.TH = th;
    a.bands = bands;
    const int lds = (th + 4) * Wp * PITCH + 8 * ASC_STG_BYTES;
    auto k = as_conv2_kernel<T, C, 5>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3((unsigned)(a.B * bands)), dim3(512), lds, s, a);
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk

using namespace mlpk;

extern "C" int mlpk_as_conv2_supported(int dtype, int H, int W, int C, int kernel_size) {
    if (dtype != MLPK_F16 && dtype != MLPK_BF16) return 0;
    if (kernel_size != 5 || (C != 96 && C != 192)) return 0;
    const int budget = 160 * 1024 - 8 * ASC_STG_BYTES;
    return H >= 1 && W >= 1 && budget / ((W + 4) * (2 * C + 16)) - 4 >= 1;
}

extern "C" int mlpk_as_conv2(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                             void* stream) {
    if (!t || !y || !mean || !rstd || !gamma || !beta || !w1 || !w2 || !b1 || !b2) return MLPK_ENULL;
    if (B <= 0 || H <= 0 || W <= 0 || ldw < C || ldw % 8) return MLPK_ESHAPE;
    if (!mlpk_as_conv2_supported(dtype, H, W, C, kernel_size)) return MLPK_ESHAPE;
    if (t == y) return MLPK_ESHAPE;                                          // a band reads its neighbours' rows
    if (((uintptr_t)t | (uintptr_t)y | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)gamma | (uintptr_t)beta) & 15) return MLPK_EALIGN;
    AsConvArgs a;
    a.t = t; a.y = y; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2;
    a.B = B; a.H = H; a.W = W; a.ldw = ldw; a.TH = 0; a.bands = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MLPK_BF16) return C == 96 ? as_conv2_launch<bf16_t, 96>(a, s) : as_conv2_launch<bf16_t, 192>(a, s);
    return C == 96 ? as_conv2_launch<f16_t, 96>(a, s) : as_conv2_launch<f16_t, 192>(a, s);
}
