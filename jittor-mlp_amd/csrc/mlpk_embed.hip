// Patch gather: the im2col half of a kernel == stride convolution (patch embedding, S2 stage
// embedding, AS-MLP PatchMerging).  The other half is mlpk_gemm_nt.  No im2col buffer beyond
// the GEMM's own A operand is ever materialised, and the dtype conversion of the input image
// (fp32 NCHW -> bf16/f16 tokens) happens here, in the same pass.
#include "mlpk_common.h"

namespace mlpk {

struct PatchArgs {
    const void* src;
    void* out;
    int B, Cin, H, W, ph, pw, sh, sw, pad, px_stride, ldo, order, layout;      // ph x pw window, sh x sw stride
    int Hp, Wp, K;
    int64_t rows;
};

template <typename TS, typename TD>
__global__ void __launch_bounds__(256) patchify_kernel(const PatchArgs p) {
    const TS* __restrict__ src = reinterpret_cast<const TS*>(p.src);
    TD* __restrict__ out = reinterpret_cast<TD*>(p.out);
    const int chunks = p.ldo / 8;
    const int64_t total = p.rows * chunks;
    const bool small = total <= 0x7fffffffll;             // 32-bit index arithmetic where it fits (64-bit divisions cost ~100 instructions each)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t row;
        int k0, wp, hp, b;
        if (small) {
            const unsigned i32 = (unsigned)idx, r32 = i32 / (unsigned)chunks, hw = r32 / (unsigned)p.Wp;
            row = r32;
            k0 = (int)(i32 - r32 * (unsigned)chunks) * 8;
            wp = (int)(r32 - hw * (unsigned)p.Wp);
            b = (int)(hw / (unsigned)p.Hp);
            hp = (int)(hw - (unsigned)b * (unsigned)p.Hp);
        } else {
            row = idx / chunks;
            k0 = (int)(idx - row * chunks) * 8;
            wp = (int)(row % p.Wp);
            hp = (int)((row / p.Wp) % p.Hp);
            b = (int)(row / ((int64_t)p.Wp * p.Hp));
        }
        TD e[8];
        if (p.layout == MLPK_LAYOUT_NHWC && (p.Cin & 7) == 0 && k0 < p.K) {
            // 8 consecutive k share the patch offset (i,j): one contiguous 8-channel run
            const int q = k0 / p.Cin;
            const int ci = k0 - q * p.Cin;
            const int i = p.order == 1 ? (q & 1) : q / p.pw;
            const int j = p.order == 1 ? (q >> 1) : q - (q / p.pw) * p.pw;
            const int y = hp * p.sh + i - p.pad, x = wp * p.sw + j - p.pad;
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
                const TS* s = src + (((int64_t)b * p.H + y) * p.W + x) * p.px_stride + ci;
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = from_f32<TD>(to_f32(s[t]));
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = from_f32<TD>(0.f);
            }
        } else if (p.layout == MLPK_LAYOUT_NCHW && (p.pw & 7) == 0 && p.sw == p.pw && p.sh == p.ph && p.pad == 0 && (p.W & 7) == 0 && k0 < p.K &&
                   (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            // 8 consecutive k = 8 consecutive pixels of one image row of one channel: 16-byte loads
            const int ci = k0 / (p.ph * p.pw);
            const int rem = k0 - ci * (p.ph * p.pw);
            const int i = rem / p.pw;
            const int j = rem - i * p.pw;
            const TS* s = src + (((int64_t)b * p.Cin + ci) * p.H + (hp * p.ph + i)) * p.W + wp * p.pw + j;
            if constexpr (sizeof(TS) == 4) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
                e[0] = from_f32<TD>(a0.x); e[1] = from_f32<TD>(a0.y); e[2] = from_f32<TD>(a0.z); e[3] = from_f32<TD>(a0.w);
                e[4] = from_f32<TD>(a1.x); e[5] = from_f32<TD>(a1.y); e[6] = from_f32<TD>(a1.z); e[7] = from_f32<TD>(a1.w);
            } else {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(s);
                TS t8[8];
                __builtin_memcpy(t8, &raw, 16);
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = from_f32<TD>(to_f32(t8[t]));
            }
        } else if (p.layout == MLPK_LAYOUT_NCHW) {
            // overlapping / padded windows on NCHW (the 7 x 7 stride-4 stems of Hire-MLP and CycleMLP): k -> (ci, i, j) decoded ONCE per
            // chunk and advanced with carries (round 6: two divisions per element made this path ALU-bound, 230 us for 77 MB in / 244 MB out)
            int ci = k0 / (p.ph * p.pw);
            const int rem = k0 - ci * (p.ph * p.pw);
            int i = rem / p.pw;
            int j = rem - i * p.pw;
            const int y0 = hp * p.sh - p.pad, x0 = wp * p.sw - p.pad;
            const TS* img = src + (int64_t)b * p.Cin * p.H * p.W;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int y = y0 + i, x = x0 + j;
                const bool ok = k0 + t < p.K && y >= 0 && y < p.H && x >= 0 && x < p.W;
                const TS raw = img[ok ? ((int64_t)ci * p.H + y) * p.W + x : 0];          // unconditional load: the eight are in flight together
                e[t] = ok ? from_f32<TD>(to_f32(raw)) : from_f32<TD>(0.f);
                if (++j == p.pw) { j = 0; if (++i == p.ph) { i = 0; ++ci; } }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = k0 + t;
                float v = 0.f;
                if (k < p.K) {
                    const int q = k / p.Cin;
                    const int ci = k - q * p.Cin;
                    const int i = p.order == 1 ? (q & 1) : q / p.pw;
                    const int j = p.order == 1 ? (q >> 1) : q - (q / p.pw) * p.pw;
                    const int y = hp * p.sh + i - p.pad, x = wp * p.sw + j - p.pad;
                    if (y >= 0 && y < p.H && x >= 0 && x < p.W) v = to_f32(src[(((int64_t)b * p.H + y) * p.W + x) * p.px_stride + ci]);
                }
                e[t] = from_f32<TD>(v);
            }
        }
        TD* o = out + row * p.ldo + k0;
        if constexpr (sizeof(TD) == 2) {
            u32x4 t;
            __builtin_memcpy(&t, e, 16);
            *reinterpret_cast<u32x4*>(o) = t;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = e[t];
        }
    }
}

template <typename TS> static int patchify_from(int dst, const PatchArgs& a, unsigned grid, hipStream_t s) {
    switch (dst) {
        case MLPK_F32: hipLaunchKernelGGL((patchify_kernel<TS, float>), dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_F16: hipLaunchKernelGGL((patchify_kernel<TS, f16_t>), dim3(grid), dim3(256), 0, s, a); break;
        case MLPK_BF16: hipLaunchKernelGGL((patchify_kernel<TS, bf16_t>), dim3(grid), dim3(256), 0, s, a); break;
        default: return MLPK_EDTYPE;
    }
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk

using namespace mlpk;

static int im2col_impl(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin, int H, int W,
                       int ph, int pw, int sh, int sw, int pad, int src_px_stride, int ldo, int order, void* stream) {
    if (!src || !out) return MLPK_ENULL;
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0 || pad < 0) return MLPK_ESHAPE;
    if (src_layout != MLPK_LAYOUT_NCHW && src_layout != MLPK_LAYOUT_NHWC) return MLPK_EMODE;
    if (order != 0 && order != 1) return MLPK_EMODE;
    if (order == 1 && (src_layout != MLPK_LAYOUT_NHWC || ph != 2 || pw != 2)) return MLPK_EMODE;
    PatchArgs a;
    a.src = src; a.out = out; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.ph = ph; a.pw = pw; a.sh = sh; a.sw = sw; a.pad = pad;
    a.px_stride = src_layout == MLPK_LAYOUT_NHWC ? src_px_stride : 0;
    a.ldo = ldo; a.order = order; a.layout = src_layout;
    if (H + 2 * pad < ph || W + 2 * pad < pw) return MLPK_ESHAPE;
    a.Hp = (H + 2 * pad - ph) / sh + 1;
    a.Wp = (W + 2 * pad - pw) / sw + 1;
    a.K = Cin * ph * pw;
    if (a.Hp <= 0 || a.Wp <= 0) return MLPK_ESHAPE;
    if (ldo % 8 || ldo < a.K) return MLPK_ESHAPE;
    if (src_layout == MLPK_LAYOUT_NHWC && src_px_stride < Cin) return MLPK_ESHAPE;
    if ((uintptr_t)out & 15) return MLPK_EALIGN;
    a.rows = (int64_t)B * a.Hp * a.Wp;
    const int64_t total = a.rows * (ldo / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (src_dtype) {
        case MLPK_F32: return patchify_from<float>(dst_dtype, a, grid, s);
        case MLPK_F16: return patchify_from<f16_t>(dst_dtype, a, grid, s);
        case MLPK_BF16: return patchify_from<bf16_t>(dst_dtype, a, grid, s);
        default: return MLPK_EDTYPE;
    }
}

extern "C" int mlpk_patchify(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin,
                             int H, int W, int ph, int pw, int pad, int src_px_stride, int ldo, int order,
                             void* stream) {
    return im2col_impl(src_dtype, dst_dtype, src_layout, src, out, B, Cin, H, W, ph, pw, ph, pw, pad, src_px_stride, ldo, order, stream);
}

extern "C" int mlpk_im2col(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin,
                           int H, int W, int kh, int kw, int stride_h, int stride_w, int pad, int src_px_stride, int ldo,
                           void* stream) {
    return im2col_impl(src_dtype, dst_dtype, src_layout, src, out, B, Cin, H, W, kh, kw, stride_h, stride_w, pad, src_px_stride, ldo, 0, stream);
}
