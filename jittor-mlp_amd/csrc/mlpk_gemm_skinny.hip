// Skinny fp32 NT GEMM: C[m, n] = act(sum_k A[m, k] * B[n, k] + bias[n]) for the tiny products of the SplitAttention / re-weighting MLPs
// (vip.py:42-53, s2_mlp_v2.py:36-47, cycle_mlp.py:160-170: M = batch or batch * groups rows, a few hundred columns, K <= 1536 --
// 75 .. 300 MFLOP).  The MFMA tiles ran them as <= 72 workgroups with a serial K loop of 12-24 slabs: 15-24 us each, and up to 120 us
// when they shared the chip with a persistent GEMM (one 64 x 64 workgroup parked on a CU holds up the persistent kernel's workgroup
// for that CU).  Here the whole chip works on them for a few microseconds:
//   * a lane owns one output COLUMN n (its weight row B[n, :] is read in 16-byte pieces), a wave owns 8 output ROWS whose A values are
//     wave-uniform (scalar loads): 32 fused multiply-adds per 16-byte vector load;
//   * the four waves of a workgroup each take a quarter of K and add their partial sums through 8 KiB of LDS;
//   * grid = ceil(N / 64) x ceil(M / 8) workgroups of 256 threads (ViP's three products: 1024 / 192 / 576 workgroups).
// Exact fp32 fused multiply-adds (a fixed order: four K quarters, each ascending, added 0 + 1 + 2 + 3), so results do not depend on M:
// a row's values are the same in every batch.
#include "mlpk_common.h"
#include "mlpk_gemm_skinny.h"

namespace mlpk {

constexpr int SK_ROWS = 8;

// X = the operand whose rows the LANES own (vector loads), Y = the one whose rows are wave-uniform (scalar loads).  Usually X = B
// (a lane per output column); with few columns and many rows (ViP's first product: 8192 x 24) the roles swap -- X = A, a lane per
// output ROW -- so that all 64 lanes work.  out(l, u) goes to C[l * sl + u * su].
struct SkinnyArgs {
    const float* X;
    const float* Y;
    float* C;
    const float* bias;
    int LN, UN, K, ldx, ldy;
    long long sl, su;
    int bias_on_lane;
};

template <bool GELU>
__global__ void __launch_bounds__(256) gemm_skinny_f32_kernel(const SkinnyArgs p) {
    __shared__ float red[3][SK_ROWS][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = blockIdx.x * 64 + lane;
    const int u0 = blockIdx.y * SK_ROWS;
    const int kq = p.K >> 2;                                   // K % 16 == 0: whole 16-byte pieces per quarter
    const int k0 = wave * kq;
    const float* __restrict__ xrow = p.X + (size_t)(l < p.LN ? l : p.LN - 1) * p.ldx + k0;
    const float* __restrict__ yrow[SK_ROWS];
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) yrow[r] = p.Y + (size_t)(u0 + r < p.UN ? u0 + r : p.UN - 1) * p.ldy + k0;      // wave-uniform
    float acc[SK_ROWS];
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) acc[r] = 0.f;
    for (int k = 0; k < kq; k += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xrow + k);
#pragma unroll
        for (int r = 0; r < SK_ROWS; ++r) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(yrow[r] + k);
            acc[r] = __builtin_fmaf(y.x, x.x, acc[r]);
            acc[r] = __builtin_fmaf(y.y, x.y, acc[r]);
            acc[r] = __builtin_fmaf(y.z, x.z, acc[r]);
            acc[r] = __builtin_fmaf(y.w, x.w, acc[r]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < SK_ROWS; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && l < p.LN) {
#pragma unroll
        for (int r = 0; r < SK_ROWS; ++r) {
            if (u0 + r >= p.UN) break;
            const float bn = p.bias ? p.bias[p.bias_on_lane ? l : u0 + r] : 0.f;
            float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane] + bn;
            if (GELU) v = gelu_f(v);
            p.C[(long long)l * p.sl + (long long)(u0 + r) * p.su] = v;
        }
    }
}

bool skinny_supported(const SkinnyCall& c) {
    if (c.M <= 0 || c.N <= 0 || c.K < 16 || c.K % 16 || c.lda % 4 || c.ldb % 4) return false;
    if ((reinterpret_cast<uintptr_t>(c.A) | reinterpret_cast<uintptr_t>(c.B)) & 15) return false;
    return (c.M + SK_ROWS - 1) / SK_ROWS <= 65535 && (c.N + SK_ROWS - 1) / SK_ROWS <= 65535;
}

int skinny_launch(const SkinnyCall& c, hipStream_t stream) {
    if (!skinny_supported(c)) return MLPK_ESHAPE;
    SkinnyArgs a;
    // few columns, many rows: a lane per ROW (the decision depends on N alone, so a row's bits do not depend on the batch)
    const bool swap = c.N < 48;
    if (swap) {
        a.X = c.A; a.Y = c.B; a.LN = c.M; a.UN = c.N; a.ldx = c.lda; a.ldy = c.ldb; a.sl = c.ldc; a.su = 1; a.bias_on_lane = 0;
    } else {
        a.X = c.B; a.Y = c.A; a.LN = c.N; a.UN = c.M; a.ldx = c.ldb; a.ldy = c.lda; a.sl = 1; a.su = c.ldc; a.bias_on_lane = 1;
    }
    a.C = c.C; a.bias = c.bias; a.K = c.K;
    const dim3 grid((unsigned)((a.LN + 63) / 64), (unsigned)((a.UN + SK_ROWS - 1) / SK_ROWS));
    if (c.gelu) hipLaunchKernelGGL(gemm_skinny_f32_kernel<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gemm_skinny_f32_kernel<false>, grid, dim3(256), 0, stream, a);
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk
