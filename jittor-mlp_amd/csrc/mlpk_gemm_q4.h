// Host interface of the generated "q4" GEMM kernels (mlpk_gemm_q4.hip); called from the tile dispatch of mlpk_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace mlpk {

#define Q4_LDS_BYTES (3 * 49152 + 16384)

struct Q4Call {
    int dtype;                 // MLPK_BF16 / MLPK_F16
    int M, N, K;
    int lda, ldb, ldc, ldr;
    const void* A;
    const void* B;
    void* C;
    const void* R;             // residual (res): C = round(round(acc + bias) + R)
    const float* bias;         // [N], required
    const float* ln_mean;      // folded LayerNorm (ln): v = (acc - mean[m] * csum[n]) * rstd[m] + bias[n]
    const float* ln_rstd;
    const float* ln_csum;
    int gelu, ln, res;
    float* row_part;           // by-product (sum, sum of squares) of the stored rows per block of 64 columns (mlpk.h row_part), or null
    int row_part_ld;
    int one_group;             // tuning: a single column group
    void* prof;                // tuning: (cycles, tiles) of every workgroup, 8 bytes each, or null
    int dbg;                   // tuning ablations: 1 = no LDS-DMA, 4 = no epilogue fillers (results are wrong by construction)
};

bool q4_supported(const Q4Call& c);
// name of the generated kernel q4_launch would run for this call (nullptr: no variant for this class / its tuning bits)
const char* q4_variant_name(const Q4Call& c);
int q4_launch(const Q4Call& c, hipStream_t stream);

}  // namespace mlpk
