// Host interface of the skinny fp32 NT GEMM (mlpk_gemm_skinny.hip); called from the tile dispatch of mlpk_gemm.hip (algo 16).
#pragma once
#include <hip/hip_runtime.h>

namespace mlpk {

struct SkinnyCall {
    int M, N, K;
    int lda, ldb, ldc;
    const float* A;
    const float* B;
    float* C;
    const float* bias;     // [N] or null
    int gelu;
};

bool skinny_supported(const SkinnyCall& c);
int skinny_launch(const SkinnyCall& c, hipStream_t stream);

}  // namespace mlpk
