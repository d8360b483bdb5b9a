// Host interface of the generated "t4" fused token-mixing MLP (mlpk_tokenmlp_t4.hip); called from mlpk_token_mlp (layout 2).
#pragma once
#include <hip/hip_runtime.h>

namespace mlpk {

struct T4Call {
    int dtype;                 // MLPK_BF16 / MLPK_F16
    int M, S, G;               // rows of xt (images x channels), tokens (196), hidden groups of 32
    int ldxt, ldx, t_rows;
    const void* xt;            // (M, 224) token-transposed LayerNorm output, columns >= S zero
    const void* w1;            // (G*32, 256) zero-padded
    const void* w2;            // ((G+1)*224, 32): group-major, k slots permuted (mlpk.h layout 2 / 3), group G and token rows >= S zero
    const float* b1;           // 1024: entry 64 + t = bias of hidden t, zeros elsewhere
    const float* b2;           // 224: zeros behind S
    void* x;                   // (B*S, ldx) residual stream, updated in place
    float* stats;              // planes of 64 channels x B*S x (sum, sum of squares), or null
    // LayerNorm as the kernel's operand loader (mlpk_token_mlp_ln): xt unused, x normalised with these on the way in
    const float* ln_mean;      // (B*S) statistics of the rows of x, or null: xt is read
    const float* ln_rstd;
    const float* gamma;        // (t_rows)
    const float* beta;
    void* prof;                // tuning: shader cycles per workgroup (8 bytes each), or null
    int dbg;                   // tuning ablations (wrong results by construction)
    int h2;                    // bf16 only (mlpk.h layout 3): GELU in packed f16, the hidden kept in f16, w2 holds f16 values
};

bool t4_supported(int dtype, int S, int G, int ldxt, int M, int t_rows, int ldx);
int t4_launch(const T4Call& c, hipStream_t stream);

}  // namespace mlpk
