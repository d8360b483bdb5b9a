// "t4" fused token-mixing MLP: 4 waves per workgroup = one per SIMD, a wave owns 64 rows of xt for all 196 tokens and runs both
// products and the GELU between them in one instruction stream (the GELU of group g-1 as fillers behind the MFMAs of
// fc2(g-2) and fc1(g)); the hidden never leaves the registers.  The kernels are GENERATED (csrc/gen/t4gen.py -> t4_kernels.inc,
// emulated on the CPU by csrc/gen/t4emu.py); this file is the host side.  Numerics contract of token_mlp_rr_kernel: fp32
// accumulation, gelu16_f's operation sequence, the hidden rounded once to the storage type, one rounding after the residual add.
// Round 5, bf16 storage ("h2" kernels, mlpk.h layout 3): the GELU runs in packed f16 (q4gen.GELU_H2) and the hidden STAYS f16 -- 11 bits instead of
// bf16's 8 -- so W2 arrives as f16 values and the second product is the f16 MFMA; x, W1 and the first product are bf16 as before.
#include "mlpk_common.h"
#include "mlpk_tokenmlp_t4.h"
#include <cstdlib>

namespace mlpk {
// kernarg block read by the generated code (offsets: KA in t4gen.py)
struct T4Args {
    const void* xt;
    const void* w1;
    const void* w2;
    const float* b1;
    const float* b2;
    void* x;
    float* stats;
    void* prof;
    int M, G, ldxt, ldx;
    int ntiles, tpi, tpi_magic, grid;
    int stat_ld, nit, lead, S;
    const float* ln_mean;
    const float* ln_rstd;
    const float* gamma;
    const float* beta;
};
static_assert(sizeof(T4Args) == 144, "kernarg layout");
}  // namespace mlpk

#include "gen_out/t4_kernels.inc"

namespace mlpk {

#define T4_LDS_BYTES 145408        // LDS_BYTES of t4gen.py

static int t4_grid_cap() {
    static int cap = 0;
    if (!cap) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1) cu = 256;
        cap = cu;
    }
    return cap;
}

bool t4_supported(int dtype, int S, int G, int ldxt, int M, int t_rows, int ldx) {
    if (dtype != MLPK_BF16 && dtype != MLPK_F16) return false;
    if (S != 196 || ldxt != 224) return false;                  // 14 k-steps of the first product, 7 token blocks of the second
    if (G < 1 || G > 28) return false;                          // bias table: groups -2 .. 29
    if (t_rows <= 0 || t_rows % 256 || M <= 0 || M % t_rows) return false;      // a 256-row tile = 4 x 64 channels of one image
    if (ldx % 8 || ldx < t_rows) return false;
    return true;
}

int t4_launch(const T4Call& c, hipStream_t stream) {
    if (!t4_supported(c.dtype, c.S, c.G, c.ldxt, c.M, c.t_rows, c.ldx)) return MLPK_ESHAPE;
    // shape: the kernels whose pipeline fill / drain iterations carry only the stages that have work exist per parity of G
    // (MLPK_T4_SHAPE=0 forces the generic one: A/B aid); the ablation variants are generic
    int shape = (c.G & 1) ? (c.G >= 3 ? 1 : 0) : (c.G >= 2 ? 2 : 0);
    const char* es = getenv("MLPK_T4_SHAPE");
    if (c.dbg || (es && es[0] == '0')) shape = 0;
    const int ln = c.ln_mean != nullptr;
    if (c.h2 && (c.dtype != MLPK_BF16 || c.dbg)) return MLPK_EMODE;
    if (ln && (!shape || !c.ln_rstd || !c.gamma || !c.beta)) return MLPK_ESHAPE;
    const T4Variant* v = nullptr;
    for (const T4Variant& k : kT4Variants)
        if (k.dtype == c.dtype && k.stats == (c.stats != nullptr) && k.dbg == c.dbg && k.shape == shape && k.ln == ln && k.h2 == c.h2) { v = &k; break; }
    if (!v) return MLPK_ESHAPE;
    T4Args a;
    a.xt = c.xt; a.w1 = c.w1; a.w2 = c.w2; a.b1 = c.b1; a.b2 = c.b2; a.x = c.x; a.stats = c.stats; a.prof = c.prof;
    a.M = c.M; a.G = c.G; a.ldxt = c.ldxt; a.ldx = c.ldx;
    a.ntiles = c.M / 256;
    a.tpi = c.t_rows / 256;
    a.tpi_magic = (int)(((1ull << 31) + a.tpi - 1) / a.tpi);
    a.grid = a.ntiles < t4_grid_cap() ? a.ntiles : t4_grid_cap();
    a.stat_ld = (c.M / c.t_rows) * c.S * 2;
    a.lead = (c.G + 2) & 1;                                     // iterations come in pairs (the LDS stage parity is static)
    a.nit = c.G + 2 + a.lead;
    a.S = c.S;
    a.ln_mean = c.ln_mean; a.ln_rstd = c.ln_rstd; a.gamma = c.gamma; a.beta = c.beta;
    hipError_t e = hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    void* params[] = {&a};
    e = hipLaunchKernel(v->fn, dim3(a.grid), dim3(256), params, T4_LDS_BYTES, stream);
    if (e != hipSuccess) return (int)e;
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk
