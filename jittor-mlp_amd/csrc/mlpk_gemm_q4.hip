// "q4" NT GEMM: 4 waves per workgroup = one per SIMD with the whole 512-entry register file, 256 x 128 tiles, two accumulator
// sets; the previous tile's epilogue (bias / folded LayerNorm / GELU / round / 16-byte stores) is issued as fillers behind the
// v_mfma_f32_32x32x16 of the current tile.  The kernels are GENERATED (csrc/gen/q4gen.py -> q4_kernels.inc): each is one
// `asm volatile` block in which the generator names every register and places every instruction; this file is the host side
// (argument block, variant choice, launch arithmetic).  Same numerics contract as the other tiles: fp32 accumulation in K order,
// the epilogue formulas of gemm_epilogue / p8_store_direct (gelu16_f's operation sequence), one rounding to the storage type.
#include "mlpk_common.h"
#include "mlpk_gemm_q4.h"

namespace mlpk {
// kernarg block read by the generated code (offsets: KA in q4gen.py)
struct Q4Args {
    const void* A;
    const void* B;
    void* C;
    const void* R;
    const float* bias;
    const float* ln_mean;
    const float* ln_rstd;
    const float* ln_csum;
    int lda, ldb, ldc, ldr;
    int nk, cg, cg_magic, U;
    int Q, log2X, m_base, grid;
    void* prof;          // tuning: per-workgroup (cycles, tiles) pairs, or null
    float* row_part;     // by-product row statistics (classes with stats): planes of 64 columns
    int row_part_ld;
    int pad[3];
};
static_assert(sizeof(Q4Args) == 144, "kernarg layout");
}  // namespace mlpk

#include "gen_out/q4_kernels.inc"

namespace mlpk {

static int q4_grid_cap() {
    static int cap = 0;
    if (!cap) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
        cap = cu & ~7;
    }
    return cap;
}

// column groups: as p8_cgroups -- the smallest power of two G | tiles_n whose group of weight panels fits ~2.5 MiB of an XCD's L2
static int q4_cgroups(int tiles_n, int K) {
    const double panel = 128.0 * K * 2;
    if (panel * tiles_n <= 3.0 * 1048576.0) return 1;
    for (int g = 2; g <= 8; g *= 2)
        if (tiles_n % g == 0 && panel * (tiles_n / g) <= 2.5 * 1048576.0) return g;
    for (int g = 8; g >= 2; g /= 2)
        if (tiles_n % g == 0) return g;
    return 1;
}

bool q4_supported(const Q4Call& c) {
    if (c.dtype != MLPK_BF16 && c.dtype != MLPK_F16) return false;
    if (c.M % 256 || c.N % 128 || c.K % 64 || c.K < 192) return false;
    if (!c.bias) return false;
    if (c.res && (c.gelu || c.ln)) return false;
    // statistics: the bias + residual class, and GELU + folded LayerNorm (q4_variant_name says whether a kernel exists)
    if (c.row_part && (!(c.res || (c.gelu && c.ln)) || c.row_part_ld < c.M || (reinterpret_cast<uintptr_t>(c.row_part) & 7))) return false;
    if (c.lda % 8 || c.ldb % 8 || c.ldc % 8 || (c.res && c.ldr % 8)) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(c.A) | reinterpret_cast<uintptr_t>(c.B) | reinterpret_cast<uintptr_t>(c.C) |
                         reinterpret_cast<uintptr_t>(c.R) | reinterpret_cast<uintptr_t>(c.bias) | reinterpret_cast<uintptr_t>(c.ln_csum);
    if (al & 15) return false;
    // 32-bit per-lane offsets inside a tile's panels
    if ((long long)256 * c.lda * 2 >= (1ll << 31) || (long long)256 * c.ldc * 2 >= (1ll << 31)) return false;
    return true;
}

// the kernel for a call: the one built for exactly this K where there is one (round 4, Q4.static in q4gen.py: stage rotation, DMA
// addressing and the tile switch resolved at generation time; MLPK_Q4_STATIC=0 switches them off for A/B runs), otherwise the
// general kernel with the most unrolled (filler-carrying) iterations that K allows
static const Q4Variant* q4_pick(const Q4Call& c, int force_nkf) {
    static const bool use_static = !(getenv("MLPK_Q4_STATIC") && atoi(getenv("MLPK_Q4_STATIC")) == 0);
    const int nk = c.K / 64;
    const Q4Variant* best = nullptr;
    for (const Q4Variant& v : kQ4Variants) {
        if (v.dtype != c.dtype || v.gelu != c.gelu || v.ln != c.ln || v.res != c.res || v.stats != (c.row_part != nullptr) || v.dbg != c.dbg) continue;
        if (v.is_static) {
            if (use_static && !force_nkf && v.nkf == nk) return &v;
            continue;
        }
        if (v.nkf > nk) continue;
        if (force_nkf && v.nkf != force_nkf) continue;
        if (!best || v.nkf > best->nkf) best = &v;
    }
    return best;
}

static const Q4Variant* q4_variant(const Q4Call& c) {
    const int force_nkf = getenv("MLPK_Q4_NKF") ? atoi(getenv("MLPK_Q4_NKF")) : 0;       // tuning: unrolled (filler) iterations
    const Q4Variant* v = q4_pick(c, force_nkf);
    if (!v && force_nkf) v = q4_pick(c, 0);
    return v;
}

const char* q4_variant_name(const Q4Call& c) {
    if (!q4_supported(c)) return nullptr;
    const Q4Variant* v = q4_variant(c);
    return v ? v->name : nullptr;
}

int q4_launch(const Q4Call& c, hipStream_t stream) {
    if (!q4_supported(c)) return MLPK_ESHAPE;
    const Q4Variant* v = q4_variant(c);
    if (!v) return MLPK_ESHAPE;
    const int tiles_n = c.N / 128;
    const int cgroups = c.one_group ? 1 : q4_cgroups(tiles_n, c.K);
    const int X = 8 / cgroups;
    const int cg = tiles_n / cgroups;
    const int panels = c.M / 256;
    const int U = panels * cg;
    const int Q = (U + X - 1) / X;
    const int cap = q4_grid_cap();
    const int grid = 8 * (Q < cap / 8 ? Q : cap / 8);
    Q4Args a;
    a.A = c.A; a.B = c.B; a.C = c.C; a.R = c.R;
    a.bias = c.bias; a.ln_mean = c.ln_mean; a.ln_rstd = c.ln_rstd; a.ln_csum = c.ln_csum;
    a.lda = c.lda; a.ldb = c.ldb; a.ldc = c.ldc; a.ldr = c.ldr;
    a.nk = c.K / 64; a.cg = cg;
    a.cg_magic = (int)(((1ull << 31) + cg - 1) / cg);
    a.U = U; a.Q = Q;
    int lg = 0;
    while ((1 << lg) < X) ++lg;
    a.log2X = lg; a.m_base = 0; a.grid = grid;
    a.prof = c.prof;
    a.row_part = c.row_part; a.row_part_ld = c.row_part_ld;
    a.pad[0] = a.pad[1] = a.pad[2] = 0;
    hipError_t e = hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    void* params[] = {&a};
    e = hipLaunchKernel(v->fn, dim3(grid), dim3(256), params, Q4_LDS_BYTES, stream);
    if (e != hipSuccess) return (int)e;
    MLPK_LAUNCH_CHECK();
    return 0;
}

}  // namespace mlpk
