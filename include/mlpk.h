/*
 * mlpk.h -- C ABI of the MI355X (gfx950) native kernels for the vision-MLP forward path.
 *
 * Drop-in boundary.  The reference has exactly one native-op seam on the
 * PyTorch side: `models_pytorch/utils/shift_cuda.py:106-129` -- a Python autograd.Function that
 * launches a raw-pointer kernel on torch's current stream, with the output allocated by the
 * caller (`input.new(...)`, :112).  Every entry point here follows that contract:
 *
 *   - plain pointers and sizes only (no torch types); the CALLER owns all memory, the library
 *     allocates nothing and keeps no state between calls;
 *   - all tensors are dense device buffers; "ld*" are row strides in ELEMENTS;
 *   - `dtype` selects the storage/MFMA operand type of activations and weights
 *     (MLPK_F32 / MLPK_F16 / MLPK_BF16); accumulation, statistics, GELU and every
 *     per-channel vector (bias, scale, shift, gamma, beta, mean, rstd) are float32;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream), launches are
 *     asynchronous;
 *   - return value: 0 = launched, negative = MLPK_E* argument error (nothing launched),
 *     positive = hipError_t from the launch.
 *
 * Reference interfaces replaced (file:line relative to the reference repository):
 *   mlpk_gemm_nt        nn.Linear / Conv1d(k=1) / Conv2d(1x1) + the elementwise ops the reference
 *                       runs after them (GELU, residual add, BatchNorm-eval affine, SGU gate):
 *                       mlp_mixer.py:16-27,6-13; g_mlp.py:17-22,32-39; res_mlp.py:52-57;
 *                       vip.py:65-90; s2_mlp_v2.py:60-69,76-85; as_mlp.py:8-24,55-95; conv_mixer.py:29-31
 *   mlpk_token_mlp_ln   the token-mixing PreNormResidual (LayerNorm + both Conv1d(k=1) + GELU + residual) in ONE kernel: mlp_mixer.py:6-13,16-27,34
 *   mlpk_token_mlp      both Conv1d(k=1) of the Mixer token-mixing FeedForward + GELU + residual in ONE kernel
 *                       (mlp_mixer.py:16-27,34,37), hidden activations never leave the CU
 *   mlpk_token_gemm     one token-mixing product with the transposed epilogue: gMLP SGU (g_mlp.py:17-22), ResMLP cross-patch (res_mlp.py:52-55)
 *   mlpk_token_gemm_ln  the same with the LayerNorm / Aff in front of it (g_mlp.py:19; res_mlp.py:17-19,53) as the kernel's operand loader
 *   mlpk_patchify       the im2col half of nn.Conv2d(k=stride=patch): mlp_mixer.py:58-60,68-71;
 *                       conv_mixer.py:18; s2_mlp_v2.py:119; as_mlp.py:319,330; PatchMerging as_mlp.py:207-211
 *   mlpk_row_stats      statistics of nn.LayerNorm (mlp_mixer.py:10) and nn.GroupNorm(1,C) (as_mlp.py:343-344)
 *   mlpk_layernorm_transpose  statistics + affine + per-image transpose of the token-mixing LayerNorm (mlp_mixer.py:34) in one pass
 *   mlpk_norm_apply     the normalise+affine half of LayerNorm/GroupNorm/Aff (res_mlp.py:17-19), fused with
 *                       GELU (as_mlp.py:64-66) and with the layout change the next GEMM needs:
 *                       token-major transpose (Conv1d over tokens) or the ViP rearranges (vip.py:69,74)
 *   mlpk_vip_unpermute  einops Rearrange back (vip.py:71,76)
 *   mlpk_pool_mean      x.mean(dim=1) / Reduce('b h w c -> b c') / AdaptiveAvgPool2d, optionally through
 *                       the final LayerNorm: mlp_mixer.py:73-74; vip.py:160-163; s2_mlp_v2.py:125; as_mlp.py:435-437
 *   mlpk_shift_nchw     Shift / _shift.forward / shift_forward_kernel: utils/shift_cuda.py:44-72,106-129,177-192
 *   mlpk_shift_nchw_backward   _shift.backward / shift_backward_grad_input_kernel: utils/shift_cuda.py:75-103,131-162
 *   mlpk_gelu_elementwise, mlpk_layernorm_backward, mlpk_col_sum, mlpk_transpose_batched, mlpk_broadcast_rows
 *                              the autograd of the Mixer path (train mode): mlp_mixer.py:6-27,34-38,62-75 (round 5)
 *   mlpk_shift_nhwc     the same remap on the channel-last layout used internally for AS-MLP
 *   mlpk_norm_shift_nhwc  AxialShift's GroupNorm + GELU + both shifts as one index-remapping pass (as_mlp.py:64-66,84-95)
 *   mlpk_as_conv2       AxialShift's core in ONE kernel: GroupNorm + GELU, both axial shifts, conv2_1 and conv2_2 with their GELUs and the sum
 *                       (as_mlp.py:64-66,84-93; utils/shift_cuda.py:49-69): the shifts are LDS read addresses of the MFMA operands
 *   mlpk_channel_mlp    fc1 + GELU + fc2 + residual of a channel MLP on narrow (C <= 192) channel-last rows in ONE kernel (as_mlp.py:36-52)
 *   mlpk_linear_gelu    a short-K (<= 512) Linear + GELU with its rows resident in registers (g_mlp.py:28,35; the fc1 of the K = 384 channel MLPs)
 *   mlpk_swin_spatial   LayerNorm + window partition + multi-head spatial MLP + merge + residual of a Swin-MLP block in ONE kernel (swin_mlp.py:97-151)
 *   mlpk_cycle_shift    the sampling half of CycleFC (cycle_mlp.py:104-131: deform_conv2d with a 1 x 1 kernel and fixed integer
 *                       offsets = a per-channel cyclic pixel shift with zero fill); the 1 x 1 convolution is mlpk_gemm_nt
 *   mlpk_split_sum      the reduction of SplitAttention (vip.py:49-50; s2_mlp_v2.py:43-44), with the
 *                       S2 spatial shifts (s2_mlp_v2.py:15-29) applied on load
 *   mlpk_split_softmax  softmax over the k=3 branches (vip.py:52-53)
 *   mlpk_split_apply    attention * x_all summed over k (vip.py:54-56), shifts applied on load
 *   mlpk_vip_split_apply  the weighted sum with the inverse ViP rearranges (vip.py:71,76) as load addresses (8 x 8 pixel tiles staged in
 *                       LDS); the reduction of ViP's SplitAttention needs no pass at all: mlpk_norm_desc.sum_ph / sum_pw + linearity
 *   mlpk_vip_branch     LayerNorm + rearrange + Linear of ViP's h / w branch in ONE kernel, the rearrange as LDS staging order (vip.py:66-76)
 *   mlpk_s2_shift       Spatial_Shift (s2_mlp_v1.py:19-25), out of place
 *   mlpk_dwconv_nhwc    depthwise Conv2d(k, groups=dim, padding="same") + GELU + BatchNorm(eval) + residual:
 *                       conv_mixer.py:5-11,24-28
 */
#ifndef MLPK_H
#define MLPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dtypes --------------------------------------------------------------------------- */
#define MLPK_F32 0
#define MLPK_F16 1
#define MLPK_BF16 2

/* ---- error codes (negative) ----------------------------------------------------------- */
#define MLPK_OK 0
#define MLPK_EDTYPE (-1)   /* unknown dtype */
#define MLPK_ESHAPE (-2)   /* size/stride violates a documented constraint */
#define MLPK_EALIGN (-3)   /* pointer or leading dimension not 16-byte aligned where required */
#define MLPK_ENULL (-4)    /* required pointer is NULL */
#define MLPK_EMODE (-5)    /* unknown mode/flag value */

/* ABI version; bumped on any signature change. */
int mlpk_abi_version(void);
/* Human-readable message for a return code (static storage). */
const char* mlpk_strerror(int code);

/* ---- GEMM with fused epilogue ----------------------------------------------------------
 * acc[m,n] = sum_k A[m*lda + k] * B[n*ldb + k]           (A: MxK, B: NxK, both K-contiguous)
 * v = (acc - ln_mean[m]*ln_csum[n]) * ln_rstd[m]   (all three NULL -> v = acc).  This folds a LayerNorm
 *     of the A rows into the GEMM: with W' = W*diag(gamma) as B, ln_csum[n] = sum_k W'[n,k] and the
 *     LayerNorm's beta folded into `bias`, A can be the UN-normalised activation (ROWMAJOR output only).
 * v = v + bias[n]                       (bias may be NULL)
 * v = gelu(v)   if act == MLPK_ACT_GELU (exact erf form)
 * v = v * cscale[n] + cshift[n]         (either may be NULL)
 * v = v * rscale[m % rperiod]           (rscale may be NULL)
 * v = v + R[...]  (res_mode ADD)  |  v = v * R[...]  (res_mode MUL)
 * C[...] = (dtype) v
 * Addressing of C (and R with ldr):
 *   out_mode ROWMAJOR : C[m*ldc + n]
 *   out_mode TOKEN_T  : rows are (image b, channel c) pairs, m = b*t_rows + c, and columns are
 *                       tokens n; element goes to C[(b*t_tokens + n)*ldc + c]  (the per-image
 *                       transpose that turns the token-mixing Conv1d into this NT GEMM).
 * Constraints: K % (16/sizeof(dtype) * 2) == 0 is NOT required; K, lda, ldb must be multiples of
 * 16/sizeof(dtype) elements (16-byte chunks); A, B 16-byte aligned.  TOKEN_T needs t_rows % 4 == 0.
 * R may alias C.  `algo` 0 = automatic tile choice.
 */
#define MLPK_ACT_NONE 0
#define MLPK_ACT_GELU 1
#define MLPK_RES_NONE 0
#define MLPK_RES_ADD 1
#define MLPK_RES_MUL 2
#define MLPK_RES_ADD_AFFINE 3   /* mlpk_token_gemm_ln only (ABI 9): out = round(gamma x + beta) + rscale * (...): the residual is the affine output, rebuilt in the kernel */
#define MLPK_OUT_ROWMAJOR 0
#define MLPK_OUT_TOKEN_T 1

typedef struct mlpk_gemm_desc {
    int32_t dtype;
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr;
    const void* A;
    const void* B;
    void* C;
    const void* R;        /* residual / gate source, same dtype as C, or NULL */
    const float* bias;    /* [N] or NULL */
    const float* cscale;  /* [N] or NULL */
    const float* cshift;  /* [N] or NULL */
    const float* rscale;  /* [rperiod] or NULL */
    const float* ln_mean; /* [ceil(M / ln_group)] or NULL: folded LayerNorm / GroupNorm(1,C) means */
    const float* ln_rstd; /* [ceil(M / ln_group)] or NULL */
    const float* ln_csum; /* [N] or NULL: row sums of B */
    int32_t rperiod;
    int32_t act;
    int32_t res_mode;
    int32_t out_mode;
    int32_t t_rows;       /* TOKEN_T: rows (channels) per image */
    int32_t t_tokens;     /* TOKEN_T: tokens per image (row count of one image in C) */
    int32_t algo;         /* 0 auto; otherwise a tile-config id, see mlpk_gemm_algo_count */
    int32_t ln_group;     /* rows that share one folded statistic: 0 / 1 = LayerNorm (one per row); H*W = GroupNorm(1, C) on channel-last
                             rows (one per sample, as_mlp.py:343-344) */
    int32_t reserved;     /* 0.  Tuning bits of the persistent tile (A/B runs only; results are bit-identical with every
                             combination): 16 = 256-row tiles only (no mixed tile heights), 64 = LDS-staged epilogue,
                             128 = a single column group */
    /* Unused since ABI 5 (mlpk_gemm_workspace_bytes() == 0): every tile is computed by one workgroup, in one K order,
       so results never depend on the batch a row is computed in.  Kept so that descriptors stay layout-compatible. */
    void* workspace;
    int64_t workspace_bytes;
    /* ABI 6.  By-product row statistics (optional; 16-bit row-major outputs with N % 8 == 0, 16-byte aligned rows of C and R):
       the statistics pass of the LayerNorm that FOLLOWS this GEMM (vip.py:66,82; g_mlp.py:40; s2_mlp_v2.py:60,78: every one of
       them reads a tensor a GEMM has just written) comes out of the store epilogue instead of a second pass over C:
         row_part[(q * row_part_ld + m) * 2 + {0, 1}] = sum / sum of squares, over the q-th block of 32 columns, of the
         values WRITTEN to C[m, :] (after rounding, after the residual), q < nparts = ceil(N / 32)
       -- planar, one plane of row_part_ld >= M pairs per column block.  EVERY tile reduces a block in the same order (fp32): the 8
       values of a 16-byte chunk by four two-term dot products in column order, then (c0 + c1) + (c2 + c3) over the block's four
       chunks; mlpk_stats_finalize_planar adds the planes in ascending order in fp64.  So a row's mean / rstd do not depend on the
       tile that stored it, i.e. not on the batch the row is computed in (SURVEY.md section 4 tier 6: a sharded forward equals the
       single forward on the concatenated batch, row for row).  mlpk_gemm_row_parts() answers nparts for a descriptor. */
    float* row_part;
    int32_t row_part_ld;
    int32_t reserved2;    /* 0 */
} mlpk_gemm_desc;

int mlpk_gemm_nt(const mlpk_gemm_desc* d, void* stream);
/* round 6 (ABI 12): a strided k x k convolution on a channel-last tensor as ONE product whose A operand is read through the window -- no gathered operand
 * (mlpk_im2col wrote B Ho Wo x kh kw Cin values for mlpk_gemm_nt to read back): the 3 x 3 stride-2 pad-1 transitions of Hire-MLP (hire_mlp.py:161) and
 * CycleMLP (cycle_mlp.py:220-231).  d: the product's descriptor with A = the dense (B, H, W, Cin) input, M = B Ho Wo, K = kh kw Cin in mlpk_im2col's NHWC
 * order (tap-major, then channel), lda = K (unused), a row-major output; every epilogue option of mlpk_gemm_nt (bias, residual, by-product statistics).
 * 16-bit, Cin % 32 == 0 (a 64-byte slab of K is 32 channels of one tap; taps outside the map read a zero slab).  The 128 x 128 "s3" tile in the K order of
 * mlpk_gemm_nt: the bits of mlpk_im2col + mlpk_gemm_nt on that tile. */
int mlpk_conv_gemm_nhwc_supported(int dtype, int Cin, int kh, int kw, int stride, int pad);
int mlpk_conv_gemm_nhwc(const mlpk_gemm_desc* d, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, void* stream);
/* round 6 (ABI 12): LayerNorm statistics (two-pass, biased variance, fp32) of the rows PatchMerging normalises -- the concatenation of a 2 x 2 window's four
 * pixels of a channel-last (B, H, W, C) tensor (swin_mlp.py:203-210, sparse_mlp.py:40-48; the order of the four does not matter to the statistics) -- without
 * the concatenated tensor: row (b, oy, ox) of B (H/2) (W/2).  16-bit, even H and W, C % 8 == 0, C <= 1536.  With mlpk_conv_gemm_nhwc (k = stride = 2) and
 * the LayerNorm folded into the reduction weight the merged tensor is never stored. */
int mlpk_merge2x2_row_stats(int dtype, const void* x, int B, int H, int W, int C, float eps, float* mean, float* rstd, void* stream);
/* ... or from the per-pixel LayerNorm statistics (mean, 1 / sqrt(var + eps_in) over the SAME C channels per pixel) a producer already delivered: the
 * merged row's mean = the average of the four pixels' means, its variance = avg(var_q + mean_q^2) - mean^2; combined in fp64. */
int mlpk_merge2x2_stats_combine(const float* mean, const float* rstd, int B, int H, int W, float eps_in, float eps_out, float* out_mean, float* out_rstd,
                                void* stream);
/* round 6 (ABI 12): two INDEPENDENT products in one launch where the dispatch gives both the same 16-bit "s3" tile family (algo 11..13) -- workgroups
 * [0, tiles of d0) compute d0, the rest d1; every tile exactly as mlpk_gemm_nt computes it (same bits).  Otherwise the two calls one after the other.
 * For the short, latency-bound products of sibling branches (Hire-MLP's proj_h / proj_w pairs, hire_mlp.py:139-143): half the launches, no side stream.
 * The two outputs must not overlap each other or the other call's operands. */
int mlpk_gemm_nt_pair(const mlpk_gemm_desc* d0, const mlpk_gemm_desc* d1, void* stream);
/* planes (pairs per row) that mlpk_gemm_nt would write for this descriptor (row_part may still be NULL); an error code when the
   descriptor cannot deliver statistics (fp32, token-transposed output, unaligned rows, an explicit algo with 64-column tiles) */
int mlpk_gemm_row_parts(const mlpk_gemm_desc* d, int* nparts);
/* ABI 8.  Name of the kernel mlpk_gemm_nt would launch for this descriptor (tile family, generated variant, tile heights), for
   measurement labels: bench.py's roofline.kernel is what the dispatch answers, not a constant.  Nothing is launched. */
int mlpk_gemm_kernel_name(const mlpk_gemm_desc* d, char* buf, int len);
/* 0 (no kernel needs scratch) */
long long mlpk_gemm_workspace_bytes(void);
/* number of tile configurations (valid algo ids are 1..count) and dynamic LDS bytes of one */
int mlpk_gemm_algo_count(void);
int mlpk_gemm_algo_info(int algo, int* bm, int* bn, int* threads, int* lds_bytes);

/* ---- fused token-mixing MLP (MLP-Mixer) ------------------------------------------------------
 * x[b,s,c] += sum_t W2[s,t] * gelu(sum_s' W1[t,s'] * xt[b*C + c, s'] + b1[t]) + b2[s]      (mlp_mixer.py:16-27,34,37)
 * with the hidden kept on chip (never written to HBM).  16-bit dtypes only.
 *   xt : (M = B*C, ldxt) token-transposed LayerNorm output (mlpk_norm_apply out_tt), ldxt = K of the first
 *        product: a multiple of 32, <= 224, columns >= S zero;
 *   w1 : (nchunks*CH, ldw1 = 256) rows >= 4S and columns >= S zero, CH = mlpk_token_mlp_chunk() = 32;
 *   b1 : (nchunks*CH) zero-padded;  nchunks*CH <= 1024;
 *   w2 : (S, ldw2), ldw2 >= nchunks*CH, columns >= 4S zero;  b2: (S);  S <= 208;
 *   x  : (B*S, ldx) residual stream updated in place, t_rows = C channels per image (M % t_rows == 0).
 */
int mlpk_token_mlp_chunk(void);
int mlpk_token_mlp(int dtype, const void* xt, int ldxt, int M, int S, const void* w1, int ldw1, const float* b1,
                   const void* w2, int ldw2, const float* b2, int nchunks, void* x, int ldx, int t_rows, float* stats,
                   int layout, void* stream);
/* `layout` = what mlpk_token_mlp_layout(S, nchunks) returned when the weights were packed:
 *   0  W2 in natural column order (128-row tiles, hidden handed between waves through LDS);
 *   1  inside every group of 32 hidden columns, column slot 8 f + e (f = 0..3, e = 0..7) holds original column
 *      (e < 4 ? 4 f + e : 16 + 4 f + e - 4): the order in which the first product's accumulators already are the second
 *      product's operands, so the hidden never leaves the registers (256-row tiles; S <= 208);
 *   2  (ABI 7) the generated one-wave-per-SIMD kernel (csrc/gen/t4gen.py): S == 196, ldxt == 224, t_rows % 256 == 0, nchunks <= 28.
 *      w2 : ((nchunks + 1) * 224, ldw2 = 32) GROUP-MAJOR -- group g holds 224 token rows (rows >= S zero) of the 32 hidden units
 *           of group g, k slot 16 kk + 8 h + e (kk, h = 0..1, e = 0..7) = hidden 16 kk + 8 (e >> 2) + 4 h + (e & 3); group nchunks
 *           is all zeros (what the pipeline's fill iterations multiply by);
 *      b1 : 1024 floats, entry 64 + t = bias of hidden unit t, zeros elsewhere;   b2 : 224 floats, zeros behind S;
 *      stats: planes of 64 channels (t_rows / 64 planes of B*S pairs) instead of 128.
 *   3  (ABI 9, bf16 storage only) layout 2 with the HIDDEN KEPT IN f16: the kernel evaluates the GELU in packed f16 and hands the second product
 *      f16 operands (11 mantissa bits instead of bf16's 8; values beyond +-65504 saturate there), so `w2` holds IEEE f16 values -- W2 rounded
 *      to f16 by the packer -- in exactly the arrangement of layout 2.  x, xt, w1 and everything stored stay bf16.
 * mlpk_token_mlp_layout_for additionally knows the channels per image and answers 3 (bf16) / 2 (f16) when the generated kernel takes the shape. */
int mlpk_token_mlp_layout(int S, int nchunks);
int mlpk_token_mlp_layout_for(int dtype, int S, int nchunks, int t_rows);
/* The whole token-mixing PreNormResidual in ONE kernel (ABI 7; mlp_mixer.py:34 with :6-13 and :16-27):
 *   x[b,s,c] += sum_t W2[s,t] * gelu(sum_s' W1[t,s'] * LN_C(x[b,s',:])[c] + b1[t]) + b2[s]
 * -- mlpk_layernorm_transpose + mlpk_token_mlp(layout 2) without the xt tensor between them: the generated kernel reads its rows of x
 * (token-major 128-byte lines), normalises them with the given row statistics (ln_mean / ln_rstd over B*S rows: mlpk_row_stats, or a
 * producer GEMM's row_part through mlpk_stats_finalize_planar) and gamma / beta (t_rows floats), and transposes through LDS into its
 * operand registers.  Weights, b1, b2, stats exactly as for layout 2 / 3 (`layout`, ABI 9: the answer of mlpk_token_mlp_layout_for the weights
 * were packed with); the shapes of layout 2 with nchunks >= 2. */
int mlpk_token_mlp_ln(int dtype, void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                      const float* beta, const void* w1, int ldw1, const float* b1, const void* w2, int ldw2, const float* b2,
                      int nchunks, int t_rows, float* stats, int layout, void* stream);

/* ---- single token-mixing product with the per-image transpose in the epilogue -----------------------------------------------
 * out[b,t,c] = R[b,t,c] (+ | *) rscale[c] * ( sum_s W[t,s] * xt[b*t_rows + c, s] + bias[t] )     (res_mode ADD | MUL; NONE: no R)
 * gMLP's spatial gating unit (g_mlp.py:17-22: R = u, MUL) and ResMLP's cross-patch sublayer (res_mlp.py:52-55: R = x, ADD,
 * rscale = gamma_1).  The same operation as mlpk_gemm_nt with MLPK_OUT_TOKEN_T, as a persistent kernel that keeps its rows of xt
 * in registers.  16-bit dtypes; xt (M = B*t_rows, ldxt) with ldxt % 32 == 0, ldxt <= 224 (K); w (ngroups*32, 256): output-token
 * rows padded to whole groups of 32 (ngroups <= 8), K zero-padded to 256; bias (ngroups*32) or NULL; rscale indexed by
 * (row % rperiod) or NULL; out / R rows are (b, t) with strides ldo / ldr. */
int mlpk_token_gemm(int dtype, const void* xt, int ldxt, int M, int S, const void* w, int ldw, const float* bias, int ngroups,
                    const float* rscale, int rperiod, const void* R, int ldr, int res_mode, void* out, int ldo, int t_rows,
                    void* stream);
/* ABI 8.  The same product with its operand built on the fly: xt[b*t_rows + c, s] = (x[b*S + s, c] - mean[b*S + s]) * rstd[b*S + s] *
 * gamma[c] + beta[c] (rounded to the storage type once) -- the LayerNorm of gMLP's spatial gating unit (g_mlp.py:19) or, with mean /
 * rstd NULL (0 / 1), ResMLP's Aff (res_mlp.py:17-19) -- read straight from the token-major x (B*S rows of stride ldx; the pointer
 * addresses channel 0 of the t_rows channels, so a column slice of a wider tensor is fine) and transposed through LDS inside the
 * kernel: no xt tensor, no normalise-and-transpose pass.  t_rows % 32 == 0, S <= 224; everything else as mlpk_token_gemm.
 * res_mode MLPK_RES_ADD_AFFINE (ABI 9; no statistics, R NULL or == x): the residual is the affine output itself, round(gamma x + beta) rebuilt
 * in the kernel (res_mlp.py:53-55 adds the cross-patch product onto the POST-affine tensor); out may be x (every element is read by the
 * workgroup that writes it, before it is written).  Plain MLPK_RES_ADD adds R as stored and refuses R == x (MLPK_EMODE): until ABI 8 that
 * aliasing silently selected the affine residual. */
int mlpk_token_gemm_ln(int dtype, const void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                       const float* beta, const void* w, int ldw, const float* bias, int ngroups, const float* rscale, int rperiod,
                       const void* R, int ldr, int res_mode, void* out, int ldo, int t_rows, void* stream);
/* mlpk_token_mlp's `stats` (optional, t_rows % 128 == 0): the statistics of the LayerNorm that follows (mlp_mixer.py:38) come out of the epilogue:
 * stats[(tile*B*S + b*S + s)*2 + {0,1}] = sum / sum of squares over the tile's 128 channels of the values written to x[b,s,:]
 * (planar: t_rows/128 planes of B*S pairs, B = M / t_rows).  mlpk_stats_finalize_planar(stats, B*S, t_rows/128, B*S, 1, t_rows, ..)
 * reduces the partials of a row to mean / rstd. */
/* mean / rstd from planar (sum, sum of squares) pairs -- mlpk_token_mlp's `stats`, mlpk_gemm_nt's row_part: statistic r covers
 * rows [r*group, (r+1)*group) of all `nplanes` planes, pair (q, m) at part[(q*plane_stride + m)*2]; count = elements per
 * statistic (group * row length).  group = 1: LayerNorm of the producer's rows; group = H*W: GroupNorm(1,C) per sample on
 * channel-last rows (as_mlp.py:343-344).  rstd = 1 / sqrt(max(0, S2/count - mean^2) + eps). */
/* mlpk_token_gemm_ln with a per-channel affine applied to what is stored (round 5): out = post_scale[c] * round(result) + post_shift[c], rounded
 * again -- ResMLP's post_affine (res_mlp.py:56), which overwrites the cross-patch sublayer's output.  Only on the pipelined kernel (>= 3 groups of 32
 * tokens, an even token count, t_rows <= 1024): MLPK_ESHAPE otherwise, and the caller applies the affine with mlpk_norm_apply.  NULL, NULL = mlpk_token_gemm_ln. */
int mlpk_token_gemm_ln_post(int dtype, const void* x, int ldx, int M, int S, const float* ln_mean, const float* ln_rstd, const float* gamma,
                            const float* beta, const void* w, int ldw, const float* bias, int ngroups, const float* rscale, int rperiod, const void* R,
                            int ldr, int res_mode, const float* post_scale, const float* post_shift, void* out, int ldo, int t_rows, void* stream);
int mlpk_stats_finalize_planar(const float* part, int64_t rows, int nplanes, int64_t plane_stride, int group, int64_t count, float eps,
                               float* mean, float* rstd, void* stream);
/* Tuning hook (tools/tokenmlp_timeline.py), not part of the forward path: when `buf` is non-NULL, later mlpk_token_mlp
 * launches log per-workgroup s_memtime stamps into it (64 x uint64 per workgroup); NULL switches the logging off.
 * The only library-held state, and off by default. */
void mlpk_token_mlp_debug(void* buf);

/* ---- patch gather (im2col of a kernel==stride convolution) -----------------------------
 * out[(b*Hp + hp)*Wp + wp][k], row stride ldo (>= K, pad columns [K, ldo) are zero-filled).
 * src_layout NCHW : src is (B,Cin,H,W) of src_dtype; k = ci*ph*pw + i*pw + j (conv weight flattening)
 * src_layout NHWC : src is (B,H,W,Cin) of src_dtype with pixel stride lds_px elements;
 *                   k = (i*pw + j)*Cin + ci  (the host permutes the weight once)
 * Pixel (hp*ph + i - pad, wp*pw + j - pad); outside the image -> 0.  Hp = (H + 2*pad - ph)/ph + 1.
 * order 0: patch offsets enumerate i (rows) outer, j inner.
 * order 1 (NHWC only): PatchMerging order of as_mlp.py:207-211 -- (i,j) in (0,0),(1,0),(0,1),(1,1).
 */
#define MLPK_LAYOUT_NCHW 0
#define MLPK_LAYOUT_NHWC 1
int mlpk_patchify(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out,
                  int B, int Cin, int H, int W, int ph, int pw, int pad, int src_px_stride,
                  int ldo, int order, void* stream);

/* round 6 (ABI 12) -- the 4 x 4 patch embedding of the hierarchical families in one kernel: out[(b, py, px), :] = [LayerNorm](W patch + bias), patch = the
 * 3 x 4 x 4 window of the NCHW image in Conv2d's (channel, row, column) order (swin_mlp.py:324-333, ms_mlp.py:255-262, as_mlp.py:319,330, sparse_mlp.py).
 * x: (B, 3, H, W), src_dtype = dst_dtype or fp32 (converted on load, as mlpk_patchify does); w: (C, ldw >= 48) 16-bit; C = 32 / 64 / 96 / 128; gamma = beta =
 * NULL: no LayerNorm.  With it: the product is rounded to the storage type, two-pass statistics of the rounded values, (v - mean) rstd gamma + beta, one
 * more rounding -- the arithmetic of mlpk_gemm_nt + mlpk_row_stats + mlpk_norm_apply, which it replaces (not their bits: the product's summation order
 * is the 32 x 32 x 16 MFMA's).  mlpk_patch_embed4_supported says whether the call takes a shape. */
int mlpk_patch_embed4_supported(int src_dtype, int dst_dtype, int Cin, int H, int W, int C);
int mlpk_patch_embed4(int src_dtype, int dst_dtype, const void* x, int B, int Cin, int H, int W, const void* w, int ldw, const float* bias,
                      const float* gamma, const float* beta, float eps, void* out, int ldo, int C, void* stream);

/* round 6 (ABI 12) -- the 7 x 7 stride-4 stems (hire_mlp.py:21 pad 3, cycle_mlp.py:261 pad 2) as a direct convolution: out[(b, oy, ox), :] = W window + bias,
 * x: (B, 3, H, W) NCHW (dst dtype or fp32), W % 8 == 0, at most 64 output columns; w: (C, 176) 16-bit with k = (ci * 7 + i) * 8 + j (zero for j = 7 and
 * k >= 168: engine.pack_stem7); C = 32 .. 128 in steps of 32.  Replaces mlpk_im2col + mlpk_gemm_nt (same arithmetic, the 32 x 32 x 16 MFMA's summation order). */
int mlpk_stem7_supported(int src_dtype, int dst_dtype, int Cin, int H, int W, int pad, int C);
int mlpk_stem7(int src_dtype, int dst_dtype, const void* x, int B, int Cin, int H, int W, int pad, const void* w, const float* bias, void* out,
               int ldo, int C, float* out_mean, float* out_rstd, float eps, void* stream);   /* out_mean / out_rstd (or NULL, NULL): LayerNorm statistics of the rows written */

/* ---- row statistics ---------------------------------------------------------------------
 * For each of `rows` rows of `len` contiguous elements (row r starts at x + r*ldx):
 * mean[r], rstd[r] = 1/sqrt(biased_var + eps).  Two-pass (mean, then centred squares), fp32.
 * LayerNorm: rows = B*S, len = C.  GroupNorm(1,C): rows = B, len = C*H*W.
 */
int mlpk_row_stats(int dtype, const void* x, int64_t rows, int64_t len, int64_t ldx,
                   float eps, float* mean, float* rstd, void* stream);

/* ---- normalise / affine / activation + layout change -------------------------------------
 * y = (x[r,c] - mean[sr]) * rstd[sr] * gamma[c] + beta[c]      (mean/rstd NULL -> plain affine;
 * gamma/beta NULL -> 1/0);  y = gelu(y) if act;  sr = r / stat_group (stat_group = 1 for
 * LayerNorm; = H*W for GroupNorm(1,C) on channel-last data where one stat covers a whole sample).
 * x: rows x C, row stride ldx.  Outputs (any subset, NULL to skip):
 *   out_rm : row-major rows x C, stride ld_rm
 *   out_tt : token-transposed (B, C, ld_tt) with out_tt[(b*C + c)*ld_tt + s], r = b*S + s;
 *            columns [S, ld_tt) are zero-filled (they are the K padding of the token GEMM)
 *   out_ph / out_pw : the ViP rearranges (vip.py:69 / :74) of x viewed as (B,H,W,C=G*seg):
 *            out_ph[((b*W + w)*G + g)*ld_p + h*seg + j] = y[b,h,w,g*seg + j]
 *            out_pw[((b*H + h)*G + g)*ld_p + w*seg + j] = y[b,h,w,g*seg + j]; pad columns zero-filled
 */
typedef struct mlpk_norm_desc {
    int32_t dtype;
    int32_t act;
    int64_t rows;
    int32_t C;
    int32_t ldx;
    int32_t stat_group;
    int32_t S;            /* tokens per image (out_tt) */
    int32_t H, W, seg;    /* ViP geometry (out_ph/out_pw) */
    int32_t ld_rm, ld_tt, ld_p;
    const void* x;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    void* out_rm;
    void* out_tt;
    void* out_ph;
    void* out_pw;
    /* optional by-products of the out_ph / out_pw passes (fp32; 16-bit fast path only): the sums of the ROUNDED normalised values over
     * the axis the pass walks, laid out as the reduced operand of the OTHER branch --
     *   sum_ph[((b*G + g)*ld_sum) + w*seg + j] = sum_h y[b,h,w,g*seg+j]   (written by the out_ph pass, which walks h for a fixed w)
     *   sum_pw[((b*G + g)*ld_sum) + h*seg + j] = sum_w y[b,h,w,g*seg+j]   (written by the out_pw pass)
     * With them SplitAttention's sum over all pixels of the three branch OUTPUTS (vip.py:49) follows from the linearity of the
     * branch Linears without reading those outputs: sum_rows(A W^T + b) = (sum_rows A) W^T + rows * b. */
    float* sum_ph;
    float* sum_pw;
    int32_t ld_sum;
    int32_t reserved;
} mlpk_norm_desc;
int mlpk_norm_apply(const mlpk_norm_desc* d, void* stream);

/* Token-mixing LayerNorm of MLP-Mixer in one pass (mlp_mixer.py:34 with :6-13): statistics over C, affine, and the per-image
 * transpose the token GEMMs read:  out_tt[(b*C + c)*ld_tt + s] = LayerNorm_C(x[b,s,:])[c], columns S..ld_tt-1 written as zeros.
 * x is (nimg*S, C) with row stride ldx.  16-bit dtypes, C % 128 == 0, C <= 2048, ldx % 8 == 0, ld_tt % 8 == 0, 16-byte aligned
 * pointers; other shapes: mlpk_row_stats + mlpk_norm_apply(out_tt). */
int mlpk_layernorm_transpose(int dtype, const void* x, int64_t nimg, int S, int C, int ldx, const float* gamma, const float* beta,
                             float eps, void* out_tt, int ld_tt, void* stream);

/* ViP inverse rearranges (vip.py:71 / :76): z is the GEMM output in the permuted layout.
 * which = 0: out[b,h,w,g*seg+q] = z[((b*W + w)*G + g)*ldz + h*seg + q]
 * which = 1: out[b,h,w,g*seg+q] = z[((b*H + h)*G + g)*ldz + w*seg + q]        (out row stride C) */
int mlpk_vip_unpermute(int dtype, int which, const void* z, void* out, int B, int H, int W, int C,
                       int seg, int ldz, void* stream);

/* ---- mean over tokens --------------------------------------------------------------------
 * out[b,c] = mean_s y[b,s,c], y = x or the LayerNorm of x (mean/rstd per row given, gamma/beta
 * per channel); x is (B,S,C) with row stride ldx.  stat_group as in mlpk_norm_apply.
 * out dtype = dtype, row stride ldo.
 */
int mlpk_pool_mean(int dtype, const void* x, int B, int S, int C, int ldx, const float* mean,
                   const float* rstd, int stat_group, const float* gamma, const float* beta,
                   void* out, int ldo, void* stream);

/* ---- AS-MLP axial shift (the reference's one native op) -----------------------------------
 * group = ceil(C / kernel_size); s = kernel_size/2 - c/group;
 * NCHW: out[n,c,h,w] = in[n,c,h+s,w] (dim 2) | in[n,c,h,w+s] (dim 3), zero outside.
 * NHWC: same on (N,H,W,C).  kernel_size must be odd and >= 3, dim in {2,3}.
 */
int mlpk_shift_nchw(int dtype, const void* in, void* out, int N, int C, int H, int W,
                    int kernel_size, int dim, void* stream);
int mlpk_shift_nhwc(int dtype, const void* in, void* out, int N, int H, int W, int C,
                    int kernel_size, int dim, void* stream);
/* The op's backward (_shift.backward / shift_backward_grad_input_kernel: utils/shift_cuda.py:75-103,131-162):
 * grad_in[n,c,h,w] = grad_out[n,c,h-s,w] (dim 2) | grad_out[n,c,h,w-s] (dim 3), zero outside -- the adjoint of the forward gather.
 * Same argument checks as the forward; the caller allocates grad_in (shift_cuda.py:147). */
int mlpk_shift_nchw_backward(int dtype, const void* grad_out, void* grad_in, int N, int C, int H, int W,
                             int kernel_size, int dim, void* stream);

/* AS-MLP: t = act(GroupNorm(1,C)(in)) read through BOTH axial shifts in one pass, t itself never stored (as_mlp.py:64-66,84-95):
 *   out_w[n,h,w,c] = t[n,h,w+s,c],  out_h[n,h,w,c] = t[n,h+s,w,c],  t = act((in - mean[n]) * rstd[n] * gamma[c] + beta[c]),
 * s as in mlpk_shift_nhwc, zero outside the map.  16-bit dtypes, C % 8 == 0, ceil(C / kernel_size) >= 8. */
int mlpk_norm_shift_nhwc(int dtype, const void* in, void* out_w, void* out_h, int N, int H, int W, int C, int kernel_size,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, int act, void* stream);

/* ABI 8.  y = gelu(conv2_1(shift_W(u)) + b1) + gelu(conv2_2(shift_H(u)) + b2),  u = gelu((t - mean[b]) * rstd[b] * gamma[c] + beta[c])
 * (as_mlp.py:64-66,84-93) on channel-last t, y (B*H*W, C) in one kernel: a workgroup stages a band of image rows of u (with its zero
 * halo) in LDS and applies the per-channel-group pixel offsets  s(c) = k/2 - c / ceil(C/k)  (utils/shift_cuda.py:49-69) as LDS read
 * addresses of the matrix-core operands; mlpk_norm_shift_nhwc's two shifted copies and the two GEMM launches that read them are gone.
 * Results are bit-equal to that three-kernel sequence.  16-bit dtypes, kernel_size 5, C = 96 or 192 (AS-MLP-T / -S / -B stages 1-2;
 * mlpk_as_conv2_supported answers for a shape), w1 / w2 (C, ldw) = the Conv2d(C, C, 1) weights (out, in), y != t. */
int mlpk_as_conv2_supported(int dtype, int H, int W, int C, int kernel_size);
int mlpk_as_conv2(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                  void* stream);
/* ABI 10 (round 6).  mlpk_as_conv2 that also delivers the GroupNorm(1, C) statistics of y -- AxialShift's norm2 (as_mlp.py:52,94) -- so that
 * no statistics pass reads y back: every step of image rows leaves (sum, sum of squares) of the values it STORED (the rounded ones) at
 * part[2 (b * steps + s)], steps = mlpk_as_conv2_steps(...) (a function of dtype, C and the map only, so an image's statistics do not
 * depend on the batch it is in), and the image's pairs are added in step order in fp64 by one thread inside the kernel:
 * mean_out[b], rstd_out[b] = 1 / sqrt(var + eps), var = E[y^2] - mean^2 clamped at 0, over H*W*C values.  Small batches cut an image into
 * row segments for several workgroups: the one that finishes an image last (counter[b], B zeroed ints, left zeroed) adds the pairs.
 * mean_out / rstd_out must not be mean / rstd (other workgroups still read those).  part: B * steps * 2 floats, 8-byte aligned. */
int mlpk_as_conv2_steps(int dtype, int H, int W, int C, int kernel_size);
int mlpk_as_conv2_stats(int dtype, const void* t, void* y, int B, int H, int W, int C, int kernel_size, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2, const float* b2, int ldw,
                        float* part, float* mean_out, float* rstd_out, int* counter, float eps, void* stream);

/* ---- fused channel MLP for narrow channel-last tensors (ABI 8, round 4) -------------------------------------------------------
 * out[m, :] = R[m, :] + W2 . gelu( W1 . norm(x[m, :]) + b1 ) + b2      (as_mlp.py:36-52 with :343-344; the Mlp / FeedForward of every
 * hierarchical family while its stage is at most 192 channels wide): both products, the GELU and the residual in ONE kernel, the
 * hidden (M x 4C) never written.  16-bit dtypes, C % 32 == 0, 64 <= C <= 192, hidden = 32 nchunks <= 1024.
 *   x (M, ldx), R (M, ldr) or NULL, out (M, ldo) -- out may be x and / or R (a workgroup reads its 256 rows before it writes them);
 *   norm: ln_mean / ln_rstd NULL = none; else statistic m / ln_group of row m (LayerNorm: ln_group = 1; GroupNorm(1, C): H*W) applied
 *     on the first accumulator as in mlpk_gemm_nt: v = (acc - mean * csum[h]) * rstd + b1[h] with gamma folded into w1, beta into b1,
 *     csum[h] = the row sums of the rounded folded w1 (nchunks*32 floats, zero-padded);
 *   w1 (nchunks*32, ldw1 = 256): hidden rows, K zero-padded to 256 (the W1 of mlpk_token_mlp); b1 (nchunks*32);
 *   w2 (C, ldw2 >= nchunks*32), zero-padded, with
 *     - the COLUMN order of mlpk_token_mlp's layout 1 inside every group of 32 hidden units: slot 8 f + e <- unit (e < 4 ? 4 f + e : 16 + 4 f + e - 4),
 *     - the ROW order: inside every group of 32 output channels, row 16 h + 4 f + r (h < 2, f < 4, r < 4) <- channel 8 f + 4 h + r
 *       (a lane's accumulators are then 8 consecutive channels of one row: 16-byte stores straight from the registers);
 *   b2 (C) in natural channel order;
 *   row_part (optional): (sum, sum of squares) of the C values WRITTEN to row m (after the rounding) at row_part[2 m], [2 m + 1] -- one
 *     plane of mlpk_stats_finalize_planar (nplanes = 1) for the LayerNorm / GroupNorm that follows; a row is summed inside one wave
 *     in one fixed order, so the pair does not depend on the batch.
 * Numerics: fp32 accumulation in K order starting from R + b2; GELU and roundings as in the GEMM epilogues (one rounding of the hidden
 * to the storage type, one of the result). */
/* The FIRST product alone, same machinery (rows resident in registers, two waves per SIMD alternating between MFMAs and the GELU, no
 * epilogue): out[m, n] = gelu( norm-fold(x[m, :] . w1[n, :]) + b1[n] ) for a short K -- gMLP's channel_proj1 (g_mlp.py:28,35), the fc1 of
 * the K <= 512 channel MLPs (vip.py:82-88, res_mlp.py:21-32, s2_mlp_v2.py:78-84, as_mlp.py stage 3).  16-bit dtypes, M % 256 == 0,
 * K in {128, 192, 256, 384, 512}, N = 32 nchunks <= 4096.
 *   w1 (N, ldw1 >= K), zero-padded, ROWS of every group of 32 stored as [row 16 j + 4 f + r <- output column 8 f + 4 j + r] (j < 2, f < 4,
 *   r < 4); b1 and csum (N floats each) in the same order; norm as for mlpk_channel_mlp (ln_mean NULL: none);
 *   row_part (optional): by-product planes of 32 output columns in the canonical order of mlpk_gemm_desc.row_part -- pair of row m in
 *   plane g at row_part[2 (g M + m)] -- so mlpk_stats_finalize_planar(row_part, M, N / 32, M, ...) (or any sub-range of planes) applies. */
int mlpk_linear_gelu_supported(int dtype, int M, int K, int N);
int mlpk_linear_gelu(int dtype, const void* x, int ldx, int M, int K, const float* ln_mean, const float* ln_rstd, int ln_group,
                     const float* csum, const void* w1, int ldw1, const float* b1, int nchunks, void* out, int ldo, float* row_part,
                     void* stream);
int mlpk_channel_mlp_supported(int dtype, int C, int hidden);
int mlpk_channel_mlp(int dtype, const void* x, int ldx, int M, int C, const float* ln_mean, const float* ln_rstd, int ln_group,
                     const float* csum, const void* w1, int ldw1, const float* b1, const void* w2, int ldw2, const float* b2,
                     int nchunks, const void* R, int ldr, void* out, int ldo, float* row_part, void* stream);
/* ---- CycleFC sampling (CycleMLP) ---------------------------------------------------------------
 * in: (B,H,W,C) channel-last with pixel stride ldi.  d(c) = (c + k/2) % k - k/2  (gen_offset, cycle_mlp.py:104-120):
 *   out_h[b,y,x,c] = in[b, y, x + d(c), c]     the operand of `sfc_h` = CycleFC(kernel (1,k))
 *   out_w[b,y,x,c] = in[b, y + d(c), x, c]     the operand of `sfc_w` = CycleFC(kernel (k,1))
 * zero where the source pixel lies outside the map; either output may be NULL; pixel stride ldo; k odd.
 */
/* (round 5) mlpk_cycle_shift_ln: the same on LayerNorm(in) without storing it -- `in` un-normalised, mean / rstd per pixel, gamma / beta per channel,
 * applied to every element with the statistics of the pixel it comes from ((x - mean) rstd gamma + beta, one rounding: mlpk_norm_apply's expression);
 * 16-bit storage, C % 8 == 0, k = 3 / 5 / 7 (MLPK_ESHAPE otherwise). */
int mlpk_cycle_shift_ln(int dtype, const void* in, const float* mean, const float* rstd, const float* gamma, const float* beta, void* out_h, void* out_w,
                        int B, int H, int W, int C, int k, int ldi, int ldo, void* stream);
int mlpk_cycle_shift(int dtype, const void* in, void* out_h, void* out_w, int B, int H, int W, int C, int k,
                     int ldi, int ldo, void* stream);

/* ---- split attention (ViP / S2-MLPv2) -----------------------------------------------------
 * Three branch tensors x_k (B,H,W,C), k=0..2, each with its own pixel stride ld_k (so they may be
 * column slices of one (B,H,W,3C) buffer).  shift_mode selects a gather applied to branches 0/1
 * while loading (s2_mlp_v2.py:15-29); branch 2 is never shifted:
 *   MLPK_SHIFT_NONE      ViP
 *   MLPK_SHIFT_S2        branch0 = spatial_shift1, branch1 = spatial_shift2, clean 1-pixel shift
 *   MLPK_SHIFT_S2_REF    same, with the reference's deterministic in-place ("smear") behaviour
 * split_sum:    a[b,c] = scale * sum_{k,h,w} x_k[b,h,w,c]             (fp32; scale = 1 for the SplitAttention of ViP / S2-MLPv2,
 *                                                                     1 / (H W) for CycleMLP's mean, cycle_mlp.py:169)
 * split_softmax: bar[b,k,c] = softmax_k(hat[b, k*C + c])             (fp32 in/out)
 * split_apply:  out[b,h,w,c] = sum_k bar[b,k,c] * x_k[b,h,w,c]       (out pixel stride ldo)
 */
#define MLPK_SHIFT_NONE 0
#define MLPK_SHIFT_S2 1
#define MLPK_SHIFT_S2_REF 2
int mlpk_split_sum(int dtype, const void* x0, const void* x1, const void* x2, int ld0, int ld1,
                   int ld2, int B, int H, int W, int C, int shift_mode, float scale, float* a, void* stream);
int mlpk_split_softmax(const float* hat, float* bar, int B, int C, void* stream);
int mlpk_split_apply(int dtype, const void* x0, const void* x1, const void* x2, int ld0, int ld1,
                     int ld2, int B, int H, int W, int C, int shift_mode, const float* bar,
                     void* out, int ldo, void* stream);
/* ViP: the same reduction / weighted sum reading the H- and W-branch GEMM outputs WHERE THEY LIE (the layout of
 * mlpk_vip_unpermute's input), so the inverse rearranges of vip.py:71,76 are never materialised:
 *   xH[b,h,w,g*seg+q] = zh[((b*W + w)*G + g)*ldh + h*seg + q],  xW[b,h,w,g*seg+q] = zw[((b*H + h)*G + g)*ldw + w*seg + q],
 *   xc row-major (B*H*W, ldc).  16-bit dtypes, C % 8 == 0, seg % 4 == 0, ldh % 4 == ldw % 4 == 0. */
int mlpk_vip_split_apply(int dtype, const void* zh, const void* zw, const void* xc, int ldh, int ldw, int ldc, int B, int H,
                         int W, int C, int seg, const float* bar, void* out, int ldo, void* stream);
/* ABI 9 (round 5).  One branch of ViP's WeightedPermuteMLP in ONE kernel -- LayerNorm + einops rearrange + Linear (vip.py:66-76):
 *   out[((b*O + o)*G + g)*ldz + n] = bias[n] + sum_{l,j} w[n*ldw + l*seg + j] * LN(x)[b, pixel(o, l), g*seg + j]
 * which = 0: the h branch (o = w, l = h: 'b h w (c s) -> b w c (h s)'), 1: the w branch (o = h, l = w); G = C / seg, N = K = L*seg.
 * The rearranged operand is staged in LDS in operand order and multiplied where it lies: no rearranged tensor in HBM (what
 * mlpk_norm_apply(out_ph / out_pw) + mlpk_gemm_nt moved: 2 x the activation per branch), bit-equal to that pair.  mean / rstd: the
 * LayerNorm statistics per pixel; sums (optional): sums[(b*G + g)*ld_sum + o*seg + j] = sum over l of the rounded LN(x) -- the by-product
 * SplitAttention's linearity trick reads (mlpk_norm_desc.sum_ph / sum_pw).  16-bit dtypes; C / seg == 32, seg % 4 == 0, K in {128, 256, 384}
 * (mlpk_vip_branch_supported). */
int mlpk_vip_branch_supported(int dtype, int H, int W, int C, int seg, int which);
int mlpk_vip_branch(int dtype, const void* x, int ldx, int B, int H, int W, int C, int seg, int which, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const void* w, int ldw, const float* bias, void* out, int ldz, float* sums, int ld_sum,
                    void* stream);
/* S2-MLPv1 Spatial_Shift on (B,H,W,C), out of place, same shift_mode values (NONE = copy). */
int mlpk_s2_shift(int dtype, const void* in, void* out, int B, int H, int W, int C, int ldi,
                  int ldo, int shift_mode, void* stream);

/* ---- ConvMixer depthwise half --------------------------------------------------------------
 * x, out: (B,H,W,C) channel-last.  w: float32 [k*k][C] (tap-major), bias/bn_scale/bn_shift float32 [C].
 * out = x + (gelu(dwconv_same(x) + bias) * bn_scale + bn_shift)        (conv_mixer.py:24-28, 5-11)
 * k odd, <= 13: 3 / 5 / 7 / 9 have the LDS-tiled and (16-bit, maps <= 32 x 32) matrix-core forms, the others the generic kernel.  An EVEN
 * kernel size of Conv2d(padding="same") pads (k - 1) / 2 before and k / 2 after: the caller passes it as the odd size k + 1 with a zero tap
 * row and column in front (models_pytorch/conv_mixer.py does).
 */
int mlpk_dwconv_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int k,
                     const float* w, const float* bias, const float* bn_scale,
                     const float* bn_shift, void* stream);

/* ---- Sparse-MLP depthwise step ------------------------------------------------
 * x, out: (B,H,W,C) channel-last, C a multiple of the 16-byte vector.  w: float32 [k*k][C] (tap-major), k odd.
 * out = x + dwconv_same(pre_scale[c] * x + pre_shift[c]) + bias[c], zero padding applied after the affine
 * (BatchNorm2d(eval) -> Conv2d(C, C, 3, padding=1, groups=C) inside a PreNormResidual: sparse_mlp.py:9-15, 84-87).
 */
int mlpk_dwconv_affine_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int k,
                            const float* w, const float* bias, const float* pre_scale,
                            const float* pre_shift, void* stream);

/* ---- general window gather (Hire-MLP's overlapping convolutions) ----------------
 * mlpk_patchify with a stride of its own: window kh x kw, stride (stride_h, stride_w), zero padding `pad`;
 * out rows (b, ho, wo), Ho = (H + 2 pad - kh) / stride_h + 1; column order as mlpk_patchify (order 0):
 * NCHW source ci*kh*kw + i*kw + j, NHWC source (i*kw + j)*Cin + ci.  hire_mlp.py:21 (7x7 s4 p3), :161 (3x3 s2 p1).
 */
int mlpk_im2col(int src_dtype, int dst_dtype, int src_layout, const void* src, void* out, int B, int Cin,
                int H, int W, int kh, int kw, int stride_h, int stride_w, int pad, int src_px_stride,
                int ldo, void* stream);

/* ---- Hire-MLP region remaps (hire_mlp.py:44-152) -------------------------------------------------
 * xn: (B,H,W,C) channel-last LayerNorm output.  Hp = H + (h - H % h), Wp = W + (w - W % w) (circular padding, a whole extra
 * region when the size divides: hire_mlp.py:131-133), gh = Hp / h, gw = Wp / w, step = the cross-region roll (0 = none).
 * gather : a_h[((b*gh + g)*W + x)*ld_h + hh*C + c] = xn[b, P_H((hh*gh + g - step) mod Hp), x, c]     rows = B*gh*W, K = h*C
 *          a_w[((b*H + y)*gw + g)*ld_w + ww*C + c] = xn[b, y, P_W((ww*gw + g - step) mod Wp), c]     rows = B*H*gw, K = w*C
 *          with P_N(q) = q < N ? q : q - N.  The K order is (region row, channel): the 1x1-conv weights of proj_h / proj_w
 *          are permuted from the reference's (channel, region row) order when they are packed.
 * combine: x[b,y,x',c] += y_h[((b*gh + qh % gh)*W + x')*ld_h + (qh / gh)*C + c] + y_w[((b*H + y)*gw + qw % gw)*ld_w + (qw / gw)*C + c],
 *          qh = (y + step) mod Hp, qw = (x' + step) mod Wp          (restore + roll back + crop, :144-150)
 */
int mlpk_hire_gather(int dtype, const void* xn, void* a_h, void* a_w, int B, int H, int W, int C, int h, int w,
                     int step, int ld_h, int ld_w, void* stream);
int mlpk_hire_combine(int dtype, void* x, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                      int step, int ld_h, int ld_w, void* stream);
/* round 5 -- the block without a stored LayerNorm output: gather_ln takes the UN-normalised x and applies the block's LayerNorm (hire_mlp.py:176;
 * mean / rstd per pixel, gamma / beta per channel, (x - mean) rstd gamma + beta with one rounding: mlpk_norm_apply's expression) to every vector it
 * moves; combine_from reads the tensor the branch results are added to from `src` and writes x = src + y_h + y_w (src = x + proj_c(LN(x)), produced by a
 * GEMM with the LayerNorm folded in, which cannot write in place). */
int mlpk_hire_gather_ln(int dtype, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* a_h, void* a_w,
                        int B, int H, int W, int C, int h, int w, int step, int ld_h, int ld_w, void* stream);
int mlpk_hire_combine_from(int dtype, void* x, const void* src, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                           int step, int ld_h, int ld_w, void* stream);
/* round 6 (ABI 12) -- combine_from that also delivers the LayerNorm statistics (mean, 1 / sqrt(var + eps), biased variance, per pixel, fp32) of the
 * rows it writes: what the block's second PreNormResidual needs (hire_mlp.py:181) without another pass over x.  Sums of the ROUNDED results. */
int mlpk_hire_combine_stats(int dtype, void* x, const void* src, const void* y_h, const void* y_w, int B, int H, int W, int C, int h, int w,
                            int step, int ld_h, int ld_w, float* out_mean, float* out_rstd, float eps, void* stream);

/* ---- MS-MLP mix-shift (ms_mlp.py:48-66) -------------------------------------------
 * x, out: (B,H,W,C) channel-last.  The C channels form `groups` <= 8 chunks of ceil(C/groups) channels (torch.chunk); chunk g is
 * rolled by shift[g] along W (left/right branch) and along H (top/down branch) and each rolled copy goes through its own
 * ksize[g] x ksize[g] depthwise convolution (odd, zero padding ksize/2); out = branch_lr + branch_td (biases included).
 * w_lr / w_td: float32 [kmax*kmax][C], channel c's taps in rows dy*k + dx with k = its chunk's ksize; b_lr / b_td: float32 [C].
 * The host pointers `shift`, `ksize` are read at launch time (they travel as kernel arguments).
 */
int mlpk_mixshift_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift,
                       const int* ksize, const float* w_lr, const float* b_lr, const float* w_td, const float* b_td,
                       void* stream);
/* round 6 (ABI 12) -- the same with by-product statistics for the LayerNorm that follows (ms_mlp.py:72): planar (sum, sum of squares) pairs of the
 * STORED values, one plane per 32 channels, pair (q, row) at row_part[(q * row_part_ld + row) * 2] -- what mlpk_stats_finalize_planar takes.
 * 16-bit dtypes, C % 32 == 0, kernel sizes 1 / 3 / 5 / 7, maps up to 56 columns: mlpk_mixshift_stats_planes returns the number of planes (C / 32) for a
 * shape the call takes, 0 otherwise (then: mlpk_mixshift_nhwc + mlpk_row_stats). */
int mlpk_mixshift_stats_planes(int dtype, int B, int H, int W, int C, int groups, const int* ksize);
int mlpk_mixshift_nhwc_stats(int dtype, const void* x, void* out, int B, int H, int W, int C, int groups, const int* shift,
                             const int* ksize, const float* w_lr, const float* b_lr, const float* w_td, const float* b_td,
                             float* row_part, long long row_part_ld, void* stream);

/* ---- Swin-MLP window partition / merge (swin_mlp.py:29-60, 122-151) ---------------
 * The (B,H,W,C) map is zero-padded to Hp x Wp (pad_t rows on top, pad_l columns on the left, the rest at the bottom / right;
 * Hp, Wp multiples of ws) and cut into ws x ws windows; `windows` holds rows ((b, wy, wx), (iy, ix)) of C channels.
 * gather     : windows <- padded map (zeros in the padding)
 * scatter_add: x[b,y,x',:] += windows[row of padded position (y + pad_t, x' + pad_l)]      (merge, crop, residual)
 */
int mlpk_window_gather(int dtype, const void* x, void* windows, int B, int H, int W, int C, int ws, int pad_t, int pad_l,
                       int Hp, int Wp, void* stream);
int mlpk_window_scatter_add(int dtype, void* x, const void* windows, int B, int H, int W, int C, int ws, int pad_t,
                            int pad_l, int Hp, int Wp, void* stream);

/* ---- Swin-MLP: the spatial-MLP half of a block in ONE kernel (ABI 8, round 4) ------------------------------------------------
 * x[b,y,x',:] += crop(merge(spatial_mlp(partition(pad(LayerNorm(x))))))   (swin_mlp.py:97-151): LayerNorm with the given row statistics
 * (mean / rstd over x's B*H*W rows) and gamma / beta, zero padding to (Hp, Wp) with pad_t rows above and pad_l columns left, windows of
 * ws x ws positions, the grouped Conv1d over the ws^2 positions -- one (ws^2 x ws^2) matrix per head, head h = channels [32 h, 32 h + 32) --
 * and the residual, in place.  16-bit dtypes, C == 32 * heads (heads <= 24), ws^2 <= 64.
 *   w (heads, 64, 64): w[h][t_out][t_in] zero-padded (the Conv1d weight (heads * ws^2, ws^2, 1) reshaped); bias (heads, 64) zero-padded.
 * The normalised window is rounded to the storage type once (MFMA operand), the product once, the sum with x once -- the roundings of the
 * five-pass form it replaces (mlpk_norm_apply, mlpk_window_gather, transpose, mlpk_gemm_nt, mlpk_window_scatter_add). */
int mlpk_swin_spatial_supported(int dtype, int C, int heads, int ws);
/* ... mlpk_swin_spatial that also delivers the LayerNorm statistics (mean, 1 / sqrt(var + eps); fp32, per row) of the rows it writes: what the
 * block's norm2 (swin_mlp.py:154) needs, without a statistics pass (round 5).  out_mean / out_rstd NULL = mlpk_swin_spatial. */
int mlpk_swin_spatial_stats(int dtype, void* x, int B, int H, int W, int C, int ws, int pad_t, int pad_l, int Hp, int Wp, int heads,
                            const float* mean, const float* rstd, const float* gamma, const float* beta, const void* w, const float* bias,
                            float* out_mean, float* out_rstd, float eps, void* stream);
int mlpk_swin_spatial(int dtype, void* x, int B, int H, int W, int C, int ws, int pad_t, int pad_l, int Hp, int Wp, int heads,
                      const float* mean, const float* rstd, const float* gamma, const float* beta, const void* w, const float* bias,
                      void* stream);

/* ---- backward of the path (ABI 9, round 5) --------------------------------------------------------------------
 * What autograd through the GEMM epilogues needs beside mlpk_gemm_nt itself: for y = x W^T + b (+ r) the two products of the backward
 * are mlpk_gemm_nt calls on transposed operands -- dX = gemm(dY, W^T), dW = gemm(dY^T, X^T) (K-contiguous copies from
 * mlpk_transpose_batched) -- db = mlpk_col_sum(dY), dr = dY.  Reference: the autograd of nn.Linear / Conv1d(k=1), nn.GELU, nn.LayerNorm,
 * the token mean and the rearranges of mlp_mixer.py:6-27,34-38,62-75; BatchNorm2d's batch statistics conv_mixer.py:20,28,31.
 * All kernels: fp32 math, one rounding per stored value, deterministic summation orders (no atomics). */
/* mode 0: out = gelu(a) (exact erf form: train-mode forward keeps the pre-activation);  mode 1: out = b * gelu'(a)   (row-major, rows x cols, stride ld) */
int mlpk_gelu_elementwise(int dtype, int mode, const void* a, const void* b, void* out, int64_t rows, int cols, int64_t ld, void* stream);
/* nn.LayerNorm backward over the last axis with the forward's row statistics: dx = rstd (g - mean(g) - x^ mean(g x^)), g = dy gamma;
 * `part` receives mlpk_layernorm_backward_blocks(rows) x 2 x C floats: per row block the partial sums of (dy x^, dy); their column sums
 * (mlpk_col_sum over the blocks) are dgamma / dbeta.  C <= 2048. */
int mlpk_layernorm_backward_blocks(int64_t rows);
int mlpk_layernorm_backward(int dtype, const void* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma, const void* dy,
                            int64_t lddy, void* dx, int64_t lddx, float* part, int64_t rows, int C, void* stream);
/* out[c] = sum over rows of v[r, c] (square != 0: of v^2), fp32, v = x (sub NULL) or x - sub (same shape and stride): bias gradients, LayerNorm
 * parameter gradients, BatchNorm batch sums (with `sub`: of what a kernel with a built-in residual -- mlpk_dwconv_nhwc -- added to its input) */
int mlpk_col_sum(int dtype, const void* x, const void* sub, int64_t rows, int cols, int64_t ld, int square, float* out, void* stream);
/* out[b, c, r] = in[b, r, c] (+ res[b, c, r]):  in (batch, R, ld_in), out (batch, Cc, ld_out), res like out or NULL.  The padding columns
 * [R, ld_out) of out are not written (callers that use ld_out as a GEMM's K zero them once). */
int mlpk_transpose_batched(int dtype, const void* in, int64_t ld_in, void* out, int64_t ld_out, const void* res, int64_t ld_res, int batch,
                           int R, int Cc, void* stream);
/* out[b, s, c] = scale * in[b, c]: backward of Reduce('b n c -> b c', 'mean') with scale = 1 / S */
int mlpk_broadcast_rows(int dtype, const void* in, void* out, int B, int S, int C, float scale, void* stream);
/* Sparse-MLP's sMLP block up to the concatenation, behind the block's eval-mode BatchNorm (sparse_mlp.py:61-72,92), one launch:
 *   x^ = bn_scale[c] * x + bn_shift[c];   out[(b,h,w), :] = [ proj_h(x^) | proj_w(x^) | x^ ]   (3 C columns: the operand of the 3C -> C fuse)
 *   proj_h: x_h[b,h',w,c] = sum_h wh[h',h] x^[b,h,w,c] + bh[h'];  proj_w: x_w[b,h,w',c] = sum_w ww[w',w] x^[b,h,w,c] + bw[w']
 * x: (B*H*W, ldx) channel-last rows; wh / ww: (32, 32) in the storage type, zero-padded; bh / bw: (32) fp32, zero-padded; H, W <= 32, C % 32 == 0,
 * 16-bit storage (mlpk_smlp_mix_supported).  x^ and the two mixes are rounded once each. */
int mlpk_smlp_mix_supported(int dtype, int H, int W, int C);
int mlpk_smlp_mix(int dtype, const void* x, int ldx, int B, int H, int W, int C, const float* bn_scale, const float* bn_shift, const void* wh,
                  const float* bh, const void* ww, const float* bw, void* out, int ldo, void* stream);
/* The same with the block's first sublayer in front (sparse_mlp.py:88-91), for maps up to 15 x 15 (mlpk_smlp_mix_dw_supported: the tiles of two workgroups share a CU's LDS):
 *   xres = x + dwconv3x3(dw_scale[c] * x + dw_shift[c]) + dw_bias[c]   (zero padding on the affine's OUTPUT; dw_w: (9, C) fp32, tap 3 dy + dx --
 *   the operation and the bits of mlpk_dwconv_affine_nhwc with k = 3), stored to xres (B*H*W, ldxr; not x itself), then mlpk_smlp_mix of xres. */
int mlpk_smlp_mix_dw_supported(int dtype, int H, int W, int C);
int mlpk_smlp_mix_dw(int dtype, const void* x, int ldx, int B, int H, int W, int C, const float* dw_w, const float* dw_bias, const float* dw_scale,
                     const float* dw_shift, void* xres, int ldxr, const float* bn_scale, const float* bn_shift, const void* wh, const float* bh,
                     const void* ww, const float* bw, void* out, int ldo, void* stream);
/* x[m, c] += t[m % period, c] for m < rows (x: rows x C with row stride ldx; t: period x C, fp32; one rounding): the absolute position embedding
 * of SwinMLP added to every image's tokens (swin_mlp.py:386-388,437-438: ape=True) */
int mlpk_add_periodic(int dtype, void* x, int64_t ldx, const float* t, int64_t rows, int C, int period, void* stream);

/* ---- backward of the other families (ABI 11, round 6) ----------------------------------------------------------
 * The element-wise, normalisation and remap derivatives that the train mode of gMLP (g_mlp.py:10-39), ResMLP (res_mlp.py:11-57), AS-MLP
 * (as_mlp.py:55-162,182-216) and ConvMixer (conv_mixer.py:5-39) needs beside the ABI-9 set; products stay mlpk_gemm_nt calls.  fp32 math,
 * one rounding per stored value, fixed summation orders. */
/* per-column / per-row-group combinations of row-major tensors (rows x cols; strides lda / ldb / ldo; g, h, k fp32 vectors):
 *   mode 0: out = a * g[c] + h[c]  (g NULL = 1, h NULL = 0)   Aff (res_mlp.py:17-19); the affine half of GroupNorm / BatchNorm; dx of a scale
 *   mode 1: out = a * b                                       the SGU gate u * v (g_mlp.py:21) and both its derivatives
 *   mode 2: out = a + g[c] * b     (g NULL: a + b)            x + gamma * f(x) (res_mlp.py:53,55); sums of gradient paths
 *   mode 3: out = a * g[m / period]                           stochastic depth's per-sample scale (as_mlp.py:159-160) and its derivative
 *   mode 4: out = a * g[c] + b * h[c] + k[c]                  BatchNorm backward on batch statistics (conv_mixer.py:20,28,31)
 *   mode 5: out = a * g[i, c] + h[i, c] + b,  i = m / period  (g: (rows / period) x cols; h, b optional)  one branch's term of SplitAttention's weighted
 *                                                             sum (vip.py:54-56; s2_mlp_v2.py:48-50) and its derivative w.r.t. the branch */
int mlpk_ew_cols(int dtype, int mode, const void* a, int64_t lda, const void* b, int64_t ldb, const float* g, const float* h, const float* k,
                 void* out, int64_t ldo, int64_t rows, int cols, int period, void* stream);
/* out[c] = sum over rows of x[r, c] * y[r, c], fp32: the gradient of a per-channel scale (Aff alpha, gamma_1 / gamma_2, GroupNorm / BatchNorm weight) */
int mlpk_col_dot(int dtype, const void* x, int64_t ldx, const void* y, int64_t ldy, int64_t rows, int cols, float* out, void* stream);
/* GroupNorm(1, C) backward on channel-last samples of glen = H W C contiguous values (as_mlp.py:343-344): xh = the normalised values,
 * g = dy * gamma[c];  dx = rstd[b] * (g - mean_b(g) - xh * mean_b(g xh)) */
int mlpk_group_norm_backward(int dtype, const void* xh, const void* g, const float* rstd, void* dx, int groups, int64_t glen, void* stream);
/* adjoint of mlpk_shift_nhwc: grad_in[n,h,w,c] = grad_out[n,h-s,w,c] (dim 2) | grad_out[n,h,w-s,c] (dim 3), zero outside
 * (shift_backward_grad_input_kernel, utils/shift_cuda.py:75-103, on the channel-last layout) */
int mlpk_shift_nhwc_backward(int dtype, const void* grad_out, void* grad_in, int N, int H, int W, int C, int kernel_size, int dim, void* stream);
/* PatchMerging's gather on channel-last tensors (as_mlp.py:207-211; the order of mlpk_patchify order 1) and its adjoint:
 * dir 0: src (B,H,W,C) -> dst (B,H/2,W/2,4C);  dir 1: src (B,H/2,W/2,4C) -> dst (B,H,W,C).  H, W even. */
int mlpk_merge2x2_nhwc(int dtype, int dir, const void* src, void* dst, int B, int H, int W, int C, void* stream);
/* the general form: the im2col half of any kernel == stride convolution read from channel-last rows, (B,H,W,C) -> (B,H/ph,W/pw, ph*pw*C) with
 * column ((i*pw + j) [order 0, = mlpk_patchify NHWC] | (j*ph + i) [order 1, PatchMerging]) * C + c, and back (dir 1): the stage convolutions of
 * S2-MLPv2 on the previous stage's output (s2_mlp_v2.py:118-119) and the gradient that flows back through them.  H % ph == 0, W % pw == 0. */
int mlpk_patch_rows_nhwc(int dtype, int dir, int order, const void* src, void* dst, int B, int H, int W, int C, int ph, int pw, void* stream);
/* depthwise Conv2d(k, groups = C, padding = "same") on (B,H,W,C) WITHOUT an epilogue (train mode keeps the pre-activation: conv_mixer.py:25),
 * w fp32 [k*k][C] tap-major like mlpk_dwconv_nhwc, bias fp32 [C] or NULL;  adjoint = 1: the gradient w.r.t. the input (bias ignored) */
int mlpk_dwconv_plain_nhwc(int dtype, int adjoint, const void* in, void* out, int B, int H, int W, int C, int k, const float* w, const float* bias,
                           void* stream);
/* dw[i*k + j][c] = sum over (b, y, x) of dy[b,y,x,c] * x[b, y+i-p, x+j-p, c], p = (k-1)/2: the gradient of the depthwise taps, fp32 */
int mlpk_dwconv_wgrad_nhwc(int dtype, const void* x, const void* dy, float* dw, int B, int H, int W, int C, int k, void* stream);

/* mlpk_col_dot per segment of seg_rows consecutive rows: out[i, c] = sum over the rows of segment i of x * y (the gradient of SplitAttention's
 * per-image branch weights: sum over an image's pixels of dy * x_k) */
int mlpk_col_dot_seg(int dtype, const void* x, int64_t ldx, const void* y, int64_t ldy, int segments, int64_t seg_rows, int cols, float* out, void* stream);
/* backward of mlpk_split_softmax (nn.Softmax(1) over the k = 3 branches, vip.py:52-53): fp32 [B][3][C]; dhat[k] = bar[k] (dbar[k] - sum_j bar[j] dbar[j]) */
int mlpk_split_softmax_backward(const float* bar, const float* dbar, float* dhat, int B, int C, void* stream);
/* spatial_shift1 / spatial_shift2 of S2-MLPv2 (s2_mlp_v2.py:15-29; which = 1 / 2) on (B, D1, D2, C) rows with strides ldi / ldo, out of place.
 * adjoint 0: the forward -- mode 0 the intended one-pixel shift, mode 1 the reference's in-place result (the +1 groups smear: y[i] = x[0]);
 * adjoint 1: what the reference's autograd returns for the in-place slice assignments -- the adjoint of the INTENDED shift, whatever the forward did. */
int mlpk_s2_shift2(int dtype, int which, int mode, int adjoint, const void* in, int64_t ldi, void* out, int64_t ldo, int B, int D1, int D2, int C, void* stream);

/* A remap as an index table shared by all images of a batch: dst[(b, i), :] = sum over q < kmax of src[(b, idx[i * kmax + q]), :], rows of `width`
 * contiguous elements, idx int32 (negative: no source).  kmax = 1: a gather -- the pads, rolls, window partitions, region rearranges and overlapping
 * convolution windows of Swin-MLP (swin_mlp.py:33-60,129-151), MS-MLP (ms_mlp.py:52-54), Hire-MLP (hire_mlp.py:38-152) and CycleMLP (cycle_mlp.py:
 * 104-131: CycleFC's fixed integer offsets, width = 1); the table of the inverse relation (kmax = largest multiplicity) is the adjoint -- the gradient. */
int mlpk_index_gather(int dtype, const void* src, void* dst, const int* idx, int batch, int64_t n_out, int64_t n_in, int width, int kmax, void* stream);

/* Tile-height plan of the persistent GEMM tile, process-wide (round 6): 0 (default) = mixed tile heights, the shortest SINGLE launch (Mixer-B fc2: 588 tiles on
 * 256 CUs run as 2.58 instead of 3 round-times); 1 = whole 256-row tiles only wherever they still fill a round of CUs and K >= 1024, the least TOTAL CU time (3 % less for that
 * product) -- the better plan when several forwards share the chip (parallel.InFlight) and the other request fills the last round anyway.  Same K order per output
 * element: results are bit-identical under either plan. */
int mlpk_gemm_set_plan(int mode);

/* ---- small utilities ------------------------------------------------------------------------ */
/* dst[i] = (dst_dtype) src[i], n elements */
int mlpk_convert(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLPK_H */
