"""ctypes binding of libmlpk.so (include/mlpk.h).  No CPU fallback: if the HIP library is
missing or a call fails, this raises -- the product path never silently degrades."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLPK_LIB_PATH") or os.path.join(_HERE, "lib", "libmlpk.so")    # (override: A/B builds of kernel variants, tools/build_variant.sh)

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_GELU = 0, 1
RES_NONE, RES_ADD, RES_MUL, RES_ADD_AFFINE = 0, 1, 2, 3
OUT_ROWMAJOR, OUT_TOKEN_T = 0, 1
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
SHIFT_NONE, SHIFT_S2, SHIFT_S2_REF = 0, 1, 2

c_void_p, c_int, c_i64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class GemmDesc(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("lda", ctypes.c_int32), ("ldb", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldr", ctypes.c_int32),
                ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("R", c_void_p),
                ("bias", c_void_p), ("cscale", c_void_p), ("cshift", c_void_p), ("rscale", c_void_p),
                ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("ln_csum", c_void_p),
                ("rperiod", ctypes.c_int32), ("act", ctypes.c_int32), ("res_mode", ctypes.c_int32),
                ("out_mode", ctypes.c_int32), ("t_rows", ctypes.c_int32), ("t_tokens", ctypes.c_int32),
                ("algo", ctypes.c_int32), ("ln_group", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("workspace", c_void_p), ("workspace_bytes", ctypes.c_int64),
                ("row_part", c_void_p), ("row_part_ld", ctypes.c_int32), ("reserved2", ctypes.c_int32)]


class NormDesc(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("act", ctypes.c_int32), ("rows", ctypes.c_int64), ("C", ctypes.c_int32),
                ("ldx", ctypes.c_int32), ("stat_group", ctypes.c_int32), ("S", ctypes.c_int32),
                ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("seg", ctypes.c_int32),
                ("ld_rm", ctypes.c_int32), ("ld_tt", ctypes.c_int32), ("ld_p", ctypes.c_int32),
                ("x", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("out_rm", c_void_p), ("out_tt", c_void_p), ("out_ph", c_void_p), ("out_pw", c_void_p),
                ("sum_ph", c_void_p), ("sum_pw", c_void_p), ("ld_sum", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# name -> (restype, argtypes); every symbol include/mlpk.h declares
PROTOTYPES = {
    "mlpk_abi_version": (c_int, []),
    "mlpk_strerror": (ctypes.c_char_p, [c_int]),
    "mlpk_gemm_nt": (c_int, [ctypes.POINTER(GemmDesc), c_void_p]),
    "mlpk_gemm_nt_pair": (c_int, [ctypes.POINTER(GemmDesc), ctypes.POINTER(GemmDesc), c_void_p]),
    "mlpk_conv_gemm_nhwc_supported": (c_int, [c_int] * 6),
    "mlpk_merge2x2_row_stats": (c_int, [c_int, c_void_p] + [c_int] * 4 + [c_float, c_void_p, c_void_p, c_void_p]),
    "mlpk_merge2x2_stats_combine": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "mlpk_conv_gemm_nhwc": (c_int, [ctypes.POINTER(GemmDesc)] + [c_int] * 8 + [c_void_p]),
    "mlpk_gemm_row_parts": (c_int, [ctypes.POINTER(GemmDesc), ctypes.POINTER(c_int)]),
    "mlpk_gemm_workspace_bytes": (ctypes.c_longlong, []),
    "mlpk_gemm_algo_count": (c_int, []),
    "mlpk_gemm_algo_info": (c_int, [c_int] + [ctypes.POINTER(c_int)] * 4),
    "mlpk_gemm_kernel_name": (c_int, [ctypes.POINTER(GemmDesc), ctypes.c_char_p, c_int]),
    "mlpk_token_mlp_chunk": (c_int, []),
    "mlpk_token_mlp": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mlpk_token_mlp_ln": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                  c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mlpk_token_mlp_layout": (c_int, [c_int, c_int]),
    "mlpk_token_mlp_layout_for": (c_int, [c_int, c_int, c_int, c_int]),
    "mlpk_token_gemm": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_void_p, c_int, c_int, c_void_p]),
    # (dtype, x, ldx, M, S, ln_mean, ln_rstd, gamma, beta, w, ldw, bias, ngroups, rscale, rperiod, R, ldr, res_mode, out, ldo, t_rows, stream)
    "mlpk_token_gemm_ln": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    # (... res_mode, post_scale, post_shift, out, ldo, t_rows, stream)
    "mlpk_token_gemm_ln_post": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mlpk_layernorm_transpose": (c_int, [c_int, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "mlpk_stats_finalize_planar": (c_int, [c_void_p, c_i64, c_int, c_i64, c_int, c_i64, c_float, c_void_p, c_void_p, c_void_p]),
    "mlpk_token_mlp_debug": (None, [c_void_p]),
    "mlpk_patchify": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "mlpk_row_stats": (c_int, [c_int, c_void_p, c_i64, c_i64, c_i64, c_float, c_void_p, c_void_p, c_void_p]),
    "mlpk_norm_apply": (c_int, [ctypes.POINTER(NormDesc), c_void_p]),
    "mlpk_vip_unpermute": (c_int, [c_int, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mlpk_pool_mean": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                               c_void_p, c_void_p, c_int, c_void_p]),
    "mlpk_shift_nchw": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mlpk_vip_branch_supported": (c_int, [c_int] * 6),
    "mlpk_vip_branch": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "mlpk_shift_nchw_backward": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mlpk_gelu_elementwise": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p]),
    "mlpk_layernorm_backward_blocks": (c_int, [c_i64]),
    "mlpk_layernorm_backward": (c_int, [c_int, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int,
                                        c_void_p]),
    "mlpk_col_sum": (c_int, [c_int, c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_void_p]),
    "mlpk_transpose_batched": (c_int, [c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "mlpk_broadcast_rows": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "mlpk_swin_spatial_stats": (c_int, [c_int, c_void_p] + [c_int] * 10 + [c_void_p] * 8 + [c_float, c_void_p]),
    "mlpk_add_periodic": (c_int, [c_int, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "mlpk_smlp_mix_supported": (c_int, [c_int] * 4),
    "mlpk_smlp_mix": (c_int, [c_int, c_void_p] + [c_int] * 5 + [c_void_p] * 7 + [c_int, c_void_p]),
    "mlpk_smlp_mix_dw_supported": (c_int, [c_int] * 4),
    "mlpk_smlp_mix_dw": (c_int, [c_int, c_void_p] + [c_int] * 5 + [c_void_p] * 5 + [c_int] + [c_void_p] * 7 + [c_int, c_void_p]),
    "mlpk_shift_nhwc": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mlpk_norm_shift_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 4 + [c_int, c_void_p]),
    "mlpk_cycle_shift": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "mlpk_cycle_shift_ln": (c_int, [c_int] + [c_void_p] * 7 + [c_int] * 7 + [c_void_p]),
    "mlpk_as_conv2_supported": (c_int, [c_int] * 5),
    "mlpk_channel_mlp_supported": (c_int, [c_int] * 3),
    "mlpk_linear_gelu_supported": (c_int, [c_int] * 4),
    "mlpk_linear_gelu": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_void_p]),
    "mlpk_swin_spatial_supported": (c_int, [c_int] * 4),
    "mlpk_swin_spatial": (c_int, [c_int, c_void_p] + [c_int] * 10 + [c_void_p] * 7),
    "mlpk_channel_mlp": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    # (dtype, t, y, B, H, W, C, kernel_size, mean, rstd, gamma, beta, w1, b1, w2, b2, ldw, stream)
    "mlpk_as_conv2": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 8 + [c_int, c_void_p]),
    "mlpk_as_conv2_steps": (c_int, [c_int] * 5),
    # (... ldw, part, mean_out, rstd_out, counter, eps, stream)
    "mlpk_as_conv2_stats": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 8 + [c_int] + [c_void_p] * 4 + [c_float, c_void_p]),
    "mlpk_split_sum": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_void_p, c_void_p]),
    "mlpk_split_softmax": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mlpk_split_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p, c_int, c_void_p]),
    "mlpk_vip_split_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p, c_int, c_void_p]),
    "mlpk_s2_shift": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "mlpk_dwconv_nhwc": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 4 + [c_void_p]),
    "mlpk_dwconv_affine_nhwc": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 4 + [c_void_p]),
    "mlpk_im2col": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "mlpk_patch_embed4_supported": (c_int, [c_int] * 6),
    "mlpk_stem7_supported": (c_int, [c_int] * 7),
    "mlpk_stem7": (c_int, [c_int, c_int, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "mlpk_patch_embed4": (c_int, [c_int, c_int, c_void_p] + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "mlpk_hire_gather": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "mlpk_hire_gather_ln": (c_int, [c_int] + [c_void_p] * 7 + [c_int] * 9 + [c_void_p]),
    "mlpk_hire_combine_from": (c_int, [c_int] + [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "mlpk_hire_combine": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "mlpk_hire_combine_stats": (c_int, [c_int] + [c_void_p] * 4 + [c_int] * 9 + [c_void_p, c_void_p, c_float, c_void_p]),
    "mlpk_mixshift_nhwc": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int)] + [c_void_p] * 4 + [c_void_p]),
    "mlpk_mixshift_nhwc_stats": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int)] + [c_void_p] * 4
                                 + [c_void_p, c_i64, c_void_p]),
    "mlpk_mixshift_stats_planes": (c_int, [c_int] * 6 + [ctypes.POINTER(c_int)]),
    "mlpk_window_gather": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "mlpk_window_scatter_add": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "mlpk_ew_cols": (c_int, [c_int, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_void_p]),
    "mlpk_col_dot": (c_int, [c_int, c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "mlpk_group_norm_backward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p]),
    "mlpk_shift_nhwc_backward": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mlpk_merge2x2_nhwc": (c_int, [c_int, c_int, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "mlpk_patch_rows_nhwc": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mlpk_gemm_set_plan": (c_int, [c_int]),
    "mlpk_index_gather": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_int, c_int, c_void_p]),
    "mlpk_dwconv_plain_nhwc": (c_int, [c_int, c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p]),
    "mlpk_dwconv_wgrad_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "mlpk_col_dot_seg": (c_int, [c_int, c_void_p, c_i64, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_void_p]),
    "mlpk_split_softmax_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mlpk_s2_shift2": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "mlpk_convert": (c_int, [c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p]),
}

_lib = None


class MlpkError(RuntimeError):
    pass


def lib():
    """Load libmlpk.so once.  Raises MlpkError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MlpkError("HIP kernel library %s is missing: run `python __graft_entry__.py build` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if handle.mlpk_abi_version() != 12:
            raise MlpkError("libmlpk.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mlpk_strerror(rc)
        raise MlpkError("%s failed (%d): %s" % (what or "mlpk call", rc, msg.decode() if msg else "?"))
