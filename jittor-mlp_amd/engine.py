"""Host-side plumbing between torch tensors and the C-ABI kernels (include/mlpk.h).

torch is used for device memory (caching allocator), streams and nothing else: every
function here hands raw device pointers + sizes to libmlpk.so on torch's current stream,
exactly like the reference's only native op does (utils/shift_cuda.py:112,122-125).
There is deliberately no CPU implementation: a non-GPU tensor raises NotImplementedError,
the same error type the reference's Shift raises on CPU (shift_cuda.py:170-173).
"""
import ctypes
import os

import threading

import torch

from . import _native as N

DT = {torch.float32: N.F32, torch.float16: N.F16, torch.bfloat16: N.BF16}


def dtype_code(dtype):
    try:
        return DT[dtype]
    except KeyError:
        raise TypeError("unsupported dtype %s (float32, float16, bfloat16 only)" % (dtype,))


def require_gpu(x, what="forward"):
    if not x.is_cuda:
        raise NotImplementedError("%s: the MI355X path needs a GPU tensor (no CPU implementation; "
                                  "the CPU oracle lives in oracle/ and is test-only)" % what)


_TLS = threading.local()          # .stream: the raw handle of the stream a running EngineModule.__call__ launches on (per thread)


def stream():
    """torch's current stream of the CURRENT device.  Every entry point that takes tensors runs under
    `on_device(x)` (EngineModule.__call__, Shift), which makes x's device current for the duration of the
    call, so launches, the stream and the device-attribute queries inside libmlpk.so all refer to the device
    the pointers live on -- also when the caller's current device is another GPU.
    Round 6: `torch.cuda.current_stream()` costs ~7 us of host time and was asked once per launch (Hire-MLP: 480 launches, 3.4 of 6.4 ms of
    host time per forward); EngineModule.__call__ asks once and keeps the handle for the duration of the call (thread-local; SideChain
    swaps it with the stream it enters).  Outside a module call (autograd's backward thread, the stand-alone ops) torch is asked as before."""
    h = getattr(_TLS, "stream", None)
    return h if h is not None else torch.cuda.current_stream().cuda_stream


_SIDE = {}
SIDE_STREAMS = True


def set_side_streams(on):
    """Process-wide switch of the side streams (SideChain then issues in line).  With several forwards in flight (parallel.InFlight) the other
    forward already fills what a small chain leaves idle, and the fork / join events only tie the streams together: Hire-MLP 9.47 -> 8.84 ms per
    step in flight without its side chain, 9.52 -> 10.10 ms one step at a time (profiles/r06_hire_combine_stats_ab.txt).  Same bits either way."""
    global SIDE_STREAMS
    SIDE_STREAMS = bool(on)


def side_stream(device):
    """One extra stream per device for small, latency-bound kernel chains that are independent of the big kernels issued next
    (ViP's SplitAttention MLP beside the branch GEMMs); None when MLPK_NO_SIDE_STREAM=1.  Fork / join are events on both sides:
        ev = torch.cuda.Event(); ev.record(); with torch.cuda.stream(side): side.wait_event(ev); ...; done.record(side)
        ...; torch.cuda.current_stream().wait_event(done)"""
    if not SIDE_STREAMS or os.environ.get("MLPK_NO_SIDE_STREAM", "0") == "1":
        return None
    key = (device.type, device.index)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=device)
    return st


class SideChain:
    """`with chain: <launches>` puts the launches on the device's side stream, ordered after everything issued so far on the
    current stream (fork event); `chain.join()` makes the current stream wait for them.  What runs inside must touch only buffers
    that nothing issued between the `with` block and join() touches.  With MLPK_NO_SIDE_STREAM=1 the launches simply stay on
    the current stream."""

    def __init__(self, ws, name, device):
        self.side = side_stream(device)
        self.main = torch.cuda.current_stream()
        if self.side is not None:
            self.fork_ev, self.join_ev = ws.get_event(name + ".fork"), ws.get_event(name + ".join")

    def __enter__(self):
        if self.side is not None:
            self.fork_ev.record(self.main)
            self.side.wait_event(self.fork_ev)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
            self._outer = getattr(_TLS, "stream", None)
            if self._outer is not None:
                _TLS.stream = self.side.cuda_stream            # (the launches inside go to the side stream: keep stream()'s cache in step)
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self.join_ev.record(self.side)
            self.ctx.__exit__(*exc)
            if self._outer is not None:
                _TLS.stream = self._outer
        return False

    def join(self):
        if self.side is not None:
            self.main.wait_event(self.join_ev)


def on_device(x):
    """Context manager: x's device current (hipSetDevice + torch's per-device current stream)."""
    return torch.cuda.device(x.device)


def ptr(t):
    return None if t is None else t.data_ptr()


def round_up(v, m):
    return (v + m - 1) // m * m


# ------------------------------------------------------------------ weight packing
def pack_matrix(w, dtype, device, kpad=8):
    """(N, K) weight -> contiguous (N, round_up(K, kpad)) in the compute dtype, zero padded."""
    w = w.detach().reshape(w.shape[0], -1)
    n, k = w.shape
    kp = round_up(k, kpad)
    out = torch.zeros((n, kp), dtype=dtype, device=device)
    out[:, :k] = w.to(device=device, dtype=dtype)
    return out


def embed_kpad(dtype):
    """K padding of a patch-embedding weight whose K is ragged (7 x 7 x 3 = 147): whole 64-byte slabs for the 16-bit dtypes, so that the product runs on the
    LDS-DMA tiles instead of the register-staged fallback (ConvMixer-1536/20's embedding: 565 us at K = 152); MLPK_EMBED_KPAD overrides (A/B aid)"""
    if os.environ.get("MLPK_EMBED_KPAD"):
        return int(os.environ["MLPK_EMBED_KPAD"])
    return 8 if dtype == torch.float32 else 32


def pack_ln_folded(w, b, gamma, beta, dtype, device, kpad=8):
    """Fold LayerNorm(gamma, beta) into the Linear(w, b) that consumes it:
    LN(x) W^T + b = rstd * (x W'^T - mu * csum) + b',  W' = W diag(gamma), csum[n] = sum_k W'[n,k]
    (taken over the ROUNDED packed values, so it cancels exactly what the MFMA accumulates),
    b' = b + W beta.  Returns (W' packed, b' fp32, csum fp32)."""
    w32 = w.detach().to(device=device, dtype=torch.float32).reshape(w.shape[0], -1)
    wf = w32 * gamma.detach().to(device=device, dtype=torch.float32).reshape(1, -1)
    packed = pack_matrix(wf, dtype, device, kpad)
    csum = packed.to(torch.float32).sum(dim=1).contiguous()
    bias = w32 @ beta.detach().to(device=device, dtype=torch.float32).reshape(-1)
    if b is not None:
        bias = bias + b.detach().to(device=device, dtype=torch.float32).reshape(-1)
    return packed, bias.contiguous(), csum


def f32(v, device):
    """Per-channel vectors always travel as float32."""
    if v is None:
        return None
    return v.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()


# ------------------------------------------------------------------ kernel wrappers
class KernelTimer:
    """Optional HIP-event timing of tagged GEMM launches on the launch stream (bench.py's roofline
    leg).  Disabled (None) in normal use: zero overhead on the product path."""

    def __init__(self):
        self.events = {}          # tag -> list of (start, end, flops)
        self.kernels = {}         # tag -> names of the kernels the dispatch chose for the timed calls (mlpk_gemm_kernel_name)

    def summary(self):
        out = {}
        for tag, evs in self.events.items():
            ms = [s.elapsed_time(e) for s, e, _ in evs]
            out[tag] = {"launches": len(evs), "avg_ms": sum(ms) / len(ms), "flops_per_launch": evs[0][2], "kernels": sorted(self.kernels.get(tag, ()))}
        return out


TIMER = None
GEMM_ALGO = {}                    # tag -> forced tile config (tuning/bench hook); default auto


def set_gemm_plan(whole_tiles):
    """mlpk_gemm_set_plan: tile-height plan of the persistent GEMM tile, process-wide.  False (default): mixed tile heights -- the shortest single
    launch (Mixer-B fc2: 2.58 instead of 3 round-times); True: whole 256-row tiles wherever they still fill a round of CUs and K >= 1024 -- 3 % less CU time in
    total, the better plan when several forwards share the chip (parallel.InFlight: Mixer-B/16, two in flight, 36.55 -> 37.3 k images/s; one at a
    time 34.8 -> 34.1 k).  Same K order per output element: same bits."""
    N.check(N.lib().mlpk_gemm_set_plan(1 if whole_tiles else 0), "mlpk_gemm_set_plan")
    global GEMM_PLAN_WHOLE
    GEMM_PLAN_WHOLE = bool(whole_tiles)


GEMM_PLAN_WHOLE = False


def gemm(A, B, C, M, Nn, K, *, lda=None, ldb=None, ldc=None, bias=None, act=N.ACT_NONE, cscale=None, cshift=None,
         rscale=None, rperiod=0, R=None, ldr=None, res=N.RES_NONE, out_mode=N.OUT_ROWMAJOR, t_rows=0, t_tokens=0,
         algo=0, tag=None, dbg=0, ln=None, ln_group=1, part=None, prof=None, _defer=False):
    """C = epilogue(A . B^T).  `part` = (workspace, name): the epilogue also delivers the row statistics of what it stores
    (mlpk.h: row_part) into a float32 buffer (nparts, M, 2) taken from the workspace; returns (buffer, nparts) for
    stats_finalize_planar, or None when the descriptor cannot deliver them (fp32, unaligned rows) and the caller runs row_stats."""
    if tag is not None and algo == 0:
        algo = GEMM_ALGO.get(tag, 0)
    if GEMM_LOG is not None:                             # tuning: the distinct GEMM calls of a forward (tools/gemm_shapes.py)
        GEMM_LOG.add((str(A.dtype), M, Nn, K, int(act), int(res), ln is not None, part is not None, cscale is not None or cshift is not None,
                      rscale is not None, int(out_mode), bias is not None))
    timed = TIMER is not None and tag is not None and not _defer
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    d = N.GemmDesc()
    d.dtype = dtype_code(A.dtype)
    d.M, d.N, d.K = M, Nn, K
    d.lda = lda if lda is not None else A.stride(0)
    d.ldb = ldb if ldb is not None else B.stride(0)
    d.ldc = ldc if ldc is not None else C.stride(-2)
    d.ldr = (ldr if ldr is not None else (R.stride(-2) if R is not None else 0))
    d.A, d.B, d.C, d.R = ptr(A), ptr(B), ptr(C), ptr(R)
    d.bias, d.cscale, d.cshift, d.rscale = ptr(bias), ptr(cscale), ptr(cshift), ptr(rscale)
    if ln is not None:                                   # (mean[M], rstd[M], csum[N]) of a folded LayerNorm
        d.ln_mean, d.ln_rstd, d.ln_csum = ptr(ln[0]), ptr(ln[1]), ptr(ln[2])
        d.ln_group = ln_group
    d.rperiod, d.act, d.res_mode, d.out_mode = rperiod, act, res, out_mode
    d.t_rows, d.t_tokens, d.algo = t_rows, t_tokens, algo
    d.reserved = dbg
    d.workspace, d.workspace_bytes = None, 0             # unused since ABI 5 (no kernel needs scratch)
    if prof is not None:                                 # tuning builds: per-workgroup cycle counters (reserved & 32 with algo 15)
        d.workspace, d.workspace_bytes = ptr(prof), prof.numel() * prof.element_size()
    out = None
    if part is not None and epilogue_stats():
        n = ctypes.c_int(0)
        if N.lib().mlpk_gemm_row_parts(ctypes.byref(d), ctypes.byref(n)) == 0:
            buf = part[0].get("%s.%d" % (part[1], n.value), (n.value, M, 2), torch.float32)     # (one buffer per plane count: no re-allocation)
            d.row_part, d.row_part_ld = ptr(buf), M
            out = (buf, n.value)
    if _defer:                                           # gemm_pair: the filled descriptor (it keeps the tensors alive through the caller's references)
        return d, out
    N.check(N.lib().mlpk_gemm_nt(ctypes.byref(d), stream()), "mlpk_gemm_nt")
    if timed:
        ev1.record()
        TIMER.events.setdefault(tag, []).append((ev0, ev1, 2.0 * M * Nn * K))
        if len(TIMER.kernels.setdefault(tag, set())) < 4:        # what actually ran, as the dispatch names it (a host-side query)
            nm = ctypes.create_string_buffer(96)
            if N.lib().mlpk_gemm_kernel_name(ctypes.byref(d), nm, 96) == 0:
                TIMER.kernels[tag].add(nm.value.decode())
    return out


def gemm_pair(first, second):
    """Two independent products in ONE launch where the dispatch gives both the same 16-bit "s3" tile (mlpk_gemm_nt_pair; otherwise one after the
    other): `first` / `second` = (args, kwargs) of engine.gemm.  Returns the two calls' `part` results.  Same bits as two calls."""
    (a0, k0), (a1, k1) = first, second
    if TIMER is not None:                                # per-call timing wants separate launches
        return gemm(*a0, **k0), gemm(*a1, **k1)
    d0, o0 = gemm(*a0, _defer=True, **k0)
    d1, o1 = gemm(*a1, _defer=True, **k1)
    N.check(N.lib().mlpk_gemm_nt_pair(ctypes.byref(d0), ctypes.byref(d1), stream()), "mlpk_gemm_nt_pair")
    return o0, o1


def conv_gemm_nhwc_supported(dtype, Cin, kh, kw, stride, pad):
    """mlpk_conv_gemm_nhwc: a strided convolution on channel-last rows as one product reading its operand through the window (MLPK_CONV_GEMM=0: window
    gather + GEMM, A/B aid)"""
    return (dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_CONV_GEMM", "1") != "0"
            and bool(N.lib().mlpk_conv_gemm_nhwc_supported(dtype_code(dtype), Cin, kh, kw, stride, pad)))


def conv_gemm_nhwc(x, w, out, B, H, W, Cin, kh, kw, stride, pad, **kw_gemm):
    """out (B Ho Wo, N) = epilogue(window(x) . w^T): x = dense channel-last (B, H, W, Cin) rows, w (N, >= kh kw Cin) in mlpk_im2col's NHWC column order;
    the keyword arguments of engine.gemm (bias, R / res, part, tag).  Returns what engine.gemm returns."""
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    K = kh * kw * Cin
    d, o = gemm(x, w, out, B * Ho * Wo, w.shape[0], K, lda=K, _defer=True, **kw_gemm)
    N.check(N.lib().mlpk_conv_gemm_nhwc(ctypes.byref(d), B, H, W, Cin, kh, kw, stride, pad, stream()), "mlpk_conv_gemm_nhwc")
    return o


def merge2x2_row_stats(x, B, H, W, C, mean, rstd, eps=1e-5):
    """LayerNorm statistics of the 2 x 2 merged rows (4 C values per window) of a channel-last tensor, without the merged tensor"""
    N.check(N.lib().mlpk_merge2x2_row_stats(dtype_code(x.dtype), ptr(x), B, H, W, C, eps, ptr(mean), ptr(rstd), stream()), "mlpk_merge2x2_row_stats")


def merge2x2_stats_combine(mean, rstd, B, H, W, out_mean, out_rstd, eps_in=1e-5, eps_out=1e-5):
    """the merged rows' LayerNorm statistics from the per-pixel statistics a producer delivered (no pass over the activations)"""
    N.check(N.lib().mlpk_merge2x2_stats_combine(ptr(mean), ptr(rstd), B, H, W, eps_in, eps_out, ptr(out_mean), ptr(out_rstd), stream()), "mlpk_merge2x2_stats_combine")


def merge_taps(w, C):
    """PatchMerging's weight columns (x0 | x1 | x2 | x3 = window positions (0,0), (1,0), (0,1), (1,1): swin_mlp.py:203-207) in mlpk_conv_gemm_nhwc's tap order
    (row-major: (0,0), (0,1), (1,0), (1,1)): the middle two blocks of C columns swapped"""
    n = w.shape[0]
    return w[:, :4 * C].reshape(n, 4, C)[:, [0, 2, 1, 3]].reshape(n, 4 * C).contiguous()


GEMM_LOG = None
CHANNEL_CHUNKS = int(os.environ.get("MLPK_CHANNEL_CHUNKS", "0"))      # 0 = by size (below); tuning override


def channel_chunks(rows, hidden_row_bytes):
    """Number of row chunks of a channel MLP: the smallest power of two whose hidden slice fits ~160 MB (the 256 MiB
    Infinity Cache minus what else streams through it), with whole 64-row groups per chunk."""
    if CHANNEL_CHUNKS > 0:
        n = CHANNEL_CHUNKS
    else:
        n = 1
        while n < 8 and rows * hidden_row_bytes / n > CHUNK_BYTES:
            n *= 2
    while n > 1 and (rows % n or (rows // n) % 64):
        n //= 2
    return max(1, n)


CHUNK_BYTES = float(os.environ.get("MLPK_CHUNK_MB", "1e9")) * 1e6      # default off (one chunk) until measured


def token_mlp(xt, ldxt, M, S, w1, b1, w2, b2, nchunks, x, ldx, t_rows, stats=None, layout=0):
    N.check(N.lib().mlpk_token_mlp(dtype_code(xt.dtype), ptr(xt), ldxt, M, S, ptr(w1), w1.stride(0), ptr(b1), ptr(w2),
                                   w2.stride(0), ptr(b2), nchunks, ptr(x), ldx, t_rows, ptr(stats), layout, stream()), "mlpk_token_mlp")


def token_mlp_ln(x, ldx, M, S, mean, rstd, gamma, beta, w1, b1, w2, b2, nchunks, t_rows, stats=None, layout=2):
    """LayerNorm + token-mixing MLP + residual in one kernel (weights packed for layout 2 / 3); x (B*S, ldx) is updated in place."""
    N.check(N.lib().mlpk_token_mlp_ln(dtype_code(x.dtype), ptr(x), ldx, M, S, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(w1), w1.stride(0), ptr(b1),
                                      ptr(w2), w2.stride(0), ptr(b2), nchunks, t_rows, ptr(stats), layout, stream()), "mlpk_token_mlp_ln")


def token_ln_fused():
    """MLPK_TOKEN_LN_FUSED=0: LayerNorm + transpose as its own pass in front of the token kernel (A/B aid)."""
    # (MLPK_T4_SHAPE=0 forces the generic token kernel, which has no LayerNorm loader: the unfused path then, not an error)
    return os.environ.get("MLPK_TOKEN_LN_FUSED", "1") != "0" and os.environ.get("MLPK_T4_SHAPE", "") != "0"


def layernorm_transpose_supported(dtype, C, ldx, ld_tt):
    return dtype in (torch.float16, torch.bfloat16) and C % 128 == 0 and C <= 2048 and ldx % 8 == 0 and ld_tt % 8 == 0 \
        and os.environ.get("MLPK_NO_FUSED_TOKEN_LN", "0") != "1"


def layernorm_transpose(x, nimg, S, C, gamma, beta, out_tt, ld_tt, eps=1e-5):
    """x: (nimg*S, >= C) with row stride x.stride(0) (a column slice of a wider tensor is fine)."""
    N.check(N.lib().mlpk_layernorm_transpose(dtype_code(x.dtype), ptr(x), nimg, S, C, x.stride(0), ptr(gamma), ptr(beta), eps,
                                             ptr(out_tt), ld_tt, stream()), "mlpk_layernorm_transpose")


def epilogue_stats():
    """MLPK_NO_EPILOGUE_STATS=1: separate statistics pass after the token kernel (A/B aid)."""
    return os.environ.get("MLPK_NO_EPILOGUE_STATS", "0") != "1"


def stats_finalize_planar(part, rows, count, mean, rstd, eps=1e-5, group=1):
    """part = the (nplanes, M, 2) buffer engine.gemm(part=...) returned; statistic r covers GEMM rows [r*group, (r+1)*group)."""
    N.check(N.lib().mlpk_stats_finalize_planar(ptr(part), rows, part.shape[0], part.shape[1], group, count, eps, ptr(mean), ptr(rstd), stream()),
            "mlpk_stats_finalize_planar")


def token_mlp_supported(dtype, S, sp, hidden=0):
    return dtype in (torch.float16, torch.bfloat16) and S <= 208 and sp <= 224 and sp % 32 == 0 and hidden <= 1024


def add_periodic(x, ldx, table, rows, C, period):
    """x[m, :] += table[m % period, :] (table fp32): SwinMLP's absolute position embedding (swin_mlp.py:437-438)"""
    N.check(N.lib().mlpk_add_periodic(dtype_code(x.dtype), ptr(x), ldx, ptr(table), rows, C, period, stream()), "mlpk_add_periodic")


def rows_to_nchw(cur, B, HW, C, out):
    """channel-last rows (B*HW, C) -> (B, C, HW): mlpk_transpose_batched (the layout a reference module returns its maps in)"""
    N.check(N.lib().mlpk_transpose_batched(dtype_code(cur.dtype), ptr(cur), C, ptr(out), HW, None, 0, B, HW, C, stream()), "mlpk_transpose_batched")
    return out


def smlp_mix_supported(dtype, H, W, C):
    """mlpk_smlp_mix (round 5): Sparse-MLP's BatchNorm + both axial mixes + the concatenation in one kernel (maps up to 32 x 32, 16 bit);
    MLPK_SMLP_MIX=0 keeps the four-launch path (A/B aid)"""
    if os.environ.get("MLPK_SMLP_MIX") == "0" or dtype not in (torch.float16, torch.bfloat16):
        return False
    return bool(N.lib().mlpk_smlp_mix_supported(dtype_code(dtype), H, W, C))


def smlp_mix_dw_supported(dtype, H, W, C):
    """mlpk_smlp_mix_dw: the same with the block's depthwise 3 x 3 sublayer in front, for maps whose raw tile fits beside the transposed
    copies (14 x 14, 7 x 7); MLPK_SMLP_MIX_DW=0: two kernels (A/B aid)"""
    if os.environ.get("MLPK_SMLP_MIX_DW") == "0" or not smlp_mix_supported(dtype, H, W, C):
        return False
    return bool(N.lib().mlpk_smlp_mix_dw_supported(dtype_code(dtype), H, W, C))


def smlp_mix_dw(x, ldx, B, H, W, C, dw_w, dw_b, dw_s, dw_h, xres, ldxr, bn_s, bn_h, wh, bh, ww, bw, out, ldo):
    """xres = x + dwconv3x3(dw_s * x + dw_h) + dw_b (sparse_mlp.py:88-91), then smlp_mix of xres"""
    N.check(N.lib().mlpk_smlp_mix_dw(dtype_code(x.dtype), ptr(x), ldx, B, H, W, C, ptr(dw_w), ptr(dw_b), ptr(dw_s), ptr(dw_h), ptr(xres), ldxr,
                                     ptr(bn_s), ptr(bn_h), ptr(wh), ptr(bh), ptr(ww), ptr(bw), ptr(out), ldo, stream()), "mlpk_smlp_mix_dw")


def pack_smlp_mix(w, b, dtype, device):
    """(S, S) axial mixing weight of nn.Linear(S, S) (sparse_mlp.py:64-65) -> (32, 32) zero-padded in the storage type, bias -> (32,) fp32"""
    S = w.shape[0]
    wp = torch.zeros((32, 32), dtype=dtype, device=device)
    wp[:S, :S] = w.detach().to(device=device, dtype=dtype)
    bp = torch.zeros((32,), dtype=torch.float32, device=device)
    bp[:S] = b.detach().to(device=device, dtype=torch.float32)
    return wp, bp


def smlp_mix(x, ldx, B, H, W, C, bn_s, bn_h, wh, bh, ww, bw, out, ldo):
    """out (B*H*W, 3C) = [proj_h(x^) | proj_w(x^) | x^], x^ = bn_s * x + bn_h (sparse_mlp.py:66-71 behind the block's BatchNorm)"""
    N.check(N.lib().mlpk_smlp_mix(dtype_code(x.dtype), ptr(x), ldx, B, H, W, C, ptr(bn_s), ptr(bn_h), ptr(wh), ptr(bh), ptr(ww), ptr(bw), ptr(out), ldo,
                                  stream()), "mlpk_smlp_mix")


def token_gemm_supported(dtype, S, sp):
    return dtype in (torch.float16, torch.bfloat16) and S <= 224 and sp <= 224 and sp % 32 == 0 and os.environ.get("MLPK_NO_TOKEN_GEMM", "0") != "1"


def pack_token_gemm(w, b, dtype, device):
    """(S_out, S_in) token-mixing weight -> (groups*32, 256) for mlpk_token_gemm, bias padded to groups*32."""
    w = w.detach().reshape(w.shape[0], -1)
    so, si = w.shape
    ng = (so + 31) // 32
    wp = torch.zeros((ng * 32, 256), dtype=dtype, device=device)
    wp[:so, :si] = w.to(device=device, dtype=dtype)
    bp = torch.zeros((ng * 32,), dtype=torch.float32, device=device)
    if b is not None:
        bp[:so] = b.detach().to(device=device, dtype=torch.float32).reshape(-1)
    return wp, bp, ng


def token_gemm(xt, ldxt, M, S, wp, bp, ng, out, ldo, t_rows, *, R=None, ldr=0, res=N.RES_NONE, rscale=None, rperiod=0):
    N.check(N.lib().mlpk_token_gemm(dtype_code(xt.dtype), ptr(xt), ldxt, M, S, ptr(wp), wp.stride(0), ptr(bp), ng, ptr(rscale), rperiod,
                                    ptr(R), ldr, res, ptr(out), ldo, t_rows, stream()), "mlpk_token_gemm")


def token_gemm_ln_supported(dtype, S, t_rows, ldx):
    """mlpk_token_gemm_ln: the LayerNorm / Aff in front of a token-mixing product as the kernel's operand loader
    (MLPK_TOKEN_GEMM_LN=0: the separate normalise-and-transpose pass, A/B aid)"""
    return (token_gemm_supported(dtype, S, round_up(S, 32)) and t_rows % 32 == 0 and ldx % 8 == 0
            and os.environ.get("MLPK_TOKEN_GEMM_LN", "1") != "0")


def token_gemm_ln_post_supported(dtype, S, t_rows, ldx):
    """mlpk_token_gemm_ln_post: the per-channel affine that FOLLOWS the sublayer applied where its result is stored -- the pipelined kernel only
    (>= 3 groups of 32 tokens, an even token count, <= 1024 channels per image); MLPK_TOKEN_GEMM_POST=0: a separate mlpk_norm_apply (A/B aid)"""
    return (token_gemm_ln_supported(dtype, S, t_rows, ldx) and S > 64 and S % 2 == 0 and t_rows <= 1024
            and os.environ.get("MLPK_TOKEN_GEMM_POST", "1") != "0" and os.environ.get("MLPK_TOKEN_GEMM_PIPE", "1") != "0")


def token_gemm_ln(x, ldx, M, S, mean, rstd, gamma, beta, wp, bp, ng, out, ldo, t_rows, *, R=None, ldr=0, res=N.RES_NONE, rscale=None, rperiod=0,
                  post=None):
    """x: token-major (B*S, >= t_rows) view (a column slice is fine: pass its stride as ldx); mean / rstd per token row or None.
    post = (scale, shift) per channel: out = scale * round(result) + shift (token_gemm_ln_post_supported)."""
    if post is not None:
        N.check(N.lib().mlpk_token_gemm_ln_post(dtype_code(x.dtype), ptr(x), ldx, M, S, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(wp), wp.stride(0),
                                                ptr(bp), ng, ptr(rscale), rperiod, ptr(R), ldr, res, ptr(post[0]), ptr(post[1]), ptr(out), ldo, t_rows,
                                                stream()), "mlpk_token_gemm_ln_post")
        return
    N.check(N.lib().mlpk_token_gemm_ln(dtype_code(x.dtype), ptr(x), ldx, M, S, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(wp), wp.stride(0),
                                       ptr(bp), ng, ptr(rscale), rperiod, ptr(R), ldr, res, ptr(out), ldo, t_rows, stream()), "mlpk_token_gemm_ln")


def token_mlp_stat_planes(C, layout):
    """planes of mlpk_token_mlp's `stats` buffer: one per 128 channels, one per 64 for the generated kernel (layout 2)"""
    return C // 64 if layout in (2, 3) else C // 128


def pack_token_mlp(w1, b1, w2, b2, dtype, device, sp, layout=None, t_rows=None):
    """Weights of the fused token-mixing kernel: W1 rows / b1 / W2 columns zero-padded to whole hidden groups
    of mlpk_token_mlp_chunk() = 32, W1's K axis zero-padded to 256 (8 LDS planes per group).  t_rows (channels per image) lets
    the library pick the generated one-wave-per-SIMD kernel (layout 2) when the shape allows."""
    ch = N.lib().mlpk_token_mlp_chunk()
    w1 = w1.detach().reshape(w1.shape[0], -1)
    w2 = w2.detach().reshape(w2.shape[0], -1)
    T, S = w1.shape
    nch = (T + ch - 1) // ch
    w1p = torch.zeros((nch * ch, 256), dtype=dtype, device=device)
    w1p[:T, :S] = w1.to(device=device, dtype=dtype)
    b1p = torch.zeros((nch * ch,), dtype=torch.float32, device=device)
    b1p[:T] = b1.detach().to(device=device, dtype=torch.float32)
    w2p = torch.zeros((S, nch * ch), dtype=dtype, device=device)
    w2p[:, :T] = w2.to(device=device, dtype=dtype)
    if layout is None:
        layout = N.lib().mlpk_token_mlp_layout_for(dtype_code(dtype), S, nch, t_rows) if t_rows else N.lib().mlpk_token_mlp_layout(S, nch)
    if layout == 3:
        # layout 3 stores W2 as f16 (the hidden stays f16 on chip): weights outside f16's range would overflow / flush where the
        # reference's bf16 keeps them -- such a model takes layout 2 (bf16 W2, fp32 GELU), same kernel family (advisor, round 5)
        amax = float(w2.abs().max()) if w2.numel() else 0.0
        if not amax <= 6.0e4:
            layout = 2
    if layout in (2, 3):
        # include/mlpk.h: W2 group-major -- (nch + 1) groups x 224 token rows x 32 k slots, slot 16 kk + 8 h + e of a group <- hidden
        # 16 kk + 8 (e >> 2) + 4 h + (e & 3); group nch and the token rows behind S are zeros; b1 / b2 as padded tables
        slot = torch.arange(32)
        kk, hh, e = slot // 16, (slot // 8) % 2, slot % 8
        src = (16 * kk + 8 * (e // 4) + 4 * hh + (e % 4)).to(device)
        # layout 3 (bf16 storage): the kernel keeps the hidden in f16 and multiplies it by f16 weights -- W2 rounded from fp32 to f16 HERE
        w2dt = torch.float16 if layout == 3 else dtype
        if layout == 3:
            assert dtype == torch.bfloat16
            w2p = torch.zeros((S, nch * ch), dtype=w2dt, device=device)
            w2p[:, :T] = w2.to(device=device, dtype=w2dt)
        w2g = torch.zeros((nch + 1, 224, 32), dtype=w2dt, device=device)
        w2g[:nch, :S] = w2p.view(S, nch, 32)[:, :, src].permute(1, 0, 2)
        b1t = torch.zeros((1024,), dtype=torch.float32, device=device)
        b1t[64:64 + nch * ch] = b1p
        b2t = torch.zeros((224,), dtype=torch.float32, device=device)
        b2t[:S] = f32(b2, device)
        return w1p, b1t, w2g.view((nch + 1) * 224, 32), b2t, nch, layout
    if layout == 1:
        # column slot 8 f + e of every 32-column group <- column (e < 4 ? 4 f + e : 16 + 4 f + e - 4)   (include/mlpk.h)
        slot = torch.arange(32)
        f, e = slot // 8, slot % 8
        src = torch.where(e < 4, 4 * f + e, 16 + 4 * f + e - 4).to(device)
        w2p = w2p.view(S, nch, 32)[:, :, src].reshape(S, nch * ch).contiguous()
    return w1p, b1p, w2p, f32(b2, device), nch, layout


def linear_gelu_enabled():
    """the opt-in switch of mlpk_linear_gelu; the models build its second copy of every fc1 weight only when it is on (advisor, round 4)"""
    return os.environ.get("MLPK_LINEAR_GELU", "0") == "1"


def linear_gelu_supported(dtype, M, K, Nn):
    """mlpk_linear_gelu: a short-K Linear + GELU with its rows resident in registers.  OPT-IN (MLPK_LINEAR_GELU=1): measured slower than the
    GEMM tiles with a GELU epilogue on every model shape (profiles/r04_linear_gelu_ab.txt: gMLP-S 11.33 vs 10.22 ms, ViP-S7 30.39 vs 30.17)."""
    return (dtype in (torch.float16, torch.bfloat16) and linear_gelu_enabled()
            and bool(N.lib().mlpk_linear_gelu_supported(dtype_code(dtype), M, K, round_up(Nn, 32))))


def pack_linear_gelu(w, b, dtype, device, gamma=None, beta=None):
    """Weights of mlpk_linear_gelu: W (N, K) [x diag(gamma)] with the rows of every group of 32 in the kernel's order (row 16 j + 4 f + r <-
    output column 8 f + 4 j + r), K zero-padded to 256 / 512, N to whole groups; b [+ W beta] and the row sums of the rounded folded W in the
    same order (csum None without a norm).  Returns (wp, bp, csum, nch)."""
    w = w.detach().to(device=device, dtype=torch.float32).reshape(w.shape[0], -1)
    Nn, K = w.shape
    nch = (Nn + 31) // 32
    kp = 256 if K <= 256 else 512
    bp = torch.zeros((nch * 32,), dtype=torch.float32, device=device)
    if b is not None:
        bp[:Nn] = b.detach().to(device=device, dtype=torch.float32).reshape(-1)
    wf = w
    if gamma is not None:
        wf = w * gamma.detach().to(device=device, dtype=torch.float32).reshape(1, -1)
        bp[:Nn] += w @ beta.detach().to(device=device, dtype=torch.float32).reshape(-1)
    wp = torch.zeros((nch * 32, kp), dtype=dtype, device=device)
    wp[:Nn, :K] = wf.to(dtype)
    csum = wp.to(torch.float32).sum(dim=1) if gamma is not None else None
    slot = torch.arange(32)
    j, f, r = slot // 16, (slot // 4) % 4, slot % 4
    src = (8 * f + 4 * j + r).to(device)                                         # output column at row 16 j + 4 f + r
    wp = wp.view(nch, 32, kp)[:, src].reshape(nch * 32, kp).contiguous()
    bp = bp.view(nch, 32)[:, src].reshape(-1).contiguous()
    if csum is not None:
        csum = csum.view(nch, 32)[:, src].reshape(-1).contiguous()
    return wp, bp, csum, nch


def linear_gelu(x, rows, K, pack, out, *, ln=None, ln_group=1, part=None):
    """out = gelu(norm(x) W^T + b); pack = pack_linear_gelu(...); ln = (mean, rstd) or None; part = (workspace, name): also deliver the row
    statistics planes of `out` (32 columns each) -- returns (buffer (N / 32, rows, 2), N / 32) as engine.gemm(part=...) does."""
    wp, bp, csum, nch = pack
    buf = None
    if part is not None and epilogue_stats():
        buf = part[0].get("%s.%d" % (part[1], nch), (nch, rows, 2), torch.float32)
    N.check(N.lib().mlpk_linear_gelu(dtype_code(x.dtype), ptr(x), x.stride(0), rows, K, ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None, ln_group,
                                     ptr(csum) if ln else None, ptr(wp), wp.stride(0), ptr(bp), nch, ptr(out), out.stride(0), ptr(buf), stream()),
            "mlpk_linear_gelu")
    return (buf, nch) if buf is not None else None


def swin_spatial_supported(dtype, C, heads, ws):
    """mlpk_swin_spatial: the spatial-MLP half of a Swin-MLP block in one kernel (MLPK_SWIN_SPATIAL_FUSED=0: the five passes, A/B aid)"""
    return (dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_SWIN_SPATIAL_FUSED", "1") != "0"
            and bool(N.lib().mlpk_swin_spatial_supported(dtype_code(dtype), C, heads, ws)))


def pack_swin_spatial(weight, bias, heads, ws, dtype, device):
    """grouped Conv1d weight (heads * ws^2, ws^2, 1) / bias (heads * ws^2) -> (heads, 64, 64) [t_out][t_in] and (heads, 64), zero-padded"""
    t = ws * ws
    w = weight.detach().to(device=device, dtype=torch.float32).reshape(heads, t, t)
    wp = torch.zeros((heads, 64, 64), dtype=dtype, device=device)
    wp[:, :t, :t] = w.to(dtype)
    bp = torch.zeros((heads, 64), dtype=torch.float32, device=device)
    if bias is not None:
        bp[:, :t] = bias.detach().to(device=device, dtype=torch.float32).reshape(heads, t)
    return wp.contiguous(), bp.contiguous()


def swin_spatial(x, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, mean, rstd, gamma, beta, wp, bp, out_stats=None, eps=1e-5):
    """out_stats = (mean, rstd) fp32 per row: the kernel also delivers the LayerNorm statistics of the rows it writes"""
    if out_stats is not None:
        N.check(N.lib().mlpk_swin_spatial_stats(dtype_code(x.dtype), ptr(x), B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, ptr(mean), ptr(rstd), ptr(gamma),
                                                ptr(beta), ptr(wp), ptr(bp), ptr(out_stats[0]), ptr(out_stats[1]), eps, stream()), "mlpk_swin_spatial_stats")
        return
    N.check(N.lib().mlpk_swin_spatial(dtype_code(x.dtype), ptr(x), B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, ptr(mean), ptr(rstd), ptr(gamma),
                                      ptr(beta), ptr(wp), ptr(bp), stream()), "mlpk_swin_spatial")


def channel_mlp_fused_supported(dtype, C, hidden):
    """mlpk_channel_mlp: the whole channel MLP of a narrow stage in one kernel (MLPK_CHANNEL_MLP_FUSED=0: the two GEMMs, A/B aid)"""
    return (dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_CHANNEL_MLP_FUSED", "1") != "0"
            and bool(N.lib().mlpk_channel_mlp_supported(dtype_code(dtype), C, round_up(hidden, 32))))


def pack_channel_mlp_fused(w1, b1, w2, b2, dtype, device, gamma=None, beta=None, cscale=None):
    """Weights of mlpk_channel_mlp (include/mlpk.h): W1 (hidden, C) [x diag(gamma)] zero-padded to (nch*32, 256), b1 [+ W1 beta], the
    row sums of the rounded folded W1 (None without a norm), W2 (C, hidden) with its columns in mlpk_token_mlp's layout-1 order
    and its rows in the kernel's store order, b2.  cscale (C): a per-output-channel scale of the second Linear (a layer scale),
    folded into W2's rows and b2.  Returns (w1p, b1p, csum, w2p, b2, nch)."""
    w1 = w1.detach().to(device=device, dtype=torch.float32).reshape(w1.shape[0], -1)
    w2 = w2.detach().to(device=device, dtype=torch.float32).reshape(w2.shape[0], -1)
    T, C = w1.shape
    nch = (T + 31) // 32
    b1p = torch.zeros((nch * 32,), dtype=torch.float32, device=device)
    if b1 is not None:
        b1p[:T] = b1.detach().to(device=device, dtype=torch.float32).reshape(-1)
    wf = w1
    if gamma is not None:
        wf = w1 * gamma.detach().to(device=device, dtype=torch.float32).reshape(1, -1)
        b1p[:T] += w1 @ beta.detach().to(device=device, dtype=torch.float32).reshape(-1)
    w1p = torch.zeros((nch * 32, 256), dtype=dtype, device=device)
    w1p[:T, :C] = wf.to(dtype)
    csum = w1p.to(torch.float32).sum(dim=1).contiguous() if gamma is not None else None
    b2v = b2.detach().to(device=device, dtype=torch.float32).reshape(-1) if b2 is not None else torch.zeros((C,), dtype=torch.float32, device=device)
    if cscale is not None:
        cs = cscale.detach().to(device=device, dtype=torch.float32).reshape(-1)
        w2 = w2 * cs.view(-1, 1)
        b2v = b2v * cs
    w2p = torch.zeros((C, nch * 32), dtype=dtype, device=device)
    w2p[:, :T] = w2.to(dtype)
    slot = torch.arange(32)
    f, e = slot // 8, slot % 8
    col = torch.where(e < 4, 4 * f + e, 16 + 4 * f + e - 4).to(device)          # hidden unit at k slot 8 f + e
    h, f4, r = slot // 16, (slot // 4) % 4, slot % 4
    row = (8 * f4 + 4 * h + r).to(device)                                        # output channel at row 16 h + 4 f + r
    w2p = w2p.view(C // 32, 32, nch, 32)[:, row][:, :, :, col].reshape(C, nch * 32).contiguous()
    return w1p, b1p, csum, w2p, b2v.contiguous(), nch


def channel_mlp_fused(x, rows, C, pack, out, *, R=None, ln=None, ln_group=1, part=None):
    """out = R + fc2(gelu(fc1(norm(x)))) on channel-last rows; pack = pack_channel_mlp_fused(...); ln = (mean, rstd) or None.
    part = (workspace, name): also deliver the row statistics of `out` -- returns (buffer (1, rows, 2), 1) as engine.gemm(part=...)
    does, for finalize_stats / stats_finalize_planar; None otherwise."""
    w1p, b1p, csum, w2p, b2p, nch = pack
    buf = None
    if part is not None and epilogue_stats():
        buf = part[0].get("%s.1" % part[1], (1, rows, 2), torch.float32)
    N.check(N.lib().mlpk_channel_mlp(dtype_code(x.dtype), ptr(x), x.stride(0), rows, C, ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None,
                                     ln_group, ptr(csum) if ln else None, ptr(w1p), w1p.stride(0), ptr(b1p), ptr(w2p), w2p.stride(0), ptr(b2p), nch,
                                     ptr(R), R.stride(0) if R is not None else 0, ptr(out), out.stride(0), ptr(buf), stream()), "mlpk_channel_mlp")
    return (buf, 1) if buf is not None else None


def patchify(src, out, B, Cin, H, W, ph, pw, pad, ldo, layout=N.LAYOUT_NCHW, px_stride=0, order=0):
    N.check(N.lib().mlpk_patchify(dtype_code(src.dtype), dtype_code(out.dtype), layout, ptr(src), ptr(out), B, Cin, H, W,
                                  ph, pw, pad, px_stride, ldo, order, stream()), "mlpk_patchify")


def patch_embed4_supported(src_dtype, dst_dtype, cin, H, W, C):
    """mlpk_patch_embed4: Conv2d(3 -> C, k = stride = 4) (+ LayerNorm) in one kernel (MLPK_PATCH_EMBED4=0: gather + GEMM + passes, A/B aid)"""
    return (dst_dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_PATCH_EMBED4", "1") != "0"
            and bool(N.lib().mlpk_patch_embed4_supported(dtype_code(src_dtype), dtype_code(dst_dtype), cin, H, W, C)))


def patch_embed4(x, w, bias, out, B, H, W, C, gamma=None, beta=None, eps=1e-5):
    N.check(N.lib().mlpk_patch_embed4(dtype_code(x.dtype), dtype_code(out.dtype), ptr(x), B, 3, H, W, ptr(w), w.stride(0), ptr(bias), ptr(gamma), ptr(beta), eps,
                                      ptr(out), out.stride(0), C, stream()), "mlpk_patch_embed4")


def stem7_supported(src_dtype, dst_dtype, cin, H, W, pad, C):
    """mlpk_stem7: Conv2d(3 -> C, k = 7, stride = 4) as a direct convolution (MLPK_STEM7=0: window gather + GEMM, A/B aid)"""
    return (dst_dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_STEM7", "1") != "0"
            and bool(N.lib().mlpk_stem7_supported(dtype_code(src_dtype), dtype_code(dst_dtype), cin, H, W, pad, C)))


def pack_stem7(weight, dtype, device):
    """Conv2d weight (C, 3, 7, 7) -> (C, 176): k = (ci * 7 + i) * 8 + j, zero for j = 7 and k >= 168 (mlpk_stem7's operand order)"""
    w = weight.detach().to(device=device, dtype=torch.float32)
    C = w.shape[0]
    wp = torch.zeros((C, 176), dtype=dtype, device=device)
    wp[:, :168] = torch.nn.functional.pad(w, (0, 1)).reshape(C, 168).to(dtype)
    return wp.contiguous()


def stem7(x, w7, bias, out, B, H, W, pad, C, out_stats=None, eps=1e-5):
    """out_stats = (mean, rstd) fp32 per output pixel: the LayerNorm statistics of the rows written, for the block that follows"""
    N.check(N.lib().mlpk_stem7(dtype_code(x.dtype), dtype_code(out.dtype), ptr(x), B, 3, H, W, pad, ptr(w7), ptr(bias), ptr(out), out.stride(0), C,
                               ptr(out_stats[0]) if out_stats else None, ptr(out_stats[1]) if out_stats else None, eps, stream()), "mlpk_stem7")


def row_stats(x, rows, length, ldx, mean, rstd, eps=1e-5):
    N.check(N.lib().mlpk_row_stats(dtype_code(x.dtype), ptr(x), rows, length, ldx, eps, ptr(mean), ptr(rstd), stream()),
            "mlpk_row_stats")


def norm_apply(x, rows, C, ldx, *, mean=None, rstd=None, gamma=None, beta=None, act=N.ACT_NONE, stat_group=1,
               out_rm=None, ld_rm=0, out_tt=None, S=0, ld_tt=0, out_ph=None, out_pw=None, H=0, W=0, seg=0, ld_p=0,
               sum_ph=None, sum_pw=None, ld_sum=0):
    d = N.NormDesc()
    d.dtype, d.act, d.rows, d.C, d.ldx, d.stat_group = dtype_code(x.dtype), act, rows, C, ldx, stat_group
    d.S, d.H, d.W, d.seg = S, H, W, seg
    d.ld_rm, d.ld_tt, d.ld_p = ld_rm, ld_tt, ld_p
    d.x, d.mean, d.rstd, d.gamma, d.beta = ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta)
    d.out_rm, d.out_tt, d.out_ph, d.out_pw = ptr(out_rm), ptr(out_tt), ptr(out_ph), ptr(out_pw)
    d.sum_ph, d.sum_pw, d.ld_sum = ptr(sum_ph), ptr(sum_pw), ld_sum
    N.check(N.lib().mlpk_norm_apply(ctypes.byref(d), stream()), "mlpk_norm_apply")


def vip_unpermute(which, z, out, B, H, W, C, seg, ldz):
    N.check(N.lib().mlpk_vip_unpermute(dtype_code(z.dtype), which, ptr(z), ptr(out), B, H, W, C, seg, ldz, stream()),
            "mlpk_vip_unpermute")


def vip_branch_supported(dtype, H, W, C, seg):
    """mlpk_vip_branch for BOTH branches of a ViP block (MLPK_VIP_BRANCH=0: the two-kernel path, A/B aid)"""
    return (dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_VIP_BRANCH", "1") != "0"
            and bool(N.lib().mlpk_vip_branch_supported(dtype_code(dtype), H, W, C, seg, 0))
            and bool(N.lib().mlpk_vip_branch_supported(dtype_code(dtype), H, W, C, seg, 1)))


def vip_branch(which, x, ldx, B, H, W, C, seg, mean, rstd, gamma, beta, w, bias, out, ldz, sums=None, ld_sum=0):
    N.check(N.lib().mlpk_vip_branch(dtype_code(x.dtype), ptr(x), ldx, B, H, W, C, seg, which, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(w), w.stride(0),
                                    ptr(bias), ptr(out), ldz, ptr(sums), ld_sum, stream()), "mlpk_vip_branch")


def pool_mean(x, B, S, C, ldx, out, ldo, *, mean=None, rstd=None, stat_group=1, gamma=None, beta=None):
    N.check(N.lib().mlpk_pool_mean(dtype_code(x.dtype), ptr(x), B, S, C, ldx, ptr(mean), ptr(rstd), stat_group,
                                   ptr(gamma), ptr(beta), ptr(out), ldo, stream()), "mlpk_pool_mean")


def shift_nchw(x, out, kernel_size, dim):
    n, c, h, w = x.shape
    N.check(N.lib().mlpk_shift_nchw(dtype_code(x.dtype), ptr(x), ptr(out), n, c, h, w, kernel_size, dim, stream()),
            "mlpk_shift_nchw")


def shift_nchw_backward(grad_out, grad_in, kernel_size, dim):
    n, c, h, w = grad_out.shape
    N.check(N.lib().mlpk_shift_nchw_backward(dtype_code(grad_out.dtype), ptr(grad_out), ptr(grad_in), n, c, h, w, kernel_size, dim, stream()),
            "mlpk_shift_nchw_backward")


def shift_nhwc(x, out, n, h, w, c, kernel_size, dim):
    N.check(N.lib().mlpk_shift_nhwc(dtype_code(x.dtype), ptr(x), ptr(out), n, h, w, c, kernel_size, dim, stream()),
            "mlpk_shift_nhwc")


def norm_shift_nhwc(x, out_w, out_h, n, h, w, c, kernel_size, mean, rstd, gamma, beta, act):
    N.check(N.lib().mlpk_norm_shift_nhwc(dtype_code(x.dtype), ptr(x), ptr(out_w), ptr(out_h), n, h, w, c, kernel_size, ptr(mean), ptr(rstd),
                                         ptr(gamma), ptr(beta), act, stream()), "mlpk_norm_shift_nhwc")


def as_conv2_supported(dtype, H, W, C, kernel_size):
    """mlpk_as_conv2 takes the shape (MLPK_ASMLP_FUSED_CONV2=0: the three-kernel sequence, A/B aid)"""
    return (dtype in (torch.float16, torch.bfloat16) and os.environ.get("MLPK_ASMLP_FUSED_CONV2", "1") != "0"
            and bool(N.lib().mlpk_as_conv2_supported(dtype_code(dtype), H, W, C, kernel_size)))


def as_conv2(t, y, B, H, W, C, kernel_size, mean, rstd, gamma, beta, w1, b1, w2, b2, stats=None, eps=1e-5):
    """y = gelu(conv2_1(shift_W(u)) + b1) + gelu(conv2_2(shift_H(u)) + b2), u = gelu(GroupNorm affine of t): AxialShift's core in one kernel.
    stats = (workspace, name): the kernel also finishes the GroupNorm(1, C) statistics of y (mlpk_as_conv2_stats: one pair per step of
    rows, added in step order inside the kernel -- no statistics pass, no finalize launch); returns (mean, rstd) of y, else None."""
    if stats is None or os.environ.get("MLPK_ASCONV_STATS", "1") == "0":
        N.check(N.lib().mlpk_as_conv2(dtype_code(t.dtype), ptr(t), ptr(y), B, H, W, C, kernel_size, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                      ptr(w1), ptr(b1), ptr(w2), ptr(b2), w1.stride(0), stream()), "mlpk_as_conv2")
        return None
    ws, name = stats
    steps = N.lib().mlpk_as_conv2_steps(dtype_code(t.dtype), H, W, C, kernel_size)
    part = ws.get(name + ".part", (B, steps, 2), torch.float32)
    counter = ws.get(name + ".count", (B,), torch.int32, fill=0)
    mo, ro = ws.get(name + ".mean", (B,), torch.float32), ws.get(name + ".rstd", (B,), torch.float32)
    N.check(N.lib().mlpk_as_conv2_stats(dtype_code(t.dtype), ptr(t), ptr(y), B, H, W, C, kernel_size, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                        ptr(w1), ptr(b1), ptr(w2), ptr(b2), w1.stride(0), ptr(part), ptr(mo), ptr(ro), ptr(counter), eps, stream()),
            "mlpk_as_conv2_stats")
    return mo, ro


def cycle_shift_ln(x, mean, rstd, gamma, beta, out_h, out_w, B, H, W, C, k, ldi, ldo):
    """mlpk_cycle_shift on LayerNorm(x) without storing it: x un-normalised, (mean, rstd) per pixel, gamma / beta per channel"""
    N.check(N.lib().mlpk_cycle_shift_ln(dtype_code(x.dtype), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(out_h), ptr(out_w), B, H, W, C, k,
                                        ldi, ldo, stream()), "mlpk_cycle_shift_ln")


def cycle_shift(x, out_h, out_w, B, H, W, C, k, ldi, ldo):
    N.check(N.lib().mlpk_cycle_shift(dtype_code(x.dtype), ptr(x), ptr(out_h), ptr(out_w), B, H, W, C, k, ldi, ldo, stream()),
            "mlpk_cycle_shift")


def split_sum(x0, x1, x2, ld0, ld1, ld2, B, H, W, C, mode, a, scale=1.0):
    N.check(N.lib().mlpk_split_sum(dtype_code(x0.dtype), ptr(x0), ptr(x1), ptr(x2), ld0, ld1, ld2, B, H, W, C, mode,
                                   scale, ptr(a), stream()), "mlpk_split_sum")


def split_softmax(hat, bar, B, C):
    N.check(N.lib().mlpk_split_softmax(ptr(hat), ptr(bar), B, C, stream()), "mlpk_split_softmax")


def split_apply(x0, x1, x2, ld0, ld1, ld2, B, H, W, C, mode, bar, out, ldo):
    N.check(N.lib().mlpk_split_apply(dtype_code(x0.dtype), ptr(x0), ptr(x1), ptr(x2), ld0, ld1, ld2, B, H, W, C, mode,
                                     ptr(bar), ptr(out), ldo, stream()), "mlpk_split_apply")


def vip_split_apply(zh, zw, xc, ldh, ldw, ldc, B, H, W, C, seg, bar, out, ldo):
    N.check(N.lib().mlpk_vip_split_apply(dtype_code(zh.dtype), ptr(zh), ptr(zw), ptr(xc), ldh, ldw, ldc, B, H, W, C, seg, ptr(bar),
                                         ptr(out), ldo, stream()), "mlpk_vip_split_apply")


def s2_shift(x, out, B, H, W, C, ldi, ldo, mode):
    N.check(N.lib().mlpk_s2_shift(dtype_code(x.dtype), ptr(x), ptr(out), B, H, W, C, ldi, ldo, mode, stream()),
            "mlpk_s2_shift")


def dwconv_nhwc(x, out, B, H, W, C, k, w, bias, bn_scale, bn_shift):
    N.check(N.lib().mlpk_dwconv_nhwc(dtype_code(x.dtype), ptr(x), ptr(out), B, H, W, C, k, ptr(w), ptr(bias),
                                     ptr(bn_scale), ptr(bn_shift), stream()), "mlpk_dwconv_nhwc")


def dwconv_affine_nhwc(x, out, B, H, W, C, k, w, bias, pre_scale, pre_shift):
    N.check(N.lib().mlpk_dwconv_affine_nhwc(dtype_code(x.dtype), ptr(x), ptr(out), B, H, W, C, k, ptr(w), ptr(bias),
                                            ptr(pre_scale), ptr(pre_shift), stream()), "mlpk_dwconv_affine_nhwc")


def im2col(src, out, B, Cin, H, W, kh, kw, sh, sw, pad, ldo, layout=N.LAYOUT_NCHW, px_stride=0):
    N.check(N.lib().mlpk_im2col(dtype_code(src.dtype), dtype_code(out.dtype), layout, ptr(src), ptr(out), B, Cin, H, W,
                                kh, kw, sh, sw, pad, px_stride, ldo, stream()), "mlpk_im2col")


def hire_gather(xn, a_h, a_w, B, H, W, C, h, w, step, ld_h, ld_w):
    N.check(N.lib().mlpk_hire_gather(dtype_code(xn.dtype), ptr(xn), ptr(a_h), ptr(a_w), B, H, W, C, h, w, step, ld_h, ld_w,
                                     stream()), "mlpk_hire_gather")


def hire_gather_ln(x, mean, rstd, gamma, beta, a_h, a_w, B, H, W, C, h, w, step, ld_h, ld_w):
    N.check(N.lib().mlpk_hire_gather_ln(dtype_code(x.dtype), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(a_h), ptr(a_w), B, H, W, C, h, w,
                                        step, ld_h, ld_w, stream()), "mlpk_hire_gather_ln")


def hire_combine_from(x, src, y_h, y_w, B, H, W, C, h, w, step, ld_h, ld_w):
    N.check(N.lib().mlpk_hire_combine_from(dtype_code(x.dtype), ptr(x), ptr(src), ptr(y_h), ptr(y_w), B, H, W, C, h, w, step, ld_h, ld_w,
                                           stream()), "mlpk_hire_combine_from")


def hire_combine_stats(x, src, y_h, y_w, B, H, W, C, h, w, step, ld_h, ld_w, mean, rstd, eps=1e-5):
    """hire_combine_from that also delivers (mean, rstd) of the rows it writes (round 6): the next LayerNorm needs no statistics pass"""
    N.check(N.lib().mlpk_hire_combine_stats(dtype_code(x.dtype), ptr(x), ptr(src), ptr(y_h), ptr(y_w), B, H, W, C, h, w, step, ld_h, ld_w,
                                            ptr(mean), ptr(rstd), eps, stream()), "mlpk_hire_combine_stats")


def hire_combine(x, y_h, y_w, B, H, W, C, h, w, step, ld_h, ld_w):
    N.check(N.lib().mlpk_hire_combine(dtype_code(x.dtype), ptr(x), ptr(y_h), ptr(y_w), B, H, W, C, h, w, step, ld_h, ld_w,
                                      stream()), "mlpk_hire_combine")


def mixshift_nhwc(x, out, B, H, W, C, shift, ksize, w_lr, b_lr, w_td, b_td, part=None):
    """part = (workspace, name): also deliver the by-product statistics planes of `out` (32 channels each) when the kernel takes the shape --
    returns (buffer (C / 32, rows, 2), C / 32) as engine.gemm(part=...) does, for finalize_stats; None otherwise (the caller runs row_stats)."""
    g = len(shift)
    arr = ctypes.c_int * g
    if part is not None and epilogue_stats() and x.dtype != torch.float32:
        nq = N.lib().mlpk_mixshift_stats_planes(dtype_code(x.dtype), B, H, W, C, g, arr(*ksize))
        if nq > 0:
            rows = B * H * W
            buf = part[0].get("%s.%d" % (part[1], nq), (nq, rows, 2), torch.float32)
            N.check(N.lib().mlpk_mixshift_nhwc_stats(dtype_code(x.dtype), ptr(x), ptr(out), B, H, W, C, g, arr(*shift), arr(*ksize), ptr(w_lr), ptr(b_lr),
                                                     ptr(w_td), ptr(b_td), ptr(buf), rows, stream()), "mlpk_mixshift_nhwc_stats")
            return buf, nq
    N.check(N.lib().mlpk_mixshift_nhwc(dtype_code(x.dtype), ptr(x), ptr(out), B, H, W, C, g, arr(*shift), arr(*ksize), ptr(w_lr), ptr(b_lr),
                                       ptr(w_td), ptr(b_td), stream()), "mlpk_mixshift_nhwc")
    return None


def window_gather(x, windows, B, H, W, C, ws, pad_t, pad_l, Hp, Wp):
    N.check(N.lib().mlpk_window_gather(dtype_code(x.dtype), ptr(x), ptr(windows), B, H, W, C, ws, pad_t, pad_l, Hp, Wp, stream()),
            "mlpk_window_gather")


def window_scatter_add(x, windows, B, H, W, C, ws, pad_t, pad_l, Hp, Wp):
    N.check(N.lib().mlpk_window_scatter_add(dtype_code(x.dtype), ptr(x), ptr(windows), B, H, W, C, ws, pad_t, pad_l, Hp, Wp, stream()),
            "mlpk_window_scatter_add")


def convert(src, dst, n):
    N.check(N.lib().mlpk_convert(dtype_code(src.dtype), dtype_code(dst.dtype), ptr(src), ptr(dst), n, stream()),
            "mlpk_convert")


# ------------------------------------------------------------------ per-model caches
class Workspace:
    """Named device buffers for one (batch, dtype, device) shape of a model.  torch owns the
    memory; buffers are zero-initialised once (GEMM K-padding columns rely on that) and never
    re-purposed for another role."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.t = {}

    def get(self, name, shape, dtype=None, fill=0.0):
        t = self.t.get(name)
        if t is None:
            t = torch.full(shape, fill, dtype=dtype or self.dtype, device=self.device)
            self.t[name] = t
        elif tuple(t.shape) != tuple(shape):
            # the same role at another size (a model without a fixed image size called at a second resolution with
            # a workspace key that did not separate them): a fresh zero-initialised buffer, never a reinterpretation.
            # Said once per workspace: two roles sharing one name would land here on every call and keep re-allocating.
            if not self.t.get(("warned", "resize")):
                import warnings
                warnings.warn("workspace buffer %r requested as %s after %s: re-allocated (workspace keys should separate these)"
                              % (name, tuple(shape), tuple(t.shape)))
                self.t[("warned", "resize")] = True
            t = torch.full(shape, fill, dtype=dtype or self.dtype, device=self.device)
            self.t[name] = t
        return t

    def get_event(self, name):
        """A named, reusable torch.cuda.Event (fork / join of a side stream)."""
        ev = self.t.get(("event", name))
        if ev is None:
            ev = self.t[("event", name)] = torch.cuda.Event()
        return ev


class EngineModule(torch.nn.Module):
    """Base of the drop-in models: caches packed weights per (dtype, device) and workspaces per
    (batch, dtype, device); caches are dropped whenever parameters change identity/version."""

    def __init__(self):
        super().__init__()
        self.__dict__["_packs"] = {}
        self.__dict__["_spaces"] = {}
        self.__dict__["_compute_dtype"] = None
        self.__dict__["_in_shape"] = None
        self.__dict__["_warned_train"] = False

    def __call__(self, *args, **kwargs):
        # run the whole forward with the input's device current: launches go to THAT device's stream
        x = args[0] if args else None
        if torch.is_tensor(x) and x.is_cuda:
            with on_device(x):
                outer = getattr(_TLS, "stream", None)
                _TLS.stream = torch.cuda.current_stream().cuda_stream       # asked once per call, not once per launch (engine.stream)
                try:
                    return super().__call__(*args, **kwargs)
                finally:
                    _TLS.stream = outer
        return super().__call__(*args, **kwargs)

    def set_compute_dtype(self, dtype):
        """Run the MFMA path in `dtype` regardless of the input dtype (input is converted while
        the patches are gathered; logits come back in the input dtype)."""
        self.__dict__["_compute_dtype"] = dtype
        return self

    def _param_stamp(self):
        # every parameter and buffer of the tree, by an explicit walk over _modules (nn.Module.parameters() is a recursive generator chain:
        # 17 000 generator frames per forward on Hire-MLP's 1 300 modules)
        out, stack, seen = [], [self], set()
        while stack:
            m = stack.pop()
            if id(m) in seen:
                continue
            seen.add(id(m))
            for t in m._parameters.values():
                if t is not None:
                    out.append((t.data_ptr(), t._version))
            for t in m._buffers.values():
                if t is not None:
                    out.append((t.data_ptr(), t._version))
            stack.extend(c for c in m._modules.values() if c is not None)
        return tuple(out)

    def _get_pack(self, dtype, device):
        key = (dtype, str(device))
        stamp = self._param_stamp()
        hit = self._packs.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, self._pack(dtype, device))
            self._packs[key] = hit
        return hit[1]

    def _get_space(self, batch, dtype, device):
        # keyed by the input's spatial shape too: ConvMixer / Hire-MLP / ... take any resolution, like the reference
        # ... and by the stream the call is issued on: two streams running the same module at once (batch shards of one
        # device, tools/two_stream_probe.py) must not share scratch buffers
        key = (batch, self._in_shape, dtype, str(device), stream())
        ws = self._spaces.get(key)
        if ws is None:
            if len(self._spaces) >= 4:          # bound resident workspaces (288 GB is big, not infinite)
                self._spaces.pop(next(iter(self._spaces)))
            ws = Workspace(device, dtype)
            self._spaces[key] = ws
        return ws

    def _resolve(self, x):
        require_gpu(x, type(self).__name__ + ".forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        self.__dict__["_in_shape"] = tuple(x.shape[1:])
        if self.training and not self._warned_train and not getattr(self, "_train_forward", False):
            import warnings
            warnings.warn("%s is inference-only: forward() ignores train mode (Dropout / DropPath are identity, BatchNorm uses "
                          "its running statistics) and the outputs carry no grad_fn; call .eval()" % type(self).__name__,
                          stacklevel=3)
            self.__dict__["_warned_train"] = True
        elif self.training and not self._warned_train and getattr(self, "_train_forward", False) == "forward-only":
            import warnings
            warnings.warn("%s.train(): the train-mode FORWARD is implemented (batch statistics / stochastic depth), the backward is not -- "
                          "the outputs carry no grad_fn and a frozen sub-module contributes no gradients" % type(self).__name__, stacklevel=3)
            self.__dict__["_warned_train"] = True
        cd = self._compute_dtype or x.dtype
        dtype_code(cd)
        dtype_code(x.dtype)
        return cd

    def _pack(self, dtype, device):  # pragma: no cover - abstract
        raise NotImplementedError
