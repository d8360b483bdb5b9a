"""Autograd through the HIP path: `torch.autograd.Function`s whose forward AND backward are kernels of
libmlpk -- the pattern of the reference's one native op, `_shift` (utils/shift_cuda.py:106-162: an autograd.Function around a forward and
a backward kernel on raw device pointers).  torch supplies the tape, the memory and the stream; every product and every element-wise
derivative below is a call through the C ABI:

    y = x W^T + b (+ r)   forward  mlpk_gemm_nt (bias / residual epilogue)                           nn.Linear, Conv1d(k=1), Conv2d(k=stride)
                          backward dX = mlpk_gemm_nt(dY, W^T),  dW = mlpk_gemm_nt(dY^T, X^T)   (K-contiguous copies: mlpk_transpose_batched),
                                   db = mlpk_col_sum(dY),  dr = dY
    gelu                  mlpk_gelu_elementwise (the pre-activation is kept: train mode does not fuse the activation into the GEMM)
    LayerNorm             mlpk_row_stats + mlpk_norm_apply / mlpk_layernorm_backward (+ mlpk_col_sum for d gamma, d beta)
    token <-> channel     mlpk_transpose_batched (the residual add of the token-mixing block rides on the way back)
    token mean            mlpk_pool_mean / mlpk_broadcast_rows

Used by the train-mode forward of MLPMixerForImageClassification (mlp_mixer.py:30-75); inference keeps its fused kernels.  Parameter
gradients come back in fp32 whatever the compute dtype (the GEMMs accumulate in fp32 and round dW once to the compute dtype).
"""
import weakref

import torch

from . import _native as N
from . import engine as E


def _epc(dtype):
    return 4 if dtype == torch.float32 else 8


def _pad_cols(t, cols):
    """(M, k) -> contiguous (M, cols) with zero columns behind k (a GEMM's K must be whole 16-byte chunks)"""
    if t.shape[1] == cols and t.is_contiguous():
        return t
    out = torch.zeros((t.shape[0], cols), dtype=t.dtype, device=t.device)
    out[:, :t.shape[1]].copy_(t)
    return out


def transpose(x, batch, R, Cc, ld_out=None, res=None):
    """x: (batch * R, Cc) row-major -> (batch * Cc, ld_out) with out[b, c, r] = x[b, r, c] (+ res); padding columns zero"""
    ld_out = ld_out or R
    out = (torch.zeros if ld_out != R else torch.empty)((batch * Cc, ld_out), dtype=x.dtype, device=x.device)
    N.check(N.lib().mlpk_transpose_batched(E.dtype_code(x.dtype), E.ptr(x), x.stride(0), E.ptr(out), ld_out, E.ptr(res), res.stride(0) if res is not None else 0,
                                           batch, R, Cc, E.stream()), "mlpk_transpose_batched")
    return out


def col_sum(x, rows, cols, square=False, sub=None):
    out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    assert sub is None or (sub.stride(0) == x.stride(0) and sub.dtype == x.dtype)
    N.check(N.lib().mlpk_col_sum(E.dtype_code(x.dtype), E.ptr(x), E.ptr(sub), rows, cols, x.stride(0), int(square), E.ptr(out), E.stream()), "mlpk_col_sum")
    return out


_PACKS = {}


def _packed(w, w2, cd, dev, kp=None):
    """the packed (compute-dtype, K-padded) copy of parameter w, cached per parameter version like EngineModule._get_pack (a training
    step used to re-pack every weight in every forward).  An entry belongs to ONE live tensor object (weak reference): the allocator
    hands a freed parameter's address -- with version 0 again -- to the next model's parameter, so the address alone is not an identity."""
    key = (id(w), cd, str(dev), kp)
    hit = _PACKS.get(key)
    if hit is None or hit[0]() is not w or hit[1] != (w.data_ptr(), w._version) or hit[2].shape[0] != w2.shape[0]:
        if len(_PACKS) > 512:
            _PACKS.clear()
        packed = E.pack_matrix(w2, cd, dev, kpad=_epc(cd))
        if kp is not None and kp > packed.shape[1]:                      # an operand with more zero padding columns than the dtype needs (im2col: 16-byte rows)
            packed = _pad_cols(packed, kp)
        hit = (weakref.ref(w), (w.data_ptr(), w._version), packed)
        _PACKS[key] = hit
    return hit[2]


class Linear(torch.autograd.Function):
    """y = x W^T + b (+ r).  x: (M, K) or (M, K_pad) with zero padding columns; w: parameter (N, K, ...) in fp32; r: (M, N) or None."""

    @staticmethod
    def forward(ctx, x, w, b, r):
        cd, dev = x.dtype, x.device
        w2 = w.reshape(w.shape[0], -1)
        n, k = w2.shape
        kp = E.round_up(k, _epc(cd))
        if x.shape[1] > kp and x.shape[1] % _epc(cd) == 0:
            kp = x.shape[1]                                                 # wider zero padding (mlpk_im2col pads rows to 16 bytes whatever the dtype)
        assert x.dim() == 2 and x.shape[1] in (k, kp), (tuple(x.shape), k)
        with E.on_device(x):
            xp = _pad_cols(x, kp)
            wp = _packed(w, w2, cd, dev, kp)
            y = torch.empty((x.shape[0], n), dtype=cd, device=dev)
            E.gemm(xp, wp, y, x.shape[0], n, kp, bias=E.f32(b, dev), R=r, res=N.RES_ADD if r is not None else N.RES_NONE)
        ctx.save_for_backward(xp, wp)
        ctx.meta = (tuple(w.shape), k, x.shape[1], b is not None, r is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        wshape, k, xcols, has_b, has_r = ctx.meta
        cd, dev = xp.dtype, xp.device
        m, n = dy.shape
        epc = _epc(cd)
        np_, mp = E.round_up(n, epc), E.round_up(m, epc)
        dy = dy.contiguous()
        dx = dw = db = None
        with E.on_device(dy):
            if ctx.needs_input_grad[0]:
                dyp = _pad_cols(dy, np_)
                wt = transpose(wp, 1, n, wp.shape[1], ld_out=np_)                    # (K_pad, N_pad): W^T, contraction axis contiguous
                dx = (torch.zeros if xcols != k else torch.empty)((m, xcols), dtype=cd, device=dev)
                E.gemm(dyp, wt, dx, m, k, np_, ldc=xcols)
            if ctx.needs_input_grad[1]:
                dyt = transpose(dy, 1, m, n, ld_out=mp)                             # (N, M_pad)
                xt = transpose(xp, 1, m, k, ld_out=mp)                              # (K, M_pad)
                # the weight gradient is produced in fp32 like the reference's autograd / AMP master gradients: a 16-bit store of the
                # product would underflow (fp16, loss scaling) or keep 8 bits (bf16).  16-bit runs multiply the fp32 copies of the two
                # transposed operands on the exact-f32 MFMA tile (advisor, round 5)
                if cd != torch.float32:
                    dyt, xt = dyt.float(), xt.float()
                dwc = torch.empty((n, k), dtype=torch.float32, device=dev)
                E.gemm(dyt, xt, dwc, n, k, mp)
                dw = dwc.reshape(wshape)
            if has_b and ctx.needs_input_grad[2]:
                db = col_sum(dy, m, n)
        return dx, dw, db, (dy if has_r else None)


class Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre):
        out = torch.empty_like(pre)
        with E.on_device(pre):
            N.check(N.lib().mlpk_gelu_elementwise(E.dtype_code(pre.dtype), 0, E.ptr(pre), None, E.ptr(out), pre.shape[0], pre.shape[1], pre.stride(0), E.stream()),
                    "mlpk_gelu_elementwise")
        ctx.save_for_backward(pre)
        return out

    @staticmethod
    def backward(ctx, dy):
        (pre,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(pre)
        with E.on_device(pre):
            N.check(N.lib().mlpk_gelu_elementwise(E.dtype_code(pre.dtype), 1, E.ptr(pre), E.ptr(dy), E.ptr(dx), pre.shape[0], pre.shape[1], pre.stride(0), E.stream()),
                    "mlpk_gelu_elementwise")
        return dx


class LayerNorm(torch.autograd.Function):
    """nn.LayerNorm over the last axis of (M, C) rows"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        m, c = x.shape
        dev = x.device
        with E.on_device(x):
            mean = torch.empty((m,), dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            g32, b32 = E.f32(gamma, dev), E.f32(beta, dev)
            E.row_stats(x, m, c, x.stride(0), mean, rstd, eps=eps)
            y = torch.empty((m, c), dtype=x.dtype, device=dev)
            E.norm_apply(x, m, c, x.stride(0), mean=mean, rstd=rstd, gamma=g32, beta=b32, out_rm=y, ld_rm=c)
        ctx.save_for_backward(x, mean, rstd, g32)
        ctx.pshape = (tuple(gamma.shape), tuple(beta.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, g32 = ctx.saved_tensors
        m, c = x.shape
        dy = dy.contiguous()
        with E.on_device(x):
            nb = N.lib().mlpk_layernorm_backward_blocks(m)
            part = torch.empty((nb, 2 * c), dtype=torch.float32, device=x.device)
            dx = torch.empty_like(x)
            N.check(N.lib().mlpk_layernorm_backward(E.dtype_code(x.dtype), E.ptr(x), x.stride(0), E.ptr(mean), E.ptr(rstd), E.ptr(g32), E.ptr(dy), dy.stride(0),
                                                    E.ptr(dx), dx.stride(0), E.ptr(part), m, c, E.stream()), "mlpk_layernorm_backward")
            sums = col_sum(part, nb, 2 * c)
        return dx, sums[:c].reshape(ctx.pshape[0]), sums[c:].reshape(ctx.pshape[1]), None


class TokensToRows(torch.autograd.Function):
    """(B*S, C) token-major -> (B*C, S_pad): rows = (image, channel), the token axis contiguous and zero-padded to a GEMM's K
    (the rearrange in front of the token-mixing FeedForward, mlp_mixer.py:34: Conv1d(k=1) over the patch axis)"""

    @staticmethod
    def forward(ctx, x, B, S):
        ctx.dims = (B, S, x.shape[1])
        with E.on_device(x):
            return transpose(x, B, S, x.shape[1], ld_out=E.round_up(S, _epc(x.dtype)))

    @staticmethod
    def backward(ctx, dxt):
        B, S, C = ctx.dims
        dxt = dxt.contiguous()
        with E.on_device(dxt):
            return transpose(dxt, B, C, S, ld_out=C), None, None


class RowsToTokensAdd(torch.autograd.Function):
    """out[b, s, c] = x[b, s, c] + y[b, c, s]: the token-mixing block's result back in token-major order, plus the residual (mlp_mixer.py:12)"""

    @staticmethod
    def forward(ctx, y, x, B, S):
        C = x.shape[1]
        ctx.dims = (B, S, C)
        with E.on_device(x):
            return transpose(y, B, C, S, ld_out=C, res=x)

    @staticmethod
    def backward(ctx, dout):
        B, S, C = ctx.dims
        dout = dout.contiguous()
        with E.on_device(dout):
            return transpose(dout, B, S, C, ld_out=S), dout, None, None


class TokenMean(torch.autograd.Function):
    """Reduce('b n c -> b c', 'mean') (mlp_mixer.py:63)"""

    @staticmethod
    def forward(ctx, x, B, S):
        C = x.shape[1]
        ctx.dims = (B, S, C)
        out = torch.empty((B, C), dtype=x.dtype, device=x.device)
        with E.on_device(x):
            E.pool_mean(x, B, S, C, x.stride(0), out, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, S, C = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((B * S, C), dtype=dy.dtype, device=dy.device)
        with E.on_device(dy):
            N.check(N.lib().mlpk_broadcast_rows(E.dtype_code(dy.dtype), E.ptr(dy), E.ptr(dx), B, S, C, 1.0 / S, E.stream()), "mlpk_broadcast_rows")
        return dx, None, None


def batch_stats(x, rows, cols, sub=None):
    """per-column (mean, biased variance) over the rows of x (- sub), fp64 on the host side of two fp32 column sums: BatchNorm2d's batch
    statistics on a channel-last tensor (conv_mixer.py:20,28,31)"""
    with E.on_device(x):
        s1 = col_sum(x, rows, cols, sub=sub).double()
        s2 = col_sum(x, rows, cols, square=True, sub=sub).double()
    mean = s1 / rows
    var = (s2 / rows - mean * mean).clamp_min_(0.0)
    return mean, var


def batchnorm_train_affine(bn, mean, var, rows):
    """(scale, shift) of BatchNorm2d in train mode from the batch statistics, and the running-statistics update torch performs
    (momentum, unbiased variance, num_batches_tracked): conv_mixer.py:20,28,31 with nn.BatchNorm2d's defaults"""
    with torch.no_grad():
        scale = bn.weight.detach().double() / torch.sqrt(var + bn.eps)
        shift = bn.bias.detach().double() - mean * scale
        if bn.track_running_stats:
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1 - mom).add_(mom * mean.to(bn.running_mean.dtype))
            bn.running_var.mul_(1 - mom).add_(mom * (var * (rows / max(rows - 1, 1))).to(bn.running_var.dtype))
            bn.num_batches_tracked += 1
    return scale.float().contiguous(), shift.float().contiguous()


# ---- round 6: the pieces the train mode of gMLP, ResMLP, AS-MLP and ConvMixer adds (ABI 11) ----------------------------------------------
def _ew(mode, a, b=None, g=None, h=None, k=None, period=1, out=None):
    """mlpk_ew_cols on (rows, cols) tensors (any row stride); g / h / k fp32 vectors"""
    rows, cols = a.shape
    out = torch.empty((rows, cols), dtype=a.dtype, device=a.device) if out is None else out
    assert a.stride(1) == 1 and (b is None or (b.stride(1) == 1 and b.shape == a.shape and b.dtype == a.dtype))
    N.check(N.lib().mlpk_ew_cols(E.dtype_code(a.dtype), mode, E.ptr(a), a.stride(0), E.ptr(b), b.stride(0) if b is not None else 0, E.ptr(g), E.ptr(h),
                                 E.ptr(k), E.ptr(out), out.stride(0), rows, cols, period, E.stream()), "mlpk_ew_cols")
    return out


def col_dot(x, y):
    rows, cols = x.shape
    assert x.stride(1) == 1 and y.stride(1) == 1 and x.shape == y.shape and x.dtype == y.dtype
    out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    N.check(N.lib().mlpk_col_dot(E.dtype_code(x.dtype), E.ptr(x), x.stride(0), E.ptr(y), y.stride(0), rows, cols, E.ptr(out), E.stream()), "mlpk_col_dot")
    return out


def _rows(t):
    """a gradient as the kernels take it: unit column stride (a chunk of a wider tensor keeps its row stride)"""
    return t if t.stride(1) == 1 else t.contiguous()


class Affine(torch.autograd.Function):
    """y = x * alpha[c] + beta[c] on (M, C) rows: Aff (res_mlp.py:11-19), the affine half of GroupNorm / BatchNorm.  beta may be None."""

    @staticmethod
    def forward(ctx, x, alpha, beta):
        with E.on_device(x):
            a32 = E.f32(alpha.reshape(-1), x.device)
            y = _ew(0, x, g=a32, h=E.f32(beta.reshape(-1), x.device) if beta is not None else None)
        ctx.save_for_backward(x, a32)
        ctx.pshape = (tuple(alpha.shape), tuple(beta.shape) if beta is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a32 = ctx.saved_tensors
        dy = _rows(dy)
        dx = da = db = None
        with E.on_device(x):
            if ctx.needs_input_grad[0]:
                dx = _ew(0, dy, g=a32)
            if ctx.needs_input_grad[1]:
                da = col_dot(dy, x).reshape(ctx.pshape[0])
            if ctx.pshape[1] is not None and ctx.needs_input_grad[2]:
                db = col_sum(dy, dy.shape[0], dy.shape[1]).reshape(ctx.pshape[1])
        return dx, da, db


class ScaleAdd(torch.autograd.Function):
    """out = x + gamma[c] * z (res_mlp.py:53,55: x + gamma_1 * token_mix(x));  gamma None: x + z"""

    @staticmethod
    def forward(ctx, x, z, gamma):
        with E.on_device(x):
            g32 = E.f32(gamma.reshape(-1), x.device) if gamma is not None else None
            out = _ew(2, x, b=_rows(z), g=g32)
        ctx.save_for_backward(z, g32)
        ctx.gshape = tuple(gamma.shape) if gamma is not None else None
        return out

    @staticmethod
    def backward(ctx, dy):
        z, g32 = ctx.saved_tensors
        dy = _rows(dy)
        dz = dg = None
        with E.on_device(dy):
            if ctx.needs_input_grad[1]:
                dz = _ew(0, dy, g=g32) if g32 is not None else dy
            if g32 is not None and ctx.needs_input_grad[2]:
                dg = col_dot(dy, _rows(z)).reshape(ctx.gshape)
        return dy, dz, dg


class Mul(torch.autograd.Function):
    """out = u * v (the SGU gate, g_mlp.py:21); u, v: (M, C) rows, possibly the two halves of one wider tensor"""

    @staticmethod
    def forward(ctx, u, v):
        with E.on_device(u):
            out = _ew(1, u, b=v)
        ctx.save_for_backward(u, v)
        return out

    @staticmethod
    def backward(ctx, dy):
        u, v = ctx.saved_tensors
        dy = _rows(dy)
        with E.on_device(dy):
            return _ew(1, dy, b=v), _ew(1, dy, b=u)


class RowScale(torch.autograd.Function):
    """out[m] = x[m] * s[m // period]: stochastic depth's keep / (1 - p) per sample (as_mlp.py:159-160); s fp32, no gradient"""

    @staticmethod
    def forward(ctx, x, s, period):
        ctx.save_for_backward(s)
        ctx.period = period
        with E.on_device(x):
            return _ew(3, x, g=s, period=period)

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        with E.on_device(dy):
            return _ew(3, _rows(dy), g=s, period=ctx.period), None, None


class RowsToTokens(torch.autograd.Function):
    """(B*C, S) rows back to token-major (B*S, C) -- RowsToTokensAdd without the residual (res_mlp.py:53: the product is scaled first)"""

    @staticmethod
    def forward(ctx, y, B, S, C):
        ctx.dims = (B, S, C)
        with E.on_device(y):
            return transpose(y, B, C, S, ld_out=C)

    @staticmethod
    def backward(ctx, dout):
        B, S, C = ctx.dims
        with E.on_device(dout):
            return transpose(dout.contiguous(), B, S, C, ld_out=S), None, None, None


class GroupNorm1(torch.autograd.Function):
    """nn.GroupNorm(1, C) (as_mlp.py:343-344 MyNorm) on channel-last samples: x (B * HW, C) contiguous, one statistic per sample"""

    @staticmethod
    def forward(ctx, x, gamma, beta, B, eps):
        rows, C = x.shape
        HW = rows // B
        dev = x.device
        x = x.contiguous()
        with E.on_device(x):
            mean = torch.empty((B,), dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            E.row_stats(x, B, HW * C, HW * C, mean, rstd, eps=eps)
            xh = torch.empty_like(x)
            E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, stat_group=HW, out_rm=xh, ld_rm=C)
            g32 = E.f32(gamma, dev)
            y = _ew(0, xh, g=g32, h=E.f32(beta, dev))
        ctx.save_for_backward(xh, rstd, g32)
        ctx.meta = (B, HW * C, tuple(gamma.shape), tuple(beta.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, rstd, g32 = ctx.saved_tensors
        B, glen, gs, bs = ctx.meta
        dy = dy.contiguous()
        dx = dg = db = None
        with E.on_device(dy):
            if ctx.needs_input_grad[0]:
                g = _ew(0, dy, g=g32)
                dx = torch.empty_like(xh)
                N.check(N.lib().mlpk_group_norm_backward(E.dtype_code(xh.dtype), E.ptr(xh), E.ptr(g), E.ptr(rstd), E.ptr(dx), B, glen, E.stream()),
                        "mlpk_group_norm_backward")
            if ctx.needs_input_grad[1]:
                dg = col_dot(dy, xh).reshape(gs)
            if ctx.needs_input_grad[2]:
                db = col_sum(dy, dy.shape[0], dy.shape[1]).reshape(bs)
        return dx, dg, db, None, None


class ShiftNHWC(torch.autograd.Function):
    """Shift (utils/shift_cuda.py:165-192; torch_shift :177-189) on channel-last rows (B*H*W, C): the reference's one native op and its backward kernel"""

    @staticmethod
    def forward(ctx, x, B, H, W, ksz, dim):
        ctx.meta = (B, H, W, x.shape[1], ksz, dim)
        x = x.contiguous()
        out = torch.empty_like(x)
        with E.on_device(x):
            E.shift_nhwc(x, out, B, H, W, x.shape[1], ksz, dim)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, ksz, dim = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        with E.on_device(dy):
            N.check(N.lib().mlpk_shift_nhwc_backward(E.dtype_code(dy.dtype), E.ptr(dy), E.ptr(dx), B, H, W, C, ksz, dim, E.stream()), "mlpk_shift_nhwc_backward")
        return dx, None, None, None, None, None


class Merge2x2(torch.autograd.Function):
    """PatchMerging's strided gather + concatenation (as_mlp.py:207-211) on channel-last rows: (B*H*W, C) -> (B*H/2*W/2, 4C)"""

    @staticmethod
    def forward(ctx, x, B, H, W):
        C = x.shape[1]
        ctx.meta = (B, H, W, C)
        x = x.contiguous()
        out = torch.empty((B * (H // 2) * (W // 2), 4 * C), dtype=x.dtype, device=x.device)
        with E.on_device(x):
            N.check(N.lib().mlpk_merge2x2_nhwc(E.dtype_code(x.dtype), 0, E.ptr(x), E.ptr(out), B, H, W, C, E.stream()), "mlpk_merge2x2_nhwc")
        return out

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty((B * H * W, C), dtype=dy.dtype, device=dy.device)
        with E.on_device(dy):
            N.check(N.lib().mlpk_merge2x2_nhwc(E.dtype_code(dy.dtype), 1, E.ptr(dy), E.ptr(dx), B, H, W, C, E.stream()), "mlpk_merge2x2_nhwc")
        return dx, None, None, None


def _taps(w, dev):
    """depthwise Conv2d weight (C, 1, k, k) -> fp32 [k*k][C] tap-major, the layout of mlpk_dwconv_nhwc"""
    C, k = w.shape[0], w.shape[-1]
    return w.detach().to(device=dev, dtype=torch.float32).reshape(C, k * k).t().contiguous()


class DepthwiseConv(torch.autograd.Function):
    """Conv2d(C, C, k, groups=C, padding="same") on channel-last rows (B*H*W, C), no epilogue (conv_mixer.py:25): the pre-activation is kept"""

    @staticmethod
    def forward(ctx, x, w, b, B, H, W):
        C, k = x.shape[1], w.shape[-1]
        x = x.contiguous()
        out = torch.empty_like(x)
        with E.on_device(x):
            wt = _taps(w, x.device)
            N.check(N.lib().mlpk_dwconv_plain_nhwc(E.dtype_code(x.dtype), 0, E.ptr(x), E.ptr(out), B, H, W, C, k, E.ptr(wt), E.ptr(E.f32(b, x.device)), E.stream()),
                    "mlpk_dwconv_plain_nhwc")
        ctx.save_for_backward(x, wt)
        ctx.meta = (B, H, W, C, k, tuple(w.shape), b is not None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        B, H, W, C, k, wshape, has_b = ctx.meta
        dy = dy.contiguous()
        dx = dw = db = None
        with E.on_device(dy):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                N.check(N.lib().mlpk_dwconv_plain_nhwc(E.dtype_code(x.dtype), 1, E.ptr(dy), E.ptr(dx), B, H, W, C, k, E.ptr(wt), None, E.stream()),
                        "mlpk_dwconv_plain_nhwc")
            if ctx.needs_input_grad[1]:
                dwt = torch.empty((k * k, C), dtype=torch.float32, device=x.device)
                N.check(N.lib().mlpk_dwconv_wgrad_nhwc(E.dtype_code(x.dtype), E.ptr(x), E.ptr(dy), E.ptr(dwt), B, H, W, C, k, E.stream()), "mlpk_dwconv_wgrad_nhwc")
                dw = dwt.t().reshape(wshape)
            if has_b and ctx.needs_input_grad[2]:
                db = col_sum(dy, dy.shape[0], C)
        return dx, dw, db, None, None, None


class BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d in train mode on channel-last rows (M, C): batch statistics (biased variance) in the forward, the running statistics
    updated by the caller (batchnorm_train_affine), the full backward through the statistics (conv_mixer.py:20,28,31)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps):
        dev = x.device
        with E.on_device(x):
            rs = (1.0 / torch.sqrt(var + eps)).float().contiguous()
            xh = _ew(0, x, g=rs, h=(-mean * rs.double()).float().contiguous())
            g32 = E.f32(gamma, dev)
            y = _ew(0, xh, g=g32, h=E.f32(beta, dev))
        ctx.save_for_backward(xh, rs, g32)
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, rs, g32 = ctx.saved_tensors
        dy = _rows(dy)
        m, c = dy.shape
        dx = dg = db = None
        with E.on_device(dy):
            s1 = col_sum(dy, m, c)
            s2 = col_dot(dy, xh)
            if ctx.needs_input_grad[0]:
                a = (g32 * rs).contiguous()
                dx = _ew(4, dy, b=xh, g=a, h=(-a * s2 / m).contiguous(), k=(-a * s1 / m).contiguous())
            if ctx.needs_input_grad[1]:
                dg = s2
            if ctx.needs_input_grad[2]:
                db = s1
        return dx, dg, db, None, None, None


# ---- ViP / S2-MLP: rearranges, SplitAttention, the spatial shifts (round 6) ----------------------------------------------------------------
class VipPermute(torch.autograd.Function):
    """einops Rearrange 'b h w (c s) -> b w c (h s)' (which 0, vip.py:69) / '-> b h c (w s)' (which 1, vip.py:74) on channel-last rows:
    (B*H*W, C) -> (B*L*G, K_pad), K = H*seg or W*seg, zero padding columns up to the GEMM's K.  Backward: the inverse rearrange."""

    @staticmethod
    def forward(ctx, x, B, H, W, seg, which):
        C = x.shape[1]
        G = C // seg
        K = (H if which == 0 else W) * seg
        kp = E.round_up(K, _epc(x.dtype))
        ctx.meta = (B, H, W, C, seg, which, kp)
        x = x.contiguous()
        z = torch.zeros((B * (W if which == 0 else H) * G, kp), dtype=x.dtype, device=x.device)
        with E.on_device(x):
            if which == 0:
                E.norm_apply(x, B * H * W, C, C, out_ph=z, H=H, W=W, seg=seg, ld_p=kp)
            else:
                E.norm_apply(x, B * H * W, C, C, out_pw=z, H=H, W=W, seg=seg, ld_p=kp)
        return z

    @staticmethod
    def backward(ctx, dz):
        B, H, W, C, seg, which, kp = ctx.meta
        dz = dz.contiguous()
        dx = torch.empty((B * H * W, C), dtype=dz.dtype, device=dz.device)
        with E.on_device(dz):
            E.vip_unpermute(which, dz, dx, B, H, W, C, seg, dz.stride(0))
        return dx, None, None, None, None, None


class VipUnpermute(torch.autograd.Function):
    """the Rearrange back, 'b w c (h s) -> b h w (c s)' (which 0, vip.py:71) / 'b h c (w s) -> b h w (c s)' (which 1, vip.py:76)"""

    @staticmethod
    def forward(ctx, z, B, H, W, C, seg, which):
        ctx.meta = (B, H, W, C, seg, which, z.shape[1])
        z = z.contiguous()
        out = torch.empty((B * H * W, C), dtype=z.dtype, device=z.device)
        with E.on_device(z):
            E.vip_unpermute(which, z, out, B, H, W, C, seg, z.stride(0))
        return out

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, seg, which, K = ctx.meta
        dy = dy.contiguous()
        dz = torch.empty((B * (W if which == 0 else H) * (C // seg), K), dtype=dy.dtype, device=dy.device)
        with E.on_device(dy):
            if which == 0:
                E.norm_apply(dy, B * H * W, C, C, out_ph=dz, H=H, W=W, seg=seg, ld_p=K)
            else:
                E.norm_apply(dy, B * H * W, C, C, out_pw=dz, H=H, W=W, seg=seg, ld_p=K)
        return dz, None, None, None, None, None, None


class ImageSum(torch.autograd.Function):
    """(B*S, C) -> (B, C): the sum over an image's pixels (SplitAttention's reduction, vip.py:49-50; s2_mlp_v2.py:43-44)"""

    @staticmethod
    def forward(ctx, x, B, S):
        C = x.shape[1]
        ctx.dims = (B, S, C)
        x = _rows(x)
        out = torch.empty((B, C), dtype=x.dtype, device=x.device)
        with E.on_device(x):
            E.pool_mean(x, B, S, C, x.stride(0), out, C)
            return _ew(0, out, g=torch.full((C,), float(S), dtype=torch.float32, device=x.device))

    @staticmethod
    def backward(ctx, dy):
        B, S, C = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((B * S, C), dtype=dy.dtype, device=dy.device)
        with E.on_device(dy):
            N.check(N.lib().mlpk_broadcast_rows(E.dtype_code(dy.dtype), E.ptr(dy), E.ptr(dx), B, S, C, 1.0, E.stream()), "mlpk_broadcast_rows")
        return dx, None, None


class SoftmaxBranches(torch.autograd.Function):
    """nn.Softmax(1) over the k = 3 branches of hat_a viewed (B, 3, C) (vip.py:51-53), fp32 in and out"""

    @staticmethod
    def forward(ctx, hat, B, C):
        hat = hat.contiguous()
        bar = torch.empty_like(hat)
        with E.on_device(hat):
            E.split_softmax(hat, bar, B, C)
        ctx.save_for_backward(bar)
        ctx.dims = (B, C)
        return bar

    @staticmethod
    def backward(ctx, dbar):
        (bar,) = ctx.saved_tensors
        B, C = ctx.dims
        dbar = dbar.contiguous()
        dhat = torch.empty_like(bar)
        with E.on_device(bar):
            N.check(N.lib().mlpk_split_softmax_backward(E.ptr(bar), E.ptr(dbar), E.ptr(dhat), B, C, E.stream()), "mlpk_split_softmax_backward")
        return dhat, None, None


class WeightedSum3(torch.autograd.Function):
    """out[b, n, :] = sum_k bar[b, k, :] * x_k[b, n, :] (vip.py:54-56; s2_mlp_v2.py:48-50); bar fp32 (B, 3C); the branches may be column
    slices of one wider tensor (S2-MLPv2: the thirds of mlp1's output)"""

    @staticmethod
    def forward(ctx, x0, x1, x2, bar, B, S):
        C = x0.shape[1]
        bar3 = bar.reshape(B, 3, C)
        parts = [bar3[:, k].contiguous() for k in range(3)]
        with E.on_device(x0):
            out = _ew(5, _rows(x0), g=parts[0], period=S)
            out = _ew(5, _rows(x1), b=out, g=parts[1], period=S)
            out = _ew(5, _rows(x2), b=out, g=parts[2], period=S)
        ctx.save_for_backward(x0, x1, x2, *parts)
        ctx.dims = (B, S, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        x0, x1, x2, p0, p1, p2 = ctx.saved_tensors
        B, S, C = ctx.dims
        dy = _rows(dy)
        grads = []
        dbar = torch.empty((B, 3, C), dtype=torch.float32, device=dy.device)
        with E.on_device(dy):
            for k, (xk, pk_) in enumerate(((x0, p0), (x1, p1), (x2, p2))):
                grads.append(_ew(5, dy, g=pk_, period=S) if ctx.needs_input_grad[k] else None)
                xk = _rows(xk)
                seg = torch.empty((B, C), dtype=torch.float32, device=dy.device)
                N.check(N.lib().mlpk_col_dot_seg(E.dtype_code(dy.dtype), E.ptr(dy), dy.stride(0), E.ptr(xk), xk.stride(0), B, S, C, E.ptr(seg), E.stream()),
                        "mlpk_col_dot_seg")
                dbar[:, k] = seg
        return grads[0], grads[1], grads[2], dbar.reshape(B, 3 * C), None, None


def split_attention(x0, x1, x2, sa, B, S):
    """SplitAttention.forward (vip.py:46-57 = s2_mlp_v2.py:40-51) on three (B*S, C) branches as autograd.Functions"""
    cd = x0.dtype
    C = x0.shape[1]
    a = ScaleAdd.apply(ScaleAdd.apply(ImageSum.apply(x0, B, S), ImageSum.apply(x1, B, S), None), ImageSum.apply(x2, B, S), None)
    hat = Linear.apply(Gelu.apply(Linear.apply(a, sa.mlp1.weight, None, None)), sa.mlp2.weight, None, None)          # (B, 3C)
    bar = SoftmaxBranches.apply(hat.float(), B, C)
    return WeightedSum3.apply(x0, x1, x2, bar, B, S).to(cd)


class S2Shift(torch.autograd.Function):
    """spatial_shift1 / spatial_shift2 (s2_mlp_v2.py:15-29; which 1 / 2) on (B*D1*D2, C) rows -- possibly a column slice of a wider tensor.
    Forward: the reference's in-place result (mode "reference_inplace": the +1 groups smear) or the intended shift; backward: what the
    reference's autograd returns for the slice assignments -- the adjoint of the INTENDED shift in both modes (mlpk.h mlpk_s2_shift2)."""

    @staticmethod
    def forward(ctx, x, B, D1, D2, which, smear):
        C = x.shape[1]
        ctx.meta = (B, D1, D2, C, which)
        x = _rows(x)
        out = torch.empty((x.shape[0], C), dtype=x.dtype, device=x.device)
        with E.on_device(x):
            N.check(N.lib().mlpk_s2_shift2(E.dtype_code(x.dtype), which, 1 if smear else 0, 0, E.ptr(x), x.stride(0), E.ptr(out), C, B, D1, D2, C, E.stream()),
                    "mlpk_s2_shift2")
        return out

    @staticmethod
    def backward(ctx, dy):
        B, D1, D2, C, which = ctx.meta
        dy = _rows(dy)
        dx = torch.empty((dy.shape[0], C), dtype=dy.dtype, device=dy.device)
        with E.on_device(dy):
            N.check(N.lib().mlpk_s2_shift2(E.dtype_code(dy.dtype), which, 0, 1, E.ptr(dy), dy.stride(0), E.ptr(dx), C, B, D1, D2, C, E.stream()), "mlpk_s2_shift2")
        return dx, None, None, None, None, None


class PatchRowsNHWC(torch.autograd.Function):
    """the im2col half of Conv2d(kernel = stride = (ph, pw)) read from channel-last rows (B*H*W, C) -> (B*H/ph*W/pw, K_pad), K = ph*pw*C in
    mlpk_patchify's NHWC order 0, zero padding columns up to the GEMM's K; backward: the inverse permutation (s2_mlp_v2.py:118-119)"""

    @staticmethod
    def forward(ctx, x, B, H, W, ph, pw):
        C = x.shape[1]
        K = ph * pw * C
        kp = E.round_up(K, _epc(x.dtype))
        ctx.meta = (B, H, W, C, ph, pw, K, kp)
        x = x.contiguous()
        rows = B * (H // ph) * (W // pw)
        out = torch.empty((rows, K), dtype=x.dtype, device=x.device)
        with E.on_device(x):
            N.check(N.lib().mlpk_patch_rows_nhwc(E.dtype_code(x.dtype), 0, 0, E.ptr(x), E.ptr(out), B, H, W, C, ph, pw, E.stream()), "mlpk_patch_rows_nhwc")
        return _pad_cols(out, kp)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, ph, pw, K, kp = ctx.meta
        dy = dy[:, :K].contiguous()
        dx = torch.empty((B * H * W, C), dtype=dy.dtype, device=dy.device)
        with E.on_device(dy):
            N.check(N.lib().mlpk_patch_rows_nhwc(E.dtype_code(dy.dtype), 1, 0, E.ptr(dy), E.ptr(dx), B, H, W, C, ph, pw, E.stream()), "mlpk_patch_rows_nhwc")
        return dx, None, None, None, None, None


# ---- remaps as index tables (round 6: Swin-MLP, MS-MLP, Hire-MLP, CycleMLP) ----------------------------------------------------------------
class IndexTable:
    """A batch-independent remap `dst[i] = src[idx[i]]` (idx < 0: zero) over rows of `width` elements, with the table of its adjoint.
    Built from a LongTensor of source positions per destination position -- produced by running the reference's own index arithmetic
    (torch.roll, F.pad, view / permute) on a tensor of positions, so the table IS the reference's remap."""

    def __init__(self, idx, n_in, width, device):
        idx = idx.reshape(-1).to(torch.int64).cpu()
        self.n_out, self.n_in, self.width = idx.numel(), int(n_in), int(width)
        self.fwd = idx.to(torch.int32).to(device).contiguous()
        # the inverse relation: for every source position the destinations that read it, padded with -1 to the largest multiplicity
        valid = idx >= 0
        dst_pos = torch.nonzero(valid).reshape(-1)
        src_pos = idx[valid]
        order = torch.argsort(src_pos, stable=True)
        src_sorted, dst_sorted = src_pos[order], dst_pos[order]
        counts = torch.bincount(src_sorted, minlength=self.n_in)
        self.kmax = max(int(counts.max()) if counts.numel() else 1, 1)
        starts = torch.cumsum(counts, 0) - counts
        rank = torch.arange(src_sorted.numel()) - starts[src_sorted]
        inv = torch.full((self.n_in, self.kmax), -1, dtype=torch.int64)
        inv[src_sorted, rank] = dst_sorted
        self.inv = inv.to(torch.int32).to(device).contiguous()


def _index_gather(src, table, kmax, batch, n_out, n_in, width):
    out = torch.empty((batch * n_out * width,), dtype=src.dtype, device=src.device)
    N.check(N.lib().mlpk_index_gather(E.dtype_code(src.dtype), E.ptr(src), E.ptr(out), E.ptr(table), batch, n_out, n_in, width, kmax, E.stream()), "mlpk_index_gather")
    return out


class IndexMap(torch.autograd.Function):
    """x: (B * n_in * width / cols, cols) contiguous -> (B * n_out * width / out_cols, out_cols): the remap of an IndexTable; backward: its adjoint"""

    @staticmethod
    def forward(ctx, x, tab, B, out_cols):
        x = x.contiguous()
        assert x.numel() == B * tab.n_in * tab.width, (tuple(x.shape), B, tab.n_in, tab.width)
        ctx.tab, ctx.B, ctx.in_cols = tab, B, x.shape[1]
        with E.on_device(x):
            return _index_gather(x, tab.fwd, 1, B, tab.n_out, tab.n_in, tab.width).view(-1, out_cols)

    @staticmethod
    def backward(ctx, dy):
        tab = ctx.tab
        dy = dy.contiguous()
        with E.on_device(dy):
            return _index_gather(dy, tab.inv, tab.kmax, ctx.B, tab.n_in, tab.n_out, tab.width).view(-1, ctx.in_cols), None, None, None


class ConcatCols(torch.autograd.Function):
    """torch.cat over the channel axis of channel-last rows (sparse_mlp.py:71, ms_mlp.py:58-59): the parts copied into column slices of one buffer"""

    @staticmethod
    def forward(ctx, *parts):
        widths = [p_.shape[1] for p_ in parts]
        ctx.widths = widths
        out = torch.empty((parts[0].shape[0], sum(widths)), dtype=parts[0].dtype, device=parts[0].device)
        c0 = 0
        with E.on_device(out):
            for p_, wd in zip(parts, widths):
                _ew(0, _rows(p_), out=out[:, c0:c0 + wd])
                c0 += wd
        return out

    @staticmethod
    def backward(ctx, dy):
        outs, c0 = [], 0
        for wd in ctx.widths:
            outs.append(dy[:, c0:c0 + wd])
            c0 += wd
        return tuple(outs)


class AddPeriodic(torch.autograd.Function):
    """x[b, l, :] + t[l, :] (SwinMLP's absolute position embedding, swin_mlp.py:437-438); t: parameter (1, L, C)"""

    @staticmethod
    def forward(ctx, x, t, L):
        out = x.clone()
        ctx.meta = (tuple(t.shape), L)
        with E.on_device(x):
            E.add_periodic(out, out.stride(0), E.f32(t, x.device), out.shape[0], out.shape[1], L)
        return out

    @staticmethod
    def backward(ctx, dy):
        tshape, L = ctx.meta
        dy = dy.contiguous()
        Bn = dy.shape[0] // L
        with E.on_device(dy):
            dt = col_sum(dy.view(Bn, L * dy.shape[1]), Bn, L * dy.shape[1])
        return dy, dt.reshape(tshape), None


def drop_add(owner, t, z, rate, B, period):
    """t + drop_path(z) in train mode (the per-sample keep / (1 - p) scale of conv_mlp.py:27-34 on the draws of owner.drop_path_uniform);
    rate 0: the plain sum"""
    rate = float(rate)
    if rate == 0.0:
        return ScaleAdd.apply(t, z, None)
    keep = 1.0 - rate
    u = owner.drop_path_uniform(B, t.dtype, t.device)
    scale = (torch.floor(keep + u.reshape(B).float()) / keep).contiguous()
    return ScaleAdd.apply(t, RowScale.apply(z, scale, period), None)


def position_table(fn, n_in, width, device, cache, key):
    """IndexTable of a remap given as torch index arithmetic: fn(pos) is applied to a float64 tensor of source positions + 1 (0 = padding)
    and returns the destination layout; cached per key"""
    tab = cache.get(key)
    if tab is None:
        pos = torch.arange(1, n_in + 1, dtype=torch.float64)
        idx = fn(pos).reshape(-1).round().to(torch.int64) - 1
        tab = IndexTable(idx, n_in, width, device)
        cache[key] = tab
    return tab


def conv_window_table(H, W, C, k, stride, pad, device, cache):
    """IndexTable of the im2col of Conv2d(kernel k, stride, padding pad) on channel-last rows: destination rows (ho, wo), columns (kh, kw, c) --
    pixel granularity (width C), -1 where the window hangs over the border; its inverse table sums the overlapping windows (col2im)"""
    key = ("convwin", H, W, C, k, stride, pad)
    tab = cache.get(key)
    if tab is None:
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        yy = (torch.arange(Ho) * stride - pad).view(Ho, 1, 1, 1) + torch.arange(k).view(1, 1, k, 1)
        xx = (torch.arange(Wo) * stride - pad).view(1, Wo, 1, 1) + torch.arange(k).view(1, 1, 1, k)
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        idx = torch.where(ok, yy * W + xx, torch.full_like(yy * W + xx, -1))
        tab = IndexTable(idx, H * W, C, device)
        tab.out_hw = (Ho, Wo)
        cache[key] = tab
    return tab
