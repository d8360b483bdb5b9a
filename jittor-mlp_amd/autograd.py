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


def _packed(w, w2, cd, dev):
    """the packed (compute-dtype, K-padded) copy of parameter w, cached per parameter version like EngineModule._get_pack (a training
    step used to re-pack every weight in every forward).  An entry belongs to ONE live tensor object (weak reference): the allocator
    hands a freed parameter's address -- with version 0 again -- to the next model's parameter, so the address alone is not an identity."""
    key = (id(w), cd, str(dev))
    hit = _PACKS.get(key)
    if hit is None or hit[0]() is not w or hit[1] != (w.data_ptr(), w._version) or hit[2].shape[0] != w2.shape[0]:
        if len(_PACKS) > 512:
            _PACKS.clear()
        hit = (weakref.ref(w), (w.data_ptr(), w._version), E.pack_matrix(w2, cd, dev, kpad=_epc(cd)))
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
        assert x.dim() == 2 and x.shape[1] in (k, kp), (tuple(x.shape), k)
        with E.on_device(x):
            xp = _pad_cols(x, kp)
            wp = _packed(w, w2, cd, dev)
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
