"""ConvMixer, drop-in for the reference's models_pytorch/conv_mixer.py (eval-mode forward:
BatchNorm uses its running statistics, as in the reference's own harness, compare.py:141).

Layer (conv_mixer.py:23-32): x <- x + BN(gelu(dwconv_kxk_same(x) + b));  x <- BN(gelu(W_pw x + b)).
GELU comes BEFORE BatchNorm, so BN cannot be folded into the conv weights; it is applied as a
per-channel scale/shift after GELU inside the producing kernel's epilogue.  Activations are
channel-last: the depthwise half is one stencil kernel (residual, GELU and BN fused), the
pointwise half is the NT GEMM with bias + GELU + scale/shift epilogue; the stem conv
(7x7, stride 7, padding 3) is the patch gather + the same GEMM epilogue.
"""
import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import Block, BlockSequential, adopt_blocks, Holder, head_linear


class Residual(Block):
    """fn(x) + x (conv_mixer.py:5-11).  Inside a ConvMixer (`model.blocks[i][0]`: depthwise convolution -> GELU -> BatchNorm, plus x) it runs
    on its own like the reference's, on (B, dim, H, W) -- the depthwise kernel has the residual built in; round 6."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


def _check_k(k):
    if not 1 <= k <= 13:
        raise NotImplementedError("depthwise kernel sizes 1 .. 13 are built; got %d" % k)


def _keff(k):
    """the odd size the kernels run: Conv2d(padding="same") pads (k - 1) // 2 before and k // 2 after, so an even k is the odd k + 1 with
    a zero tap row and column in FRONT (conv_mixer.py:26 takes any kernel_size; round 5: 1 .. 13, odd or even)"""
    return k if k % 2 else k + 1


def _dw_taps(weight, dim, k, device):
    """(dim, 1, k, k) depthwise weight -> tap-major float32 [keff * keff][dim] (include/mlpk.h mlpk_dwconv_nhwc)"""
    w = weight.detach().to(device=device, dtype=torch.float32).reshape(dim, k, k)
    ke = _keff(k)
    if ke != k:
        wz = torch.zeros((dim, ke, ke), dtype=torch.float32, device=device)
        wz[:, 1:, 1:] = w
        w = wz
    return w.reshape(dim, ke * ke).t().contiguous()


def _bn_affine(bn, device):
    scale = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return E.f32(scale.float(), device), E.f32(shift.float(), device)


class ConvMixer(E.EngineModule):
    """Same signature and defaults as the reference (conv_mixer.py:14)."""
    _train_forward = True                    # train(): BatchNorm on batch statistics + running-statistics update; with gradients enabled, autograd (_forward_autograd)

    def __init__(self, dim, depth, kernel_size=9, patch_size=7, n_classes=1000):
        super().__init__()
        self.embedding = nn.Sequential(
            nn.Conv2d(3, dim, kernel_size=patch_size, stride=patch_size, padding=patch_size // 2), nn.GELU(), nn.BatchNorm2d(dim))
        self.blocks = nn.Sequential(*[BlockSequential(
            Residual(nn.Sequential(nn.Conv2d(dim, dim, kernel_size, groups=dim, padding="same"), nn.GELU(), nn.BatchNorm2d(dim))),
            nn.Conv2d(dim, dim, kernel_size=1), nn.GELU(), nn.BatchNorm2d(dim)) for i in range(depth)])
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(dim, n_classes))
        self._cfg = (dim, depth, kernel_size, patch_size, n_classes)
        adopt_blocks(self, self.blocks)                            # lets `model.blocks[i](x)` run (common.BlockSequential)
        for i, blk in enumerate(self.blocks):
            blk[0].__dict__["_owner"] = (self, (i, "res"))         # ... and `model.blocks[i][0](x)`, the Residual (round 6)

    def _pack(self, dtype, device):
        dim, depth, k, patch, _ = self._cfg
        _check_k(k)
        pk = {}
        pk["embed.w"] = E.pack_matrix(self.embedding[0].weight, dtype, device, kpad=E.embed_kpad(dtype))
        pk["embed.b"] = E.f32(self.embedding[0].bias, device)
        pk["embed.s"], pk["embed.h"] = _bn_affine(self.embedding[2], device)
        for i, blk in enumerate(self.blocks):
            p = "b%d." % i
            dw, bn_a = blk[0].fn[0], blk[0].fn[2]
            pk[p + "dw.w"] = _dw_taps(dw.weight, dim, k, device)
            pk[p + "dw.b"] = E.f32(dw.bias, device)
            pk[p + "dw.s"], pk[p + "dw.h"] = _bn_affine(bn_a, device)
            pk[p + "pw.w"] = E.pack_matrix(blk[1].weight, dtype, device)
            pk[p + "pw.b"] = E.f32(blk[1].bias, device)
            pk[p + "pw.s"], pk[p + "pw.h"] = _bn_affine(blk[3], device)
        pk["head.w"] = E.pack_matrix(self.classifier[2].weight, dtype, device)
        pk["head.b"] = E.f32(self.classifier[2].bias, device)
        return pk

    def _run_single(self, i, x):
        """block i alone on (B, dim, H, W), as `model.blocks[i](x)` in the reference (conv_mixer.py:23-32)"""
        E.require_gpu(x, "ConvMixer block")
        E.dtype_code(x.dtype)
        residual_only = isinstance(i, tuple)                           # (i, "res"): blocks[i][0] alone (round 6)
        if residual_only:
            i = i[0]
        dim, _, k, _, _ = self._cfg
        if x.dim() != 4 or x.shape[1] != dim:
            raise ValueError("expected a (B, %d, H, W) tensor" % dim)
        B, _, H, W = x.shape
        rows = B * H * W
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws = self._get_space(("block", B, H, W, dim), x.dtype, x.device)
            cur = ws.get("blk.x", (rows, dim))
            cur.copy_(x.permute(0, 2, 3, 1).reshape(rows, dim))                        # channel-last rows, as forward() keeps them
            tmp = ws.get("blk.y", (rows, dim))
            p = "b%d." % i
            E.dwconv_nhwc(cur, tmp, B, H, W, dim, _keff(k), pk[p + "dw.w"], pk[p + "dw.b"], pk[p + "dw.s"], pk[p + "dw.h"])
            if residual_only:
                return tmp.reshape(B, H, W, dim).permute(0, 3, 1, 2).contiguous()
            E.gemm(tmp, pk[p + "pw.w"], cur, rows, dim, dim, bias=pk[p + "pw.b"], act=N.ACT_GELU, cscale=pk[p + "pw.s"], cshift=pk[p + "pw.h"])
            return cur.reshape(B, H, W, dim).permute(0, 3, 1, 2).contiguous()

    def _forward_train(self, x):
        """Train mode: BatchNorm2d normalises with the statistics of the BATCH and updates its running statistics
        (conv_mixer.py:20,28,31).  Every BatchNorm layer is two passes of the inference kernel: first with the identity affine, whose
        output gives the batch statistics (mlpk_col_sum; the depthwise kernel has its residual built in, so there the sums are taken of
        out - x), then with the scale / shift those statistics give -- the same epilogue arithmetic as eval mode.  Forward only: the
        result carries no grad_fn (the depthwise convolution's backward is not built)."""
        from .. import autograd as AG
        E.require_gpu(x, "ConvMixer.forward")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        self.__dict__["_in_shape"] = ("train",) + tuple(x.shape[1:])
        dim, depth, k, patch, n_classes = self._cfg
        _check_k(k)
        B, cin, H_in, W_in = x.shape
        dev = x.device
        pad = patch // 2
        H, W = (H_in + 2 * pad - patch) // patch + 1, (W_in + 2 * pad - patch) // patch + 1
        rows = B * H * W
        ones, zeros = torch.ones(dim, dtype=torch.float32, device=dev), torch.zeros(dim, dtype=torch.float32, device=dev)
        with torch.no_grad():
            ws = self._get_space(B, cd, dev)
            w = E.pack_matrix(self.embedding[0].weight, cd, dev)
            kp = w.shape[1]
            patches = ws.get("embed.patches", (rows, kp))
            E.patchify(x.contiguous(), patches, B, cin, H_in, W_in, patch, patch, pad, kp)
            cur, tmp = ws.get("x", (rows, dim)), ws.get("y", (rows, dim))

            def gemm_bn(a, conv, bn, out, kk):
                wp, b = E.pack_matrix(conv.weight, cd, dev), E.f32(conv.bias, dev)
                E.gemm(a, wp, out, rows, dim, kk, bias=b, act=N.ACT_GELU)
                s, h = AG.batchnorm_train_affine(bn, *AG.batch_stats(out, rows, dim), rows)
                E.gemm(a, wp, out, rows, dim, kk, bias=b, act=N.ACT_GELU, cscale=s, cshift=h)
            gemm_bn(patches, self.embedding[0], self.embedding[2], cur, kp)
            for blk in self.blocks:
                dw, bn_a = blk[0].fn[0], blk[0].fn[2]
                wd = _dw_taps(dw.weight, dim, k, dev)
                bd = E.f32(dw.bias, dev)
                E.dwconv_nhwc(cur, tmp, B, H, W, dim, _keff(k), wd, bd, ones, zeros)                # tmp = cur + gelu(dwconv(cur) + b)
                s, h = AG.batchnorm_train_affine(bn_a, *AG.batch_stats(tmp, rows, dim, sub=cur), rows)
                E.dwconv_nhwc(cur, tmp, B, H, W, dim, _keff(k), wd, bd, s, h)
                gemm_bn(tmp, blk[1], blk[3], cur, dim)
            pooled = ws.get("pooled", (B, dim))
            E.pool_mean(cur, B, H * W, dim, dim, pooled, dim)
            return head_linear(ws, pooled, B, dim, E.pack_matrix(self.classifier[2].weight, cd, dev), E.f32(self.classifier[2].bias, dev), n_classes, x.dtype)

    def _forward_autograd(self, x):
        """Train mode WITH autograd (round 6): conv_mixer.py:17-39 as autograd.Functions of `..autograd` whose forward and backward
        are C-ABI calls -- the patch / pointwise convolutions = mlpk_gemm_nt (+ the two GEMMs of their backward), the depthwise convolution =
        mlpk_dwconv_plain_nhwc (its adjoint for dX, mlpk_dwconv_wgrad_nhwc for the taps), GELU = mlpk_gelu_elementwise, BatchNorm2d on BATCH
        statistics with the full backward through them (mlpk_col_sum / mlpk_col_dot + one mlpk_ew_cols pass) and the running-statistics update,
        the Residual = mlpk_ew_cols.  No gradient w.r.t. the input image."""
        from .. import autograd as AG
        E.require_gpu(x, "ConvMixer.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        dim, depth, k, patch, n_classes = self._cfg
        _check_k(k)
        B, cin, H_in, W_in = x.shape
        pad = patch // 2
        H, W = (H_in + 2 * pad - patch) // patch + 1, (W_in + 2 * pad - patch) // patch + 1
        rows = B * H * W
        kp = E.round_up(cin * patch * patch, 4 if cd == torch.float32 else 8)
        with E.on_device(x):
            patches = torch.zeros((rows, kp), dtype=cd, device=x.device)
            E.patchify(x.contiguous(), patches, B, cin, H_in, W_in, patch, patch, pad, kp)

        def bn(a, mod):
            mean, var = AG.batch_stats(a.detach(), rows, dim)
            AG.batchnorm_train_affine(mod, mean, var, rows)                     # (the running statistics: momentum, unbiased variance)
            return AG.BatchNormTrain.apply(a, mod.weight, mod.bias, mean, var, mod.eps)

        e = self.embedding
        t = bn(AG.Gelu.apply(AG.Linear.apply(patches, e[0].weight, e[0].bias, None)), e[2])
        for blk in self.blocks:
            dw, bn_a = blk[0].fn[0], blk[0].fn[2]
            y = bn(AG.Gelu.apply(AG.DepthwiseConv.apply(t, dw.weight, dw.bias, B, H, W)), bn_a)
            t = AG.ScaleAdd.apply(y, t, None)                                   # Residual: fn(x) + x (conv_mixer.py:10)
            t = bn(AG.Gelu.apply(AG.Linear.apply(t, blk[1].weight, blk[1].bias, None)), blk[3])
        pooled = AG.TokenMean.apply(t, B, H * W)
        head = self.classifier[2]
        logits = AG.Linear.apply(pooled, head.weight, head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training:
            return self._forward_autograd(x) if torch.is_grad_enabled() else self._forward_train(x)
        cd = self._resolve(x)
        dim, depth, k, patch, n_classes = self._cfg
        B, cin, H_in, W_in = x.shape
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        x = x.contiguous()
        pad = patch // 2
        H, W = (H_in + 2 * pad - patch) // patch + 1, (W_in + 2 * pad - patch) // patch + 1
        rows = B * H * W
        kp = pk["embed.w"].shape[1]
        patches = ws.get("embed.patches", (rows, kp))
        E.patchify(x, patches, B, cin, H_in, W_in, patch, patch, pad, kp)
        cur = ws.get("x", (rows, dim))
        tmp = ws.get("y", (rows, dim))
        E.gemm(patches, pk["embed.w"], cur, rows, dim, kp, bias=pk["embed.b"], act=N.ACT_GELU, cscale=pk["embed.s"], cshift=pk["embed.h"])
        for i in range(depth):
            p = "b%d." % i
            E.dwconv_nhwc(cur, tmp, B, H, W, dim, _keff(k), pk[p + "dw.w"], pk[p + "dw.b"], pk[p + "dw.s"], pk[p + "dw.h"])
            E.gemm(tmp, pk[p + "pw.w"], cur, rows, dim, dim, bias=pk[p + "pw.b"], act=N.ACT_GELU, cscale=pk[p + "pw.s"],
                   cshift=pk[p + "pw.h"], tag="convmixer_pw")
        pooled = ws.get("pooled", (B, dim))
        E.pool_mean(cur, B, H * W, dim, dim, pooled, dim)
        return head_linear(ws, pooled, B, dim, pk["head.w"], pk["head.b"], n_classes, x.dtype)
