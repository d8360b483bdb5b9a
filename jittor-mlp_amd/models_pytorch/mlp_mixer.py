"""MLP-Mixer, drop-in for the reference's models_pytorch/mlp_mixer.py (same constructors, same
state_dict keys), with forward() running on hand-written gfx950 kernels.

Per block (reference mlp_mixer.py:36-39, formulas SURVEY.md Appendix E):
  token mixing   x <- x + W2 . gelu(W1 . LN(x) + b1) + b2     contraction over tokens (Conv1d k=1)
  channel mixing x <- x + gelu(LN(x) W3^T + b3) W4^T + b4     contraction over channels (Linear)
Mapping to kernels:
  * LN statistics: mlpk_row_stats; the normalised copy is written TOKEN-TRANSPOSED (B, C, S_pad)
    by mlpk_norm_apply so the token contraction becomes the same K-contiguous NT GEMM as the
    channel one: rows = (image, channel) pairs, K = tokens;
  * token fc1: GEMM (B*C, S) x (4S, S)^T, epilogue bias+GELU -> hidden (B*C, 4S);
  * token fc2: GEMM (B*C, 4S) x (S, 4S)^T, epilogue bias + residual, stored through the per-image
    transpose straight back into x (B, S, C) -- no einops/permute copy ever exists;
  * channel fc1/fc2: GEMM with bias+GELU / bias+residual epilogues;
  * head: LN folded into the token mean (mlpk_pool_mean) then one small GEMM.
"""
import os
from functools import partial

import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import channel_mlp, embed_patches, finalize_stats, head_linear, layernorm_stats, pack_channel_mlp
from .utils.tools import check_sizes, pair


class _SubModule(E.EngineModule):
    """Base of the Mixer sub-modules that the reference lets a caller run on their own (`model.model[i]`, `model.model[i][0]`,
    `...fn`): their own packed weights and workspace, the same HIP kernels as the enclosing model's forward."""

    def _begin(self, x, width):
        E.require_gpu(x, type(self).__name__ + ".forward")
        E.dtype_code(x.dtype)
        if x.dim() < 2 or x.shape[-1] != width:
            raise ValueError("expected a (..., %d) tensor" % width)
        return self._get_pack(x.dtype, x.device)


class FeedForward(_SubModule):
    """dense -> GELU -> Dropout -> dense -> Dropout (mlp_mixer.py:16-27); parameters at net.0 / net.3.
    dense = nn.Linear: acts on the last dimension.  dense = Conv1d(kernel_size=1): acts on dimension 1 of a (B, S, C) tensor
    (the token-mixing MLP: the S tokens are the convolution's channels)."""

    def __init__(self, dim, hidden_dim, dropout=0., dense=nn.Linear):
        super().__init__()
        self.net = nn.Sequential(dense(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 dense(hidden_dim, dim), nn.Dropout(dropout))
        self.__dict__["_token"] = not isinstance(self.net[0], nn.Linear)
        self.__dict__["_dims"] = (dim, hidden_dim)

    def _pack(self, dtype, device):
        kpad = 32 if self._token else 8
        return {"fc1.w": E.pack_matrix(self.net[0].weight, dtype, device, kpad=kpad), "fc1.b": E.f32(self.net[0].bias, device),
                "fc2.w": E.pack_matrix(self.net[3].weight, dtype, device, kpad=kpad), "fc2.b": E.f32(self.net[3].bias, device)}

    def _token_products(self, ws, pk, xt, B, S, C, out, R=None):
        """out[b, t, c] (+= R) = sum_h W2[t, h] gelu(sum_s W1[h, s] xt[b*C + c, s] + b1[h]) + b2[t]; xt = (B*C, S_pad), zero padded."""
        th = self._dims[1]
        thp = pk["fc2.w"].shape[1]
        ht = ws.get("ff.h", (B * C, thp))
        E.gemm(xt, pk["fc1.w"], ht, B * C, th, xt.shape[1], bias=pk["fc1.b"], act=N.ACT_GELU)
        E.gemm(ht, pk["fc2.w"], out, B * C, S, thp, ldc=C, bias=pk["fc2.b"], R=R, ldr=C if R is not None else None,
               res=N.RES_ADD if R is not None else N.RES_NONE, out_mode=N.OUT_TOKEN_T, t_rows=C, t_tokens=S)

    def forward(self, x):
        dim, hidden = self._dims
        if not self._token:
            pk = self._begin(x, dim)
            rows = x.numel() // dim
            ws = self._get_space(rows, x.dtype, x.device)
            xb = ws.get("ff.x", (rows, pk["fc1.w"].shape[1]))            # K zero-padded to whole 16-byte chunks
            xb[:, :dim].copy_(x.reshape(rows, dim))
            h = ws.get("ff.hid", (rows, pk["fc2.w"].shape[1]))
            y = ws.get("ff.y", (rows, dim))
            E.gemm(xb, pk["fc1.w"], h, rows, hidden, xb.shape[1], bias=pk["fc1.b"], act=N.ACT_GELU)
            E.gemm(h, pk["fc2.w"], y, rows, dim, h.shape[1], bias=pk["fc2.b"])
            return y.reshape(x.shape).clone()
        if x.dim() != 3 or x.shape[1] != dim:
            raise ValueError("expected a (B, %d, C) tensor" % dim)
        B, S, C = x.shape
        pk = self._begin(x, C)
        ws = self._get_space(B * S * C, x.dtype, x.device)
        xb = ws.get("ff.x", (B * S, C))
        xb.copy_(x.reshape(B * S, C))
        xt = ws.get("ff.xt", (B * C, pk["fc1.w"].shape[1]))
        E.norm_apply(xb, B * S, C, C, out_tt=xt, S=S, ld_tt=xt.shape[1])    # plain per-image transpose
        y = ws.get("ff.y", (B * S, C))
        self._token_products(ws, pk, xt, B, S, C, y)
        return y.reshape(B, S, C).clone()


class PreNormResidual(_SubModule):
    """fn(LayerNorm(x)) + x  (mlp_mixer.py:6-13): holds `fn` and `norm`; callable on (B, S, C) tokens like the reference's."""

    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)

    def _pack(self, dtype, device):
        pk = {}
        if isinstance(self.fn, FeedForward) and not self.fn._token:
            pack_channel_mlp(pk, "", self.norm, self.fn.net[0], self.fn.net[3], dtype, device)
        else:
            pk["ln.g"], pk["ln.b"] = E.f32(self.norm.weight, device), E.f32(self.norm.bias, device)
        return pk

    def forward(self, x):
        if not isinstance(self.fn, FeedForward):
            raise NotImplementedError("PreNormResidual runs on its own around a FeedForward only; call the enclosing model")
        C = self.norm.normalized_shape[0]
        pk = self._begin(x, C)
        rows = x.numel() // C
        ws = self._get_space(rows, x.dtype, x.device)
        xb = ws.get("pn.x", (rows, C))
        xb.copy_(x.reshape(rows, C))
        if not self.fn._token:
            channel_mlp(ws, xb, rows, C, pk, "", self.fn._dims[1], eps=self.norm.eps)
            return xb.reshape(x.shape).clone()
        if x.dim() != 3 or x.shape[1] != self.fn._dims[0]:
            raise ValueError("expected a (B, %d, %d) tensor" % (self.fn._dims[0], C))
        B, S, _ = x.shape
        fpk = self.fn._get_pack(x.dtype, x.device)
        mean, rstd = layernorm_stats(ws, xb, rows, C, eps=self.norm.eps)
        xt = ws.get("pn.xt", (B * C, fpk["fc1.w"].shape[1]))
        E.norm_apply(xb, rows, C, C, mean=mean, rstd=rstd, gamma=pk["ln.g"], beta=pk["ln.b"], out_tt=xt, S=S, ld_tt=xt.shape[1])
        self.fn._token_products(ws, fpk, xt, B, S, C, xb, R=xb)
        return xb.reshape(B, S, C).clone()


class MLPMixer(E.EngineModule):
    """Backbone on tokens (B, S, C) (mlp_mixer.py:30-43)."""

    def __init__(self, num_patches, d_model, depth, expansion_factor=4, dropout=0.):
        super().__init__()
        # train(): autograd through the HIP path (MLPMixerForImageClassification._forward_train); Dropout is not implemented, and the bare
        # backbone has no train path: it warns like every inference-only module
        self.__dict__["_train_forward"] = dropout == 0. and hasattr(self, "_forward_train")
        chan_first, chan_last = partial(nn.Conv1d, kernel_size=1), nn.Linear
        self.model = nn.Sequential(*[
            nn.Sequential(
                PreNormResidual(d_model, FeedForward(num_patches, num_patches * expansion_factor, dropout, chan_first)),
                PreNormResidual(d_model, FeedForward(d_model, d_model * expansion_factor, dropout, chan_last)))
            for _ in range(depth)])
        self._dims = (num_patches, d_model, depth, expansion_factor)
        self.__dict__["fused_token_mlp"] = os.environ.get("MLPK_NO_FUSED_TOKEN", "0") != "1"

    # ---- widths that are not whole 16-byte chunks (round 6; the reference takes any d_model, mlp_mixer.py:46-54) ----
    # The activations carry round_up(C, 8) channels; the extra ones are EXACTLY zero everywhere: zero rows in every weight that produces
    # channels (patch embedding, channel fc2) with zero bias, zero LayerNorm gamma / beta, zero weight columns wherever channels are
    # contracted -- and a 0 / 1 channel mask as the per-row scale of the token-mixing product's epilogue (mlpk.h rscale: a padded channel
    # would otherwise pick up W2 gelu(b1) + b2, the same constant for every image).  LayerNorm statistics are taken over the C real
    # channels (mlpk_row_stats with length C on rows of pitch Cp).  Every product is the unfused GEMM path: a drop-in hole closed, not
    # a fast path.
    def _cpad(self):
        C = self._dims[1]
        return E.round_up(C, 8) if C % 8 else None

    def _pack_blocks_padded(self, pk, dtype, device, Cp):
        S, C, depth, ef = self._dims
        hid = C * ef
        hidp = E.round_up(hid, 8)

        def padv(v, n):
            out = torch.zeros((n,), dtype=torch.float32, device=device)
            out[:v.numel()] = v.detach().to(device=device, dtype=torch.float32).reshape(-1)
            return out
        pk["mask"] = padv(torch.ones(C), Cp)
        for i, blk in enumerate(self.model):
            tok, ch = blk[0], blk[1]
            p = "b%d." % i
            pk[p + "tok.ln.g"], pk[p + "tok.ln.b"] = padv(tok.norm.weight, Cp), padv(tok.norm.bias, Cp)
            pk[p + "tok.fc1.w"] = E.pack_matrix(tok.fn.net[0].weight, dtype, device, kpad=32)
            pk[p + "tok.fc1.b"] = E.f32(tok.fn.net[0].bias, device)
            pk[p + "tok.fc2.w"] = E.pack_matrix(tok.fn.net[3].weight, dtype, device, kpad=32)
            pk[p + "tok.fc2.b"] = E.f32(tok.fn.net[3].bias, device)
            w1 = torch.zeros((hidp, C), dtype=torch.float32, device=device)
            w1[:hid] = ch.fn.net[0].weight.detach().to(device=device, dtype=torch.float32)
            w2 = torch.zeros((Cp, hid), dtype=torch.float32, device=device)
            w2[:C] = ch.fn.net[3].weight.detach().to(device=device, dtype=torch.float32)
            pk[p + "ch.fc1.w"], pk[p + "ch.fc1.b"], pk[p + "ch.fc1.csum"] = E.pack_ln_folded(w1, padv(ch.fn.net[0].bias, hidp), ch.norm.weight, ch.norm.bias,
                                                                                          dtype, device)
            pk[p + "ch.fc2.w"] = E.pack_matrix(w2, dtype, device)
            pk[p + "ch.fc2.b"] = padv(ch.fn.net[3].bias, Cp)

    def _run_blocks_padded(self, ws, pk, x, B, Cp):
        """x: (B*S, Cp) channel-last tokens with zero padding channels, updated in place."""
        S, C, depth, ef = self._dims
        rows = B * S
        sp = E.round_up(S, 32)
        th = S * ef
        thp = E.round_up(th, 32)
        hidp = E.round_up(C * ef, 8)
        xt = ws.get("tok.xt", (B * Cp, sp))
        ht = ws.get("tok.h", (B * Cp, thp))
        for i in range(depth):
            p = "b%d." % i
            mean, rstd = layernorm_stats(ws, x, rows, C)
            E.norm_apply(x, rows, Cp, Cp, mean=mean, rstd=rstd, gamma=pk[p + "tok.ln.g"], beta=pk[p + "tok.ln.b"], out_tt=xt, S=S, ld_tt=sp)
            E.gemm(xt, pk[p + "tok.fc1.w"], ht, B * Cp, th, sp, bias=pk[p + "tok.fc1.b"], act=N.ACT_GELU, tag="token_fc1")
            E.gemm(ht, pk[p + "tok.fc2.w"], x, B * Cp, S, thp, ldc=Cp, bias=pk[p + "tok.fc2.b"], R=x, ldr=Cp, res=N.RES_ADD,
                   out_mode=N.OUT_TOKEN_T, t_rows=Cp, t_tokens=S, rscale=pk["mask"], rperiod=Cp, tag="token_fc2")
            channel_mlp(ws, x, rows, Cp, pk, p + "ch.", hidp, stats=layernorm_stats(ws, x, rows, C, tag="cm.ln"))
        return x

    # ---- weight packing: compute-dtype matrices (K zero-padded to 16 B), fp32 vectors ----
    def _pack_blocks(self, pk, dtype, device):
        if self._cpad():
            return self._pack_blocks_padded(pk, dtype, device, self._cpad())
        for i, blk in enumerate(self.model):
            tok, ch = blk[0], blk[1]
            p = "b%d." % i
            pk[p + "tok.ln.g"], pk[p + "tok.ln.b"] = E.f32(tok.norm.weight, device), E.f32(tok.norm.bias, device)
            pk[p + "tok.fc1.w"] = E.pack_matrix(tok.fn.net[0].weight, dtype, device, kpad=32)     # (4S, S_pad)
            pk[p + "tok.fc1.b"] = E.f32(tok.fn.net[0].bias, device)
            pk[p + "tok.fc2.w"] = E.pack_matrix(tok.fn.net[3].weight, dtype, device, kpad=32)     # (S, 4S_pad)
            pk[p + "tok.fc2.b"] = E.f32(tok.fn.net[3].bias, device)
            S = tok.fn.net[0].weight.shape[1]
            if self.fused_token_mlp and E.token_mlp_supported(dtype, S, E.round_up(S, 32), tok.fn.net[0].weight.shape[0]):
                pk[p + "tok.fused"] = E.pack_token_mlp(tok.fn.net[0].weight, tok.fn.net[0].bias, tok.fn.net[3].weight,
                                                       tok.fn.net[3].bias, dtype, device, E.round_up(S, 32),
                                                       t_rows=tok.norm.weight.shape[0])
            pack_channel_mlp(pk, p + "ch.", ch.norm, ch.fn.net[0], ch.fn.net[3], dtype, device)

    def _pack(self, dtype, device):
        pk = {}
        self._pack_blocks(pk, dtype, device)
        return pk

    def _run_blocks(self, ws, pk, x, B):
        """x: (B*S, C) channel-last tokens, updated in place."""
        S, C, depth, ef = self._dims
        rows = B * S
        sp = E.round_up(S, 32)        # token K padding: whole half-slabs -> direct-to-LDS GEMM tiles
        th = S * ef
        thp = E.round_up(th, 32)
        nxt = None                                             # statistics of x's rows out of the previous block's fc2 epilogue
        for i in range(depth):
            p = "b%d." % i
            fused = pk.get(p + "tok.fused")
            if (fused is not None and fused[5] in (2, 3) and fused[4] >= 2 and E.token_ln_fused() and C % 128 == 0
                    and (p + "ch.fc1.csum") in pk and E.epilogue_stats()):
                # the whole token-mixing PreNormResidual in ONE kernel: the LayerNorm + transpose is the token kernel's operand loader
                # (no xt tensor, x read once); its row statistics come out of the previous block's fc2 epilogue (first block: one pass)
                w1f, b1f, w2f, b2f, nch, lay = fused
                mean, rstd = nxt if nxt is not None else layernorm_stats(ws, x, rows, C, tag="tok.ln1")
                part = ws.get("tok.stats", (E.token_mlp_stat_planes(C, lay), rows, 2), torch.float32)
                E.token_mlp_ln(x, C, B * C, S, mean, rstd, pk[p + "tok.ln.g"], pk[p + "tok.ln.b"], w1f, b1f, w2f, b2f, nch, C, stats=part, layout=lay)
                stats = (ws.get("cm.ln.mean", (rows,), torch.float32), ws.get("cm.ln.rstd", (rows,), torch.float32))
                E.stats_finalize_planar(part, rows, C, stats[0], stats[1])
                got = channel_mlp(ws, x, rows, C, pk, p + "ch.", C * ef, stats=stats, part=(ws, "tok.lnpart") if i + 1 < depth else None)
                nxt = finalize_stats(ws, got, rows, C, tag="tok.ln1f") if i + 1 < depth else None
                continue
            nxt = None
            xt = ws.get("tok.xt", (B * C, sp))                 # LN(x) transposed per image, zero K-padding
            if E.layernorm_transpose_supported(x.dtype, C, C, sp):
                E.layernorm_transpose(x, B, S, C, pk[p + "tok.ln.g"], pk[p + "tok.ln.b"], xt, sp)     # statistics + apply, one pass
            else:
                mean, rstd = layernorm_stats(ws, x, rows, C)
                E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "tok.ln.g"], beta=pk[p + "tok.ln.b"],
                             out_tt=xt, S=S, ld_tt=sp)
            if fused is not None:
                # both token-mixing products + GELU + residual in one kernel; the hidden stays in LDS
                w1f, b1f, w2f, b2f, nch, lay = fused
                stats = None
                if C % 128 == 0 and (p + "ch.fc1.csum") in pk and E.epilogue_stats():
                    # the statistics of the channel LayerNorm come out of the token kernel's epilogue (no pass over x)
                    part = ws.get("tok.stats", (E.token_mlp_stat_planes(C, lay), rows, 2), torch.float32)
                    E.token_mlp(xt, sp, B * C, S, w1f, b1f, w2f, b2f, nch, x, C, C, stats=part, layout=lay)
                    stats = (ws.get("cm.ln.mean", (rows,), torch.float32), ws.get("cm.ln.rstd", (rows,), torch.float32))
                    E.stats_finalize_planar(part, rows, C, stats[0], stats[1])
                else:
                    E.token_mlp(xt, sp, B * C, S, w1f, b1f, w2f, b2f, nch, x, C, C, layout=lay)
                channel_mlp(ws, x, rows, C, pk, p + "ch.", C * ef, stats=stats)
                continue
            ht = ws.get("tok.h", (B * C, thp))
            E.gemm(xt, pk[p + "tok.fc1.w"], ht, B * C, th, sp, bias=pk[p + "tok.fc1.b"], act=N.ACT_GELU, tag="token_fc1")
            E.gemm(ht, pk[p + "tok.fc2.w"], x, B * C, S, thp, ldc=C, bias=pk[p + "tok.fc2.b"], R=x, ldr=C,
                   res=N.RES_ADD, out_mode=N.OUT_TOKEN_T, t_rows=C, t_tokens=S, tag="token_fc2")
            channel_mlp(ws, x, rows, C, pk, p + "ch.", C * ef)
        return x

    def forward(self, x):
        """tokens (B, S, C) -> (B, S, C), as the reference backbone (mlp_mixer.py:42-43)."""
        E.require_gpu(x, "MLPMixer.forward")
        S, C, _, _ = self._dims
        if x.dim() != 3 or x.shape[1] != S or x.shape[2] != C:
            raise ValueError("expected tokens of shape (B, %d, %d)" % (S, C))
        B = x.shape[0]
        pk = self._get_pack(x.dtype, x.device)
        ws = self._get_space(B, x.dtype, x.device)
        Cp = self._cpad()
        if Cp:
            buf = ws.get("x", (B * S, Cp))
            buf[:, :C].copy_(x.reshape(B * S, C))
            self._run_blocks_padded(ws, pk, buf, B, Cp)
            return buf[:, :C].reshape(B, S, C).clone()
        buf = ws.get("x", (B * S, C))
        buf.copy_(x.reshape(B * S, C))
        self._run_blocks(ws, pk, buf, B)
        return buf.reshape(B, S, C).clone()


class MLPMixerForImageClassification(MLPMixer):
    """Same signature and defaults as the reference (mlp_mixer.py:45-54)."""

    def __init__(self, in_channels=3, d_model=512, num_classes=1000, patch_size=16, image_size=224, depth=12,
                 expansion_factor=4):
        num_patches = check_sizes(image_size, patch_size)
        super().__init__(num_patches, d_model, depth, expansion_factor)
        self.patcher = nn.Sequential(nn.Conv2d(in_channels, d_model, kernel_size=patch_size, stride=patch_size))
        self.active = nn.LayerNorm(d_model)
        self.mlp_head = nn.Sequential(nn.Linear(d_model, num_classes))
        self._patch = pair(patch_size)
        self._num_classes = num_classes

    def _pack(self, dtype, device):
        pk = {}
        self._pack_blocks(pk, dtype, device)
        Cp = self._cpad()
        if Cp:
            C = self._dims[1]
            conv = self.patcher[0]
            wz = torch.zeros((Cp, conv.weight[0].numel()), dtype=torch.float32, device=device)
            wz[:C] = conv.weight.detach().to(device=device, dtype=torch.float32).reshape(C, -1)
            pk["embed.w"] = E.pack_matrix(wz, dtype, device)
            for key, v in (("embed.b", conv.bias), ("active.g", self.active.weight), ("active.b", self.active.bias)):
                pk[key] = torch.zeros((Cp,), dtype=torch.float32, device=device)
                pk[key][:C] = v.detach().to(device=device, dtype=torch.float32)
            pk["head.w"] = E.pack_matrix(self.mlp_head[0].weight, dtype, device)            # (classes, C -> Cp zero columns)
            pk["head.b"] = E.f32(self.mlp_head[0].bias, device)
            return pk
        pk["embed.w"] = E.pack_matrix(self.patcher[0].weight, dtype, device)
        pk["embed.b"] = E.f32(self.patcher[0].bias, device)
        pk["active.g"], pk["active.b"] = E.f32(self.active.weight, device), E.f32(self.active.bias, device)
        pk["head.w"] = E.pack_matrix(self.mlp_head[0].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[0].bias, device)
        return pk

    def _forward_train(self, x):
        """Train mode: the same mathematics as forward() with every step an autograd.Function of `..autograd` -- forward
        and backward both through the C ABI -- so that `loss.backward()` fills the parameters' .grad like the reference's autograd does
        (mlp_mixer.py:30-75; Dropout(p=0) is the identity there too).  Unfused on purpose: the pre-activations and the normalised tensors
        are what the backward needs.  The gradient w.r.t. the input image is not produced (no consumer)."""
        from .. import autograd as AG
        E.require_gpu(x, "MLPMixerForImageClassification.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        if x.requires_grad and not self.__dict__.get("_warned_input_grad"):
            import warnings
            warnings.warn("MLPMixerForImageClassification.train(): the gradient with respect to the input image is not produced "
                          "(parameter gradients only)", stacklevel=3)
            self.__dict__["_warned_input_grad"] = True
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        S, C, depth, _ = self._dims
        B, cin, H, W = x.shape
        ph, pw = self._patch
        if (H // ph) * (W // pw) != S:
            raise ValueError("input size gives %d patches, the model was built for %d" % ((H // ph) * (W // pw), S))
        conv = self.patcher[0]
        kp = E.round_up(cin * ph * pw, 4 if cd == torch.float32 else 8)
        with E.on_device(x):
            patches = torch.zeros((B * S, kp), dtype=cd, device=x.device)
            E.patchify(x.contiguous(), patches, B, cin, H, W, ph, pw, 0, kp)
        t = AG.Linear.apply(patches, conv.weight, conv.bias, None)                                   # (B*S, C)
        for blk in self.model:
            tok, ch = blk[0], blk[1]
            n1 = AG.LayerNorm.apply(t, tok.norm.weight, tok.norm.bias, tok.norm.eps)
            h = AG.Linear.apply(AG.TokensToRows.apply(n1, B, S), tok.fn.net[0].weight, tok.fn.net[0].bias, None)     # (B*C, 4S)
            y = AG.Linear.apply(AG.Gelu.apply(h), tok.fn.net[3].weight, tok.fn.net[3].bias, None)                     # (B*C, S)
            t = AG.RowsToTokensAdd.apply(y, t, B, S)
            n2 = AG.LayerNorm.apply(t, ch.norm.weight, ch.norm.bias, ch.norm.eps)
            h = AG.Linear.apply(n2, ch.fn.net[0].weight, ch.fn.net[0].bias, None)
            t = AG.Linear.apply(AG.Gelu.apply(h), ch.fn.net[3].weight, ch.fn.net[3].bias, t)
        nf = AG.LayerNorm.apply(t, self.active.weight, self.active.bias, self.active.eps)
        logits = AG.Linear.apply(AG.TokenMean.apply(nf, B, S), self.mlp_head[0].weight, self.mlp_head[0].bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        S, C, _, _ = self._dims
        B = x.shape[0]
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        x = x.contiguous()
        Cp = self._cpad()
        if Cp:
            # any d_model (mlp_mixer.py:46-54): zero padding channels up to whole 16-byte chunks, see _cpad
            tokens, hp, wp = embed_patches(ws, "embed", x, pk["embed.w"], pk["embed.b"], cd, self._patch, out=ws.get("x", (B * S, Cp)))
            if hp * wp != S:
                raise ValueError("input size gives %d patches, the model was built for %d" % (hp * wp, S))
            self._run_blocks_padded(ws, pk, tokens, B, Cp)
            mean, rstd = layernorm_stats(ws, tokens, B * S, C)
            pooled = ws.get("pooled", (B, Cp))
            E.pool_mean(tokens, B, S, Cp, Cp, pooled, Cp, mean=mean, rstd=rstd, gamma=pk["active.g"], beta=pk["active.b"])
            return head_linear(ws, pooled, B, Cp, pk["head.w"], pk["head.b"], self._num_classes, x.dtype)
        tokens, hp, wp = embed_patches(ws, "embed", x, pk["embed.w"], pk["embed.b"], cd, self._patch,
                                       out=ws.get("x", (B * S, C)))
        if hp * wp != S:
            raise ValueError("input size gives %d patches, the model was built for %d" % (hp * wp, S))
        self._run_blocks(ws, pk, tokens, B)
        mean, rstd = layernorm_stats(ws, tokens, B * S, C)
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(tokens, B, S, C, C, pooled, C, mean=mean, rstd=rstd, gamma=pk["active.g"], beta=pk["active.b"])
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self._num_classes, x.dtype)
