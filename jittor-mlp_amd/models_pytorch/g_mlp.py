"""gMLP, drop-in for the reference's models_pytorch/g_mlp.py.

Block (g_mlp.py:17-22, 32-39):  h = gelu(LN(x) P1^T + p1)  (width 2F);  u, v = halves of h;
  v' = Wsp . LN_F(v) + bsp   (contraction over tokens);   x <- x + (u * v') P2^T + p2.
Kernels: LN -> row-major copy; proj1 GEMM (bias+GELU epilogue) writes h once; the LayerNorm of the
v half is written token-transposed (reads the half in place through its row stride, no chunk()
copy); the spatial GEMM's epilogue adds the per-token bias, multiplies by u (read in place from h)
and stores through the per-image transpose; proj2 GEMM adds bias + residual.
"""
import contextlib
import os

import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import Block, Holder, adopt_blocks, embed_patches, finalize_stats, head_linear, layernorm_stats, standalone_space
from .utils.tools import check_sizes, pair


class SpatialGatingUnit(Holder):
    """g_mlp.py:10-22; spatial_proj bias initialised to 1.0 (:15), weight default-initialised."""

    def __init__(self, d_ffn, seq_len):
        super().__init__()
        self.norm = nn.LayerNorm(d_ffn)
        self.spatial_proj = nn.Conv1d(seq_len, seq_len, kernel_size=1)
        nn.init.constant_(self.spatial_proj.bias, 1.0)

    def forward(self, x):
        """g_mlp.py:17-22 on x (B, S, 2 F): u, v = chunk; v = spatial_proj(norm(v)); u * v.  The halves are read in place, the
        LayerNorm output is written token-transposed, the spatial product runs as the NT GEMM whose epilogue stores through the
        per-image transpose and multiplies by u (the kernels of the model's own block, unfused from proj1 / proj2)."""
        if x.dim() != 3 or x.shape[2] % 2:
            raise ValueError("expected (B, S, 2 * d_ffn) tokens")
        B, S, F2 = x.shape
        F = F2 // 2
        rows = B * S
        sp = E.round_up(S, 32)
        ws = standalone_space(x)
        with E.on_device(x):
            h = x.contiguous().view(rows, F2)
            v = h[:, F:]
            vt = ws.get("vt", (B * F, sp))
            mean, rstd = ws.get("m", (rows,), torch.float32), ws.get("r", (rows,), torch.float32)
            E.row_stats(v, rows, F, F2, mean, rstd)
            E.norm_apply(v, rows, F, F2, mean=mean, rstd=rstd, gamma=E.f32(self.norm.weight, x.device), beta=E.f32(self.norm.bias, x.device),
                         out_tt=vt, S=S, ld_tt=sp)
            out = torch.empty((rows, F), dtype=x.dtype, device=x.device)
            E.gemm(vt, E.pack_matrix(self.spatial_proj.weight, x.dtype, x.device, kpad=32), out, B * F, S, sp, ldc=F,
                   bias=E.f32(self.spatial_proj.bias, x.device), R=h, ldr=F2, res=N.RES_MUL, out_mode=N.OUT_TOKEN_T, t_rows=F, t_tokens=S)
        return out.view(B, S, F)


class gMLPBlock(Block):
    """g_mlp.py:24-39; channel_proj1 is 2*d_ffn wide (:28).  Callable on (B, S, C) tokens once it sits in a gMLP backbone."""

    def __init__(self, d_model, d_ffn, seq_len):
        super().__init__()
        self.norm = nn.LayerNorm(d_model)
        self.channel_proj1 = nn.Linear(d_model, d_ffn * 2)
        self.channel_proj2 = nn.Linear(d_ffn, d_model)
        self.sgu = SpatialGatingUnit(d_ffn, seq_len)


# tuning: how channel_proj1 delivers the SGU LayerNorm's statistics -- "split" (u | v halves, the v half with by-product statistics),
# "full" (one launch, statistics of all columns), "rowstats" (a statistics pass over the v half)
P1_MODE = os.environ.get("MLPK_GMLP_P1", "full")      # measured (profiles/r04_gmlp_p1_ab.txt): full 10.23 ms, rowstats 10.85, split 10.96-11.16


class gMLP(E.EngineModule):
    """Backbone on tokens (g_mlp.py:41-49)."""

    def __init__(self, d_model=256, d_ffn=1536, seq_len=256, depth=30):
        super().__init__()
        self.model = nn.Sequential(*[gMLPBlock(d_model, d_ffn, seq_len) for _ in range(depth)])
        self._dims = (seq_len, d_model, d_ffn, depth)
        adopt_blocks(self, self.model)

    def _pack_blocks(self, pk, dtype, device):
        for i, blk in enumerate(self.model):
            p = "b%d." % i
            # block LayerNorm folded into channel_proj1 (gamma -> weights, beta -> bias, stats in the epilogue)
            pk[p + "p1.w"], pk[p + "p1.b"], pk[p + "p1.csum"] = E.pack_ln_folded(
                blk.channel_proj1.weight, blk.channel_proj1.bias, blk.norm.weight, blk.norm.bias, dtype, device)
            w1 = blk.channel_proj1.weight
            if E.linear_gelu_enabled() and w1.shape[1] in (128, 192, 256, 384, 512) and w1.shape[0] % 32 == 0 and dtype in (torch.float16, torch.bfloat16):
                # round 4: channel_proj1 + GELU with its rows resident in registers (mlpk_linear_gelu), statistics planes included
                pk[p + "p1.rr"] = E.pack_linear_gelu(w1, blk.channel_proj1.bias, dtype, device, blk.norm.weight, blk.norm.bias)
            pk[p + "p2.w"] = E.pack_matrix(blk.channel_proj2.weight, dtype, device)
            pk[p + "p2.b"] = E.f32(blk.channel_proj2.bias, device)
            pk[p + "sgu.g"], pk[p + "sgu.b"] = E.f32(blk.sgu.norm.weight, device), E.f32(blk.sgu.norm.bias, device)
            pk[p + "sp.w"] = E.pack_matrix(blk.sgu.spatial_proj.weight, dtype, device, kpad=32)      # (S, S_pad)
            pk[p + "sp.b"] = E.f32(blk.sgu.spatial_proj.bias, device)
            S = blk.sgu.spatial_proj.weight.shape[0]
            if E.token_gemm_supported(dtype, S, E.round_up(S, 32)):
                pk[p + "sp.tg"] = E.pack_token_gemm(blk.sgu.spatial_proj.weight, blk.sgu.spatial_proj.bias, dtype, device)

    def _pack(self, dtype, device):
        pk = {}
        self._pack_blocks(pk, dtype, device)
        return pk

    def _run_blocks(self, ws, pk, x, B, only=None):
        S, C, F, depth = self._dims
        rows = B * S
        sp = E.round_up(S, 32)        # token K padding: whole half-slabs -> direct-to-LDS GEMM tiles
        nxt = None
        for i in (range(depth) if only is None else only):
            p = "b%d." % i
            # the block's LayerNorm (g_mlp.py:40) reads what the previous block's proj_out + residual GEMM wrote: its statistics
            # come out of that epilogue (mlpk.h row_part)
            mean, rstd = nxt if nxt is not None else layernorm_stats(ws, x, rows, C)
            h = ws.get("h", (rows, 2 * F))
            v = h[:, F:]                                        # second half, row stride 2F (g_mlp.py:18)
            tg = pk.get(p + "sp.tg")
            if tg is not None and E.token_gemm_ln_supported(v.dtype, S, F, 2 * F):
                # round 4: the SGU LayerNorm (g_mlp.py:19) is the spatial product's operand loader: ONE kernel reads v, normalises,
                # transposes through LDS, multiplies, gates with u and stores -- the token-transposed tensor and the normalise-and-
                # transpose pass (17 % of gMLP-S at 256 images) are gone.  Its row statistics come out of channel_proj1's epilogue:
                # the product is issued as its two halves (u | v, g_mlp.py:18), the v half with the by-product statistics of what it
                # stores (mlpk.h row_part; a statistics pass over v when the tile cannot deliver them)
                w1, b1, cs1 = pk[p + "p1.w"], pk[p + "p1.b"], pk[p + "p1.csum"]
                vst = None
                if P1_MODE == "split":
                    E.gemm(x, w1[:F], h, rows, F, C, ldc=2 * F, bias=b1[:F], act=N.ACT_GELU, ln=(mean, rstd, cs1[:F]), tag="gmlp_proj1")
                    got = E.gemm(x, w1[F:], v, rows, F, C, ldc=2 * F, bias=b1[F:], act=N.ACT_GELU, ln=(mean, rstd, cs1[F:]), tag="gmlp_proj1",
                                 part=(ws, "p1.part"))
                    vst = finalize_stats(ws, got, rows, F, tag="v")
                elif P1_MODE == "full":
                    # one launch with the statistics of all 2F columns; the planes of the v half are the second half of the buffer
                    if (p + "p1.rr") in pk and E.linear_gelu_supported(x.dtype, rows, C, 2 * F):
                        got = E.linear_gelu(x, rows, C, pk[p + "p1.rr"], h, ln=(mean, rstd), part=(ws, "p1.part"))
                    else:
                        got = E.gemm(x, w1, h, rows, 2 * F, C, bias=b1, act=N.ACT_GELU, ln=(mean, rstd, cs1), tag="gmlp_proj1", part=(ws, "p1.part"))
                    if got is not None:
                        vst = finalize_stats(ws, (got[0][got[1] // 2:], got[1] // 2), rows, F, tag="v")
                else:
                    E.gemm(x, w1, h, rows, 2 * F, C, bias=b1, act=N.ACT_GELU, ln=(mean, rstd, cs1), tag="gmlp_proj1")
                if vst is None:
                    vst = (ws.get("v.mean", (rows,), torch.float32), ws.get("v.rstd", (rows,), torch.float32))
                    E.row_stats(v, rows, F, 2 * F, vst[0], vst[1])
                vmean, vrstd = vst
                g = ws.get("gate", (rows, F))
                E.token_gemm_ln(v, 2 * F, B * F, S, vmean, vrstd, pk[p + "sgu.g"], pk[p + "sgu.b"], tg[0], tg[1], tg[2], g, F, F,
                                R=h, ldr=2 * F, res=N.RES_MUL)
                got = E.gemm(g, pk[p + "p2.w"], x, rows, C, F, bias=pk[p + "p2.b"], R=x, res=N.RES_ADD,
                             part=(ws, "p2.part") if only is None and i + 1 < depth else None)
                nxt = finalize_stats(ws, got, rows, C)
                continue
            E.gemm(x, pk[p + "p1.w"], h, rows, 2 * F, C, bias=pk[p + "p1.b"], act=N.ACT_GELU, ln=(mean, rstd, pk[p + "p1.csum"]),
                   tag="gmlp_proj1")
            vt = ws.get("vt", (B * F, sp))
            if E.layernorm_transpose_supported(v.dtype, F, 2 * F, sp):
                # the SGU LayerNorm (g_mlp.py:19) in one pass: statistics + affine + per-image transpose
                E.layernorm_transpose(v, B, S, F, pk[p + "sgu.g"], pk[p + "sgu.b"], vt, sp)
            else:
                vmean = ws.get("v.mean", (rows,), torch.float32)
                vrstd = ws.get("v.rstd", (rows,), torch.float32)
                E.row_stats(v, rows, F, 2 * F, vmean, vrstd)
                E.norm_apply(v, rows, F, 2 * F, mean=vmean, rstd=vrstd, gamma=pk[p + "sgu.g"], beta=pk[p + "sgu.b"],
                             out_tt=vt, S=S, ld_tt=sp)
            g = ws.get("gate", (rows, F))
            # out[b,t,f] = u[b,t,f] * (sum_s Wsp[t,s] v^[b,s,f] + bsp[t]);  u = h[:, :F] read in place
            tg = pk.get(p + "sp.tg")
            if tg is not None:
                E.token_gemm(vt, sp, B * F, S, tg[0], tg[1], tg[2], g, F, F, R=h, ldr=2 * F, res=N.RES_MUL)
            else:
                E.gemm(vt, pk[p + "sp.w"], g, B * F, S, sp, ldc=F, bias=pk[p + "sp.b"], R=h, ldr=2 * F, res=N.RES_MUL,
                       out_mode=N.OUT_TOKEN_T, t_rows=F, t_tokens=S)
            got = E.gemm(g, pk[p + "p2.w"], x, rows, C, F, bias=pk[p + "p2.b"], R=x, res=N.RES_ADD, part=(ws, "p2.part") if only is None and i + 1 < depth else None)
            nxt = finalize_stats(ws, got, rows, C)
        return x

    def forward(self, x, _only=None):
        E.require_gpu(x, "gMLP.forward")
        S, C, _, _ = self._dims
        if x.dim() != 3 or x.shape[1] != S or x.shape[2] != C:
            raise ValueError("expected tokens of shape (B, %d, %d)" % (S, C))
        B = x.shape[0]
        pk = self._get_pack(x.dtype, x.device)
        ws = self._get_space(B, x.dtype, x.device)
        buf = ws.get("x", (B * S, C))
        buf.copy_(x.reshape(B * S, C))
        self._run_blocks(ws, pk, buf, B, only=_only)
        return buf.reshape(B, S, C).clone()

    def _run_single(self, i, x):
        """block i alone, (B, S, C) -> (B, S, C): what `model.model[i](x)` computes in the reference (g_mlp.py:33-39)"""
        with E.on_device(x) if x.is_cuda else contextlib.nullcontext():
            return gMLP.forward(self, x, _only=[i])       # (the image-classification subclass overrides forward)


class gMLPForImageClassification(gMLP):
    """Same signature and defaults as the reference (g_mlp.py:52-62); note image_size defaults to 256."""

    def __init__(self, image_size=256, patch_size=16, in_channels=3, num_classes=1000, d_model=256, d_ffn=1536, depth=30):
        num_patches = check_sizes(image_size, patch_size)
        super().__init__(d_model, d_ffn, num_patches, depth)
        self.patcher = nn.Sequential(nn.Conv2d(in_channels, d_model, kernel_size=patch_size, stride=patch_size))
        self.mlp_head = nn.Sequential(nn.Linear(d_model, num_classes))
        self._patch = pair(patch_size)
        self._num_classes = num_classes

    def _pack(self, dtype, device):
        pk = {}
        self._pack_blocks(pk, dtype, device)
        pk["embed.w"] = E.pack_matrix(self.patcher[0].weight, dtype, device)
        pk["embed.b"] = E.f32(self.patcher[0].bias, device)
        pk["head.w"] = E.pack_matrix(self.mlp_head[0].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[0].bias, device)
        return pk

    _train_forward = True

    def _forward_train(self, x):
        """Train mode with autograd (round 6): g_mlp.py:10-39,64-82 as autograd.Functions of `..autograd`, forward and backward
        through the C ABI -- the projections and the spatial Conv1d(k=1) = mlpk_gemm_nt (+ the two GEMMs of their backward), both LayerNorms
        = mlpk_row_stats + mlpk_norm_apply / mlpk_layernorm_backward (the SGU's on the v half of h IN PLACE: a row stride of 2 d_ffn), the
        gate u * v and its two derivatives = mlpk_ew_cols, the token <-> channel rearranges = mlpk_transpose_batched.  Unfused on purpose."""
        from .. import autograd as AG
        E.require_gpu(x, "gMLPForImageClassification.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        S, C, F, _ = self._dims
        B, cin, H, W = x.shape
        ph, pw = self._patch
        if (H // ph) * (W // pw) != S:
            raise ValueError("input size gives %d patches, the model was built for %d" % ((H // ph) * (W // pw), S))
        conv = self.patcher[0]
        kp = E.round_up(cin * ph * pw, 4 if cd == torch.float32 else 8)
        with E.on_device(x):
            patches = torch.zeros((B * S, kp), dtype=cd, device=x.device)
            E.patchify(x.contiguous(), patches, B, cin, H, W, ph, pw, 0, kp)
        t = AG.Linear.apply(patches, conv.weight, conv.bias, None)
        for blk in self.model:
            n = AG.LayerNorm.apply(t, blk.norm.weight, blk.norm.bias, blk.norm.eps)
            h = AG.Gelu.apply(AG.Linear.apply(n, blk.channel_proj1.weight, blk.channel_proj1.bias, None))            # (B*S, 2F)
            u, v = h[:, :F], h[:, F:]                                                                                  # chunk(2, dim=-1): views
            vn = AG.LayerNorm.apply(v, blk.sgu.norm.weight, blk.sgu.norm.bias, blk.sgu.norm.eps)
            sp = blk.sgu.spatial_proj
            vs = AG.RowsToTokens.apply(AG.Linear.apply(AG.TokensToRows.apply(vn, B, S), sp.weight, sp.bias, None), B, S, F)
            t = AG.Linear.apply(AG.Mul.apply(u, vs), blk.channel_proj2.weight, blk.channel_proj2.bias, t)
        head = self.mlp_head[0]
        logits = AG.Linear.apply(AG.TokenMean.apply(t, B, S), head.weight, head.bias, None)
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
        tokens, hp, wp = embed_patches(ws, "embed", x, pk["embed.w"], pk["embed.b"], cd, self._patch,
                                       out=ws.get("x", (B * S, C)))
        if hp * wp != S:
            raise ValueError("input size gives %d patches, the model was built for %d" % (hp * wp, S))
        self._run_blocks(ws, pk, tokens, B)
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(tokens, B, S, C, C, pooled, C)               # no final norm (g_mlp.py:79)
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self._num_classes, x.dtype)
