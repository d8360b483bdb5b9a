"""Sparse-MLP, drop-in for the reference's models_pytorch/sparse_mlp.py (eval-mode forward:
BatchNorm uses its running statistics).

One entry of sMLPStage.model (sparse_mlp.py:84-104), on channel-last activations (B*H*W, C):
  * x <- x + dwconv3x3(BN(x)) + b                 one stencil kernel (mlpk_dwconv_affine_nhwc): the BatchNorm affine
                                                  is applied to the taps, padding stays zero, residual = the raw x;
  * x <- x + fuse(cat[proj_h(x^), proj_w(x^), x^]),  x^ = BN(x)  (sMLPBlock, :61-74):
      - one mlpk_norm_apply writes x^ row-major into the right half of a (rows, 2C) buffer AND its per-(b,h) token
        transpose ((B*H)*C, W); a second one writes the per-image transpose over H ((B)*(W*C), H) -- the two axial
        mixes are then the same K-contiguous NT GEMM as every other token mix on this path, stored straight back
        channel-last through the per-image transposed epilogue (no permute copy):
          x_w[(b,h), w', c] = sum_w Ww[w', w] x^[(b,h), w, c] + bw[w']      tokens = W, channels = C
          x_h[b, h', (w,c)] = sum_h Wh[h', h] x^[b, h, (w,c)] + bh[h']      tokens = H, channels = W*C
      - the 1x1 fuse conv over the concatenation is two accumulating GEMMs (K = C on x_h, K = 2C on [x_w | x^]),
        bias and the residual in the epilogues;
  * x <- x + FF(LN(x))                            LayerNorm folded into fc1 (common.channel_mlp).
PatchMerging (:17-52) = 2x2 space-to-depth gather + LayerNorm(4C) folded into the bias-free reduction GEMM.
Head (:151-157) = LayerNorm folded into the token mean, then one small GEMM.
"""
import os

import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import Block, BlockSequential, Holder, channel_mlp, finalize_stats, embed_patches, head_linear, layernorm_stats, pack_channel_mlp
from .conv_mixer import _bn_affine
from .utils import pair


class PreNormResidual(Block):
    """fn(norm(x)) + x (sparse_mlp.py:8-15).  Inside a SparseMLP the three of a block run on their own like the reference's (round 6):
    `layers[l].model[b][0]` (BatchNorm + depthwise 3 x 3) and `[b][1]` (BatchNorm + sMLPBlock) on (B, C, H, W), `[b][3]` (LayerNorm + MLP) on
    channel-last (B, H, W, C)."""

    def __init__(self, dim, fn, norm=nn.LayerNorm):
        super().__init__()
        self.fn = fn
        self.norm = norm(dim)


class PatchMerging(Block):
    """sparse_mlp.py:17-31.  Inside a SparseMLP it runs on its own like the reference's (:33-50): channel-last (B, H, W, C) -> (B, H/2, W/2, 2C);
    round 5."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)


class sMLPBlock(Block):
    """sparse_mlp.py:61-74.  Inside a SparseMLP it runs on its own like the reference's (:68-74): (B, C, H, W) -> fuse([proj_h x | proj_w x | x]),
    no normalisation, no residual; round 6."""

    def __init__(self, h=224, w=224, d_model=3):
        super().__init__()
        self.proj_h = nn.Linear(h, h)
        self.proj_w = nn.Linear(w, w)
        self.fuse = nn.Conv2d(3 * d_model, d_model, kernel_size=1)


class sMLPStage(Block):
    """sparse_mlp.py:76-104.  `patch_merge` exists in every stage (also where pooling is False), like the reference's,
    so that the state_dict keys match.  Inside a SparseMLP a stage runs on its own like the reference's (:106-110): (B, C, H, W) through its
    blocks and, where pooling, its PatchMerging -> (B, 2C, H/2, W/2); round 5."""

    def __init__(self, height, width, d_model, depth, expansion_factor=2, dropout=0., pooling=False):
        super().__init__()
        self.pooling = pooling
        self.patch_merge = nn.Sequential(nn.Identity(), PatchMerging((height, width), d_model), nn.Identity())
        self.model = nn.Sequential(*[BlockSequential(
            PreNormResidual(d_model, nn.Sequential(nn.Conv2d(d_model, d_model, kernel_size=3, padding=1, groups=d_model)),
                            norm=nn.BatchNorm2d),
            PreNormResidual(d_model, nn.Sequential(sMLPBlock(height, width, d_model)), norm=nn.BatchNorm2d),
            nn.Identity(),
            PreNormResidual(d_model, nn.Sequential(nn.Linear(d_model, d_model * expansion_factor), nn.GELU(), nn.Dropout(dropout),
                                                   nn.Linear(d_model * expansion_factor, d_model), nn.Dropout(dropout)),
                            norm=nn.LayerNorm),
            nn.Identity()) for _ in range(depth)])
        self.geom = (height, width, d_model, depth, expansion_factor)


class SparseMLP(E.EngineModule):
    """Same signature and defaults as the reference (sparse_mlp.py:106-117)."""

    def __init__(self, image_size=224, patch_size=4, in_channels=3, num_classes=1000, d_model=96, depth=[2, 10, 24, 2],
                 expansion_factor=2, patcher_norm=False):
        image_size = pair(image_size)
        patch_size = pair(patch_size)
        assert (image_size[0] % patch_size[0]) == 0, 'image must be divisible by patch size'
        assert (image_size[1] % patch_size[1]) == 0, 'image must be divisible by patch size'
        height = image_size[0] // patch_size[0]
        width = image_size[1] // patch_size[1]
        super().__init__()
        self.patcher = nn.Sequential(
            nn.Conv2d(in_channels, d_model, kernel_size=patch_size, stride=patch_size),
            nn.Identity() if (not patcher_norm) else nn.Sequential(nn.Identity(), nn.LayerNorm(d_model), nn.Identity()))
        self.layers = nn.ModuleList()
        for i_layer in range(len(depth)):
            self.layers.append(sMLPStage(height // (2 ** i_layer), width // (2 ** i_layer), d_model, depth[i_layer],
                                         expansion_factor=expansion_factor, pooling=((i_layer + 1) < len(depth))))
            if (i_layer + 1) < len(depth):
                d_model = d_model * 2
        self.mlp_head = nn.Sequential(nn.Identity(), nn.LayerNorm(d_model), nn.Identity(), nn.Linear(d_model, num_classes))
        self._cfg = (image_size, patch_size, in_channels, num_classes, patcher_norm)
        for li, stage in enumerate(self.layers):
            for bi, blk in enumerate(stage.model):
                blk.__dict__["_owner"] = (self, (li, bi))          # lets `model.layers[l].model[b](x)` run (common.BlockSequential)
                blk[0].__dict__["_owner"] = (self, (li, (bi, "pre0")))    # ... its PreNormResidual thirds and the sMLPBlock (round 6)
                blk[1].__dict__["_owner"] = (self, (li, (bi, "pre1")))
                blk[3].__dict__["_owner"] = (self, (li, (bi, "pre3")))
                blk[1].fn[0].__dict__["_owner"] = (self, (li, (bi, "smlp")))
            stage.__dict__["_owner"] = (self, (li, "layer"))       # ... `model.layers[l](x)`: the blocks, then the PatchMerging where pooling
            stage.patch_merge[1].__dict__["_owner"] = (self, (li, "merge"))

    def _pack(self, dtype, device):
        pk = {}
        pk["embed.w"] = E.pack_matrix(self.patcher[0].weight, dtype, device)
        pk["embed.b"] = E.f32(self.patcher[0].bias, device)
        if self._cfg[4]:
            pk["embed.g"], pk["embed.be"] = E.f32(self.patcher[1][1].weight, device), E.f32(self.patcher[1][1].bias, device)
        for li, stage in enumerate(self.layers):
            H, W, C, depth, ef = stage.geom
            for bi, blk in enumerate(stage.model):
                p = "l%d.b%d." % (li, bi)
                dw = blk[0].fn[0]
                pk[p + "dw.w"] = dw.weight.detach().reshape(C, 9).t().contiguous().to(device=device, dtype=torch.float32)
                pk[p + "dw.b"] = E.f32(dw.bias, device)
                pk[p + "dw.s"], pk[p + "dw.h"] = _bn_affine(blk[0].norm, device)
                sm = blk[1].fn[0]
                s, h = _bn_affine(blk[1].norm, device)
                pk[p + "bn.s"], pk[p + "bn.h"] = s, h
                pk[p + "bn.sw"], pk[p + "bn.hw"] = s.repeat(W).contiguous(), h.repeat(W).contiguous()   # per (w, c) "channel"
                pk[p + "ph.w"], pk[p + "ph.b"] = E.pack_matrix(sm.proj_h.weight, dtype, device), E.f32(sm.proj_h.bias, device)
                pk[p + "pw.w"], pk[p + "pw.b"] = E.pack_matrix(sm.proj_w.weight, dtype, device), E.f32(sm.proj_w.bias, device)
                if E.token_gemm_supported(dtype, max(sm.proj_h.weight.shape[0], sm.proj_w.weight.shape[0]), 32):
                    # the persistent single-product token kernel (gMLP / ResMLP) serves the two axial mixes as well
                    pk[p + "ph.tg"] = E.pack_token_gemm(sm.proj_h.weight, sm.proj_h.bias, dtype, device)
                    pk[p + "pw.tg"] = E.pack_token_gemm(sm.proj_w.weight, sm.proj_w.bias, dtype, device)
                if E.smlp_mix_supported(dtype, H, W, C):
                    # round 5: BatchNorm + both mixes + the concatenation in one kernel, the 3C -> C fuse as ONE GEMM on [x_h | x_w | x^]
                    pk[p + "mix.wh"], pk[p + "mix.bh"] = E.pack_smlp_mix(sm.proj_h.weight, sm.proj_h.bias, dtype, device)
                    pk[p + "mix.ww"], pk[p + "mix.bw"] = E.pack_smlp_mix(sm.proj_w.weight, sm.proj_w.bias, dtype, device)
                    pk[p + "fu.w3"] = E.pack_matrix(sm.fuse.weight.detach().reshape(C, 3 * C), dtype, device)
                wf = sm.fuse.weight.detach().reshape(C, 3 * C)
                pk[p + "fu.wh"] = E.pack_matrix(wf[:, :C], dtype, device)
                pk[p + "fu.wr"] = E.pack_matrix(wf[:, C:], dtype, device)                                # [x_w | x^] columns
                pk[p + "fu.b"] = E.f32(sm.fuse.bias, device)
                ff = blk[3]
                pack_channel_mlp(pk, p + "ff.", ff.norm, ff.fn[0], ff.fn[3], dtype, device)
            if stage.pooling:
                pm = stage.patch_merge[1]
                p = "l%d.merge." % li
                pk[p + "w"], pk[p + "b"], pk[p + "csum"] = E.pack_ln_folded(pm.reduction.weight, None, pm.norm.weight, pm.norm.bias,
                                                                             dtype, device)
                if dtype != torch.float32:
                    pk[p + "wc"] = E.merge_taps(pk[p + "w"], pm.norm.weight.shape[0] // 4)       # round 6: the reduction as an implicit-convolution product
        pk["head.g"], pk["head.be"] = E.f32(self.mlp_head[1].weight, device), E.f32(self.mlp_head[1].bias, device)
        pk["head.w"] = E.pack_matrix(self.mlp_head[3].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[3].bias, device)
        return pk

    def _part(self, ws, pk, li, bi, stage, cur, tmp, B, part):
        """One inner module of block `layers[li].model[bi]` on its own (round 6): "pre0" = x + dwconv(BN x), "pre1" = x + sMLPBlock(BN x),
        "smlp" = sMLPBlock(x) (identity in place of the BatchNorm affine, no residual), "pre3" = x + MLP(LN x).  The unfused kernels of the
        block, one sublayer at a time; returns the result buffer."""
        H, W, C, depth, ef = stage.geom
        rows = B * H * W
        p = "l%d.b%d." % (li, bi)
        if part == "pre0":
            E.dwconv_affine_nhwc(cur, tmp, B, H, W, C, 3, pk[p + "dw.w"], pk[p + "dw.b"], pk[p + "dw.s"], pk[p + "dw.h"])
            return tmp
        if part == "pre3":
            channel_mlp(ws, cur, rows, C, pk, p + "ff.", C * ef, tag="l%d.cm" % li)
            return cur
        plain = part == "smlp"
        one = ws.get("part.one", (max(C, W * C),), torch.float32, fill=1.0)
        zero = ws.get("part.zero", (max(C, W * C),), torch.float32)
        bs, bh = (one[:C], zero[:C]) if plain else (pk[p + "bn.s"], pk[p + "bn.h"])
        if (p + "mix.wh") in pk:
            cat3 = ws.get("l%d.cat3" % li, (rows, 3 * C))
            E.smlp_mix(cur, C, B, H, W, C, bs, bh, pk[p + "mix.wh"], pk[p + "mix.bh"], pk[p + "mix.ww"], pk[p + "mix.bw"], cat3, 3 * C)
            E.gemm(cat3, pk[p + "fu.w3"], tmp, rows, C, 3 * C, bias=pk[p + "fu.b"], R=None if plain else cur, res=N.RES_NONE if plain else N.RES_ADD,
                   tag="smlp_fuse")
            return tmp
        tgk = ("l%d.b0.pw.tg" % li) in pk
        hp, wp = E.round_up(H, 32 if tgk else 8), E.round_up(W, 32 if tgk else 8)
        xh = ws.get("l%d.xh" % li, (rows, C))
        cat = ws.get("l%d.cat" % li, (rows, 2 * C))
        xt_w = ws.get("l%d.xtw" % li, (B * H * C, wp))
        xt_h = ws.get("l%d.xth" % li, (B * W * C, hp))
        E.norm_apply(cur, rows, C, C, gamma=bs, beta=bh, out_rm=cat[:, C:], ld_rm=2 * C, out_tt=xt_w, S=W, ld_tt=wp)
        E.norm_apply(cur, B * H, W * C, W * C, gamma=one[:W * C] if plain else pk[p + "bn.sw"], beta=zero[:W * C] if plain else pk[p + "bn.hw"],
                     out_tt=xt_h, S=H, ld_tt=hp)
        if tgk:
            tw, th = pk[p + "pw.tg"], pk[p + "ph.tg"]
            E.token_gemm(xt_w, wp, B * H * C, W, tw[0], tw[1], tw[2], cat, 2 * C, C)
            E.token_gemm(xt_h, hp, B * W * C, H, th[0], th[1], th[2], xh, W * C, W * C)
        else:
            E.gemm(xt_w, pk[p + "pw.w"], cat, B * H * C, W, wp, ldc=2 * C, bias=pk[p + "pw.b"], out_mode=N.OUT_TOKEN_T, t_rows=C, t_tokens=W, tag="smlp_w")
            E.gemm(xt_h, pk[p + "ph.w"], xh, B * W * C, H, hp, ldc=W * C, bias=pk[p + "ph.b"], out_mode=N.OUT_TOKEN_T, t_rows=W * C, t_tokens=H,
                   tag="smlp_h")
        E.gemm(xh, pk[p + "fu.wh"], tmp, rows, C, C, bias=pk[p + "fu.b"], R=None if plain else cur, res=N.RES_NONE if plain else N.RES_ADD, tag="smlp_fuse")
        E.gemm(cat, pk[p + "fu.wr"], tmp, rows, C, 2 * C, R=tmp, res=N.RES_ADD, tag="smlp_fuse")
        return tmp

    def _block(self, ws, pk, li, bi, stage, cur, tmp, B):
        """Block `layers[li].model[bi]` on channel-last rows `cur` (B*H*W, C) with `tmp` as the other half of the ping-pong pair (a stencil
        cannot run in place); returns (result buffer, the other one)."""
        H, W, C, depth, ef = stage.geom
        rows = B * H * W
        p = "l%d.b%d." % (li, bi)
        if (p + "mix.wh") in pk:
            # round 5 (maps up to 32 x 32): x + dwconv, then ONE kernel for BN + proj_h + proj_w + cat (mlpk_smlp_mix: the tile of an image's 32
            # channels is transposed inside LDS instead of through two transposed tensors in HBM), then the fuse as one K = 3C GEMM
            cat3 = ws.get("l%d.cat3" % li, (rows, 3 * C))
            if E.smlp_mix_dw_supported(cur.dtype, H, W, C):
                # ... and on maps up to 15 x 15 the depthwise sublayer in the same kernel: its output goes to `tmp` (the fuse GEMM's residual)
                E.smlp_mix_dw(cur, C, B, H, W, C, pk[p + "dw.w"], pk[p + "dw.b"], pk[p + "dw.s"], pk[p + "dw.h"], tmp, C,
                              pk[p + "bn.s"], pk[p + "bn.h"], pk[p + "mix.wh"], pk[p + "mix.bh"], pk[p + "mix.ww"], pk[p + "mix.bw"], cat3, 3 * C)
                cur, tmp = tmp, cur
            else:
                E.dwconv_affine_nhwc(cur, tmp, B, H, W, C, 3, pk[p + "dw.w"], pk[p + "dw.b"], pk[p + "dw.s"], pk[p + "dw.h"])
                cur, tmp = tmp, cur
                E.smlp_mix(cur, C, B, H, W, C, pk[p + "bn.s"], pk[p + "bn.h"], pk[p + "mix.wh"], pk[p + "mix.bh"], pk[p + "mix.ww"], pk[p + "mix.bw"],
                           cat3, 3 * C)
            got = E.gemm(cat3, pk[p + "fu.w3"], cur, rows, C, 3 * C, bias=pk[p + "fu.b"], R=cur, res=N.RES_ADD, tag="smlp_fuse",
                         part=(ws, "l%d.fu.part" % li))
            channel_mlp(ws, cur, rows, C, pk, p + "ff.", C * ef, tag="l%d.cm" % li, stats=finalize_stats(ws, got, rows, C, tag="l%d.cm.ln" % li))
            return cur, tmp
        tgk = ("l%d.b0.pw.tg" % li) in pk                            # token kernel: K padded to whole 64-byte slabs
        hp, wp = E.round_up(H, 32 if tgk else 8), E.round_up(W, 32 if tgk else 8)
        xh = ws.get("l%d.xh" % li, (rows, C))
        cat = ws.get("l%d.cat" % li, (rows, 2 * C))                  # [x_w | x^]
        xt_w = ws.get("l%d.xtw" % li, (B * H * C, wp))
        xt_h = ws.get("l%d.xth" % li, (B * W * C, hp))
        p = "l%d.b%d." % (li, bi)
        # x + dwconv3x3(BN(x)) + b    (ping-pong cur <-> tmp: a stencil cannot run in place)
        E.dwconv_affine_nhwc(cur, tmp, B, H, W, C, 3, pk[p + "dw.w"], pk[p + "dw.b"], pk[p + "dw.s"], pk[p + "dw.h"])
        cur, tmp = tmp, cur
        # x^ = BN(x): row-major into cat[:, C:], transposed over W per (b, h); transposed over H per image
        E.norm_apply(cur, rows, C, C, gamma=pk[p + "bn.s"], beta=pk[p + "bn.h"], out_rm=cat[:, C:], ld_rm=2 * C,
                     out_tt=xt_w, S=W, ld_tt=wp)
        E.norm_apply(cur, B * H, W * C, W * C, gamma=pk[p + "bn.sw"], beta=pk[p + "bn.hw"], out_tt=xt_h, S=H, ld_tt=hp)
        if tgk:
            tw, th = pk[p + "pw.tg"], pk[p + "ph.tg"]
            E.token_gemm(xt_w, wp, B * H * C, W, tw[0], tw[1], tw[2], cat, 2 * C, C)
            E.token_gemm(xt_h, hp, B * W * C, H, th[0], th[1], th[2], xh, W * C, W * C)
        else:
            E.gemm(xt_w, pk[p + "pw.w"], cat, B * H * C, W, wp, ldc=2 * C, bias=pk[p + "pw.b"], out_mode=N.OUT_TOKEN_T,
                   t_rows=C, t_tokens=W, tag="smlp_w")
            E.gemm(xt_h, pk[p + "ph.w"], xh, B * W * C, H, hp, ldc=W * C, bias=pk[p + "ph.b"], out_mode=N.OUT_TOKEN_T,
                   t_rows=W * C, t_tokens=H, tag="smlp_h")
        E.gemm(xh, pk[p + "fu.wh"], cur, rows, C, C, bias=pk[p + "fu.b"], R=cur, res=N.RES_ADD, tag="smlp_fuse")
        # the LayerNorm of the channel MLP reads what this GEMM writes: its statistics come out of the epilogue (mlpk.h row_part)
        got = E.gemm(cat, pk[p + "fu.wr"], cur, rows, C, 2 * C, R=cur, res=N.RES_ADD, tag="smlp_fuse", part=(ws, "l%d.fu.part" % li))
        channel_mlp(ws, cur, rows, C, pk, p + "ff.", C * ef, tag="l%d.cm" % li, stats=finalize_stats(ws, got, rows, C, tag="l%d.cm.ln" % li))
        return cur, tmp

    def _merge(self, ws, pk, li, cur, B, H, W, C):
        """PatchMerging (sparse_mlp.py:33-50): 2 x 2 gather + LayerNorm folded into the bias-free reduction GEMM"""
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."                      # sparse_mlp.py:38
        p = "l%d.merge." % li
        H2, W2 = H // 2, W // 2
        nxt = ws.get("l%d.x" % (li + 1), (B * H2 * W2, 2 * C))
        # (opt-in, MLPK_MERGE_IMPLICIT=1: measured neutral to -0.9 % here -- the statistics pass over the windows costs what the gather saved;
        #  AS-MLP, whose GroupNorm statistics are already there, uses the same product by default: profiles/r06_conv_gemm_ab.txt)
        if (os.environ.get("MLPK_MERGE_IMPLICIT") == "1" and (p + "wc") in pk and pk[p + "w"].shape[1] == 4 * C
                and E.conv_gemm_nhwc_supported(cur.dtype, C, 2, 2, 2, 0)):
            # round 6: no merged tensor (mlpk_merge2x2_row_stats + mlpk_conv_gemm_nhwc, the weight's column blocks in its tap order)
            mean = ws.get("l%d.merge.ln.mean" % li, (B * H2 * W2,), torch.float32)
            rstd = ws.get("l%d.merge.ln.rstd" % li, (B * H2 * W2,), torch.float32)
            E.merge2x2_row_stats(cur, B, H, W, C, mean, rstd, eps=self.layers[li].patch_merge[1].norm.eps)
            E.conv_gemm_nhwc(cur, pk[p + "wc"], nxt, B, H, W, C, 2, 2, 2, 0, bias=pk[p + "b"], ln=(mean, rstd, pk[p + "csum"]), tag="smlp_merge")
            return nxt
        merged = ws.get("l%d.merged" % li, (B * H2 * W2, 4 * C))
        E.patchify(cur, merged, B, C, H, W, 2, 2, 0, 4 * C, layout=N.LAYOUT_NHWC, px_stride=C, order=1)
        mean, rstd = layernorm_stats(ws, merged, B * H2 * W2, 4 * C, tag="l%d.merge.ln" % li)
        E.gemm(merged, pk[p + "w"], nxt, B * H2 * W2, 2 * C, 4 * C, bias=pk[p + "b"], ln=(mean, rstd, pk[p + "csum"]), tag="smlp_merge")
        return nxt

    def _run_single(self, key, x):
        """An inner module alone, as calling it does in the reference: block `layers[l].model[b]` and stage `layers[l]` on (B, C, H, W) with the
        stage's H x W (sparse_mlp.py:84-110), `layers[l].patch_merge[1]` (PatchMerging, :33-50) on channel-last (B, H, W, C)"""
        li, bi = key
        E.require_gpu(x, "SparseMLP inner module")
        E.dtype_code(x.dtype)
        stage = self.layers[li]
        H, W, C = stage.geom[:3]
        if bi == "merge":
            if (("l%d.merge.w" % li) not in self._get_pack(x.dtype, x.device)):
                raise NotImplementedError("this stage does not pool: its PatchMerging only holds parameters, like the reference's unused one")
            if x.dim() != 4 or tuple(x.shape[1:]) != (H, W, C):
                raise ValueError("expected a channel-last (B, %d, %d, %d) tensor" % (H, W, C))
            B = x.shape[0]
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)
                cur = ws.get("blk.x", (B * H * W, C))
                cur.copy_(x.reshape(B * H * W, C))
                return self._merge(ws, pk, li, cur, B, H, W, C).reshape(B, H // 2, W // 2, 2 * C).clone()
        if bi == "layer":
            if x.dim() != 4 or tuple(x.shape[1:]) != (C, H, W):
                raise ValueError("expected a (B, %d, %d, %d) tensor" % (C, H, W))
            B = x.shape[0]
            rows = B * H * W
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)
                cur = ws.get("blk.x", (rows, C))
                cur.copy_(x.permute(0, 2, 3, 1).reshape(rows, C))
                tmp = ws.get("blk.tmp", (rows, C))
                for b_i in range(stage.geom[3]):
                    cur, tmp = self._block(ws, pk, li, b_i, stage, cur, tmp, B)
                if stage.pooling:
                    cur = self._merge(ws, pk, li, cur, B, H, W, C)
                    H, W, C = H // 2, W // 2, 2 * C
                return cur.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        if isinstance(bi, tuple) and bi[1] == "pre3":             # the channel MLP third: channel-last in and out (sparse_mlp.py:93-101)
            if x.dim() != 4 or tuple(x.shape[1:]) != (H, W, C):
                raise ValueError("expected a channel-last (B, %d, %d, %d) tensor" % (H, W, C))
            B = x.shape[0]
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)
                cur = ws.get("blk.x", (B * H * W, C))
                cur.copy_(x.reshape(B * H * W, C))
                return self._part(ws, pk, li, bi[0], stage, cur, None, B, "pre3").reshape(B, H, W, C).clone()
        if x.dim() != 4 or tuple(x.shape[1:]) != (C, H, W):
            raise ValueError("expected a (B, %d, %d, %d) tensor" % (C, H, W))
        B = x.shape[0]
        rows = B * H * W
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)     # (C: blocks of different stages can meet at one map size)
            cur = ws.get("blk.x", (rows, C))
            cur.copy_(x.permute(0, 2, 3, 1).reshape(rows, C))                          # channel-last rows, as the stages keep them
            if isinstance(bi, tuple):                                  # an inner module of block bi[0] (round 6)
                out = self._part(ws, pk, li, bi[0], stage, cur, ws.get("blk.tmp", (rows, C)), B, bi[1])
                return out.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
            cur, _ = self._block(ws, pk, li, bi, stage, cur, ws.get("blk.tmp", (rows, C)), B)
            return cur.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()

    _train_forward = True

    def _forward_train(self, x):
        """Train mode with autograd (round 6): sparse_mlp.py:6-175 as autograd.Functions of `..autograd`, forward and backward through the
        C ABI.  BatchNorm2d runs on BATCH statistics with the full backward through them and updates its running statistics (the two
        PreNormResidual(norm = BatchNorm2d) sublayers); proj_h / proj_w contract over H resp. W: the ViP rearranges with one-channel segments
        (mlpk_norm_apply out_ph / out_pw, mlpk_vip_unpermute) around mlpk_gemm_nt; the concatenation in front of `fuse` = column slices of one
        buffer; the depthwise 3 x 3 = mlpk_dwconv_plain_nhwc (+ adjoint, + mlpk_dwconv_wgrad_nhwc); PatchMerging = mlpk_merge2x2_nhwc."""
        from .. import autograd as AG
        E.require_gpu(x, "SparseMLP.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        image_size, patch, cin, num_classes, patcher_norm = self._cfg
        B, _, H_in, W_in = x.shape
        ph, pw = patch
        H, W = H_in // ph, W_in // pw
        if (H, W) != self.layers[0].geom[:2]:
            raise ValueError("input size gives a %dx%d grid, the model was built for %dx%d" % ((H, W) + self.layers[0].geom[:2]))
        kp = E.round_up(cin * ph * pw, 4 if cd == torch.float32 else 8)
        with E.on_device(x):
            patches = torch.zeros((B * H * W, kp), dtype=cd, device=x.device)
            E.patchify(x.contiguous(), patches, B, cin, H_in, W_in, ph, pw, 0, kp)

        def ln(t, norm):
            return AG.LayerNorm.apply(t, norm.weight, norm.bias, norm.eps)

        def bn(t, mod):
            rows, dim = t.shape
            mean, var = AG.batch_stats(t.detach(), rows, dim)
            AG.batchnorm_train_affine(mod, mean, var, rows)
            return AG.BatchNormTrain.apply(t, mod.weight, mod.bias, mean, var, mod.eps)

        conv = self.patcher[0]
        t = AG.Linear.apply(patches, conv.weight, conv.bias, None)
        if patcher_norm:
            t = ln(t, self.patcher[1][1])
        for stage in self.layers:
            C = stage.geom[2]
            for blk in stage.model:
                dw_pre, mix_pre, mlp = blk[0], blk[1], blk[3]
                dw = dw_pre.fn[0]
                t = AG.ScaleAdd.apply(AG.DepthwiseConv.apply(bn(t, dw_pre.norm), dw.weight, dw.bias, B, H, W), t, None)
                sm = mix_pre.fn[0]
                n = bn(t, mix_pre.norm)
                x_h = AG.VipUnpermute.apply(AG.Linear.apply(AG.VipPermute.apply(n, B, H, W, 1, 0), sm.proj_h.weight, sm.proj_h.bias, None), B, H, W, C, 1, 0)
                x_w = AG.VipUnpermute.apply(AG.Linear.apply(AG.VipPermute.apply(n, B, H, W, 1, 1), sm.proj_w.weight, sm.proj_w.bias, None), B, H, W, C, 1, 1)
                t = AG.Linear.apply(AG.ConcatCols.apply(x_h, x_w, n), sm.fuse.weight, sm.fuse.bias, t)
                fc1, fc2 = mlp.fn[0], mlp.fn[3]
                t = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(ln(t, mlp.norm), fc1.weight, fc1.bias, None)), fc2.weight, fc2.bias, t)
            if stage.pooling:
                pm = stage.patch_merge[1]
                t = AG.Linear.apply(ln(AG.Merge2x2.apply(t, B, H, W), pm.norm), pm.reduction.weight, None, None)
                H, W = H // 2, W // 2
        head_ln, head = self.mlp_head[1], self.mlp_head[3]
        logits = AG.Linear.apply(AG.TokenMean.apply(ln(t, head_ln), B, H * W), head.weight, head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        image_size, patch, cin, num_classes, patcher_norm = self._cfg
        B = x.shape[0]
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        x = x.contiguous()
        C0 = self.layers[0].geom[2]
        cur, H, W = embed_patches(ws, "embed", x, pk["embed.w"], pk["embed.b"], cd, patch, out=ws.get("l0.x", (B * self.layers[0].geom[0] * self.layers[0].geom[1], C0)))
        if (H, W) != self.layers[0].geom[:2]:
            raise ValueError("input size gives a %dx%d grid, the model was built for %dx%d" % ((H, W) + self.layers[0].geom[:2]))
        if patcher_norm:
            mean, rstd = layernorm_stats(ws, cur, B * H * W, C0, tag="embed.ln")
            E.norm_apply(cur, B * H * W, C0, C0, mean=mean, rstd=rstd, gamma=pk["embed.g"], beta=pk["embed.be"], out_rm=cur, ld_rm=C0)
        for li, stage in enumerate(self.layers):
            H, W, C, depth, ef = stage.geom
            rows = B * H * W
            tmp = ws.get("l%d.tmp" % li, (rows, C))
            for bi in range(depth):
                cur, tmp = self._block(ws, pk, li, bi, stage, cur, tmp, B)
            if stage.pooling:
                cur = self._merge(ws, pk, li, cur, B, H, W, C)
        H, W, C = self.layers[-1].geom[:3]
        mean, rstd = layernorm_stats(ws, cur, B * H * W, C, tag="head.ln")
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C, mean=mean, rstd=rstd, gamma=pk["head.g"], beta=pk["head.be"])
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], num_classes, x.dtype)
