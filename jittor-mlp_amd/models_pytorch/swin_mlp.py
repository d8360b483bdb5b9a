"""Swin-MLP, drop-in for the reference's models_pytorch/swin_mlp.py (eval mode: DropPath is the identity;
`ape=True` -- the absolute position embedding of swin_mlp.py:386-388,437-438 -- is one mlpk_add_periodic pass behind the patch embedding).

SwinMLPBlock (swin_mlp.py:63-157) on channel-last tokens (B*H*W, C):
  * LayerNorm -> mlpk_window_gather: zero padding of the shifted blocks (:101-102, 122-124) and the window partition (:29-42)
    as one index map, rows ((b, wy, wx), token) of C channels;
  * the multi-head spatial MLP -- a grouped Conv1d over the ws^2 window positions, one (ws^2 x ws^2) matrix per head
    (:105-108) -- is ONE token GEMM per block: the window rows are viewed as (window, (token, head)) x (C / heads)
    channels, transposed per window by mlpk_norm_apply, and multiplied by the block-diagonal (token, head) x (token, head)
    matrix built once from the Conv1d weight; the transposed epilogue stores the result back window-major.  (The dense
    block-diagonal form spends heads x the useful flops on a part that is < 2 % of the model's.)
  * mlpk_window_scatter_add: window merge (:45-60), crop of the padding (:148-149) and the residual in one pass;
  * the channel MLP folds its LayerNorm into fc1.
PatchMerging (:178-212) = 2x2 gather + LayerNorm(4C) folded into the bias-free reduction; head = LayerNorm folded into the
token mean, then the classifier GEMM.
"""
import os

import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import Block, Holder, LinearMlp, StochasticDepth, channel_mlp, finalize_stats, embed_patches, head_linear, layernorm_stats, pack_channel_mlp


def to_2tuple(v):
    return v if isinstance(v, (tuple, list)) else (v, v)


class Mlp(LinearMlp):
    """swin_mlp.py:12-26; callable on its own like the reference's (common.LinearMlp); inside a model a parameter container whose weights the
    block packs into its fused channel MLP."""


class SwinMLPBlock(Block):
    """swin_mlp.py:79-111.  Callable on (B, H*W, C) like the reference's (swin_mlp.py:113-157) once it sits in a SwinMLP."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.padding = [self.window_size - self.shift_size, self.shift_size,
                        self.window_size - self.shift_size, self.shift_size]  # P_l,P_r,P_t,P_b
        self.norm1 = norm_layer(dim)
        self.spatial_mlp = nn.Conv1d(self.num_heads * self.window_size ** 2, self.num_heads * self.window_size ** 2, kernel_size=1,
                                     groups=self.num_heads)
        self.drop_path = nn.Identity()                 # DropPath(p): identity in eval mode; train mode: SwinMLP._block (round 6)
        self.drop_path_rate = drop_path
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)


class PatchMerging(Block):
    """swin_mlp.py:178-191.  Inside a SwinMLP it runs on its own like the reference's (:193-212): (B, H*W, C) -> (B, H/2*W/2, 2C); round 5."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)


class BasicLayer(Block):
    """swin_mlp.py:231-256.  Inside a SwinMLP a stage runs on its own like the reference's (:258-266): its blocks, then its PatchMerging; round 5."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., drop=0., drop_path=0., norm_layer=nn.LayerNorm,
                 downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            SwinMLPBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                         shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio, drop=drop,
                         drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer) for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None


class PatchEmbed(Block):
    """swin_mlp.py:296-322.  Inside a SwinMLP it runs on its own like the reference's (:324-333): (B, 3, H, W) -> (B, H/4*W/4, embed_dim); round 5."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.img_size = img_size
        self.patch_size = patch_size
        self.patches_resolution = patches_resolution
        self.num_patches = patches_resolution[0] * patches_resolution[1]
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None


class SwinMLP(StochasticDepth, E.EngineModule):
    """Same signature and defaults as the reference (swin_mlp.py:374-379).

    train() (round 6): the forward applies the blocks' stochastic depth (swin_mlp.py:105,154-155: the same DropPath in front of
    both residual additions of a block) -- see common.StochasticDepth; LayerNorm has no batch statistics, Dropout has p = 0.  Forward only:
    the outputs carry no grad_fn."""
    _train_forward = True

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                 window_size=7, mlp_ratio=4., drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = ape
        self.patch_norm = patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if self.patch_norm else None)
        num_patches = self.patch_embed.num_patches
        patches_resolution = self.patch_embed.patches_resolution
        self.patches_resolution = patches_resolution
        if self.ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i_layer in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i_layer),
                input_resolution=(patches_resolution[0] // (2 ** i_layer), patches_resolution[1] // (2 ** i_layer)),
                depth=depths[i_layer], num_heads=num_heads[i_layer], window_size=window_size, mlp_ratio=self.mlp_ratio, drop=drop_rate,
                drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if (i_layer < self.num_layers - 1) else None, use_checkpoint=use_checkpoint))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        for li, layer in enumerate(self.layers):
            for bi, blk in enumerate(layer.blocks):
                blk.__dict__["_owner"] = (self, (li, bi))          # lets `model.layers[l].blocks[b](x)` run (common.Block)
            layer.__dict__["_owner"] = (self, (li, "layer"))       # ... `model.layers[l](x)`: the blocks, then the PatchMerging
            if layer.downsample is not None:
                layer.downsample.__dict__["_owner"] = (self, (li, "down"))
        self.patch_embed.__dict__["_owner"] = (self, ("embed", None))
        self.apply(self._init_weights)

    def _init_weights(self, m):
        # swin_mlp.py:419-426
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _pack(self, dtype, device):
        pk = {}
        pe = self.patch_embed
        pk["embed.w"] = E.pack_matrix(pe.proj.weight, dtype, device)
        pk["embed.b"] = E.f32(pe.proj.bias, device)
        if pe.norm is not None:
            pk["embed.g"], pk["embed.be"] = E.f32(pe.norm.weight, device), E.f32(pe.norm.bias, device)
        if self.ape:
            pk["ape"] = E.f32(self.absolute_pos_embed.detach().reshape(-1, self.embed_dim), device)
        for li, layer in enumerate(self.layers):
            for bi, blk in enumerate(layer.blocks):
                p = "l%d.b%d." % (li, bi)
                nh, t = blk.num_heads, blk.window_size ** 2
                pk[p + "n1.g"], pk[p + "n1.b"] = E.f32(blk.norm1.weight, device), E.f32(blk.norm1.bias, device)
                # grouped Conv1d weight (heads*t, t, 1): [h][t_out][t_in]  ->  block-diagonal over (token, head) x (token, head)
                w = blk.spatial_mlp.weight.detach().float().reshape(nh, t, t).cpu()
                bd = torch.zeros((t, nh, t, nh), dtype=torch.float32)
                for h in range(nh):
                    bd[:, h, :, h] = w[h]
                pk[p + "sp.w"] = E.pack_matrix(bd.reshape(t * nh, t * nh), dtype, device)
                pk[p + "sp.b"] = E.f32(blk.spatial_mlp.bias.detach().reshape(nh, t).t().reshape(-1), device)      # index t_out * heads + h
                if E.swin_spatial_supported(dtype, blk.dim, nh, blk.window_size):
                    # round 4: LayerNorm + partition + the grouped Conv1d + merge + residual in one kernel (mlpk_swin_spatial)
                    pk[p + "sp.fw"], pk[p + "sp.fb"] = E.pack_swin_spatial(blk.spatial_mlp.weight, blk.spatial_mlp.bias, nh, blk.window_size, dtype, device)
                pack_channel_mlp(pk, p + "ff.", blk.norm2, blk.mlp.fc1, blk.mlp.fc2, dtype, device)
            if layer.downsample is not None:
                pm = layer.downsample
                p = "l%d.merge." % li
                pk[p + "w"], pk[p + "b"], pk[p + "csum"] = E.pack_ln_folded(pm.reduction.weight, None, pm.norm.weight, pm.norm.bias, dtype, device)
                if dtype != torch.float32:
                    pk[p + "wc"] = E.merge_taps(pk[p + "w"], pm.norm.weight.shape[0] // 4)       # round 6: the reduction as an implicit-convolution product
        pk["norm.g"], pk["norm.b"] = E.f32(self.norm.weight, device), E.f32(self.norm.bias, device)
        if isinstance(self.head, nn.Linear):
            pk["head.w"] = E.pack_matrix(self.head.weight, dtype, device)
            pk["head.b"] = E.f32(self.head.bias, device)
        return pk

    def _block(self, ws_, pk, li, bi, blk, cur, B, H, W, C, st):
        """One SwinMLPBlock in place on channel-last rows `cur` (B*H*W, C); st = (mean, rstd) of cur's rows when the GEMM that wrote
        them delivered the statistics (else None); returns the statistics of the result the same way."""
        rows = B * H * W
        p = "l%d.b%d." % (li, bi)
        ws, nh = blk.window_size, blk.num_heads
        d = C // nh
        pad_l, pad_r, pad_t, pad_b = blk.padding if blk.shift_size > 0 else (0, 0, 0, 0)
        Hp, Wp = H + pad_t + pad_b, W + pad_l + pad_r
        nwin = B * (Hp // ws) * (Wp // ws)
        tk = ws * ws * nh                                     # "tokens" of the per-window GEMM: (window position, head)
        kp = E.round_up(tk, 8)
        tag = "l%d.s%d." % (li, 1 if blk.shift_size > 0 else 0)
        mean, rstd = st if st is not None else layernorm_stats(ws_, cur, rows, C, tag="l%d.ln" % li)
        # train mode: x = shortcut + drop_path(spatial branch), x = x + drop_path(mlp(norm2(x))) (swin_mlp.py:154-155) -- two draws per block
        dp1 = self._drop_scale(blk.drop_path_rate, B, (Hp // ws) * (Wp // ws) * d, cur.dtype, cur.device)
        dp2 = self._drop_scale(blk.drop_path_rate, B, H * W, cur.dtype, cur.device)
        if dp1 is None and dp2 is None and (p + "sp.fw") in pk and E.swin_spatial_supported(cur.dtype, C, nh, ws):
            # round 5: the kernel holds whole rows, so it also delivers norm2's statistics of what it writes (no statistics pass;
            # MLPK_SWIN_SPATIAL_STATS=0: the pass, A/B aid)
            st2 = None
            if os.environ.get("MLPK_SWIN_SPATIAL_STATS") != "0":
                st2 = (ws_.get("l%d.cm.mean" % li, (rows,), torch.float32), ws_.get("l%d.cm.rstd" % li, (rows,), torch.float32))
            E.swin_spatial(cur, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, nh, mean, rstd, pk[p + "n1.g"], pk[p + "n1.b"], pk[p + "sp.fw"], pk[p + "sp.fb"],
                           out_stats=st2)
            got = channel_mlp(ws_, cur, rows, C, pk, p + "ff.", int(C * self.mlp_ratio), tag="l%d.cm" % li, part=(ws_, "l%d.fc2.part" % li), stats=st2)
            return finalize_stats(ws_, got, rows, C, tag="l%d.ln" % li)
        xn = ws_.get("l%d.xn" % li, (rows, C))
        xw = ws_.get(tag + "xw", (nwin * ws * ws, C))
        xt = ws_.get(tag + "xt", (nwin * d, kp))
        E.norm_apply(cur, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "n1.g"], beta=pk[p + "n1.b"], out_rm=xn, ld_rm=C)
        E.window_gather(xn, xw, B, H, W, C, ws, pad_t, pad_l, Hp, Wp)
        # rows (window, token, head) x d channels  ->  per window transposed: ((window, channel), (token, head))
        E.norm_apply(xw, nwin * tk, d, d, out_tt=xt, S=tk, ld_tt=kp)
        # (stochastic depth: the branch scaled per sample where the product stores it -- GEMM row = (window, channel), the windows of an
        # image are consecutive -- before the windows are added back)
        E.gemm(xt, pk[p + "sp.w"], xw, nwin * d, tk, kp, ldc=d, bias=pk[p + "sp.b"], out_mode=N.OUT_TOKEN_T, t_rows=d, t_tokens=tk,
               tag="swin_spatial", rscale=dp1, rperiod=nwin * d if dp1 is not None else 0)
        E.window_scatter_add(cur, xw, B, H, W, C, ws, pad_t, pad_l, Hp, Wp)
        got = channel_mlp(ws_, cur, rows, C, pk, p + "ff.", int(C * self.mlp_ratio), tag="l%d.cm" % li, part=(ws_, "l%d.fc2.part" % li), rscale=dp2)
        st = finalize_stats(ws_, got, rows, C, tag="l%d.ln" % li)
        return st

    def _embed(self, ws_, pk, x, B, cd):
        """PatchEmbed (swin_mlp.py:324-333): Conv2d(k = stride = patch) + flatten + transpose (+ LayerNorm) -> channel-last tokens"""
        pe = self.patch_embed
        C = self.embed_dim
        cur, H, W = embed_patches(ws_, "embed", x, pk["embed.w"], pk["embed.b"], cd, tuple(pe.patch_size),
                                  out=ws_.get("l0.x", (B * pe.patches_resolution[0] * pe.patches_resolution[1], C)),
                                  ln=(pk["embed.g"], pk["embed.be"], pe.norm.eps) if pe.norm is not None else None)
        return cur, H, W

    def _merge(self, ws_, pk, li, cur, B, H, W, C, st=None):
        """PatchMerging (swin_mlp.py:193-212): 2 x 2 gather + LayerNorm folded into the bias-free reduction GEMM; returns (next, statistics)"""
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."                                 # swin_mlp.py:201
        p = "l%d.merge." % li
        H2, W2 = H // 2, W // 2
        nxt = ws_.get("l%d.x" % (li + 1), (B * H2 * W2, 2 * C))
        # round 6: no merged tensor where the last block's GEMM delivered the per-pixel statistics of `cur` (st): the merged rows' LayerNorm statistics are
        # combined from them (mlpk_merge2x2_stats_combine: 1.6 MB instead of a pass over the activations) and the reduction reads `cur` through the 2 x 2
        # window (mlpk_conv_gemm_nhwc, the weight's column blocks in its tap order).  Without st: MLPK_MERGE_IMPLICIT=1 takes the statistics from a pass
        # over the windows (mlpk_merge2x2_row_stats: measured neutral), the default is the gather.
        implicit = (p + "wc") in pk and pk[p + "w"].shape[1] == 4 * C and E.conv_gemm_nhwc_supported(cur.dtype, C, 2, 2, 2, 0) and \
            os.environ.get("MLPK_CONV_GEMM", "1") != "0" and (st is not None or os.environ.get("MLPK_MERGE_IMPLICIT") == "1")
        if implicit:
            mean = ws_.get("l%d.merge.ln.mean" % li, (B * H2 * W2,), torch.float32)
            rstd = ws_.get("l%d.merge.ln.rstd" % li, (B * H2 * W2,), torch.float32)
            eps_m = self.layers[li].downsample.norm.eps
            if st is not None:
                E.merge2x2_stats_combine(st[0], st[1], B, H, W, mean, rstd, eps_in=1e-5, eps_out=eps_m)
            else:
                E.merge2x2_row_stats(cur, B, H, W, C, mean, rstd, eps=eps_m)
            got = E.conv_gemm_nhwc(cur, pk[p + "wc"], nxt, B, H, W, C, 2, 2, 2, 0, bias=pk[p + "b"], ln=(mean, rstd, pk[p + "csum"]), tag="swin_merge",
                                   part=(ws_, "l%d.merge.part" % li))
        else:
            merged = ws_.get("l%d.merged" % li, (B * H2 * W2, 4 * C))
            E.patchify(cur, merged, B, C, H, W, 2, 2, 0, 4 * C, layout=N.LAYOUT_NHWC, px_stride=C, order=1)
            mean, rstd = layernorm_stats(ws_, merged, B * H2 * W2, 4 * C, tag="l%d.merge.ln" % li)
            got = E.gemm(merged, pk[p + "w"], nxt, B * H2 * W2, 2 * C, 4 * C, bias=pk[p + "b"], ln=(mean, rstd, pk[p + "csum"]), tag="swin_merge",
                         part=(ws_, "l%d.merge.part" % li))
        return nxt, finalize_stats(ws_, got, B * H2 * W2, 2 * C, tag="l%d.ln" % (li + 1))

    def _run_single(self, key, x):
        """An inner module alone, as calling it does in the reference: `model.layers[l].blocks[b](x)` (swin_mlp.py:113-157), `model.layers[l](x)`
        (a stage: :258-266), `model.layers[l].downsample(x)` (:193-212) on (B, H*W, C); `model.patch_embed(x)` (:324-333) on (B, 3, H, W)"""
        li, bi = key
        E.require_gpu(x, "SwinMLP inner module")
        E.dtype_code(x.dtype)
        if li == "embed":
            pe = self.patch_embed
            if x.dim() != 4 or x.shape[1] != pe.in_chans:
                raise ValueError("expected a (B, %d, H, W) tensor" % pe.in_chans)
            B, _, H_in, W_in = x.shape
            assert H_in == pe.img_size[0] and W_in == pe.img_size[1], \
                f"Input image size ({H_in}*{W_in}) doesn't match model ({pe.img_size[0]}*{pe.img_size[1]})."      # swin_mlp.py:327-328
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws_ = self._get_space(("embed", B, H_in, W_in), x.dtype, x.device)
                cur, H, W = self._embed(ws_, pk, x.contiguous(), B, x.dtype)
                return cur.reshape(B, H * W, self.embed_dim).clone()
        layer = self.layers[li]
        H, W = layer.input_resolution
        C = layer.dim
        if x.dim() != 3 or x.shape[1] != H * W or x.shape[2] != C:
            raise ValueError("expected a (B, %d, %d) tensor" % (H * W, C))
        B = x.shape[0]
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws_ = self._get_space(("block", B, H, W, C), x.dtype, x.device)     # (C: blocks of different stages can meet at one map size)
            cur = ws_.get("l%d.x" % li, (B * H * W, C))
            cur.copy_(x.reshape(B * H * W, C))
            st = None
            if bi != "down":
                for b_i, blk in enumerate(layer.blocks):
                    if bi == "layer" or b_i == bi:
                        st = self._block(ws_, pk, li, b_i, blk, cur, B, H, W, C, st)
            if bi == "down" or (bi == "layer" and layer.downsample is not None):
                cur, _ = self._merge(ws_, pk, li, cur, B, H, W, C)
                H, W, C = H // 2, W // 2, 2 * C
            return cur.reshape(B, H * W, C).clone()

    def _forward_train(self, x):
        """Train mode with autograd (round 6): swin_mlp.py:33-157,175-215,295-335,428-456 as autograd.Functions of `..autograd`,
        forward and backward through the C ABI.  The zero padding of the shifted blocks, the window partition and their inverses are index
        tables (mlpk_index_gather; the inverse table is the gradient), built by running the reference's own F.pad / view / permute on a tensor
        of positions; the multi-head spatial MLP (a grouped Conv1d over a window's tokens) is one mlpk_gemm_nt per head between per-window
        transposes (mlpk_transpose_batched); stochastic depth on drop_path_uniform's draws; PatchMerging = mlpk_merge2x2_nhwc."""
        import torch.nn.functional as F
        from .. import autograd as AG
        E.require_gpu(x, "SwinMLP.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        pe = self.patch_embed
        B, cin, H_in, W_in = x.shape
        assert H_in == pe.img_size[0] and W_in == pe.img_size[1], \
            f"Input image size ({H_in}*{W_in}) doesn't match model ({pe.img_size[0]}*{pe.img_size[1]})."
        ph, pw = pe.patch_size
        H, W = H_in // ph, W_in // pw
        dev = x.device
        kp = E.round_up(cin * ph * pw, 4 if cd == torch.float32 else 8)
        with E.on_device(x):
            patches = torch.zeros((B * H * W, kp), dtype=cd, device=dev)
            E.patchify(x.contiguous(), patches, B, cin, H_in, W_in, ph, pw, 0, kp)
        tables = self.__dict__.setdefault("_tables", {})

        def ln(t, norm):
            return AG.LayerNorm.apply(t, norm.weight, norm.bias, norm.eps)

        t = AG.Linear.apply(patches, pe.proj.weight, pe.proj.bias, None)
        if pe.norm is not None:
            t = ln(t, pe.norm)
        if self.ape:
            t = AG.AddPeriodic.apply(t, self.absolute_pos_embed, H * W)
        C = self.embed_dim
        for layer in self.layers:
            for blk in layer.blocks:
                ws, nH = blk.window_size, blk.num_heads
                P_l, P_r, P_t, P_b = blk.padding if blk.shift_size > 0 else (0, 0, 0, 0)
                Hp, Wp = H + P_t + P_b, W + P_l + P_r

                def part(pos, H=H, W=W, Hp=Hp, Wp=Wp, ws=ws, pads=(P_l, P_r, P_t, P_b)):
                    g = F.pad(pos.view(1, H, W, 1), [0, 0, pads[0], pads[1], pads[2], pads[3]], "constant", 0)      # swin_mlp.py:129-132
                    g = g.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).contiguous()            # window_partition (:33-45)
                    return g

                def rev(pos, H=H, W=W, Hp=Hp, Wp=Wp, ws=ws, pads=(P_l, P_r, P_t, P_b)):
                    g = pos.view(1, Hp // ws, Wp // ws, ws, ws, 1).permute(0, 1, 3, 2, 4, 5).contiguous().view(1, Hp, Wp, 1)   # window_reverse (:48-60)
                    return g[:, pads[2]:Hp - pads[3], pads[0]:Wp - pads[1], :].contiguous()                        # reverse shift (:146-149)

                t_part = AG.position_table(part, H * W, C, dev, tables, ("part", H, W, ws, P_l, P_t, C))
                t_rev = AG.position_table(rev, Hp * Wp, C, dev, tables, ("rev", H, W, ws, P_l, P_t, C))
                nwb = B * (Hp // ws) * (Wp // ws)
                xw = AG.IndexMap.apply(ln(t, blk.norm1), t_part, B, C)                       # (nW*B * ws*ws, C): windows, tokens row-major
                ch = C // nH
                heads = []
                for h in range(nH):
                    # head h: the window's tokens mixed by rows [h ws^2, (h+1) ws^2) of the grouped Conv1d (swin_mlp.py:101-104,137-143)
                    wgt = blk.spatial_mlp.weight[h * ws * ws:(h + 1) * ws * ws]
                    bias = blk.spatial_mlp.bias[h * ws * ws:(h + 1) * ws * ws] if blk.spatial_mlp.bias is not None else None
                    rows_h = AG.TokensToRows.apply(xw[:, h * ch:(h + 1) * ch], nwb, ws * ws)
                    heads.append(AG.RowsToTokens.apply(AG.Linear.apply(rows_h, wgt, bias, None), nwb, ws * ws, ch))
                y = AG.IndexMap.apply(AG.ConcatCols.apply(*heads) if nH > 1 else heads[0], t_rev, B, C)
                t = AG.drop_add(self, t, y, blk.drop_path_rate if self.training else 0.0, B, H * W)
                hdn = AG.Gelu.apply(AG.Linear.apply(ln(t, blk.norm2), blk.mlp.fc1.weight, blk.mlp.fc1.bias, None))
                if float(blk.drop_path_rate) > 0.0:
                    t = AG.drop_add(self, t, AG.Linear.apply(hdn, blk.mlp.fc2.weight, blk.mlp.fc2.bias, None), blk.drop_path_rate, B, H * W)
                else:
                    t = AG.Linear.apply(hdn, blk.mlp.fc2.weight, blk.mlp.fc2.bias, t)
            if layer.downsample is not None:
                ds = layer.downsample
                t = AG.Linear.apply(ln(AG.Merge2x2.apply(t, B, H, W), ds.norm), ds.reduction.weight, None, None)
                H, W, C = H // 2, W // 2, 2 * C
        pooled = AG.TokenMean.apply(ln(t, self.norm), B, H * W)
        if not isinstance(self.head, nn.Linear):
            return pooled if pooled.dtype == x.dtype else pooled.to(x.dtype)
        logits = AG.Linear.apply(pooled, self.head.weight, self.head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        pe = self.patch_embed
        B, _, H_in, W_in = x.shape
        assert H_in == pe.img_size[0] and W_in == pe.img_size[1], \
            f"Input image size ({H_in}*{W_in}) doesn't match model ({pe.img_size[0]}*{pe.img_size[1]})."          # swin_mlp.py:327-328
        pk = self._get_pack(cd, x.device)
        ws_ = self._get_space(B, cd, x.device)
        C = self.embed_dim
        cur, H, W = self._embed(ws_, pk, x.contiguous(), B, cd)
        if self.ape:
            E.add_periodic(cur, C, pk["ape"], B * H * W, C, H * W)                       # x + absolute_pos_embed (swin_mlp.py:437-438)
        st = None            # (mean, rstd) of cur's rows when the GEMM that wrote cur delivered them (mlpk.h row_part)
        for li, layer in enumerate(self.layers):
            rows = B * H * W
            xn = ws_.get("l%d.xn" % li, (rows, C))
            for bi, blk in enumerate(layer.blocks):
                st = self._block(ws_, pk, li, bi, blk, cur, B, H, W, C, st)
            if layer.downsample is not None:
                cur, st = self._merge(ws_, pk, li, cur, B, H, W, C, st)
                H, W, C = H // 2, W // 2, 2 * C
        mean, rstd = st if st is not None else layernorm_stats(ws_, cur, B * H * W, C, tag="head.ln")
        pooled = ws_.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C, mean=mean, rstd=rstd, gamma=pk["norm.g"], beta=pk["norm.b"])
        if not isinstance(self.head, nn.Linear):
            out = pooled.clone()
            return out if out.dtype == x.dtype else out.to(x.dtype)
        return head_linear(ws_, pooled, B, C, pk["head.w"], pk["head.b"], self.num_classes, x.dtype)
