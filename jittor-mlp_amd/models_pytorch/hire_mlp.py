"""Hire-MLP, drop-in for the reference's models_pytorch/hire_mlp.py, padding_type 'circular'
(the reference default; the other torch padding modes are not built).

HireMLPBlock (hire_mlp.py:96-152) on the channel-last LayerNorm output xn (B*H*W, C):
  * the circular pad to whole regions (:131-133 -- a whole extra region when the size already divides), the cross-region
    roll (:44-51) and the einops fold of the h (w) rows (columns) one region-count apart into the channel axis (:53-93) are
    index arithmetic inside mlpk_hire_gather, which writes the operand rows of the two branch MLPs directly:
        A_h rows (b, g, x), K = (region row hh, channel c);   A_w rows (b, y, g), K = (region column ww, channel c)
    (the 1x1-conv weights are permuted from the reference's (c, hh) order once, when packed);
  * proj_h / proj_w (FeedForward: 1x1 conv -> GELU -> 1x1 conv, hidden C/2) are two NT GEMMs each, GELU in the epilogue;
  * proj_c is a GEMM on xn with the block's residual in its epilogue; mlpk_hire_combine adds both branch results back through
    the inverse index map (restore, roll back, crop) -- nothing padded, rolled or permuted is ever materialised.
The patcher (7x7 stride-4 pad-3 conv, :203) and the stage transitions (3x3 stride-2 pad-1 conv, :161) are window gathers
(mlpk_im2col) + GEMM; the channel MLP folds its LayerNorm into fc1; the head folds its LayerNorm into the token mean.
"""
import contextlib
import os

import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import Block, BlockSequential, Holder, SubModule, channel_mlp, finalize_stats, head_linear, layernorm_stats, pack_channel_mlp
from .utils import pair


class PreNormResidual(Block):
    """fn(norm(x)) + x (hire_mlp.py:8-15).  Inside a HireMLP both halves of a block -- `layers[l].model[b][0]` (around the HireMLPBlock) and
    `[b][1]` (around the channel MLP) -- run on their own like the reference's, on channel-last (B, H, W, C); round 6."""

    def __init__(self, dim, fn, norm=nn.LayerNorm):
        super().__init__()
        self.fn = fn
        self.norm = norm(dim)


class PatchEmbedding(Block):
    """hire_mlp.py:17-31.  Inside a HireMLP -- as `model.patcher` (7 x 7, stride = patch, padding 3) and as a pooling stage's `patch_merge[1]`
    (3 x 3, stride 2, padding 1) -- it runs on its own like the reference's: NCHW in, NCHW out; round 5."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, padding, norm_layer=False):
        super().__init__()
        self.reduction = nn.Sequential(
            nn.Conv2d(dim_in, dim_out, kernel_size=kernel_size, stride=stride, padding=padding),
            nn.Identity() if (not norm_layer) else nn.Sequential(nn.Identity(), nn.LayerNorm(dim_out), nn.Identity()))


class FeedForward(SubModule):
    """hire_mlp.py:33-42: Conv2d(1 x 1) -> GELU -> Conv2d(1 x 1) on (B, C, H, W).  Inside a model a parameter container (the block packs
    `proj_h` / `proj_w` in its gathered order); on its own callable like the reference's (round 6): channel-last inside, the two 1 x 1
    convolutions are NT GEMMs with the GELU in the first epilogue."""

    def __init__(self, dim_in, hidden_dim, dim_out):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(dim_in, hidden_dim, kernel_size=1), nn.GELU(), nn.Conv2d(hidden_dim, dim_out, kernel_size=1))

    def _pack(self, dtype, device):
        c1, c2 = self.net[0], self.net[2]
        return {"w1": E.pack_matrix(c1.weight, dtype, device), "b1": E.f32(c1.bias, device),
                "w2": E.pack_matrix(c2.weight, dtype, device), "b2": E.f32(c2.bias, device)}

    def forward(self, x):
        c1, c2 = self.net[0], self.net[2]
        cin, hid, cout = c1.in_channels, c1.out_channels, c2.out_channels
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        pk = self._begin(x, cin, axis=1)
        B, _, H, W = x.shape
        rows = B * H * W
        with E.on_device(x):
            ws = self._get_space((rows, cin), x.dtype, x.device)
            xb = ws.get("ff.x", (rows, pk["w1"].shape[1]))              # K zero-padded to whole 16-byte chunks
            xb[:, :cin].copy_(x.permute(0, 2, 3, 1).reshape(rows, cin))
            t = ws.get("ff.t", (rows, pk["w2"].shape[1]))
            E.gemm(xb, pk["w1"], t, rows, hid, pk["w1"].shape[1], bias=pk["b1"], act=N.ACT_GELU)
            y = torch.empty((rows, cout), dtype=x.dtype, device=x.device)
            E.gemm(t, pk["w2"], y, rows, cout, pk["w2"].shape[1], bias=pk["b2"])
            return y.reshape(B, H, W, cout).permute(0, 3, 1, 2).contiguous()


class HireMLPBlock(Block):
    """hire_mlp.py:96-152 (the rearrange / roll sub-modules hold no parameters and are index arithmetic here).  Inside a HireMLP it runs on its
    own like the reference's (:128-152): channel-last (B, H, W, C) in and out, no LayerNorm, no residual; round 6."""

    def __init__(self, h, w, d_model, cross_region_step=1, cross_region_id=0, cross_region_interval=2, padding_type='circular'):
        super().__init__()
        assert (padding_type in ['constant', 'reflect', 'replicate', 'circular'])
        if padding_type != 'circular':
            raise NotImplementedError("padding_type %r: only the reference default 'circular' is built" % padding_type)
        self.padding_type = padding_type
        self.w = w
        self.h = h
        self.cross_region = (cross_region_id % cross_region_interval == 0)
        self.step = cross_region_step if self.cross_region else 0
        self.proj_h = FeedForward(h * d_model, d_model // 2, h * d_model)
        self.proj_w = FeedForward(w * d_model, d_model // 2, w * d_model)
        self.proj_c = nn.Conv2d(d_model, d_model, kernel_size=1)


class HireMLPStage(Block):
    """hire_mlp.py:154-187; `patch_merge` exists in every stage, like the reference's.  Inside a HireMLP a stage runs on its own like the
    reference's (:182-186): channel-last (B, H, W, C) through its blocks and, where pooling, its PatchEmbedding; round 5."""

    def __init__(self, h, w, d_model_in, d_model_out, depth, cross_region_step, cross_region_interval, expansion_factor=2,
                 dropout=0., pooling=False, padding_type='circular'):
        super().__init__()
        self.pooling = pooling
        self.patch_merge = nn.Sequential(nn.Identity(),
                                         PatchEmbedding(d_model_in, d_model_out, kernel_size=3, stride=2, padding=1, norm_layer=False),
                                         nn.Identity())
        self.model = nn.Sequential(*[BlockSequential(
            PreNormResidual(d_model_in, nn.Sequential(HireMLPBlock(h, w, d_model_in, cross_region_step=cross_region_step,
                                                                   cross_region_id=i_depth + 1, cross_region_interval=cross_region_interval,
                                                                   padding_type=padding_type)), norm=nn.LayerNorm),
            PreNormResidual(d_model_in, nn.Sequential(nn.Linear(d_model_in, d_model_in * expansion_factor), nn.GELU(), nn.Dropout(dropout),
                                                      nn.Linear(d_model_in * expansion_factor, d_model_in), nn.Dropout(dropout)),
                            norm=nn.LayerNorm)) for i_depth in range(depth)])
        self.geom = (h, w, d_model_in, d_model_out, depth, expansion_factor)


def _perm_in(wt, c, r):
    """1x1-conv weight (out, c*r [,1,1]) with input index c_i * r + rr  ->  (out, r*c) with index rr * c + c_i."""
    o = wt.shape[0]
    return wt.detach().reshape(o, c, r).permute(0, 2, 1).reshape(o, r * c)


def _perm_out(wt, b, c, r):
    """1x1-conv weight (c*r, in [,1,1]) / bias (c*r) with output index c_o * r + rr  ->  rows rr * c + c_o."""
    i = wt.shape[1]
    return (wt.detach().reshape(c, r, i).permute(1, 0, 2).reshape(r * c, i), b.detach().reshape(c, r).t().reshape(r * c))


class HireMLP(E.EngineModule):
    """Same signature and defaults as the reference (hire_mlp.py:188-201)."""

    def __init__(self, patch_size=4, in_channels=3, num_classes=1000, d_model=[64, 128, 320, 512], h=[4, 3, 3, 2], w=[4, 3, 3, 2],
                 cross_region_step=[2, 2, 1, 1], cross_region_interval=2, depth=[4, 6, 24, 3], expansion_factor=2,
                 patcher_norm=False, padding_type='circular'):
        patch_size = pair(patch_size)
        super().__init__()
        self.patcher = PatchEmbedding(dim_in=in_channels, dim_out=d_model[0], kernel_size=7, stride=patch_size, padding=3,
                                      norm_layer=patcher_norm)
        self.layers = nn.ModuleList()
        for i_layer in range(len(depth)):
            self.layers.append(HireMLPStage(
                h[i_layer], w[i_layer], d_model[i_layer],
                d_model_out=d_model[i_layer + 1] if (i_layer + 1 < len(depth)) else d_model[-1], depth=depth[i_layer],
                cross_region_step=cross_region_step[i_layer], cross_region_interval=cross_region_interval,
                expansion_factor=expansion_factor, pooling=((i_layer + 1) < len(depth)), padding_type=padding_type))
        self.mlp_head = nn.Sequential(nn.LayerNorm(d_model[-1]), nn.Identity(), nn.Linear(d_model[-1], num_classes))
        self._cfg = (patch_size, in_channels, num_classes, patcher_norm)
        for li, stage in enumerate(self.layers):
            for bi, blk in enumerate(stage.model):
                blk.__dict__["_owner"] = (self, (li, bi))          # lets `model.layers[l].model[b](x)` run (common.BlockSequential)
                blk[0].__dict__["_owner"] = (self, (li, (bi, "pre0")))    # ... its two PreNormResidual halves and the HireMLPBlock (round 6)
                blk[1].__dict__["_owner"] = (self, (li, (bi, "pre1")))
                blk[0].fn[0].__dict__["_owner"] = (self, (li, (bi, "hire")))
            stage.__dict__["_owner"] = (self, (li, "layer"))       # ... `model.layers[l](x)`: the blocks, then the PatchEmbedding where pooling
            stage.patch_merge[1].__dict__["_owner"] = (self, (li, "merge"))
        self.patcher.__dict__["_owner"] = (self, ("embed", None))

    def _pack(self, dtype, device):
        pk = {}
        conv = self.patcher.reduction[0]
        pk["embed.w"] = E.pack_matrix(conv.weight, dtype, device)                           # (ci, i, j) order == the NCHW window gather
        if dtype != torch.float32 and self._cfg[0] == (4, 4) and self._cfg[1] == 3:
            pk["embed.w7"] = E.pack_stem7(conv.weight, dtype, device)                      # round 6: the stem as a direct convolution (mlpk_stem7)
        pk["embed.b"] = E.f32(conv.bias, device)
        if self._cfg[3]:
            ln = self.patcher.reduction[1][1]
            pk["embed.g"], pk["embed.be"] = E.f32(ln.weight, device), E.f32(ln.bias, device)
        st = None            # (mean, rstd) of cur's rows when the GEMM that wrote cur delivered them (mlpk.h row_part)
        for li, stage in enumerate(self.layers):
            h, w, C, Cout, depth, ef = stage.geom
            for bi, blk in enumerate(stage.model):
                p = "l%d.b%d." % (li, bi)
                hb = blk[0].fn[0]
                pk[p + "ln.g"], pk[p + "ln.b"] = E.f32(blk[0].norm.weight, device), E.f32(blk[0].norm.bias, device)
                for tag, ff, r in (("h", hb.proj_h, h), ("w", hb.proj_w, w)):
                    pk[p + tag + "1.w"] = E.pack_matrix(_perm_in(ff.net[0].weight, C, r), dtype, device)
                    pk[p + tag + "1.b"] = E.f32(ff.net[0].bias, device)
                    w2, b2 = _perm_out(ff.net[2].weight, ff.net[2].bias, C, r)
                    pk[p + tag + "2.w"] = E.pack_matrix(w2, dtype, device)
                    pk[p + tag + "2.b"] = E.f32(b2, device)
                pk[p + "c.w"] = E.pack_matrix(hb.proj_c.weight, dtype, device)
                pk[p + "c.b"] = E.f32(hb.proj_c.bias, device)
                # round 5: proj_c with the block's LayerNorm folded in (reads x itself; the gather applies the LayerNorm to what it moves)
                pk[p + "cf.w"], pk[p + "cf.b"], pk[p + "cf.csum"] = E.pack_ln_folded(hb.proj_c.weight, hb.proj_c.bias, blk[0].norm.weight, blk[0].norm.bias,
                                                                                   dtype, device)
                ff = blk[1]
                pack_channel_mlp(pk, p + "ff.", ff.norm, ff.fn[0], ff.fn[3], dtype, device)
            if stage.pooling:
                mc = stage.patch_merge[1].reduction[0]
                # channel-last window gather: K order (i, j, ci)
                pk["l%d.merge.w" % li] = E.pack_matrix(mc.weight.detach().permute(0, 2, 3, 1).reshape(Cout, 9 * C), dtype, device)
                pk["l%d.merge.b" % li] = E.f32(mc.bias, device)
        pk["head.g"], pk["head.be"] = E.f32(self.mlp_head[0].weight, device), E.f32(self.mlp_head[0].bias, device)
        pk["head.w"] = E.pack_matrix(self.mlp_head[2].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[2].bias, device)
        return pk

    def _block(self, ws, pk, li, bi, stage, cur, B, H, W, st, part="both"):
        """Block `layers[li].model[bi]` in place on channel-last rows `cur` (B*H*W, C); st = (mean, rstd) of cur's rows when the GEMM that
        wrote them delivered the statistics (else None); returns the statistics of the result the same way.
        part (round 6, the block's inner modules on their own): "pre0" = x + HireMLPBlock(LN x) only, "pre1" = x + MLP(LN x) only,
        "hire" = HireMLPBlock(x) alone (no LayerNorm, no residual)."""
        h, w, C, Cout, depth, ef = stage.geom
        rows = B * H * W
        Hp, Wp = H + (h - H % h), W + (w - W % w)                                         # hire_mlp.py:131-133
        gh, gw = Hp // h, Wp // w
        rows_h, rows_w = B * gh * W, B * H * gw
        hid = C // 2
        hidp = E.round_up(hid, 8)
        xn = ws.get("l%d.xn" % li, (rows, C))
        a_h = ws.get("l%d.ah" % li, (rows_h, h * C))
        a_w = ws.get("l%d.aw" % li, (rows_w, w * C))
        t_h = ws.get("l%d.th" % li, (rows_h, hidp))
        t_w = ws.get("l%d.tw" % li, (rows_w, hidp))
        blk = stage.model[bi]
        p = "l%d.b%d." % (li, bi)
        step = blk[0].fn[0].step
        if part == "pre1":
            channel_mlp(ws, cur, rows, C, pk, p + "ff.", C * ef, tag="l%d.cm" % li)
            return None
        if part == "hire":
            # x is what the block's LayerNorm would have delivered: gather it as it is, out = proj_c(x) + y_h + y_w
            xn.copy_(cur)
            E.hire_gather(xn, a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)
            E.gemm(a_w, pk[p + "w1.w"], t_w, rows_w, hid, w * C, bias=pk[p + "w1.b"], act=N.ACT_GELU, tag="hire_fc1")
            E.gemm(t_w, pk[p + "w2.w"], a_w, rows_w, w * C, hidp, bias=pk[p + "w2.b"], tag="hire_fc2")
            E.gemm(a_h, pk[p + "h1.w"], t_h, rows_h, hid, h * C, bias=pk[p + "h1.b"], act=N.ACT_GELU, tag="hire_fc1")
            E.gemm(t_h, pk[p + "h2.w"], a_h, rows_h, h * C, hidp, bias=pk[p + "h2.b"], tag="hire_fc2")
            E.gemm(xn, pk[p + "c.w"], cur, rows, C, C, bias=pk[p + "c.b"], tag="hire_c")
            E.hire_combine(cur, a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)
            return None
        mean, rstd = st if st is not None else layernorm_stats(ws, cur, rows, C, tag="l%d.ln" % li)
        st2 = None                                                                        # statistics of the block's first half, when the combine delivers them
        fold = cur.dtype != torch.float32 and os.environ.get("MLPK_HIRE_LN_FOLD") != "0"
        if fold:
            # round 5: no stored LayerNorm output -- the gather normalises the vectors it moves, proj_c reads x with the LayerNorm folded in
            E.hire_gather_ln(cur, mean, rstd, pk[p + "ln.g"], pk[p + "ln.b"], a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)
        else:
            E.norm_apply(cur, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "ln.g"], beta=pk[p + "ln.b"], out_rm=xn, ld_rm=C)
            E.hire_gather(xn, a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)
        # the w-branch pair of GEMMs touches only a_w / t_w: it runs on a side stream beside the h-branch pair and proj_c
        # (short GEMMs of 20-50 us each: two kernels in flight fill the tail of each other's last wave of tiles)
        # one step at a time the w-branch pair runs on a side stream beside the h-branch pair and proj_c; with forwards in flight (no side streams:
        # engine.set_side_streams) the branches' Linears go pairwise into one launch each.  Same box: 9.31 -> 9.14 ms per step in flight with the pairs,
        # 10.08 -> 10.20 ms one at a time (there the side chain also overlaps proj_c) -- profiles/r06_hire_combine_stats_ab.txt.  MLPK_HIRE_PAIR=0/1 forces.
        pe = os.environ.get("MLPK_HIRE_PAIR")
        paired = (pe != "0") if pe is not None else E.side_stream(cur.device) is None
        chain = E.SideChain(ws, "hire.w", cur.device) if not paired else contextlib.nullcontext()
        if paired:
            # round 6: the two branches' Linears pairwise in ONE launch each (mlpk_gemm_nt_pair: the same tiles, the same bits) -- 20-30 us products,
            # launch- and latency-bound: two launches per block instead of four, and no side stream with its fork / join events
            E.gemm_pair(((a_w, pk[p + "w1.w"], t_w, rows_w, hid, w * C), dict(bias=pk[p + "w1.b"], act=N.ACT_GELU, tag="hire_fc1")),
                        ((a_h, pk[p + "h1.w"], t_h, rows_h, hid, h * C), dict(bias=pk[p + "h1.b"], act=N.ACT_GELU, tag="hire_fc1")))
            E.gemm_pair(((t_w, pk[p + "w2.w"], a_w, rows_w, w * C, hidp), dict(bias=pk[p + "w2.b"], tag="hire_fc2")),
                        ((t_h, pk[p + "h2.w"], a_h, rows_h, h * C, hidp), dict(bias=pk[p + "h2.b"], tag="hire_fc2")))     # y_w / y_h overwrite a_w / a_h
        else:
            with chain:
                E.gemm(a_w, pk[p + "w1.w"], t_w, rows_w, hid, w * C, bias=pk[p + "w1.b"], act=N.ACT_GELU, tag="hire_fc1")
                E.gemm(t_w, pk[p + "w2.w"], a_w, rows_w, w * C, hidp, bias=pk[p + "w2.b"], tag="hire_fc2")
            E.gemm(a_h, pk[p + "h1.w"], t_h, rows_h, hid, h * C, bias=pk[p + "h1.b"], act=N.ACT_GELU, tag="hire_fc1")
            E.gemm(t_h, pk[p + "h2.w"], a_h, rows_h, h * C, hidp, bias=pk[p + "h2.b"], tag="hire_fc2")    # y_h overwrites a_h
        if fold:
            E.gemm(cur, pk[p + "cf.w"], xn, rows, C, C, bias=pk[p + "cf.b"], ln=(mean, rstd, pk[p + "cf.csum"]), R=cur, res=N.RES_ADD, tag="hire_c")
            if not paired:
                chain.join()
            if part != "pre0" and os.environ.get("MLPK_HIRE_COMBINE_STATS") != "0":
                # round 6: the combine delivers the statistics of the rows it writes -- the MLP half's LayerNorm needs no pass over x
                st2 = (ws.get("l%d.cm.mean" % li, (rows,), torch.float32), ws.get("l%d.cm.rstd" % li, (rows,), torch.float32))
                E.hire_combine_stats(cur, xn, a_h, a_w, B, H, W, C, h, w, step, h * C, w * C, st2[0], st2[1], eps=blk[1].norm.eps)
            else:
                E.hire_combine_from(cur, xn, a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)  # x = (x + proj_c(LN x)) + y_h + y_w
        else:
            E.gemm(xn, pk[p + "c.w"], cur, rows, C, C, bias=pk[p + "c.b"], R=cur, res=N.RES_ADD, tag="hire_c")   # x + proj_c(xn)
            if not paired:
                chain.join()
            E.hire_combine(cur, a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)
        if part == "pre0":
            return None
        got = channel_mlp(ws, cur, rows, C, pk, p + "ff.", C * ef, tag="l%d.cm" % li, part=(ws, "l%d.fc2.part" % li), stats=st2, eps=blk[1].norm.eps)
        st = finalize_stats(ws, got, rows, C, tag="l%d.ln" % li)
        return st

    def _embed(self, ws, pk, x, B, H_in, W_in):
        """`patcher` (hire_mlp.py:203): 7 x 7 stride-patch pad-3 conv (+ LayerNorm) on the NCHW image -> channel-last rows"""
        patch, cin, num_classes, patcher_norm = self._cfg
        C = self.layers[0].geom[2]
        H, W = (H_in + 6 - 7) // patch[0] + 1, (W_in + 6 - 7) // patch[1] + 1
        kp = pk["embed.w"].shape[1]
        cur = ws.get("l0.x", (B * H * W, C))
        if "embed.w7" in pk and x.data_ptr() % 16 == 0 and E.stem7_supported(x.dtype, cur.dtype, cin, H_in, W_in, 3, C):
            E.stem7(x, pk["embed.w7"], pk["embed.b"], cur, B, H_in, W_in, 3, C)
        else:
            patches = ws.get("embed.patches", (B * H * W, kp))
            E.im2col(x, patches, B, cin, H_in, W_in, 7, 7, patch[0], patch[1], 3, kp)
            E.gemm(patches, pk["embed.w"], cur, B * H * W, C, kp, bias=pk["embed.b"])
        if patcher_norm:
            mean, rstd = layernorm_stats(ws, cur, B * H * W, C, tag="embed.ln")
            E.norm_apply(cur, B * H * W, C, C, mean=mean, rstd=rstd, gamma=pk["embed.g"], beta=pk["embed.be"], out_rm=cur, ld_rm=C)
        return cur, H, W, C

    def _merge(self, ws, pk, li, cur, B, H, W, C):
        """a pooling stage's `patch_merge` (hire_mlp.py:158-162): 3 x 3 stride-2 pad-1 conv on channel-last rows; returns (next, H2, W2, Cout, statistics)"""
        Cout = self.layers[li].geom[3]
        H2, W2 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        kp = pk["l%d.merge.w" % li].shape[1]
        nxt = ws.get("l%d.x" % (li + 1), (B * H2 * W2, Cout))
        if kp == 9 * C and E.conv_gemm_nhwc_supported(cur.dtype, C, 3, 3, 2, 1):
            # round 6: the window is the product's operand loader -- no gathered copy of the map (mlpk_conv_gemm_nhwc)
            got = E.conv_gemm_nhwc(cur, pk["l%d.merge.w" % li], nxt, B, H, W, C, 3, 3, 2, 1, bias=pk["l%d.merge.b" % li], tag="hire_merge",
                                   part=(ws, "l%d.merge.part" % li))
        else:
            cols = ws.get("l%d.cols" % li, (B * H2 * W2, kp))
            E.im2col(cur, cols, B, C, H, W, 3, 3, 2, 2, 1, kp, layout=N.LAYOUT_NHWC, px_stride=C)
            got = E.gemm(cols, pk["l%d.merge.w" % li], nxt, B * H2 * W2, Cout, kp, bias=pk["l%d.merge.b" % li], tag="hire_merge",
                         part=(ws, "l%d.merge.part" % li))
        return nxt, H2, W2, Cout, finalize_stats(ws, got, B * H2 * W2, Cout, tag="l%d.ln" % (li + 1))

    def _run_single(self, key, x):
        """An inner module alone, as calling it does in the reference: block `layers[l].model[b]` and stage `layers[l]` on channel-last
        (B, H, W, C) (hire_mlp.py:176-187); `patcher` and `layers[l].patch_merge[1]` (PatchEmbedding, :17-31) on NCHW"""
        li, bi = key
        E.require_gpu(x, "HireMLP inner module")
        E.dtype_code(x.dtype)
        if li == "embed":
            cin = self._cfg[1]
            if x.dim() != 4 or x.shape[1] != cin:
                raise ValueError("expected a (B, %d, H, W) tensor" % cin)
            B, _, H_in, W_in = x.shape
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("embed", B, H_in, W_in), x.dtype, x.device)
                cur, H, W, C = self._embed(ws, pk, x.contiguous(), B, H_in, W_in)
                return cur.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        stage = self.layers[li]
        C = stage.geom[2]
        if bi == "merge":
            if not stage.pooling:
                raise NotImplementedError("this stage does not pool: its PatchEmbedding only holds parameters, like the reference's unused one")
            if x.dim() != 4 or x.shape[1] != C:
                raise ValueError("expected a (B, %d, H, W) tensor" % C)
            B, _, H, W = x.shape
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)
                cur = ws.get("blk.x", (B * H * W, C))
                cur.copy_(x.permute(0, 2, 3, 1).reshape(B * H * W, C))
                nxt, H2, W2, Cout, _ = self._merge(ws, pk, li, cur, B, H, W, C)
                return nxt.reshape(B, H2, W2, Cout).permute(0, 3, 1, 2).contiguous()
        if x.dim() != 4 or x.shape[-1] != C:
            raise ValueError("expected a channel-last (B, H, W, %d) tensor" % C)
        B, H, W, _ = x.shape
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)     # (C: blocks of different stages can meet at one map size)
            cur = ws.get("blk.x", (B * H * W, C))
            cur.copy_(x.reshape(B * H * W, C))
            st = None
            if isinstance(bi, tuple):                                  # an inner module of block bi[0] (round 6)
                self._block(ws, pk, li, bi[0], stage, cur, B, H, W, None, part=bi[1])
                return cur.reshape(B, H, W, C).clone()
            for b_i in range(len(stage.model)):
                if bi == "layer" or b_i == bi:
                    st = self._block(ws, pk, li, b_i, stage, cur, B, H, W, st)
            if bi == "layer" and stage.pooling:
                cur, H, W, C, _ = self._merge(ws, pk, li, cur, B, H, W, C)
            return cur.reshape(B, H, W, C).clone()

    _train_forward = True

    def _forward_train(self, x):
        """Train mode with autograd (round 6): hire_mlp.py:6-215 as autograd.Functions of `..autograd`, forward and backward through the
        C ABI.  A block's circular padding + cross-region roll + inner-region rearrange, and the restore + roll back + crop, are ONE index
        table each per axis (mlpk_index_gather at element granularity; built by running the reference's own F.pad / torch.roll / einops
        patterns on a tensor of positions; the inverse table -- which sums the duplicates the circular padding makes -- is the gradient);
        the region FeedForwards and proj_c are mlpk_gemm_nt (proj_c is pointwise: it commutes with the padding and the crop); the 3 x 3
        stride-2 stage transitions are an overlapping-window table + mlpk_gemm_nt (col2im = the inverse table)."""
        import torch.nn.functional as F
        from .. import autograd as AG
        E.require_gpu(x, "HireMLP.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        patch, cin, num_classes, patcher_norm = self._cfg
        B, _, H_in, W_in = x.shape
        dev = x.device
        H, W = (H_in + 6 - 7) // patch[0] + 1, (W_in + 6 - 7) // patch[1] + 1
        kp = E.round_up(cin * 49, 8)                                                     # (mlpk_im2col: rows of whole 16-byte chunks)
        with E.on_device(x):
            patches = torch.zeros((B * H * W, kp), dtype=cd, device=dev)
            E.im2col(x.contiguous(), patches, B, cin, H_in, W_in, 7, 7, patch[0], patch[1], 3, kp)
        tables = self.__dict__.setdefault("_tables", {})

        def ln(t, norm):
            return AG.LayerNorm.apply(t, norm.weight, norm.bias, norm.eps)

        def ff(rows, m):                                                                          # FeedForward (hire_mlp.py:33-42): two 1x1 convs
            return AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(rows, m.net[0].weight, m.net[0].bias, None)), m.net[2].weight, m.net[2].bias, None)

        red = self.patcher.reduction
        t = AG.Linear.apply(patches, red[0].weight, red[0].bias, None)
        if patcher_norm:
            t = ln(t, red[1][1])
        for stage in self.layers:
            h, w, C, Cout, depth, ef = stage.geom
            for blk in stage.model:
                pre, mlp = blk[0], blk[1]
                hb = pre.fn[0]
                step = hb.step
                Hp, Wp = H + (h - H % h), W + (w - W % w)                                          # hire_mlp.py:134-136 (a whole region more when divisible)

                def padded(pos, H=H, W=W, C=C, Hp=Hp, Wp=Wp, mode=hb.padding_type):
                    g = pos.view(1, H, W, C).permute(0, 3, 1, 2)                                   # NCHW view of the channel-last positions
                    return F.pad(g, (0, Wp - W, 0, Hp - H), mode) if mode != "constant" else F.pad(g, (0, Wp - W, 0, Hp - H), "constant", 0)

                def region_h(pos, h=h, C=C, Hp=Hp, Wp=Wp, step=step):
                    g = padded(pos)
                    g = torch.roll(g, step, 2) if step else g                                      # cross_regionH
                    g = g.reshape(1, C * h, Hp // h, Wp)                                           # 'b c (h group) w -> b (c h) group w'
                    return g.permute(0, 2, 3, 1).contiguous()                                      # rows (group, w), columns (c h)

                def region_w(pos, w=w, C=C, Hp=Hp, Wp=Wp, step=step):
                    g = padded(pos)
                    g = torch.roll(g, step, 3) if step else g                                      # cross_regionW
                    g = g.view(1, C, Hp, w, Wp // w).permute(0, 1, 3, 2, 4).reshape(1, C * w, Hp, Wp // w)      # 'b c h (w group) -> b (c w) h group'
                    return g.permute(0, 2, 3, 1).contiguous()

                def restore_h(pos, h=h, C=C, H=H, W=W, Hp=Hp, Wp=Wp, step=step):
                    g = pos.view(1, Hp // h, Wp, C * h).permute(0, 3, 1, 2)                        # (b, (c h), group, w)
                    g = g.reshape(1, C, Hp, Wp)                                                    # 'b (c h) group w -> b c (h group) w'
                    g = torch.roll(g, -step, 2) if step else g
                    return g[:, :, :H, :W].permute(0, 2, 3, 1).contiguous()

                def restore_w(pos, w=w, C=C, H=H, W=W, Hp=Hp, Wp=Wp, step=step):
                    g = pos.view(1, Hp, Wp // w, C * w).permute(0, 3, 1, 2)                        # (b, (c w), h, group)
                    g = g.view(1, C, w, Hp, Wp // w).permute(0, 1, 3, 2, 4).reshape(1, C, Hp, Wp)  # 'b (c w) h group -> b c h (w group)'
                    g = torch.roll(g, -step, 3) if step else g
                    return g[:, :, :H, :W].permute(0, 2, 3, 1).contiguous()

                key = (H, W, C, h, w, step, hb.padding_type)
                if hb.padding_type in ("reflect", "replicate", "circular"):
                    pass
                th = AG.position_table(region_h, H * W * C, 1, dev, tables, ("rh",) + key)
                tw = AG.position_table(region_w, H * W * C, 1, dev, tables, ("rw",) + key)
                bh = AG.position_table(restore_h, Hp * Wp * C, 1, dev, tables, ("bh",) + key)
                bw = AG.position_table(restore_w, Hp * Wp * C, 1, dev, tables, ("bw",) + key)
                n = ln(t, pre.norm)
                x_h = AG.IndexMap.apply(ff(AG.IndexMap.apply(n, th, B, C * h), hb.proj_h), bh, B, C)
                x_w = AG.IndexMap.apply(ff(AG.IndexMap.apply(n, tw, B, C * w), hb.proj_w), bw, B, C)
                x_c = AG.Linear.apply(n, hb.proj_c.weight, hb.proj_c.bias, None)
                t = AG.ScaleAdd.apply(AG.ScaleAdd.apply(AG.ScaleAdd.apply(x_c, x_h, None), x_w, None), t, None)
                fc1, fc2 = mlp.fn[0], mlp.fn[3]
                t = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(ln(t, mlp.norm), fc1.weight, fc1.bias, None)), fc2.weight, fc2.bias, t)
            if stage.pooling:
                conv = stage.patch_merge[1].reduction[0]
                tab = AG.conv_window_table(H, W, C, 3, 2, 1, dev, tables)
                t = AG.Linear.apply(AG.IndexMap.apply(t, tab, B, 9 * C), conv.weight.permute(0, 2, 3, 1), conv.bias, None)
                H, W = tab.out_hw
        head_ln, head = self.mlp_head[0], self.mlp_head[2]
        logits = AG.Linear.apply(AG.TokenMean.apply(ln(t, head_ln), B, H * W), head.weight, head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        patch, cin, num_classes, patcher_norm = self._cfg
        B, _, H_in, W_in = x.shape
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        cur, H, W, C = self._embed(ws, pk, x.contiguous(), B, H_in, W_in)
        st = None            # (mean, rstd) of cur's rows when the GEMM that wrote cur delivered them (mlpk.h row_part)
        for li, stage in enumerate(self.layers):
            h, w, C, Cout, depth, ef = stage.geom
            for bi in range(len(stage.model)):
                st = self._block(ws, pk, li, bi, stage, cur, B, H, W, st)
            if stage.pooling:
                cur, H, W, _, st = self._merge(ws, pk, li, cur, B, H, W, C)
        C = self.layers[-1].geom[2]
        mean, rstd = st if st is not None else layernorm_stats(ws, cur, B * H * W, C, tag="head.ln")
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C, mean=mean, rstd=rstd, gamma=pk["head.g"], beta=pk["head.be"])
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], num_classes, x.dtype)
