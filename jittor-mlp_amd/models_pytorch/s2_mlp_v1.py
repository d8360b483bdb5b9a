"""S2-MLPv1, drop-in for the reference's models_pytorch/s2_mlp_v1.py.

Block (s2_mlp_v1.py:33-46): x <- x + Linear(Spatial_Shift(gelu(Linear(LN(x))))); x <- x + MLP(LN(x)).
Spatial_Shift (s2_mlp_v1.py:19-25) is spatial_shift1 of v2 applied to the full C-wide tensor; see
s2_mlp_v2.py in this package for the two `shift_mode` semantics.
"""
import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import PreNormResidualMLP, BlockSequential, Holder, channel_mlp, finalize_stats, head_linear, layernorm_stats, stage_embed, pack_channel_mlp
from .s2_mlp_v2 import SHIFT_MODES
from .utils.tools import pair


class PreNormResidual(PreNormResidualMLP):
    """fn(LayerNorm(x)) + x: a parameter container inside a model, callable on its own like the reference's (common.PreNormResidualMLP)."""


class Spatial_Shift(Holder):
    """Parameter-free (s2_mlp_v1.py:15-25); applied as a gather by mlpk_s2_shift.  Callable on (b, w, h, c) like the reference's:
    the argument is overwritten and returned (`shift_mode` as in s2_mlp_v2.SHIFT_MODES)."""
    shift_mode = "reference_inplace"

    def forward(self, x):
        E.require_gpu(x, "Spatial_Shift.forward")
        if x.dim() != 4:
            raise ValueError("expected a (b, w, h, c) tensor")
        b, hh, ww, c = x.shape
        with E.on_device(x):
            src = x.contiguous()
            out = torch.empty_like(src)
            E.s2_shift(src, out, b, hh, ww, c, c, c, {"reference_inplace": N.SHIFT_S2_REF, "shift": N.SHIFT_S2}[self.shift_mode])
        x.copy_(out)
        return x


class S2Block(E.EngineModule):
    def __init__(self, d_model, depth, expansion_factor=4, dropout=0.):
        super().__init__()
        self.model = nn.Sequential(*[BlockSequential(
            PreNormResidual(d_model, nn.Sequential(nn.Linear(d_model, d_model), nn.GELU(), Spatial_Shift(), nn.Linear(d_model, d_model))),
            PreNormResidual(d_model, nn.Sequential(nn.Linear(d_model, d_model * expansion_factor), nn.GELU(), nn.Dropout(dropout),
                                                   nn.Linear(d_model * expansion_factor, d_model), nn.Dropout(dropout))))
            for _ in range(depth)])
        self._dims = (d_model, depth, expansion_factor)

    def _pack_blocks(self, pk, dtype, device, prefix):
        for i, blk in enumerate(self.model):
            p = prefix + "b%d." % i
            sm = blk[0]
            pk[p + "l0.w"], pk[p + "l0.b"], pk[p + "l0.csum"] = E.pack_ln_folded(
                sm.fn[0].weight, sm.fn[0].bias, sm.norm.weight, sm.norm.bias, dtype, device)
            pk[p + "l3.w"] = E.pack_matrix(sm.fn[3].weight, dtype, device)
            pk[p + "l3.b"] = E.f32(sm.fn[3].bias, device)
            mlp = blk[1]
            pack_channel_mlp(pk, p + "mlp.", mlp.norm, mlp.fn[0], mlp.fn[3], dtype, device)

    def _run_blocks(self, ws, pk, x, B, H, W, prefix, mode, only=None):
        C, depth, ef = self._dims
        rows = B * H * W
        nxt = None
        for i in (range(depth) if only is None else only):
            p = prefix + "b%d." % i
            # both LayerNorms of a block read what a GEMM + residual has just written: statistics from those epilogues (mlpk.h row_part)
            mean, rstd = nxt if nxt is not None else layernorm_stats(ws, x, rows, C, tag=prefix + "ln")
            t = ws.get(prefix + "t", (rows, C))
            E.gemm(x, pk[p + "l0.w"], t, rows, C, C, bias=pk[p + "l0.b"], act=N.ACT_GELU, ln=(mean, rstd, pk[p + "l0.csum"]),
                   tag="s2v1_l0")
            ts = ws.get(prefix + "ts", (rows, C))
            E.s2_shift(t, ts, B, H, W, C, C, C, mode)
            got = E.gemm(ts, pk[p + "l3.w"], x, rows, C, C, bias=pk[p + "l3.b"], R=x, res=N.RES_ADD, tag="s2v1_l3", part=(ws, prefix + "l3.part"))
            got = channel_mlp(ws, x, rows, C, pk, p + "mlp.", C * ef, tag=prefix + "cm", stats=finalize_stats(ws, got, rows, C, tag=prefix + "cm.ln"),
                              part=(ws, prefix + "fc2.part"))
            nxt = finalize_stats(ws, got, rows, C, tag=prefix + "ln")
        return x

    shift_mode = "reference_inplace"          # SHIFT_MODES; the enclosing model's set_shift_mode keeps it in step

    def _pack(self, dtype, device):
        pk = {}
        self._pack_blocks(pk, dtype, device, "s.")
        return pk

    def forward(self, x):
        """One stage on its own, as in the reference (s2_mlp_v1.py:48-53): (B, C, H, W) in, the depth blocks on the channel-last view, (B, C, H, W) out.
        The two permutes are the module's boundary (a torch copy each way); inside a model the stage works on the resident channel-last rows."""
        cd = self._resolve(x)
        B, C, H, W = x.shape
        if C != self._dims[0]:
            raise ValueError("this stage works on %d channels" % self._dims[0])
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        buf = ws.get("stage.x", (B * H * W, C))
        buf.view(B, H, W, C).copy_(x.permute(0, 2, 3, 1))
        self._run_blocks(ws, pk, buf, B, H, W, "s.", SHIFT_MODES[self.shift_mode])
        return buf.view(B, H, W, C).permute(0, 3, 1, 2).to(x.dtype).clone()


class S2MLPv1(E.EngineModule):
    """Same signature, defaults and assertions as the reference (s2_mlp_v1.py:55-88)."""

    def __init__(self, image_size=224, patch_size=[7, 2], in_channels=3, num_classes=1000, d_model=[192, 384], depth=[4, 14],
                 expansion_factor=[3, 3]):
        image_size = pair(image_size)
        oldps = [1, 1]
        for ps in patch_size:
            ps = pair(ps)
            assert (image_size[0] % (ps[0] * oldps[0])) == 0, 'image must be divisible by patch size'
            assert (image_size[1] % (ps[1] * oldps[1])) == 0, 'image must be divisible by patch size'
            oldps[0] = oldps[0] * ps[0]
            oldps[1] = oldps[1] * ps[1]
        assert (len(patch_size) == len(depth) == len(d_model) == len(expansion_factor)), \
            'patch_size/depth/d_model/expansion_factor must be a list'
        super().__init__()
        self.stage = len(patch_size)
        self.stages = nn.Sequential(*[nn.Sequential(
            nn.Conv2d(in_channels if i == 0 else d_model[i - 1], d_model[i], kernel_size=patch_size[i], stride=patch_size[i]),
            S2Block(d_model[i], depth[i], expansion_factor[i], dropout=0.)) for i in range(self.stage)])
        self.mlp_head = nn.Sequential(Holder(), nn.Linear(d_model[-1], num_classes))
        self._patches = [pair(p) for p in patch_size]
        self._d_model = list(d_model)
        self._num_classes = num_classes
        self.shift_mode = "reference_inplace"
        for s in range(self.stage):
            for i, blk in enumerate(self.stages[s][1].model):
                blk.__dict__["_owner"] = (self, (s, i))            # lets `model.stages[s][1].model[i](x)` run (common.BlockSequential)

    def _run_single(self, key, x):
        """block `stages[s][1].model[i]` alone on channel-last (B, H, W, C), as calling it does in the reference (s2_mlp_v1.py:47-52)"""
        stage, index = key
        E.require_gpu(x, "S2MLPv1 block")
        E.dtype_code(x.dtype)
        B, H, W, C = x.shape
        if C != self._d_model[stage]:
            raise ValueError("stage %d works on %d channels" % (stage, self._d_model[stage]))
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)     # (C: blocks of different stages can meet at one map size)
            buf = ws.get("blk%d.x" % stage, (B * H * W, C))
            buf.copy_(x.reshape(B * H * W, C))
            self.stages[stage][1]._run_blocks(ws, pk, buf, B, H, W, "s%d." % stage, SHIFT_MODES[self.shift_mode], only=[index])
            return buf.reshape(B, H, W, C).clone()

    def set_shift_mode(self, mode):
        if mode not in SHIFT_MODES:
            raise ValueError("shift_mode must be one of %s" % sorted(SHIFT_MODES))
        self.shift_mode = mode
        return self

    def _pack(self, dtype, device):
        pk = {}
        for s in range(self.stage):
            conv, blk = self.stages[s][0], self.stages[s][1]
            w = conv.weight
            if s > 0:
                w = w.permute(0, 2, 3, 1)
            pk["s%d.embed.w" % s] = E.pack_matrix(w.reshape(w.shape[0], -1), dtype, device)
            pk["s%d.embed.b" % s] = E.f32(conv.bias, device)
            blk._pack_blocks(pk, dtype, device, "s%d." % s)
        pk["head.w"] = E.pack_matrix(self.mlp_head[1].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[1].bias, device)
        return pk

    _train_forward = True

    def _forward_train(self, x):
        """Train mode with autograd (round 6): s2_mlp_v1.py:6-93 as autograd.Functions of `..autograd` (see S2MLPv2._forward_train): the
        Spatial_Shift between the two Linears of the token-mixing sublayer runs in the model's shift_mode forward and returns the gradient the
        reference's autograd returns for the in-place slice assignments (the adjoint of the intended shift)."""
        from .. import autograd as AG
        E.require_gpu(x, "S2MLPv1.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        B, cin, H, W = x.shape
        smear = self.shift_mode == "reference_inplace"
        t = None
        for s in range(self.stage):
            conv, blk = self.stages[s][0], self.stages[s][1]
            ph, pw = self._patches[s]
            if s == 0:
                kp = E.round_up(cin * ph * pw, 4 if cd == torch.float32 else 8)
                with E.on_device(x):
                    patches = torch.zeros((B * (H // ph) * (W // pw), kp), dtype=cd, device=x.device)
                    E.patchify(x.contiguous(), patches, B, cin, H, W, ph, pw, 0, kp)
                t = AG.Linear.apply(patches, conv.weight, conv.bias, None)
            else:
                t = AG.Linear.apply(AG.PatchRowsNHWC.apply(t, B, H, W, ph, pw), conv.weight.permute(0, 2, 3, 1), conv.bias, None)
            H, W = H // ph, W // pw
            for b2 in blk.model:
                pre, mlp = b2[0], b2[1]
                l1, l2 = pre.fn[0], pre.fn[3]
                n = AG.LayerNorm.apply(t, pre.norm.weight, pre.norm.bias, pre.norm.eps)
                y = AG.S2Shift.apply(AG.Gelu.apply(AG.Linear.apply(n, l1.weight, l1.bias, None)), B, H, W, 1, smear)
                t = AG.Linear.apply(y, l2.weight, l2.bias, t)
                n2 = AG.LayerNorm.apply(t, mlp.norm.weight, mlp.norm.bias, mlp.norm.eps)
                fc1, fc2 = mlp.fn[0], mlp.fn[3]
                t = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(n2, fc1.weight, fc1.bias, None)), fc2.weight, fc2.bias, t)
        head = self.mlp_head[1]
        logits = AG.Linear.apply(AG.TokenMean.apply(t, B, H * W), head.weight, head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        B = x.shape[0]
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        mode = SHIFT_MODES[self.shift_mode]
        cur, H, W, C = x.contiguous(), x.shape[2], x.shape[3], x.shape[1]
        for s in range(self.stage):
            cur, H, W = stage_embed(ws, "s%d" % s, cur, B, C, H, W, pk["s%d.embed.w" % s], pk["s%d.embed.b" % s],
                                    self._patches[s], channel_last=s > 0)
            C = self._d_model[s]
            self.stages[s][1]._run_blocks(ws, pk, cur, B, H, W, "s%d." % s, mode)
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C)
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self._num_classes, x.dtype)


def S2MLPv1_deep(num_classes: int = 1000, **kwargs):
    """s2_mlp_v1.py:95-103."""
    return S2MLPv1(image_size=224, patch_size=[16], d_model=[384], depth=[36], num_classes=num_classes,
                   expansion_factor=[4], **kwargs)


def S2MLPv1_wide(num_classes: int = 1000, **kwargs):
    """s2_mlp_v1.py:105-113."""
    return S2MLPv1(image_size=224, patch_size=[16], d_model=[768], depth=[12], num_classes=num_classes,
                   expansion_factor=[4], **kwargs)
