"""S2-MLPv2, drop-in for the reference's models_pytorch/s2_mlp_v2.py.

S2Attention on channel-last (B,H,W,C) (s2_mlp_v2.py:60-69): t = LN(x) A1^T + a1 (3C wide);
x1 = spatial_shift1(t[..., :C]), x2 = spatial_shift2(t[..., C:2C]), x3 = t[..., 2C:];
split attention over (x1,x2,x3); x <- x + (.) A2^T + a2; then the channel MLP.

The two shifts are never materialised: the reduction and the weighted-sum kernels gather the
shifted pixel while loading (mlpk_split_sum / mlpk_split_apply).

Shift semantics.  The reference assigns overlapping slices IN PLACE
(s2_mlp_v2.py:17-20,25-28); with one CPU thread that deterministically "smears" the two "+1"
channel groups (y[i] = x[0]), with several threads it is a data race.  `shift_mode`:
  "reference_inplace" (default) -- the reference's deterministic single-thread result, which is
                                    what the golden logits pin;
  "shift"                        -- the intended one-pixel shift (paper Algorithm 1, Jittor twin).
Both are out-of-place and deterministic here.
"""
import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import (PreNormResidualMLP, BlockSequential, Holder, channel_mlp, finalize_stats, head_linear, layernorm_stats, split_attention_forward,
                     split_attention_weights, standalone_space, stage_embed, pack_channel_mlp)
from .utils.tools import pair

SHIFT_MODES = {"reference_inplace": N.SHIFT_S2_REF, "shift": N.SHIFT_S2}


class PreNormResidual(PreNormResidualMLP):
    """fn(LayerNorm(x)) + x: a parameter container inside a model, callable on its own like the reference's (common.PreNormResidualMLP)."""


def _spatial_shift(x, branch, mode):
    """spatial_shift1 / spatial_shift2 of s2_mlp_v2.py:15-29 on a channel-last (b, w, h, c) tensor, IN PLACE like the reference
    (the argument is overwritten and returned).  The shift itself is the gather mlpk_split_apply applies while loading branch
    `branch` -- called with unit weight on that branch, zero on the others -- so the standalone function and the fused block run the
    same kernel.  mode: SHIFT_MODES ("reference_inplace": the reference's deterministic single-thread result; "shift": the
    intended one-pixel shift)."""
    E.require_gpu(x, "spatial_shift%d" % (branch + 1))
    if x.dim() != 4:
        raise ValueError("expected a (b, w, h, c) tensor")
    b, hh, ww, c = x.shape
    with E.on_device(x):
        src = x.contiguous().view(b * hh * ww, c)
        bar = torch.zeros((b, 3 * c), dtype=torch.float32, device=x.device)
        bar[:, branch * c:(branch + 1) * c] = 1.0
        out = torch.empty_like(src)
        E.split_apply(src, src, src, c, c, c, b, hh, ww, c, SHIFT_MODES[mode], bar, out, c)
    x.copy_(out.view(b, hh, ww, c))
    return x


def spatial_shift1(x, mode="reference_inplace"):
    return _spatial_shift(x, 0, mode)


def spatial_shift2(x, mode="reference_inplace"):
    return _spatial_shift(x, 1, mode)


class SplitAttention(Holder):
    """s2_mlp_v2.py:31-51 (bias-free mlp1/mlp2).  Callable on x_all (b, 3, h, w, c) like the reference's."""

    def forward(self, x_all):
        return split_attention_forward(self, x_all)

    def __init__(self, channel=512, k=3):
        super().__init__()
        self.channel = channel
        self.k = k
        self.mlp1 = nn.Linear(channel, channel, bias=False)
        self.gelu = nn.GELU()
        self.mlp2 = nn.Linear(channel, channel * k, bias=False)
        self.softmax = nn.Softmax(1)


class S2Attention(Holder):
    """s2_mlp_v2.py:53-69.  Callable on (b, h, w, c) like the reference's: mlp1 -> the two shifts as gathers inside the
    split-attention kernels -> mlp2 (no LayerNorm / residual: those belong to the enclosing PreNormResidual)."""
    shift_mode = "reference_inplace"

    def __init__(self, channels=512):
        super().__init__()
        self.mlp1 = nn.Linear(channels, channels * 3)
        self.mlp2 = nn.Linear(channels, channels)
        self.split_attention = SplitAttention(channels)

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError("expected a (b, h, w, c) tensor")
        b, h, w, c = x.shape
        rows = b * h * w
        ws = standalone_space(x)
        mode = SHIFT_MODES[self.shift_mode]
        with E.on_device(x):
            xin = x.contiguous().view(rows, c)
            t = ws.get("t", (rows, 3 * c))
            E.gemm(xin, E.pack_matrix(self.mlp1.weight, x.dtype, x.device), t, rows, 3 * c, c, bias=E.f32(self.mlp1.bias, x.device))
            x0, x1, x2 = t[:, :c], t[:, c:2 * c], t[:, 2 * c:]
            sa = self.split_attention
            bar = split_attention_weights(ws, x0, x1, x2, 3 * c, 3 * c, 3 * c, b, h, w, c, mode, E.pack_matrix(sa.mlp1.weight, torch.float32, x.device),
                                          E.pack_matrix(sa.mlp2.weight, torch.float32, x.device))
            m = ws.get("m", (rows, c))
            E.split_apply(x0, x1, x2, 3 * c, 3 * c, 3 * c, b, h, w, c, mode, bar, m, c)
            out = torch.empty((rows, c), dtype=x.dtype, device=x.device)
            E.gemm(m, E.pack_matrix(self.mlp2.weight, x.dtype, x.device), out, rows, c, c, bias=E.f32(self.mlp2.bias, x.device))
        return out.view(b, h, w, c)


class S2Block(E.EngineModule):
    """depth x [PreNormResidual(S2Attention), PreNormResidual(MLP)] (s2_mlp_v2.py:71-92)."""

    def __init__(self, d_model, depth, expansion_factor=4, dropout=0.):
        super().__init__()
        self.model = nn.Sequential(*[BlockSequential(
            PreNormResidual(d_model, S2Attention(d_model)),
            PreNormResidual(d_model, nn.Sequential(nn.Linear(d_model, d_model * expansion_factor), nn.GELU(), nn.Dropout(dropout),
                                                   nn.Linear(d_model * expansion_factor, d_model), nn.Dropout(dropout))))
            for _ in range(depth)])
        self._dims = (d_model, depth, expansion_factor)

    def _pack_blocks(self, pk, dtype, device, prefix):
        for i, blk in enumerate(self.model):
            p = prefix + "b%d." % i
            att = blk[0].fn
            pk[p + "a1.w"], pk[p + "a1.b"], pk[p + "a1.csum"] = E.pack_ln_folded(
                att.mlp1.weight, att.mlp1.bias, blk[0].norm.weight, blk[0].norm.bias, dtype, device)
            pk[p + "a2.w"] = E.pack_matrix(att.mlp2.weight, dtype, device)
            pk[p + "a2.b"] = E.f32(att.mlp2.bias, device)
            pk[p + "sa.m1"] = E.pack_matrix(att.split_attention.mlp1.weight, torch.float32, device)
            pk[p + "sa.m2"] = E.pack_matrix(att.split_attention.mlp2.weight, torch.float32, device)
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
            t = ws.get(prefix + "t", (rows, 3 * C))
            E.gemm(x, pk[p + "a1.w"], t, rows, 3 * C, C, bias=pk[p + "a1.b"], ln=(mean, rstd, pk[p + "a1.csum"]), tag="s2_mlp1")
            x0, x1, x2 = t[:, :C], t[:, C:2 * C], t[:, 2 * C:]
            bar = split_attention_weights(ws, x0, x1, x2, 3 * C, 3 * C, 3 * C, B, H, W, C, mode, pk[p + "sa.m1"], pk[p + "sa.m2"],
                                          tag=prefix + "sa")
            m = ws.get(prefix + "m", (rows, C))
            E.split_apply(x0, x1, x2, 3 * C, 3 * C, 3 * C, B, H, W, C, mode, bar, m, C)
            got = E.gemm(m, pk[p + "a2.w"], x, rows, C, C, bias=pk[p + "a2.b"], R=x, res=N.RES_ADD, tag="s2_mlp2", part=(ws, prefix + "a2.part"))
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
        """One stage on its own, as in the reference (s2_mlp_v2.py:86-92): (B, C, H, W) in, the depth blocks on the channel-last view, (B, C, H, W) out.
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


class S2MLPv2(E.EngineModule):
    """Same signature, defaults and assertions as the reference (s2_mlp_v2.py:94-127)."""

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

    def set_shift_mode(self, mode):
        if mode not in SHIFT_MODES:
            raise ValueError("shift_mode must be one of %s" % sorted(SHIFT_MODES))
        self.shift_mode = mode
        for m in self.modules():                  # the attention modules and the stages are callable on their own: keep them in step
            if isinstance(m, (S2Attention, S2Block)):
                m.shift_mode = mode
        return self

    def _pack(self, dtype, device):
        pk = {}
        for s in range(self.stage):
            conv, blk = self.stages[s][0], self.stages[s][1]
            w = conv.weight
            if s > 0:                                              # channel-last source: k = (i*pw + j)*Cin + ci
                w = w.permute(0, 2, 3, 1)
            pk["s%d.embed.w" % s] = E.pack_matrix(w.reshape(w.shape[0], -1), dtype, device, kpad=E.embed_kpad(dtype))
            pk["s%d.embed.b" % s] = E.f32(conv.bias, device)
            blk._pack_blocks(pk, dtype, device, "s%d." % s)
        pk["head.w"] = E.pack_matrix(self.mlp_head[1].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[1].bias, device)
        return pk

    def _block_runner(self, s):
        return self.stages[s][1]._run_blocks

    def _run_single(self, key, x):
        return self.forward_block(key[0], key[1], x)

    def forward_block(self, stage, index, x):
        """One block `stages[stage][1].model[index]` on a channel-last activation (B, H, W, C) -> (B, H, W, C): what
        calling that sub-module does in the reference (s2_mlp_v2.py:86-92).  Used for teacher-forced parity checks."""
        E.require_gpu(x, "S2MLPv2.forward_block")
        B, H, W, C = x.shape
        if C != self._d_model[stage]:
            raise ValueError("stage %d works on %d channels" % (stage, self._d_model[stage]))
        cd = self._compute_dtype or x.dtype
        with E.on_device(x):
            pk = self._get_pack(cd, x.device)
            self.__dict__["_in_shape"] = ("block", H, W)
            ws = self._get_space(B, cd, x.device)
            buf = ws.get("blk%d.x" % stage, (B * H * W, C))
            buf.copy_(x.reshape(B * H * W, C))
            self._block_runner(stage)(ws, pk, buf, B, H, W, "s%d." % stage, SHIFT_MODES[self.shift_mode], only=[index])
            return buf.reshape(B, H, W, C).to(x.dtype).clone()

    _train_forward = True

    def _forward_train(self, x):
        """Train mode with autograd (round 6): s2_mlp_v2.py:6-132 as autograd.Functions of `..autograd`, forward and backward through
        the C ABI.  The spatial shifts run on the thirds of mlp1's output IN PLACE of the reference's slice assignments: the forward in the model's
        shift_mode (default: the reference's in-place result), the backward what the reference's autograd returns for those assignments -- the
        adjoint of the INTENDED shift (mlpk_s2_shift2, checked against the reference's own gradients).  Stage convolutions after the first read
        the previous stage's channel-last rows (mlpk_patch_rows_nhwc and its inverse for the gradient that flows back)."""
        from .. import autograd as AG
        E.require_gpu(x, "S2MLPv2.forward")
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
                # channel-last source: columns (i, j, ci) -- the weight viewed in that order (a permute autograd maps back)
                t = AG.Linear.apply(AG.PatchRowsNHWC.apply(t, B, H, W, ph, pw), conv.weight.permute(0, 2, 3, 1), conv.bias, None)
            H, W = H // ph, W // pw
            C = self._d_model[s]
            for b2 in blk.model:
                pre, mlp = b2[0], b2[1]
                att = pre.fn
                n = AG.LayerNorm.apply(t, pre.norm.weight, pre.norm.bias, pre.norm.eps)
                y = AG.Linear.apply(n, att.mlp1.weight, att.mlp1.bias, None)                                 # (rows, 3C)
                x1 = AG.S2Shift.apply(y[:, :C], B, H, W, 1, smear)
                x2 = AG.S2Shift.apply(y[:, C:2 * C], B, H, W, 2, smear)
                a = AG.split_attention(x1, x2, y[:, 2 * C:], att.split_attention, B, H * W)
                t = AG.Linear.apply(a, att.mlp2.weight, att.mlp2.bias, t)
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
            self._block_runner(s)(ws, pk, cur, B, H, W, "s%d." % s, mode)
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C)                 # Reduce('b c h w -> b c', 'mean'), no final norm
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self._num_classes, x.dtype)
