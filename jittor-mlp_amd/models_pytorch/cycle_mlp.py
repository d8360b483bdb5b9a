"""CycleMLP, drop-in for the reference's models_pytorch/cycle_mlp.py: same classes, constructor
signatures, state_dict keys (CycleFC's `offset` buffer included) and CycleMLP_B1..B5 factories, classification head.

CycleFC (cycle_mlp.py:54-131) hands torchvision's deform_conv2d a 1 x 1 kernel and a FIXED integer offset per input channel
(gen_offset, :104-120): channel c is sampled d(c) = (c + k/2) % k - k/2 pixels away along W (`sfc_h`, kernel (1,3)) or along H
(`sfc_w`, kernel (3,1)); bilinear sampling at an integer point is the pixel itself, or 0 outside the map.  So the operator is a
per-channel cyclic pixel shift with zero fill followed by a 1 x 1 convolution: mlpk_cycle_shift writes both gathered copies
of LN(x) in one pass, the convolutions are NT GEMMs.  (torchvision is absent from this image: the sampling rule is restated
from its published algorithm, oracle/functional.py, "parity unpinned" against torchvision itself.)

CycleMLP block (:147-175) on channel-last rows (B*H*W, C):
  h, w, c   = three GEMMs on shift_W(xn), shift_H(xn), xn
  a         = mean over pixels of (h + w + c)        mlpk_split_sum with scale 1 / (H W)
  reweight  = Mlp(C -> C/4 -> 3C), fp32, B rows; its output index is c*3 + k (reshape(B, C, 3), :170): fc2's rows are
              permuted to k*C + c once when packed, so the softmax / weighted sum kernels of ViP / S2-MLPv2 apply as they are
  x        += proj(h a0 + w a1 + c a2)               mlpk_split_apply, GEMM with the residual in its epilogue
then the channel MLP with its LayerNorm folded into fc1.  PatchEmbedOverlapping (7x7 stride 4 pad 2, :261) and Downsample
(3x3 stride 2 pad 1, :220-231) are window gathers (mlpk_im2col) + GEMM; the final LayerNorm is folded into the token mean.
"""
import math
import os

import torch
from torch import nn
from torch.nn import init

from .. import _native as N
from .. import engine as E
from .common import Block, Holder, LinearMlp, StochasticDepth, channel_mlp, finalize_stats, head_linear, layernorm_stats, pack_channel_mlp
from .utils import pair


class Mlp(LinearMlp):
    """cycle_mlp.py:35-51; callable on its own like the reference's (common.LinearMlp); inside a model a parameter container whose weights the
    block packs into its fused channel MLP."""


class CycleFC(Holder):
    """cycle_mlp.py:54-145: parameters and the registered `offset` buffer; same argument checks."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        # the reference's argument contract (cycle_mlp.py:72-79): same conditions, same exception type and message text, because
        # callers match on them
        for ok, what in ((in_channels % groups == 0, 'in_channels must be divisible by groups'),
                         (out_channels % groups == 0, 'out_channels must be divisible by groups'),
                         (stride == 1, 'stride must be 1'), (padding == 0, 'padding must be 0')):
            if not ok:
                raise ValueError(what)
        self.in_channels, self.out_channels, self.kernel_size, self.groups = in_channels, out_channels, kernel_size, groups
        self.stride, self.padding, self.dilation = pair(stride), pair(padding), pair(dilation)
        # state_dict contract: `weight` (out, in / groups, 1, 1), optional `bias` (out), buffer `offset` (1, 2 in, 1, 1)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, 1, 1))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.register_buffer('offset', self.gen_offset())
        self.reset_parameters()

    def reset_parameters(self):
        """nn.Linear's default initialisation, which is what the reference applies (cycle_mlp.py:96-102)."""
        fan_in = self.weight.shape[1] * self.weight.shape[2] * self.weight.shape[3]
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            init.uniform_(self.bias, -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    def gen_offset(self):
        """The (dy, dx) pair of every input channel as the (1, 2 * in_channels, 1, 1) tensor torchvision's deform_conv2d takes
        (values as cycle_mlp.py:104-120 produces them): channel i is displaced by ((i + k // 2) mod k) - k // 2 pixels along the
        one axis on which the kernel is longer than 1 (k its length there), and by 0 along the other."""
        kh, kw = self.kernel_size
        if kh != 1 and kw != 1:
            raise AssertionError(self.kernel_size)
        k = kw if kh == 1 else kh
        step = (torch.arange(self.in_channels) + (kh * kw) // 2) % k - k // 2
        off = torch.zeros(self.in_channels, 2)
        off[:, 1 if kh == 1 else 0] = step.to(off.dtype)
        return off.reshape(1, 2 * self.in_channels, 1, 1)

    def extra_repr(self):
        return '%d, %d, kernel_size=%s' % (self.in_channels, self.out_channels, (self.kernel_size,))


class CycleMLP(Holder):
    """cycle_mlp.py:147-158."""

    def __init__(self, dim, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.mlp_c = nn.Linear(dim, dim, bias=qkv_bias)
        self.sfc_h = CycleFC(dim, dim, (1, 3), 1, 0)
        self.sfc_w = CycleFC(dim, dim, (3, 1), 1, 0)
        self.reweight = Mlp(dim, dim // 4, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class CycleBlock(Block):
    """cycle_mlp.py:178-197 (DropPath is the identity on the forward path built here).  Callable on channel-last (B, H, W, C) like the
    reference's once it sits in a CycleNet."""

    def __init__(self, dim, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, skip_lam=1.0, mlp_fn=CycleMLP):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = mlp_fn(dim, qkv_bias=qkv_bias, qk_scale=None, attn_drop=attn_drop)
        self.drop_path = nn.Identity()                 # DropPath(p): identity in eval mode; train mode: CycleNet._block (round 6)
        self.drop_path_rate = drop_path
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer)
        self.skip_lam = skip_lam


class PatchEmbedOverlapping(Block):
    """cycle_mlp.py:200-217.  Inside a CycleNet (`model.patch_embed`) it runs on its own like the reference's (:213-215): (B, 3, H, W) -> (B, C, H', W');
    round 5."""

    def __init__(self, patch_size=16, stride=16, padding=0, in_chans=3, embed_dim=768, norm_layer=None, groups=1):
        super().__init__()
        patch_size, stride, padding = pair(patch_size), pair(stride), pair(padding)
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=padding, groups=groups)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()


class Downsample(Block):
    """cycle_mlp.py:220-231.  Inside a CycleNet (`model.network[i]`) it runs on its own like the reference's (:227-231): channel-last
    (B, H, W, C) -> (B, H', W', C'); round 5."""

    def __init__(self, in_embed_dim, out_embed_dim, patch_size):
        super().__init__()
        assert patch_size == 2, patch_size
        self.proj = nn.Conv2d(in_embed_dim, out_embed_dim, kernel_size=(3, 3), stride=(2, 2), padding=1)


def basic_blocks(dim, index, layers, mlp_ratio=3., qkv_bias=False, qk_scale=None, attn_drop=0., drop_path_rate=0., skip_lam=1.0,
                 mlp_fn=CycleMLP, **kwargs):
    """cycle_mlp.py:234-245."""
    blocks = []
    for block_idx in range(layers[index]):
        block_dpr = drop_path_rate * (block_idx + sum(layers[:index])) / (sum(layers) - 1)
        blocks.append(CycleBlock(dim, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, drop_path=block_dpr,
                                 skip_lam=skip_lam, mlp_fn=mlp_fn))
    return nn.Sequential(*blocks)


class CycleNet(StochasticDepth, E.EngineModule):
    """train() (round 6): the forward applies the blocks' stochastic depth (cycle_mlp.py:186,194-195: the same DropPath in front
    of both residual additions, then / skip_lam) -- see common.StochasticDepth; forward only, the outputs carry no grad_fn.

    Same signature as the reference (cycle_mlp.py:248-256).  fork_feat=True (round 5; :274-287, :326-334): instead of logits the forward
    returns the list of the four stage outputs, each through its own `norm{0,2,4,6}` LayerNorm (an Identity for the first one under the
    reference's FORK_LAST3 environment switch) and as (B, C, H, W)."""

    _train_forward = True

    def __init__(self, layers, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dims=None, transitions=None,
                 segment_dim=None, mlp_ratios=None, skip_lam=1.0, qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, mlp_fn=CycleMLP, fork_feat=False):
        super().__init__()
        if not fork_feat:
            self.num_classes = num_classes
        self.fork_feat = fork_feat
        self.patch_embed = PatchEmbedOverlapping(patch_size=7, stride=4, padding=2, in_chans=3, embed_dim=embed_dims[0])
        network = []
        for i in range(len(layers)):
            network.append(basic_blocks(embed_dims[i], i, layers, mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias, qk_scale=qk_scale,
                                        attn_drop=attn_drop_rate, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
                                        skip_lam=skip_lam, mlp_fn=mlp_fn))
            if i >= len(layers) - 1:
                break
            if transitions[i] or embed_dims[i] != embed_dims[i + 1]:
                network.append(Downsample(embed_dims[i], embed_dims[i + 1], 2 if transitions[i] else 1))
        self.network = nn.ModuleList(network)
        for si, stage in enumerate(self.network):
            if not isinstance(stage, Downsample):
                for bi, blk in enumerate(stage):
                    blk.__dict__["_owner"] = (self, (si, bi))      # lets `model.network[si][bi](x)` run (common.Block)
            else:
                stage.__dict__["_owner"] = (self, (si, "down"))    # ... `model.network[si](x)` for a Downsample
        self.patch_embed.__dict__["_owner"] = (self, ("embed", None))
        if self.fork_feat:
            self.out_indices = [0, 2, 4, 6]
            for i_emb, i_layer in enumerate(self.out_indices):
                if i_emb == 0 and os.environ.get("FORK_LAST3", None):
                    layer = nn.Identity()                          # cycle_mlp.py:278-283
                else:
                    layer = norm_layer(embed_dims[i_emb])
                self.add_module("norm%d" % i_layer, layer)
        else:
            self.norm = norm_layer(embed_dims[-1])
            self.head = nn.Linear(embed_dims[-1], num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self.cls_init_weights)

    def cls_init_weights(self, m):
        """cycle_mlp.py:296-306."""
        if isinstance(m, nn.Linear):
            init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, CycleFC):
            init.trunc_normal_(m.weight, std=.02)
            nn.init.constant_(m.bias, 0)

    def get_classifier(self):
        return self.head

    # ------------------------------------------------------------------ packing
    def _pack(self, dtype, device):
        pk = {}
        conv = self.patch_embed.proj
        pk["embed.w"] = E.pack_matrix(conv.weight, dtype, device)
        if dtype != torch.float32 and tuple(conv.kernel_size) == (7, 7) and tuple(conv.stride) == (4, 4) and conv.in_channels == 3:
            pk["embed.w7"] = E.pack_stem7(conv.weight, dtype, device)                      # round 6: the stem as a direct convolution (mlpk_stem7)
        pk["embed.b"] = E.f32(conv.bias, device)
        for si, stage in enumerate(self.network):
            if isinstance(stage, Downsample):
                w = stage.proj.weight
                pk["n%d.w" % si] = E.pack_matrix(w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1), dtype, device)   # (i, j, ci)
                pk["n%d.b" % si] = E.f32(stage.proj.bias, device)
                continue
            for bi, blk in enumerate(stage):
                p = "n%d.b%d." % (si, bi)
                att = blk.attn
                C = att.proj.weight.shape[0]
                lam = 1.0 / float(blk.skip_lam)
                pk[p + "ln.g"], pk[p + "ln.b"] = E.f32(blk.norm1.weight, device), E.f32(blk.norm1.bias, device)
                pk[p + "h.w"] = E.pack_matrix(att.sfc_h.weight, dtype, device)
                pk[p + "h.b"] = E.f32(att.sfc_h.bias, device)
                pk[p + "w.w"] = E.pack_matrix(att.sfc_w.weight, dtype, device)
                pk[p + "w.b"] = E.f32(att.sfc_w.bias, device)
                pk[p + "c.w"] = E.pack_matrix(att.mlp_c.weight, dtype, device)
                pk[p + "c.b"] = E.f32(att.mlp_c.bias, device)
                if dtype != torch.float32 and C % 8 == 0:
                    # round 5: mlp_c with norm1 folded in (reads x itself; the shift kernel normalises what it moves): no stored LayerNorm output
                    pk[p + "cf.w"], pk[p + "cf.b"], pk[p + "cf.csum"] = E.pack_ln_folded(att.mlp_c.weight, att.mlp_c.bias, blk.norm1.weight, blk.norm1.bias,
                                                                                       dtype, device)
                pk[p + "r1.w"] = E.pack_matrix(att.reweight.fc1.weight, torch.float32, device)
                pk[p + "r1.b"] = E.f32(att.reweight.fc1.bias, device)
                w2 = att.reweight.fc2.weight.detach()                                               # rows c*3 + k -> k*C + c
                pk[p + "r2.w"] = E.pack_matrix(w2.reshape(C, 3, -1).permute(1, 0, 2).reshape(3 * C, -1), torch.float32, device)
                pk[p + "r2.b"] = E.f32(att.reweight.fc2.bias.detach().reshape(C, 3).t().reshape(-1), device)
                pk[p + "p.w"] = E.pack_matrix(att.proj.weight.detach() * lam, dtype, device)
                pk[p + "p.b"] = E.f32(att.proj.bias.detach() * lam, device)
                pack_channel_mlp(pk, p + "ff.", blk.norm2, blk.mlp.fc1, blk.mlp.fc2, dtype, device)
                if lam != 1.0:
                    pk[p + "ff.fc2.w"] = E.pack_matrix(blk.mlp.fc2.weight.detach() * lam, dtype, device)
                    pk[p + "ff.fc2.b"] = E.f32(blk.mlp.fc2.bias.detach() * lam, device)
        if self.fork_feat:
            for i in self.out_indices:
                nl = getattr(self, "norm%d" % i)
                if isinstance(nl, nn.LayerNorm):
                    pk["fork%d.g" % i], pk["fork%d.be" % i] = E.f32(nl.weight, device), E.f32(nl.bias, device)
            return pk
        pk["head.g"], pk["head.be"] = E.f32(self.norm.weight, device), E.f32(self.norm.bias, device)
        if isinstance(self.head, nn.Linear):
            pk["head.w"] = E.pack_matrix(self.head.weight, dtype, device)
            pk["head.b"] = E.f32(self.head.bias, device)
        return pk

    # ------------------------------------------------------------------ forward
    def _fork_out(self, ws, pk, idx, cur, B, H, W, C, st, dtype):
        """fork_feat: `norm{idx}` of the stage output `cur` (channel-last rows), returned as (B, C, H, W) (cycle_mlp.py:329-332)"""
        rows = B * H * W
        src = cur
        if ("fork%d.g" % idx) in pk:
            mean, rstd = st if st is not None else layernorm_stats(ws, cur, rows, C, tag="fork%d.ln" % idx)
            src = ws.get("fork%d.n" % idx, (rows, C))
            E.norm_apply(cur, rows, C, C, mean=mean, rstd=rstd, gamma=pk["fork%d.g" % idx], beta=pk["fork%d.be" % idx], out_rm=src, ld_rm=C)
        out = torch.empty((B, C, H, W), dtype=src.dtype, device=src.device)
        E.rows_to_nchw(src, B, H * W, C, out)
        return out if out.dtype == dtype else out.to(dtype)

    def _block(self, ws, pk, p, cur, B, H, W, C, hidden, tag, stats=None, rate=0.0):
        """One CycleBlock in place.  `stats` = (mean, rstd) of cur's rows when the GEMM that wrote cur delivered them; returns the
        statistics of the result the same way (or None): both LayerNorms read what a GEMM has just written (mlpk.h row_part)."""
        rows = B * H * W
        mean, rstd = stats if stats is not None else layernorm_stats(ws, cur, rows, C, tag=tag + ".ln")
        sh, sw = ws.get(tag + ".sh", (rows, C)), ws.get(tag + ".sw", (rows, C))
        th, tw, tc = ws.get(tag + ".th", (rows, C)), ws.get(tag + ".tw", (rows, C)), ws.get(tag + ".tc", (rows, C))
        fold = (p + "cf.w") in pk and os.environ.get("MLPK_CYCLE_LN_FOLD") != "0"
        if fold:
            E.cycle_shift_ln(cur, mean, rstd, pk[p + "ln.g"], pk[p + "ln.b"], sh, sw, B, H, W, C, 3, C, C)
        else:
            xn = ws.get(tag + ".xn", (rows, C))
            E.norm_apply(cur, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "ln.g"], beta=pk[p + "ln.b"], out_rm=xn, ld_rm=C)
            E.cycle_shift(xn, sh, sw, B, H, W, C, 3, C, C)
        E.gemm(sh, pk[p + "h.w"], th, rows, C, C, bias=pk[p + "h.b"], tag="cycle_fc")
        E.gemm(sw, pk[p + "w.w"], tw, rows, C, C, bias=pk[p + "w.b"], tag="cycle_fc")
        if fold:
            E.gemm(cur, pk[p + "cf.w"], tc, rows, C, C, bias=pk[p + "cf.b"], ln=(mean, rstd, pk[p + "cf.csum"]), tag="cycle_c")
        else:
            E.gemm(xn, pk[p + "c.w"], tc, rows, C, C, bias=pk[p + "c.b"], tag="cycle_c")
        # reweight: mean over pixels -> Mlp -> softmax over the three branches (fp32, B rows)
        a = ws.get(tag + ".a", (B, C), torch.float32)
        E.split_sum(th, tw, tc, C, C, C, B, H, W, C, N.SHIFT_NONE, a, scale=1.0 / (H * W))
        hid = pk[p + "r1.w"].shape[0]
        hp = pk[p + "r2.w"].shape[1]                                   # hidden padded to whole 16-byte chunks (zero columns)
        t = ws.get(tag + ".t", (B, hp), torch.float32)
        E.gemm(a, pk[p + "r1.w"], t, B, hid, C, ldc=hp, bias=pk[p + "r1.b"], act=N.ACT_GELU)
        hat = ws.get(tag + ".hat", (B, 3 * C), torch.float32)
        E.gemm(t, pk[p + "r2.w"], hat, B, 3 * C, hp, bias=pk[p + "r2.b"])
        bar = ws.get(tag + ".bar", (B, 3 * C), torch.float32)
        E.split_softmax(hat, bar, B, C)
        m = ws.get(tag + ".m", (rows, C))
        E.split_apply(th, tw, tc, C, C, C, B, H, W, C, N.SHIFT_NONE, bar, m, C)
        # train mode: x + drop_path(attn(.)) / skip_lam, x + drop_path(mlp(.)) / skip_lam (cycle_mlp.py:194-195; 1 / skip_lam sits in the packed
        # weights): a per-row scale in the epilogues that add the residuals, two draws per block
        dp1 = self._drop_scale(rate, B, H * W, cur.dtype, cur.device)
        dp2 = self._drop_scale(rate, B, H * W, cur.dtype, cur.device)
        got = E.gemm(m, pk[p + "p.w"], cur, rows, C, C, bias=pk[p + "p.b"], R=cur, res=N.RES_ADD, tag="cycle_proj",
                     part=(ws, tag + ".p.part") if dp1 is None else None, rscale=dp1, rperiod=rows if dp1 is not None else 0)
        got = channel_mlp(ws, cur, rows, C, pk, p + "ff.", hidden, tag=tag + ".cm", stats=finalize_stats(ws, got, rows, C, tag=tag + ".cm.ln"),
                          part=(ws, tag + ".fc2.part"), rscale=dp2)
        return finalize_stats(ws, got, rows, C, tag=tag + ".ln")

    def _run_single(self, key, x):
        """CycleBlock (si, bi) alone on channel-last (B, H, W, C), as `model.network[si][bi](x)` in the reference (cycle_mlp.py:194-197)"""
        si, bi = key
        E.require_gpu(x, "CycleNet inner module")
        E.dtype_code(x.dtype)
        if si == "embed":                                          # PatchEmbedOverlapping (cycle_mlp.py:213-215): 7 x 7 stride-4 pad-2 conv, NCHW out
            if x.dim() != 4 or x.shape[1] != self.patch_embed.proj.in_channels:
                raise ValueError("expected a (B, %d, H, W) tensor" % self.patch_embed.proj.in_channels)
            B, cin, H_in, W_in = x.shape
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("embed", B, H_in, W_in), x.dtype, x.device)
                C = pk["embed.w"].shape[0]
                H, W = (H_in + 4 - 7) // 4 + 1, (W_in + 4 - 7) // 4 + 1
                kp = pk["embed.w"].shape[1]
                cur = ws.get("n0.x", (B * H * W, C))
                if "embed.w7" in pk and x.data_ptr() % 16 == 0 and E.stem7_supported(x.dtype, cur.dtype, cin, H_in, W_in, 2, C):
                    E.stem7(x.contiguous(), pk["embed.w7"], pk["embed.b"], cur, B, H_in, W_in, 2, C)
                else:
                    patches = ws.get("embed.patches", (B * H * W, kp))
                    E.im2col(x.contiguous(), patches, B, cin, H_in, W_in, 7, 7, 4, 4, 2, kp)
                    E.gemm(patches, pk["embed.w"], cur, B * H * W, C, kp, bias=pk["embed.b"])
                return cur.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        if bi == "down":                                           # Downsample (:227-231): 3 x 3 stride-2 pad-1 conv on channel-last
            C = self.network[si].proj.in_channels
            if x.dim() != 4 or x.shape[-1] != C:
                raise ValueError("expected a channel-last (B, H, W, %d) tensor" % C)
            B, H, W, _ = x.shape
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)
                cur = ws.get("blk.x", (B * H * W, C))
                cur.copy_(x.reshape(B * H * W, C))
                Cout = pk["n%d.w" % si].shape[0]
                H2, W2 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
                kp = pk["n%d.w" % si].shape[1]
                cols = ws.get("n%d.cols" % si, (B * H2 * W2, kp))
                E.im2col(cur, cols, B, C, H, W, 3, 3, 2, 2, 1, kp, layout=N.LAYOUT_NHWC, px_stride=C)
                nxt = ws.get("n%d.x" % (si + 1), (B * H2 * W2, Cout))
                E.gemm(cols, pk["n%d.w" % si], nxt, B * H2 * W2, Cout, kp, bias=pk["n%d.b" % si], tag="cycle_down")
                return nxt.reshape(B, H2, W2, Cout).clone()
        blk = self.network[si][bi]
        C = blk.norm1.normalized_shape[0]
        if x.dim() != 4 or x.shape[-1] != C:
            raise ValueError("expected a channel-last (B, H, W, %d) tensor" % C)
        B, H, W, _ = x.shape
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)     # (C: blocks of different stages can meet at one map size)
            cur = ws.get("blk.x", (B * H * W, C))
            cur.copy_(x.reshape(B * H * W, C))
            self._block(ws, pk, "n%d.b%d." % (si, bi), cur, B, H, W, C, blk.mlp.fc1.weight.shape[0], "n%d" % si, rate=blk.drop_path_rate)
            return cur.reshape(B, H, W, C).clone()

    def _forward_train(self, x):
        """Train mode with autograd (round 6): cycle_mlp.py:54-131,163-199,213-226,297-345 as autograd.Functions of `..autograd`, forward
        and backward through the C ABI.  CycleFC = deform_conv2d with a 1 x 1 kernel and fixed integer offsets = a per-channel pixel shift with
        zero fill (an index table at element granularity from CycleFC.offset itself; its inverse is the gradient) followed by mlpk_gemm_nt;
        the reweighting = per-image means, two small Linears, the softmax over the three branches (mlpk_split_softmax + backward) and the weighted
        sum (mlpk_ew_cols); stochastic depth and 1 / skip_lam as per-sample / per-channel scales; Downsample (3 x 3 stride 2) = an
        overlapping-window table + mlpk_gemm_nt.  (deform_conv2d itself: the stand-in caveat of SURVEY 8f-3 applies to the fixture, not to this code.)"""
        from .. import autograd as AG
        E.require_gpu(x, "CycleNet.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        if self.fork_feat:
            raise NotImplementedError("fork_feat=True is a dense-prediction backbone: train() with autograd is built for the classifier")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        B, cin, H_in, W_in = x.shape
        dev = x.device
        H, W = (H_in + 4 - 7) // 4 + 1, (W_in + 4 - 7) // 4 + 1
        kp = E.round_up(cin * 49, 8)                                                     # (mlpk_im2col: rows of whole 16-byte chunks)
        with E.on_device(x):
            patches = torch.zeros((B * H * W, kp), dtype=cd, device=dev)
            E.im2col(x.contiguous(), patches, B, cin, H_in, W_in, 7, 7, 4, 4, 2, kp)
        tables = self.__dict__.setdefault("_tables", {})

        def ln(t, norm):
            return AG.LayerNorm.apply(t, norm.weight, norm.bias, norm.eps)

        def shifted(fc, t, H, W, C):
            off = fc.offset.reshape(C, 2).round().to(torch.int64).cpu()                           # (dy, dx) per input channel (cycle_mlp.py:104-120)

            def fn(pos, H=H, W=W, C=C, off=off):
                g = pos.view(H, W, C)
                yy = torch.arange(H).view(H, 1, 1) + off[:, 0].view(1, 1, C)
                xx = torch.arange(W).view(1, W, 1) + off[:, 1].view(1, 1, C)
                ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
                src = g[yy.clamp(0, H - 1), xx.clamp(0, W - 1), torch.arange(C).view(1, 1, C).expand(H, W, C)]
                return torch.where(ok, src, torch.zeros_like(src))
            tab = AG.position_table(fn, H * W * C, 1, dev, tables, ("cyc", H, W, C, tuple(off.reshape(-1).tolist())))
            return AG.Linear.apply(AG.IndexMap.apply(t, tab, B, C), fc.weight, fc.bias, None)

        pe = self.patch_embed
        t = AG.Linear.apply(patches, pe.proj.weight, pe.proj.bias, None)
        C = pe.proj.weight.shape[0]
        for stage in self.network:
            if isinstance(stage, Downsample):
                tab = AG.conv_window_table(H, W, C, 3, 2, 1, dev, tables)
                t = AG.Linear.apply(AG.IndexMap.apply(t, tab, B, 9 * C), stage.proj.weight.permute(0, 2, 3, 1), stage.proj.bias, None)
                H, W = tab.out_hw
                C = stage.proj.weight.shape[0]
                continue
            for blk in stage:
                at = blk.attn
                lam = None if float(blk.skip_lam) == 1.0 else torch.full((C,), 1.0 / float(blk.skip_lam), dtype=torch.float32, device=dev)
                n = ln(t, blk.norm1)
                xh, xw = shifted(at.sfc_h, n, H, W, C), shifted(at.sfc_w, n, H, W, C)
                xc = AG.Linear.apply(n, at.mlp_c.weight, at.mlp_c.bias, None)
                a = AG.TokenMean.apply(AG.ScaleAdd.apply(AG.ScaleAdd.apply(xh, xw, None), xc, None), B, H * W)            # (h + w + c).mean over pixels
                r = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(a, at.reweight.fc1.weight, at.reweight.fc1.bias, None)),
                                    at.reweight.fc2.weight, at.reweight.fc2.bias, None)                                    # (B, 3C), column c * 3 + k
                hat = r.float().reshape(B, C, 3).permute(0, 2, 1).reshape(B, 3 * C)                                     # -> [B][k][C] (tiny; autograd-native)
                bar = AG.SoftmaxBranches.apply(hat, B, C)
                z = AG.Linear.apply(AG.WeightedSum3.apply(xh, xw, xc, bar, B, H * W).to(cd), at.proj.weight, at.proj.bias, None)
                if lam is not None:
                    z = AG.Affine.apply(z, lam, None)
                t = AG.drop_add(self, t, z, blk.drop_path_rate, B, H * W)
                z = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(ln(t, blk.norm2), blk.mlp.fc1.weight, blk.mlp.fc1.bias, None)),
                                    blk.mlp.fc2.weight, blk.mlp.fc2.bias, None)
                if lam is not None:
                    z = AG.Affine.apply(z, lam, None)
                t = AG.drop_add(self, t, z, blk.drop_path_rate, B, H * W)
        pooled = AG.TokenMean.apply(ln(t, self.norm), B, H * W)
        if not isinstance(self.head, nn.Linear):
            return pooled if pooled.dtype == x.dtype else pooled.to(x.dtype)
        logits = AG.Linear.apply(pooled, self.head.weight, self.head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        B, cin, H_in, W_in = x.shape
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        x = x.contiguous()
        C = pk["embed.w"].shape[0]
        H, W = (H_in + 4 - 7) // 4 + 1, (W_in + 4 - 7) // 4 + 1
        kp = pk["embed.w"].shape[1]
        cur = ws.get("n0.x", (B * H * W, C))
        if "embed.w7" in pk and x.data_ptr() % 16 == 0 and E.stem7_supported(x.dtype, cur.dtype, cin, H_in, W_in, 2, C):
            # round 6: the stem as a direct convolution, delivering the first block's LayerNorm statistics too
            st = (ws.get("n0.ln.mean", (B * H * W,), torch.float32), ws.get("n0.ln.rstd", (B * H * W,), torch.float32))
            E.stem7(x, pk["embed.w7"], pk["embed.b"], cur, B, H_in, W_in, 2, C, out_stats=st, eps=self.network[0][0].norm1.eps)
        else:
            patches = ws.get("embed.patches", (B * H * W, kp))
            E.im2col(x, patches, B, cin, H_in, W_in, 7, 7, 4, 4, 2, kp)
            got = E.gemm(patches, pk["embed.w"], cur, B * H * W, C, kp, bias=pk["embed.b"], part=(ws, "embed.part"))
            st = finalize_stats(ws, got, B * H * W, C, tag="n0.ln")
        outs = []
        for si, stage in enumerate(self.network):
            if self.fork_feat and si > 0 and (si - 1) in self.out_indices:
                outs.append(self._fork_out(ws, pk, si - 1, cur, B, H, W, C, st, x.dtype))
            if isinstance(stage, Downsample):
                Cout = pk["n%d.w" % si].shape[0]
                H2, W2 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
                kp = pk["n%d.w" % si].shape[1]
                nxt = ws.get("n%d.x" % (si + 1), (B * H2 * W2, Cout))
                if kp == 9 * C and E.conv_gemm_nhwc_supported(cur.dtype, C, 3, 3, 2, 1):
                    # round 6: the window is the product's operand loader -- no gathered copy of the map (mlpk_conv_gemm_nhwc)
                    got = E.conv_gemm_nhwc(cur, pk["n%d.w" % si], nxt, B, H, W, C, 3, 3, 2, 1, bias=pk["n%d.b" % si], tag="cycle_down",
                                           part=(ws, "n%d.down.part" % si))
                else:
                    cols = ws.get("n%d.cols" % si, (B * H2 * W2, kp))
                    E.im2col(cur, cols, B, C, H, W, 3, 3, 2, 2, 1, kp, layout=N.LAYOUT_NHWC, px_stride=C)
                    got = E.gemm(cols, pk["n%d.w" % si], nxt, B * H2 * W2, Cout, kp, bias=pk["n%d.b" % si], tag="cycle_down", part=(ws, "n%d.down.part" % si))
                st = finalize_stats(ws, got, B * H2 * W2, Cout, tag="n%d.ln" % (si + 1))
                cur, H, W, C = nxt, H2, W2, Cout
                continue
            for bi, blk in enumerate(stage):
                st = self._block(ws, pk, "n%d.b%d." % (si, bi), cur, B, H, W, C, blk.mlp.fc1.weight.shape[0], "n%d" % si, stats=st,
                                 rate=blk.drop_path_rate)
        if self.fork_feat:
            last = len(self.network) - 1
            if last in self.out_indices:
                outs.append(self._fork_out(ws, pk, last, cur, B, H, W, C, st, x.dtype))
            return outs
        mean, rstd = st if st is not None else layernorm_stats(ws, cur, B * H * W, C, tag="head.ln")
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C, mean=mean, rstd=rstd, gamma=pk["head.g"], beta=pk["head.be"])
        if not isinstance(self.head, nn.Linear):
            out = torch.empty((B, C), dtype=x.dtype, device=x.device)
            E.convert(pooled, out, B * C)
            return out
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self.num_classes, x.dtype)


def _factory(layers, mlp_ratios, embed_dims):
    def make(pretrained=False, **kwargs):
        return CycleNet(layers, embed_dims=embed_dims, patch_size=7, transitions=[True, True, True, True], mlp_ratios=mlp_ratios,
                        mlp_fn=CycleMLP, **kwargs)
    return make


# cycle_mlp.py:352-410
CycleMLP_B1 = _factory([2, 2, 4, 2], [4, 4, 4, 4], [64, 128, 320, 512])
CycleMLP_B2 = _factory([2, 3, 10, 3], [4, 4, 4, 4], [64, 128, 320, 512])
CycleMLP_B3 = _factory([3, 4, 18, 3], [8, 8, 4, 4], [64, 128, 320, 512])
CycleMLP_B4 = _factory([3, 8, 27, 3], [8, 8, 4, 4], [64, 128, 320, 512])
CycleMLP_B5 = _factory([3, 4, 24, 3], [4, 4, 4, 4], [96, 192, 384, 768])
