"""Vision Permutator, drop-in for the reference's models_pytorch/vip.py.

Block on channel-last (B,H,W,C), C = G*s with s = `segments` (vip.py:65-90):
  x^ = LN(x);  xH = Linear_{H*s} over the (h, j) index of x^[b,h,w,g*s+j]  (vip.py:68-72)
               xW = the same over (w, j)                                   (vip.py:73-77)
               xC = Linear_C(x^)                                           (vip.py:78)
  m = split-attention weighted sum of (xH, xW, xC)  (vip.py:24-57)  |  plain sum (vip.py:16-22)
  x <- x + Linear_C(m);   x <- x + MLP(LN(x))
Kernels: one normalise pass writes x^ row-major plus BOTH rearranged copies (each a contiguous
slab staged through LDS, so the einops copies of the reference become coalesced index remaps);
three NT GEMMs; the inverse rearranges; split attention as reduce -> two tiny fp32 GEMMs ->
softmax -> weighted apply; projection GEMM with the residual in its epilogue.
"""
import contextlib
import os
import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import (PreNormResidualMLP, BlockSequential, Holder, adopt_blocks, channel_mlp, embed_patches, finalize_stats, head_linear, layernorm_stats,
                     split_attention_forward, split_attention_weights, standalone_space, pack_channel_mlp)
from .utils.tools import pair


class PreNormResidual(PreNormResidualMLP):
    """fn(LayerNorm(x)) + x: a parameter container inside a model, callable on its own like the reference's (common.PreNormResidualMLP)."""


class ParallelSum(Holder):
    """Sum of branches (vip.py:16-22).  Inside a model a parameter container; on its own (round 6) callable like the reference's on
    (b, h, w, c): the three Permute-MLP branches as NT GEMMs, their sum by the weighted-sum kernel with unit weights (mlpk_split_apply)."""

    def __init__(self, *fns):
        super().__init__()
        self.fns = nn.ModuleList(fns)

    def forward(self, x):
        E.require_gpu(x, "ParallelSum.forward")
        if x.dim() != 4 or len(self.fns) != 3:
            raise NotImplementedError("ParallelSum runs on its own around the three Permute-MLP branches of a (b, h, w, c) tensor")
        b, h, w, c = x.shape
        with E.on_device(x):
            xs = [_branch_forward(f, x, k).reshape(b * h * w, c) for f, k in zip(self.fns, "hwc")]
            ones = torch.ones((b, 3 * c), dtype=torch.float32, device=x.device)
            out = torch.empty((b * h * w, c), dtype=x.dtype, device=x.device)
            E.split_apply(xs[0], xs[1], xs[2], c, c, c, b, h, w, c, N.SHIFT_NONE, ones, out, c)
            return out.view(b, h, w, c)


class ParallelWeightedSum(Holder):
    """Split-attention weighted sum of three branches (vip.py:24-35)."""

    def __init__(self, sa, *fns):
        super().__init__()
        self.fns = nn.ModuleList(fns)
        self.split_attention = sa

    def forward(self, x):
        """vip.py:24-35: the three branches of x (b, h, w, c), stacked, through the split attention."""
        E.require_gpu(x, "ParallelWeightedSum.forward")
        with E.on_device(x):
            xs = [_branch_forward(f, x, k) for f, k in zip(self.fns, "hwc")]
            return self.split_attention(torch.stack(xs, 1))


class SplitAttention(Holder):
    """Bias-free mlp1 (C->C), GELU, mlp2 (C->kC), softmax over k (vip.py:37-57)."""

    def __init__(self, channel=512, k=3):
        super().__init__()
        self.channel = channel
        self.k = k
        self.mlp1 = nn.Linear(channel, channel, bias=False)
        self.gelu = nn.GELU()
        self.mlp2 = nn.Linear(channel, channel * k, bias=False)
        self.softmax = nn.Softmax(1)

    def forward(self, x_all):
        """x_all (b, 3, h, w, c) -> (b, h, w, c), vip.py:47-57."""
        return split_attention_forward(self, x_all)


def _branch_forward(seq, x, which):
    """One Permute-MLP branch on its own (vip.py:69-76): Sequential(Rearrange, Linear, Rearrange) for the h / w branches, a bare
    Linear for the channel branch.  The Linear runs as mlpk_gemm_nt; the two rearranges of a lone branch are plain index
    permutations done by torch here (inside the model they are fused into the normalise pass and the weighted-sum kernel)."""
    b, h, w, c = x.shape
    lin = seq if isinstance(seq, nn.Linear) else seq[1]
    if which == "c":
        rows = x.contiguous().view(b * h * w, c)
    else:
        s = lin.weight.shape[0] // (h if which == "h" else w)
        g = c // s
        v = x.contiguous().view(b, h, w, g, s)
        rows = (v.permute(0, 2, 3, 1, 4) if which == "h" else v.permute(0, 1, 3, 2, 4)).contiguous().view(-1, lin.weight.shape[1])
    out = torch.empty((rows.shape[0], lin.weight.shape[0]), dtype=x.dtype, device=x.device)
    E.gemm(rows, E.pack_matrix(lin.weight, x.dtype, x.device), out, rows.shape[0], lin.weight.shape[0], lin.weight.shape[1], bias=E.f32(lin.bias, x.device))
    if which == "c":
        return out.view(b, h, w, c)
    if which == "h":
        return out.view(b, w, g, h, s).permute(0, 3, 1, 2, 4).contiguous().view(b, h, w, c)
    return out.view(b, h, g, w, s).permute(0, 1, 3, 2, 4).contiguous().view(b, h, w, c)


class _Rearrange(Holder):
    """Parameter-free placeholder keeping the reference's Sequential indices (the Linear sits at .1)."""

    def __init__(self, pattern):
        super().__init__()
        self.pattern = pattern


def _branches(height, width, d_model, segments):
    return (nn.Sequential(_Rearrange('b h w (c s) -> b w c (h s)'), nn.Linear(height * segments, height * segments),
                          _Rearrange('b w c (h s) -> b h w (c s)')),
            nn.Sequential(_Rearrange('b h w (c s) -> b h c (w s)'), nn.Linear(width * segments, width * segments),
                          _Rearrange('b h c (w s) -> b h w (c s)')),
            nn.Linear(d_model, d_model))


def _mlp(d_model, expansion_factor, dropout):
    return nn.Sequential(nn.Linear(d_model, d_model * expansion_factor), nn.GELU(), nn.Dropout(dropout),
                         nn.Linear(d_model * expansion_factor, d_model), nn.Dropout(dropout))


class _PermutatorBase(E.EngineModule):
    weighted = True

    def __init__(self, height, width, d_model, depth, segments, expansion_factor=4, dropout=0.):
        super().__init__()
        blocks = []
        for _ in range(depth):
            if self.weighted:
                mix = ParallelWeightedSum(SplitAttention(d_model, k=3), *_branches(height, width, d_model, segments))
            else:
                mix = ParallelSum(*_branches(height, width, d_model, segments))
            blocks.append(BlockSequential(
                PreNormResidual(d_model, nn.Sequential(mix, nn.Linear(d_model, d_model))),
                PreNormResidual(d_model, _mlp(d_model, expansion_factor, dropout))))
        self.model = nn.Sequential(*blocks)
        self._dims = (height, width, d_model, depth, segments, expansion_factor)
        adopt_blocks(self, self.model)                             # lets `backbone.model[i](x)` run (common.BlockSequential)

    def _pack_blocks(self, pk, dtype, device, prefix=""):
        for i, blk in enumerate(self.model):
            p = prefix + "b%d." % i
            mix, proj = blk[0].fn[0], blk[0].fn[1]
            pk[p + "ln.g"], pk[p + "ln.b"] = E.f32(blk[0].norm.weight, device), E.f32(blk[0].norm.bias, device)
            pk[p + "h.w"] = E.pack_matrix(mix.fns[0][1].weight, dtype, device, kpad=32)
            pk[p + "h.b"] = E.f32(mix.fns[0][1].bias, device)
            pk[p + "w.w"] = E.pack_matrix(mix.fns[1][1].weight, dtype, device, kpad=32)
            pk[p + "w.b"] = E.f32(mix.fns[1][1].bias, device)
            pk[p + "c.w"] = E.pack_matrix(mix.fns[2].weight, dtype, device)
            pk[p + "c.b"] = E.f32(mix.fns[2].bias, device)
            if dtype != torch.float32:
                # the channel branch reads x itself: its LayerNorm is folded into the GEMM (no row-major normalised copy)
                pk[p + "c.wf"], pk[p + "c.bf"], pk[p + "c.csum"] = E.pack_ln_folded(mix.fns[2].weight, mix.fns[2].bias, blk[0].norm.weight,
                                                                                 blk[0].norm.bias, dtype, device)
            if self.weighted:
                pk[p + "sa.m1"] = E.pack_matrix(mix.split_attention.mlp1.weight, torch.float32, device)
                pk[p + "sa.m2"] = E.pack_matrix(mix.split_attention.mlp2.weight, torch.float32, device)
                if dtype != torch.float32:
                    # SplitAttention's a = sum over all pixels of the three branch OUTPUTS (vip.py:49) from sums of the branch
                    # INPUTS: sum_rows(A W^T + b) = (sum_rows A) W^T + rows * b, with the reduced weights taken from the ROUNDED
                    # packed matrices (what the MFMAs multiply).  Two tiny fp32 GEMMs:
                    #   o (B*G, 2 seg) = [Ah | Aw] . W1^T + b1     columns j < seg: the h- and w-branch parts of a at channel (g, j),
                    #                                               columns seg + j: sum over h of Ah = the per-image sum of x^ at (g, j)
                    #   t (B, C) = gelu(o viewed as (B, G*2*seg) . (M1 W2)^T + M1 bc')   W2 = [identity | Wc] per group, so that
                    #   W2 o + bc' IS a; mlp1 (vip.py:51, bias-free) is composed into it at pack time.
                    H, W, C, _, seg, _ = self._dims
                    G = C // seg
                    wh, ww = pk[p + "h.w"].float(), pk[p + "w.w"].float()               # (H*seg, ldh), (W*seg, ldw)
                    ldh, ldw = wh.shape[1], ww.shape[1]
                    w1 = torch.zeros((2 * seg, ldh + ldw), dtype=torch.float32, device=device)
                    w1[:seg, :ldh] = wh.view(H, seg, ldh).sum(0)
                    w1[:seg, ldh:] = ww.view(W, seg, ldw).sum(0)
                    for j in range(seg):
                        w1[seg + j, j:H * seg:seg] = 1.0                              # sum over h of the (h, j) columns of Ah
                    b1 = torch.zeros((2 * seg,), dtype=torch.float32, device=device)
                    b1[:seg] = pk[p + "h.b"].view(H, seg).sum(0) * float(W) + pk[p + "w.b"].view(W, seg).sum(0) * float(H)
                    wc = pk[p + "c.w"].float()                                          # (C, C)
                    w2 = torch.zeros((C, G, 2 * seg), dtype=torch.float64, device=device)
                    eye = torch.eye(C, dtype=torch.float64, device=device).view(C, G, seg)
                    w2[:, :, :seg] = eye
                    w2[:, :, seg:] = wc.double().view(C, G, seg)
                    m1 = pk[p + "sa.m1"].double()
                    pk[p + "sa.w1"], pk[p + "sa.b1"] = w1.contiguous(), b1
                    pk[p + "sa.m1w2"] = (m1 @ w2.view(C, G * 2 * seg)).float().contiguous()
                    pk[p + "sa.m1b"] = (m1 @ (pk[p + "c.b"].double() * float(H * W))).float().contiguous()
            pk[p + "proj.w"] = E.pack_matrix(proj.weight, dtype, device)
            pk[p + "proj.b"] = E.f32(proj.bias, device)
            mlp = blk[1]
            pack_channel_mlp(pk, p + "mlp.", mlp.norm, mlp.fn[0], mlp.fn[3], dtype, device)

    def _pack(self, dtype, device):
        pk = {}
        self._pack_blocks(pk, dtype, device)
        return pk

    def _run_blocks(self, ws, pk, x, B, prefix="", final_stats=False, only=None):
        """final_stats: also return the (mean, rstd) of the rows of the result (for the LayerNorm of the head), or None."""
        H, W, C, depth, seg, ef = self._dims
        rows = B * H * W
        G = C // seg
        hs, wsz = H * seg, W * seg
        ldh, ldw = E.round_up(hs, 32), E.round_up(wsz, 32)
        # both LayerNorms of a block read a tensor a GEMM has just written (proj + residual, fc2 + residual): their statistics
        # come out of those epilogues (mlpk.h row_part) instead of two more passes over x per block
        nxt = None
        for i in (range(depth) if only is None else only):
            p = prefix + "b%d." % i
            mean, rstd = nxt if nxt is not None else layernorm_stats(ws, x, rows, C)
            cfold = (p + "c.csum") in pk
            xn = None if cfold else ws.get("vip.xn", (rows, C))
            # the 16-byte rearrange path (and its by-product sums) also needs its LDS slab to fit: (C / seg) rows of ld_p elements
            # (+ 32 bytes of padding) + gamma / beta -- mlpk_norm_apply's own condition, asked here so that a configuration it cannot
            # take (very tall maps) falls back to the unfused split attention instead of failing
            es = 2 if x.dtype != torch.float32 else 4
            lds_ok = G * (max(ldh, ldw) * es + 32) + 2 * C * 4 <= 160 * 1024
            fused = x.dtype != torch.float32 and seg % 4 == 0 and C % 8 == 0 and lds_ok
            lin = fused and self.weighted
            # by-product sums of the two rearrange passes, side by side: [sum_w x^ as rows (b, g) x columns (h, j) | sum_h x^ ... (w, j)]
            asum = ws.get("sa.sums", (B * G, ldh + ldw), torch.float32) if lin else None
            ldzh, ldzw = E.round_up(hs, 8), E.round_up(wsz, 8)
            zh = ws.get("vip.zh", (B * W * G, ldzh))
            zw = ws.get("vip.zw", (B * H * G, ldzw))
            # round 5: LayerNorm + rearrange + the branch Linear in ONE kernel per branch (mlpk_vip_branch): the rearranged operand is
            # staged in LDS in operand order and multiplied where it lies -- `ph` / `pw` (2 x 201 MB written and read back per block at
            # ViP-S7 / 256 images) do not exist; bit-equal to the two-kernel path below, which stays for the shapes the kernel does not take
            branch = fused and E.vip_branch_supported(x.dtype, H, W, C, seg)
            if branch:
                E.vip_branch(0, x, C, B, H, W, C, seg, mean, rstd, pk[p + "ln.g"], pk[p + "ln.b"], pk[p + "h.w"], pk[p + "h.b"], zh, ldzh,
                             sums=asum[:, ldh:] if lin else None, ld_sum=ldh + ldw)
                E.vip_branch(1, x, C, B, H, W, C, seg, mean, rstd, pk[p + "ln.g"], pk[p + "ln.b"], pk[p + "w.w"], pk[p + "w.b"], zw, ldzw,
                             sums=asum if lin else None, ld_sum=ldh + ldw)
                if xn is not None:
                    E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "ln.g"], beta=pk[p + "ln.b"], out_rm=xn, ld_rm=C)
            else:
                ph = ws.get("vip.ph", (B * W * G, ldh))
                pw = ws.get("vip.pw", (B * H * G, ldw))
                E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "ln.g"], beta=pk[p + "ln.b"], out_rm=xn, ld_rm=C,
                             out_ph=ph, H=H, W=W, seg=seg, ld_p=ldh, sum_ph=asum[:, ldh:] if lin else None, ld_sum=ldh + ldw)
                E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=pk[p + "ln.g"], beta=pk[p + "ln.b"],
                             out_pw=pw, H=H, W=W, seg=seg, ld_p=ldw, sum_pw=asum, ld_sum=ldh + ldw)
            bar, inline, side = None, False, False
            if lin:
                # SplitAttention's weights from the by-product sums: two tiny fp32 GEMMs instead of a 600 MB pass over the three
                # branch outputs, then mlp2 + softmax.  Four latency-bound kernels (~80 us in a row on a few CUs) that depend only
                # on the rearrange passes above: they run on a side stream BESIDE the three branch GEMMs below and are joined in
                # front of the weighted sum.
                # (the chain's buffers are taken BEFORE the side stream is entered: allocated -- and zero-filled -- on the stream
                # that later reads `bar`, never owned by the side stream in the caching allocator's books)
                t = ws.get("sa.t", (B, C), torch.float32)
                o = ws.get("sa.o", (B * G, 2 * seg), torch.float32)
                hat = ws.get("sa.hat", (B, 3 * C), torch.float32)
                bar = ws.get("sa.bar", (B, 3 * C), torch.float32)
                # round 5: with the one-kernel branches the sums exist only after BOTH branch kernels, and what is left beside the chain is one
                # GEMM -- shorter than the chain of three 64 x 64-tile products (25 - 75 us each on a few CUs): it runs IN LINE on the skinny
                # fp32 kernel (algo 16: ~25 us per product on the whole chip) instead; same-box: 28.5 ms with the side stream, 27.5 - 27.9
                # in line, 28.8 - 29.6 for the two-kernel branches (profiles/r05_vip_branch_ab.txt)
                inline = branch and B * G <= 16384 and (ldh + ldw) % 16 == 0 and (G * 2 * seg) % 16 == 0 and C % 16 == 0     # what algo 16 takes
                sk = dict(algo=16) if inline else {}
                side = not inline or os.environ.get("MLPK_VIP_CHAIN_SIDE") == "1"      # experiment: the skinny chain BESIDE the channel-branch GEMM
                chain = E.SideChain(ws, "sa", x.device) if side else contextlib.nullcontext()
                with chain:
                    E.gemm(asum, pk[p + "sa.w1"], o, B * G, 2 * seg, ldh + ldw, bias=pk[p + "sa.b1"], **sk)
                    E.gemm(o.view(B, G * 2 * seg), pk[p + "sa.m1w2"], t, B, C, G * 2 * seg, bias=pk[p + "sa.m1b"], act=N.ACT_GELU, **sk)
                    E.gemm(t, pk[p + "sa.m2"], hat, B, 3 * C, C, **sk)
                    E.split_softmax(hat, bar, B, C)
            if not branch:
                E.gemm(ph, pk[p + "h.w"], zh, B * W * G, hs, ldh, bias=pk[p + "h.b"], tag="vip_h")
                E.gemm(pw, pk[p + "w.w"], zw, B * H * G, wsz, ldw, bias=pk[p + "w.b"], tag="vip_w")
            xc = ws.get("vip.xc", (rows, C))
            if cfold:
                E.gemm(x, pk[p + "c.wf"], xc, rows, C, C, bias=pk[p + "c.bf"], ln=(mean, rstd, pk[p + "c.csum"]), tag="vip_c")
            else:
                E.gemm(xn, pk[p + "c.w"], xc, rows, C, C, bias=pk[p + "c.b"], tag="vip_c")
            m = ws.get("vip.m", (rows, C))
            if fused:
                # the inverse rearranges (vip.py:71,76) are load addresses of the split-attention kernels: xH / xW are never
                # written back in (B,H,W,C) order (two full-tensor passes per block fewer)
                if self.weighted:
                    if side:
                        chain.join()
                else:
                    bar = ws.get("vip.ones", (B, 3 * C), torch.float32, fill=1.0)     # plain sum (vip.py:16-22)
                E.vip_split_apply(zh, zw, xc, ldzh, ldzw, C, B, H, W, C, seg, bar, m, C)
            else:
                xh = ws.get("vip.xh", (rows, C))
                xw = ws.get("vip.xw", (rows, C))
                E.vip_unpermute(0, zh, xh, B, H, W, C, seg, ldzh)
                E.vip_unpermute(1, zw, xw, B, H, W, C, seg, ldzw)
                if self.weighted:
                    bar = split_attention_weights(ws, xh, xw, xc, C, C, C, B, H, W, C, N.SHIFT_NONE, pk[p + "sa.m1"], pk[p + "sa.m2"])
                else:
                    bar = ws.get("vip.ones", (B, 3 * C), torch.float32, fill=1.0)     # plain sum (vip.py:16-22)
                E.split_apply(xh, xw, xc, C, C, C, B, H, W, C, N.SHIFT_NONE, bar, m, C)
            got = E.gemm(m, pk[p + "proj.w"], x, rows, C, C, bias=pk[p + "proj.b"], R=x, res=N.RES_ADD, tag="vip_proj", part=(ws, "vip.proj.part"))
            got = channel_mlp(ws, x, rows, C, pk, p + "mlp.", C * ef, stats=finalize_stats(ws, got, rows, C, tag="cm.ln"),
                              part=(ws, "vip.fc2.part"))
            nxt = finalize_stats(ws, got, rows, C)
        return (x, nxt) if final_stats else x

    def forward(self, x, _only=None):
        """(B,H,W,C) -> (B,H,W,C), as the reference backbones (vip.py:92-93, 127-128)."""
        E.require_gpu(x, type(self).__name__ + ".forward")
        H, W, C = self._dims[:3]
        if x.dim() != 4 or tuple(x.shape[1:]) != (H, W, C):
            raise ValueError("expected (B, %d, %d, %d)" % (H, W, C))
        B = x.shape[0]
        pk = self._get_pack(x.dtype, x.device)
        ws = self._get_space(B, x.dtype, x.device)
        buf = ws.get("x", (B * H * W, C))
        buf.copy_(x.reshape(B * H * W, C))
        self._run_blocks(ws, pk, buf, B, only=_only)
        return buf.reshape(B, H, W, C).clone()

    def _run_single(self, i, x):
        """block i alone on (B, H, W, C): what `backbone.model[i](x)` computes in the reference (vip.py:85-93)"""
        return self(x, _only=[i])


class WeightedPermutator(_PermutatorBase):
    """vip.py:59-93."""
    weighted = True


class Permutator(_PermutatorBase):
    """vip.py:95-128."""
    weighted = False


class ViP(E.EngineModule):
    """Same signature, defaults and assertions as the reference (vip.py:130-148).  Note that the
    all-default call fails its own divisibility assert (256 % 14), exactly like the reference."""

    def __init__(self, image_size=224, patch_size=16, in_channels=3, num_classes=1000, d_model=256, depth=30, segments=14,
                 expansion_factor=4, weighted=True):
        image_size = pair(image_size)
        patch_size = pair(patch_size)
        assert (image_size[0] % patch_size[0]) == 0, 'image must be divisible by patch size'
        assert (image_size[1] % patch_size[1]) == 0, 'image must be divisible by patch size'
        assert (d_model % segments) == 0, 'dimension must be divisible by the number of segments'
        height = image_size[0] // patch_size[0]
        width = image_size[1] // patch_size[1]
        super().__init__()
        self.patcher = nn.Sequential(nn.Conv2d(in_channels, d_model, kernel_size=patch_size, stride=patch_size))
        cls = WeightedPermutator if weighted else Permutator
        self.blocks = cls(height, width, d_model, depth, segments, expansion_factor, dropout=0.)
        self.mlp_head = nn.Sequential(nn.LayerNorm(d_model), _Rearrange('b h w c -> b c (mean)'), nn.Linear(d_model, num_classes))
        self._patch = patch_size
        self._num_classes = num_classes
        self._hw = (height, width)
        self._C = d_model

    def _pack(self, dtype, device):
        pk = {}
        self.blocks._pack_blocks(pk, dtype, device)
        pk["embed.w"] = E.pack_matrix(self.patcher[0].weight, dtype, device, kpad=E.embed_kpad(dtype))
        pk["embed.b"] = E.f32(self.patcher[0].bias, device)
        pk["head.ln.g"], pk["head.ln.b"] = E.f32(self.mlp_head[0].weight, device), E.f32(self.mlp_head[0].bias, device)
        pk["head.w"] = E.pack_matrix(self.mlp_head[2].weight, dtype, device)
        pk["head.b"] = E.f32(self.mlp_head[2].bias, device)
        return pk

    _train_forward = True

    def _forward_train(self, x):
        """Train mode with autograd (round 6): vip.py:7-57,59-128,150-175 as autograd.Functions of `..autograd`, forward and backward
        through the C ABI -- the three branch Linears, the projection and the channel MLP = mlpk_gemm_nt (+ the two GEMMs of their backward),
        the einops rearranges = mlpk_norm_apply(out_ph / out_pw) and mlpk_vip_unpermute (each the other's backward), SplitAttention = per-image
        sums (mlpk_pool_mean / mlpk_broadcast_rows), two small Linears, mlpk_split_softmax (+ _backward) and the weighted sum by mlpk_ew_cols
        (weights' gradient: mlpk_col_dot_seg); ParallelSum (weighted=False) = two mlpk_ew_cols additions."""
        from .. import autograd as AG
        E.require_gpu(x, "ViP.forward")
        if x.dim() != 4:
            raise ValueError("expected a (B, C, H, W) tensor")
        cd = self._compute_dtype or x.dtype
        E.dtype_code(cd)
        H, W = self._hw
        C = self._C
        seg = self.blocks._dims[4]
        B, cin, H_in, W_in = x.shape
        ph, pw = self._patch
        if (H_in // ph, W_in // pw) != (H, W):
            raise ValueError("input size gives a %dx%d grid, the model was built for %dx%d" % (H_in // ph, W_in // pw, H, W))
        conv = self.patcher[0]
        kp = E.round_up(cin * ph * pw, 4 if cd == torch.float32 else 8)
        S = H * W
        with E.on_device(x):
            patches = torch.zeros((B * S, kp), dtype=cd, device=x.device)
            E.patchify(x.contiguous(), patches, B, cin, H_in, W_in, ph, pw, 0, kp)
        t = AG.Linear.apply(patches, conv.weight, conv.bias, None)
        for blk in self.blocks.model:
            pre, mlp = blk[0], blk[1]
            mix, proj = pre.fn[0], pre.fn[1]
            n = AG.LayerNorm.apply(t, pre.norm.weight, pre.norm.bias, pre.norm.eps)
            lh, lw, lc = mix.fns[0][1], mix.fns[1][1], mix.fns[2]
            xh = AG.VipUnpermute.apply(AG.Linear.apply(AG.VipPermute.apply(n, B, H, W, seg, 0), lh.weight, lh.bias, None), B, H, W, C, seg, 0)
            xw = AG.VipUnpermute.apply(AG.Linear.apply(AG.VipPermute.apply(n, B, H, W, seg, 1), lw.weight, lw.bias, None), B, H, W, C, seg, 1)
            xc = AG.Linear.apply(n, lc.weight, lc.bias, None)
            if self.blocks.weighted:
                m = AG.split_attention(xh, xw, xc, mix.split_attention, B, S)
            else:
                m = AG.ScaleAdd.apply(AG.ScaleAdd.apply(xh, xw, None), xc, None)
            t = AG.Linear.apply(m, proj.weight, proj.bias, t)
            n2 = AG.LayerNorm.apply(t, mlp.norm.weight, mlp.norm.bias, mlp.norm.eps)
            fc1, fc2 = mlp.fn[0], mlp.fn[3]
            t = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(n2, fc1.weight, fc1.bias, None)), fc2.weight, fc2.bias, t)
        ln, head = self.mlp_head[0], self.mlp_head[2]
        nf = AG.LayerNorm.apply(t, ln.weight, ln.bias, ln.eps)
        logits = AG.Linear.apply(AG.TokenMean.apply(nf, B, S), head.weight, head.bias, None)
        return logits if logits.dtype == x.dtype else logits.to(x.dtype)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        cd = self._resolve(x)
        H, W = self._hw
        C = self._C
        B = x.shape[0]
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        x = x.contiguous()
        tokens, hp, wp = embed_patches(ws, "embed", x, pk["embed.w"], pk["embed.b"], cd, self._patch,
                                       out=ws.get("x", (B * H * W, C)))
        if (hp, wp) != (H, W):
            raise ValueError("input size gives a %dx%d grid, the model was built for %dx%d" % (hp, wp, H, W))
        rows = B * H * W
        _, st = self.blocks._run_blocks(ws, pk, tokens, B, final_stats=True)
        mean, rstd = st if st is not None else layernorm_stats(ws, tokens, rows, C)
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(tokens, B, H * W, C, C, pooled, C, mean=mean, rstd=rstd, gamma=pk["head.ln.g"], beta=pk["head.ln.b"])
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self._num_classes, x.dtype)
