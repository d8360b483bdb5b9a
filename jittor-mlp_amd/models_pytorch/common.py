"""Building blocks shared by the drop-in models: each helper issues the HIP kernels for one
reference sub-module on channel-last activations held in a Workspace."""
import torch
from torch import nn

from .. import _native as N
from .. import engine as E


class Holder(nn.Module):
    """A parameter container.  The per-block compute runs in the fused HIP kernels issued by the
    owning model's forward(), so calling a holder on its own is not supported."""

    def forward(self, *a, **k):
        raise NotImplementedError("%s only holds parameters; call the enclosing model" % type(self).__name__)


class SubModule(E.EngineModule):
    """Base of the small inner modules the reference lets a caller run on their own (`Aff`, the families' `FeedForward` / `Mlp`): their
    own packed weights and workspace, the same HIP kernels as the enclosing model's forward.  Inside a model they are parameter
    containers: the model packs their parameters into its own fused sequence and never calls them."""

    def _begin(self, x, width, axis=-1):
        E.require_gpu(x, type(self).__name__ + ".forward")
        E.dtype_code(x.dtype)
        if x.dim() < 2 or x.shape[axis] != width:
            raise ValueError("expected %d entries along dimension %d, got a tensor of shape %s" % (width, axis, tuple(x.shape)))
        return self._get_pack(x.dtype, x.device)


class PreNormResidualMLP(SubModule):
    """fn(LayerNorm(x)) + x  (vip.py:6-13, s2_mlp_v2.py:31-38, s2_mlp_v1.py, hire_mlp.py: the families' `PreNormResidual`).  Inside a model
    it is a parameter container (`fn`, `norm`); on its own it is callable like the reference's
      * around the channel MLP  nn.Sequential(Linear, GELU, Dropout, Linear, Dropout):  LayerNorm fold + fc1 + GELU + fc2 + residual through
        the same kernels as the enclosing model's forward (mlpk_gemm_nt pair, or mlpk_channel_mlp for a narrow width);
      * around a module that is itself callable on its own (S2Attention, ParallelWeightedSum, ...): LayerNorm through mlpk_row_stats +
        mlpk_norm_apply, the module, then the residual add."""

    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)

    def _mlp(self):
        fn = self.fn
        if isinstance(fn, nn.Sequential):
            lin = [m for m in fn if isinstance(m, nn.Linear)]
            rest = [m for m in fn if not isinstance(m, (nn.Linear, nn.GELU, nn.Dropout))]
            if len(lin) == 2 and not rest and lin[0].out_features == lin[1].in_features and lin[1].out_features == lin[0].in_features:
                return lin
        return None

    def _pack(self, dtype, device):
        pk = {}
        lin = self._mlp()
        if lin is not None:
            pack_channel_mlp(pk, "", self.norm, lin[0], lin[1], dtype, device)
        else:
            pk["ln.g"], pk["ln.b"] = E.f32(self.norm.weight, device), E.f32(self.norm.bias, device)
        return pk

    def forward(self, x):
        C = self.norm.normalized_shape[0]
        pk = self._begin(x, C)
        rows = x.numel() // C
        with E.on_device(x):
            ws = self._get_space(rows, x.dtype, x.device)
            xb = ws.get("pn.x", (rows, C))
            xb.copy_(x.reshape(rows, C))
            lin = self._mlp()
            if lin is not None:
                channel_mlp(ws, xb, rows, C, pk, "", lin[0].out_features, eps=self.norm.eps)
                return xb.reshape(x.shape).clone()
            mean, rstd = layernorm_stats(ws, xb, rows, C, eps=self.norm.eps)
            xn = ws.get("pn.xn", (rows, C))
            E.norm_apply(xb, rows, C, C, mean=mean, rstd=rstd, gamma=pk["ln.g"], beta=pk["ln.b"], out_rm=xn, ld_rm=C)
            return self.fn(xn.reshape(x.shape)) + x


def two_layer_mlp(ws, pk, xb, rows, dim, hidden, out_dim):
    """out = fc2(gelu(fc1(x))) on channel-last rows xb (rows, K-padded dim): two NT GEMMs, bias + exact GELU in the first epilogue"""
    h = ws.get("mlp.hid", (rows, pk["fc2.w"].shape[1]))
    y = ws.get("mlp.y", (rows, out_dim))
    E.gemm(xb, pk["fc1.w"], h, rows, hidden, xb.shape[1], bias=pk["fc1.b"], act=N.ACT_GELU)
    E.gemm(h, pk["fc2.w"], y, rows, out_dim, h.shape[1], bias=pk["fc2.b"])
    return y


class LinearMlp(SubModule):
    """Linear -> act -> Dropout -> Linear -> Dropout on the last dimension with the members `fc1`, `act`, `fc2`, `drop` (the `Mlp` of
    swin_mlp.py:12-26 and cycle_mlp.py:35-51): callable on its own like the reference's, through two NT GEMMs (GELU in the first epilogue)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def _pack(self, dtype, device):
        return {"fc1.w": E.pack_matrix(self.fc1.weight, dtype, device), "fc1.b": E.f32(self.fc1.bias, device),
                "fc2.w": E.pack_matrix(self.fc2.weight, dtype, device), "fc2.b": E.f32(self.fc2.bias, device)}

    def forward(self, x):
        if not isinstance(self.act, nn.GELU):
            raise NotImplementedError("Mlp runs on its own with the reference's default activation (GELU) only")
        dim, hidden, out = self.fc1.in_features, self.fc1.out_features, self.fc2.out_features
        pk = self._begin(x, dim)
        rows = x.numel() // dim
        with E.on_device(x):
            ws = self._get_space(rows, x.dtype, x.device)
            xb = ws.get("mlp.x", (rows, pk["fc1.w"].shape[1]))          # K zero-padded to whole 16-byte chunks
            xb[:, :dim].copy_(x.reshape(rows, dim))
            return two_layer_mlp(ws, pk, xb, rows, dim, hidden, out).reshape(tuple(x.shape[:-1]) + (out,)).clone()


def standalone_space(x):
    """A workspace for ONE call of an inner module on its own (the sub-block boundary of the reference: g_mlp.py:17-22,
    vip.py:24-57, s2_mlp_v2.py:15-69, as_mlp.py:55-95).  Not a hot path: weights are packed per call."""
    E.require_gpu(x, "sub-module forward")
    return E.Workspace(x.device, x.dtype)


def split_attention_forward(mod, x_all):
    """SplitAttention.forward of vip.py:47-57 / s2_mlp_v2.py:41-51 on x_all (b, 3, h, w, c): whole-image reduction, the two
    bias-free Linears + GELU (fp32), softmax over the three branches, weighted sum -- mlpk_split_sum / mlpk_gemm_nt /
    mlpk_split_softmax / mlpk_split_apply, nothing shifted (MLPK_SHIFT_NONE)."""
    if x_all.dim() != 5 or x_all.shape[1] != 3 or mod.k != 3:
        raise ValueError("expected x_all of shape (b, 3, h, w, c) and k = 3")
    b, _, h, w, c = x_all.shape
    ws = standalone_space(x_all)
    with E.on_device(x_all):
        xs = [x_all[:, i].contiguous().view(b * h * w, c) for i in range(3)]
        m1 = E.pack_matrix(mod.mlp1.weight, torch.float32, x_all.device)
        m2 = E.pack_matrix(mod.mlp2.weight, torch.float32, x_all.device)
        bar = split_attention_weights(ws, xs[0], xs[1], xs[2], c, c, c, b, h, w, c, N.SHIFT_NONE, m1, m2)
        out = torch.empty((b * h * w, c), dtype=x_all.dtype, device=x_all.device)
        E.split_apply(xs[0], xs[1], xs[2], c, c, c, b, h, w, c, N.SHIFT_NONE, bar, out, c)
    return out.view(b, h, w, c)


class Block(Holder):
    """A parameter container that is one WHOLE block of a backbone (the unit the reference's nn.Sequential chains): callable on its
    own like the reference's, through the owning backbone's packed weights and kernels (`backbone._run_single(index, x)`)."""

    def forward(self, x):
        owner = self.__dict__.get("_owner")
        if owner is None:
            return super().forward(x)
        return owner[0]._run_single(owner[1], x)


class BlockSequential(nn.Sequential):
    """The same for the families whose block is a plain nn.Sequential of inner modules in the reference (ViP, S2-MLP): the module
    tree and the state_dict keys are those of nn.Sequential, and calling the block runs it through the owning model's kernels.
    Outside a model it behaves like nn.Sequential (and its parameter-container children raise)."""

    def forward(self, x):
        owner = self.__dict__.get("_owner")
        if owner is None:
            return super().forward(x)
        return owner[0]._run_single(owner[1], x)


def adopt_blocks(backbone, blocks):
    """Tell every block which backbone runs it (a plain reference kept out of the module tree: no extra state_dict keys)."""
    for i, b in enumerate(blocks):
        b.__dict__["_owner"] = (backbone, i)


def embed_patches(ws, name, x, w_packed, bias, cdtype, patch, pad=0, out=None, ln=None):
    """Conv2d(k = stride = patch) on an NCHW input -> channel-last tokens (B*Hp*Wp, Cout).
    mlp_mixer.py:58-60,68-71; conv_mixer.py:18.
    ln = (gamma, beta, eps): the LayerNorm that follows the embedding (swin_mlp.py:332-333, ms_mlp.py:261-262) is applied too -- in the same kernel
    for the 4 x 4 embeddings of three input channels (mlpk_patch_embed4, round 6), by a statistics and a normalise pass otherwise."""
    B, cin, H, W = x.shape
    ph, pw = patch
    hp, wp = (H + 2 * pad - ph) // ph + 1, (W + 2 * pad - pw) // pw + 1
    kp = w_packed.shape[1]
    cout = w_packed.shape[0]
    rows = B * hp * wp
    if out is None:
        out = ws.get(name + ".tokens", (rows, cout))
    if ((ph, pw) == (4, 4) and pad == 0 and out.stride(0) == cout and x.data_ptr() % 16 == 0
            and E.patch_embed4_supported(x.dtype, out.dtype, cin, H, W, cout)):
        g, b_, eps = ln if ln is not None else (None, None, 1e-5)
        E.patch_embed4(x, w_packed, bias, out, B, H, W, cout, gamma=g, beta=b_, eps=eps)
        return out, hp, wp
    patches = ws.get(name + ".patches", (rows, kp))
    E.patchify(x, patches, B, cin, H, W, ph, pw, pad, kp)
    E.gemm(patches, w_packed, out, rows, cout, kp, bias=bias)
    if ln is not None:
        mean, rstd = layernorm_stats(ws, out, rows, cout, tag=name + ".ln", eps=ln[2])
        E.norm_apply(out, rows, cout, cout, mean=mean, rstd=rstd, gamma=ln[0], beta=ln[1], out_rm=out, ld_rm=cout)
    return out, hp, wp


def pack_channel_mlp(pk, prefix, norm, fc1, fc2, dtype, device):
    """Pack LN -> fc1 -> GELU -> fc2 with the LayerNorm folded into fc1 (engine.pack_ln_folded)."""
    pk[prefix + "fc1.w"], pk[prefix + "fc1.b"], pk[prefix + "fc1.csum"] = E.pack_ln_folded(
        fc1.weight, fc1.bias, norm.weight, norm.bias, dtype, device)
    pk[prefix + "fc2.w"] = E.pack_matrix(fc2.weight, dtype, device)
    pk[prefix + "fc2.b"] = E.f32(fc2.bias, device)
    w1 = fc1.weight.reshape(fc1.weight.shape[0], -1)
    if E.linear_gelu_enabled() and w1.shape[1] in (128, 192, 256, 384, 512) and w1.shape[0] % 32 == 0 and dtype in (torch.float16, torch.bfloat16):
        # short K: fc1 + GELU with its rows resident in registers (mlpk_linear_gelu) instead of a GEMM tile with a GELU epilogue
        pk[prefix + "fc1.rr"] = E.pack_linear_gelu(fc1.weight, fc1.bias, dtype, device, norm.weight, norm.bias)
    if E.channel_mlp_fused_supported(dtype, w1.shape[1], w1.shape[0]) and fc2.weight.reshape(fc2.weight.shape[0], -1).shape[0] == w1.shape[1]:
        # narrow stage: the whole block in one kernel (mlpk_channel_mlp), the hidden never written
        pk[prefix + "fused"] = E.pack_channel_mlp_fused(fc1.weight, fc1.bias, fc2.weight, fc2.bias, dtype, device, norm.weight, norm.bias)


def layernorm_stats(ws, x, rows, C, tag="ln", eps=1e-5):
    mean = ws.get(tag + ".mean", (rows,), torch.float32)
    rstd = ws.get(tag + ".rstd", (rows,), torch.float32)
    E.row_stats(x, rows, C, x.stride(0), mean, rstd, eps=eps)
    return mean, rstd


def finalize_stats(ws, got, rows, count, tag="ln", eps=1e-5, group=1):
    """(mean, rstd) of the LayerNorm (group = 1) or per-sample GroupNorm(1,C) (group = H*W rows per sample) that follows a GEMM,
    from the by-product partials its epilogue delivered (`got` = what engine.gemm(part=...) returned); None stays None and the
    caller runs the statistics pass."""
    if got is None:
        return None
    part, n = got
    mean = ws.get(tag + ".mean", (rows // group,), torch.float32)
    rstd = ws.get(tag + ".rstd", (rows // group,), torch.float32)
    E.stats_finalize_planar(part, rows // group, count * group, mean, rstd, eps=eps, group=group)
    return mean, rstd


class StochasticDepth:
    """Train-mode stochastic depth (DropPath) of the hierarchical families, forward only: per sample,
    branch * floor(keep + u) / keep with u uniform in [0, 1) -- timm's drop_path as the reference repository itself restates it
    (conv_mlp.py:17-34; timm is not vendored) -- applied as a per-row scale in the epilogue of the GEMM that adds the residual
    (mlpk.h: v * rscale[m] in front of + R).  The draws come from `drop_path_uniform(B, dtype, device)` (default torch.rand on the
    input's device, one call per DropPath with a rate > 0, in the reference's order; tests replace it with a reference run's draws)."""

    def drop_path_uniform(self, B, dtype, device):
        return torch.rand((B,), dtype=dtype, device=device)

    def _drop_scale(self, rate, B, rows_per_sample, dtype, device):
        """per-row scale of one DropPath call (None: identity -- eval mode or rate 0)"""
        rate = float(rate)
        if not self.training or rate == 0.0:
            return None
        keep = 1.0 - rate
        u = self.drop_path_uniform(B, dtype, device)
        mask = torch.floor(keep + u.reshape(B).float())
        return (mask / keep).repeat_interleave(rows_per_sample).contiguous()


def channel_mlp(ws, x, rows, C, pk, prefix, hidden, *, norm=True, cscale2=None, res_src=None, tag="cm", eps=1e-5, stats=None, part=None, rscale=None):
    """x <- x + fc2(gelu(fc1(LN(x))))   (mlp_mixer.py:38; vip.py:82-88; s2_mlp_v2.py:78-84).
    LN -> row-major normalised copy; fc1 epilogue = bias + exact GELU; fc2 epilogue = bias + residual.
    `stats` = (mean, rstd) of x's rows when the producer of x already delivered them (token_mlp's epilogue, a GEMM's row_part).
    `part` = (workspace, name): fc2's epilogue delivers the row statistics of the new x for the LayerNorm that follows; the
    return value is then what finalize_stats takes (None when they could not be delivered) instead of x.
    `rscale` (rows,) fp32: a per-row scale of the branch in front of the residual addition (train-mode stochastic depth): the two
    GEMMs, the scale in fc2's epilogue."""
    if rscale is None and norm and (cscale2 is None or pk.get(prefix + "fused.scaled")) and (prefix + "fused") in pk and E.channel_mlp_fused_supported(x.dtype, C, hidden):
        mean, rstd = stats if stats is not None else layernorm_stats(ws, x, rows, C, tag=tag + ".ln", eps=eps)
        got = E.channel_mlp_fused(x, rows, C, pk[prefix + "fused"], x, R=res_src if res_src is not None else x, ln=(mean, rstd), part=part)
        return got if part is not None else x
    ln = None
    if norm and (prefix + "fc1.csum") in pk:
        # LayerNorm folded into fc1 (gamma in the weights, beta in the bias, mean/rstd applied on the
        # accumulator): x itself is the GEMM operand, only the row statistics are computed
        mean, rstd = stats if stats is not None else layernorm_stats(ws, x, rows, C, tag=tag + ".ln", eps=eps)
        ln = (mean, rstd, pk[prefix + "fc1.csum"])
        xn = x
    elif norm:
        mean, rstd = layernorm_stats(ws, x, rows, C, tag=tag + ".ln", eps=eps)
        xn = ws.get(tag + ".xn", (rows, C))
        E.norm_apply(x, rows, C, x.stride(0), mean=mean, rstd=rstd, gamma=pk[prefix + "ln.g"], beta=pk[prefix + "ln.b"],
                     out_rm=xn, ld_rm=C)
    else:
        xn = x
    h = ws.get(tag + ".h", (rows, hidden))
    res = res_src if res_src is not None else x
    # Row chunks: fc1 and fc2 of one chunk run back to back, so that the chunk's hidden activations (rows/chunks x hidden,
    # 308 MB for all of Mixer-B/16 at 256 images) are still in the 256 MiB Infinity Cache when fc2 reads them.
    nchunk = E.channel_chunks(rows, hidden * x.element_size())
    step = rows // nchunk
    for c in range(nchunk):
        r0 = c * step
        sl = slice(r0, r0 + step)
        lnc = None if ln is None else (ln[0][sl], ln[1][sl], ln[2])
        if ln is not None and (prefix + "fc1.rr") in pk and E.linear_gelu_supported(x.dtype, step, C, hidden):
            E.linear_gelu(xn[sl], step, C, pk[prefix + "fc1.rr"], h[sl], ln=(ln[0][sl], ln[1][sl]))
        else:
            E.gemm(xn[sl], pk[prefix + "fc1.w"], h[sl], step, hidden, C, bias=pk[prefix + "fc1.b"], act=N.ACT_GELU, ln=lnc, tag="channel_fc1")
        got = E.gemm(h[sl], pk[prefix + "fc2.w"], x[sl], step, C, hidden, bias=pk[prefix + "fc2.b"], cscale=cscale2,
                     R=res[sl], res=N.RES_ADD, tag="channel_fc2", part=part if nchunk == 1 and rscale is None else None,
                     rscale=rscale[sl] if rscale is not None else None, rperiod=step if rscale is not None else 0)
    return got if part is not None else x


def head_linear(ws, pooled, B, C, w, b, num_classes, out_dtype):
    logits = ws.get("logits", (B, num_classes))
    E.gemm(pooled, w, logits, B, num_classes, C, bias=b)
    out = logits
    if out.dtype != out_dtype:
        out = torch.empty((B, num_classes), dtype=out_dtype, device=logits.device)
        E.convert(logits, out, B * num_classes)
        return out
    return logits.clone()


def stage_embed(ws, name, src, B, cin, H, W, w_packed, bias, patch, channel_last):
    """Conv2d(k = stride = patch) at a stage boundary -> channel-last tokens (B*Hp*Wp, Cout).
    `src` is the NCHW input image (channel_last=False) or the previous stage's channel-last
    activation (B*H*W, cin) (channel_last=True; the weight was packed in (i, j, ci) order).
    s2_mlp_v2.py:119; s2_mlp_v1.py:82."""
    ph, pw = patch
    hp, wp = H // ph, W // pw
    kp, cout = w_packed.shape[1], w_packed.shape[0]
    rows = B * hp * wp
    patches = ws.get(name + ".patches", (rows, kp))
    if channel_last:
        E.patchify(src, patches, B, cin, H, W, ph, pw, 0, kp, layout=N.LAYOUT_NHWC, px_stride=src.stride(0))
    else:
        E.patchify(src, patches, B, cin, H, W, ph, pw, 0, kp)
    out = ws.get(name + ".x", (rows, cout))
    E.gemm(patches, w_packed, out, rows, cout, kp, bias=bias)
    return out, hp, wp


def split_attention_weights(ws, x0, x1, x2, ld0, ld1, ld2, B, H, W, C, mode, m1, m2, tag="sa"):
    """bar[b,k,c] of SplitAttention (vip.py:47-53 == s2_mlp_v2.py:41-47): whole-image reduction of the
    three branches (S2 shifts applied on load), then the two bias-free Linears + GELU in fp32 (tiny:
    B rows) and the softmax over the k = 3 branches.  Returns a (B, 3C) float32 tensor."""
    a = ws.get(tag + ".a", (B, C), torch.float32)
    E.split_sum(x0, x1, x2, ld0, ld1, ld2, B, H, W, C, mode, a)
    t = ws.get(tag + ".t", (B, C), torch.float32)
    E.gemm(a, m1, t, B, C, C, act=N.ACT_GELU)
    hat = ws.get(tag + ".hat", (B, 3 * C), torch.float32)
    E.gemm(t, m2, hat, B, 3 * C, C)
    bar = ws.get(tag + ".bar", (B, 3 * C), torch.float32)
    E.split_softmax(hat, bar, B, C)
    return bar
