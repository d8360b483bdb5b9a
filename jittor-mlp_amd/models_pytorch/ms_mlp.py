"""MS-MLP, drop-in for the reference's models_pytorch/ms_mlp.py (eval mode: DropPath is the identity).

MixShiftBlock (ms_mlp.py:11-78) on channel-last activations (B*H*W, C):
  * the channel chunks' rolls along W / H (:56-57), their depthwise k x k convolutions with per-chunk kernel sizes (:60-62),
    the concatenations and the sum of the two branches (:64-67) are ONE stencil kernel (mlpk_mixshift_nhwc): the roll is
    a modular source index, the chunk decides shift and kernel size per channel -- no chunk / roll / cat copy exists;
  * LayerNorm(eps 1e-6) folded into pwconv1 (GEMM, GELU epilogue); pwconv2 GEMM with the layer scale gamma as a per-column
    scale and the block's residual (the block INPUT, not the mixed map) in its epilogue.
Patch embedding and the stage transitions (PatchEmbed with patch_size 2, :178-180) = patch gather + GEMM + LayerNorm;
head = token mean, then LayerNorm on the pooled vector (:352-354), then the classifier GEMM.
"""
import torch
from torch import nn

from .. import _native as N
from .. import engine as E
from .common import Block, Holder, StochasticDepth, SubModule, channel_mlp, embed_patches, finalize_stats, head_linear, layernorm_stats

MS_EPS = 1e-6


def to_2tuple(v):
    return v if isinstance(v, (tuple, list)) else (v, v)


class LayerNorm(SubModule):
    """The reference's own LayerNorm class (ms_mlp.py:273-298): eps 1e-6, parameters `weight`, `bias`.  Callable on its own like the reference's --
    over the last dimension (channels_last) or over dimension 1 of (B, C, H, W) (channels_first) -- through mlpk_row_stats + mlpk_norm_apply;
    inside a model its parameters are folded into the GEMM that follows."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if self.data_format not in ["channels_last", "channels_first"]:
            raise NotImplementedError
        self.normalized_shape = (normalized_shape, )

    def _pack(self, dtype, device):
        return {"g": E.f32(self.weight, device), "b": E.f32(self.bias, device)}

    def forward(self, x):
        C = self.normalized_shape[0]
        first = self.data_format == "channels_first"
        pk = self._begin(x, C, axis=1 if first else -1)
        if first and x.dim() != 4:
            raise ValueError("channels_first expects a (B, C, H, W) tensor")
        with E.on_device(x):
            src = (x.permute(0, 2, 3, 1) if first else x).contiguous()
            rows = src.numel() // C
            ws = self._get_space(rows, x.dtype, x.device)
            mean, rstd = layernorm_stats(ws, src.view(rows, C), rows, C, tag="ln", eps=self.eps)
            out = torch.empty((rows, C), dtype=x.dtype, device=x.device)
            E.norm_apply(src.view(rows, C), rows, C, C, mean=mean, rstd=rstd, gamma=pk["g"], beta=pk["b"], out_rm=out, ld_rm=C)
            out = out.view(src.shape)
            return out.permute(0, 3, 1, 2).contiguous() if first else out


class MixShiftBlock(Block):
    """ms_mlp.py:24-46.  Callable on (B, C, H, W) like the reference's (ms_mlp.py:48-78) once it sits in an MS_MLP."""

    def __init__(self, dim, input_resolution, shift_size, shift_dist, mix_size, layer_scale_init_value=1e-6, mlp_ratio=4, drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.mlp_ratio = mlp_ratio
        self.shift_size = shift_size
        self.shift_dist = shift_dist
        self.chunk_size = [i.shape[0] for i in torch.chunk(torch.zeros(dim), self.shift_size)]
        self.kernel_size = [(ms, ms // 2) for ms in mix_size]
        self.dwconv_lr = nn.ModuleList([nn.Conv2d(cd, cd, kernel_size=ks[0], padding=ks[1], groups=cd)
                                        for cd, ks in zip(self.chunk_size, self.kernel_size)])
        self.dwconv_td = nn.ModuleList([nn.Conv2d(cd, cd, kernel_size=ks[0], padding=ks[1], groups=cd)
                                        for cd, ks in zip(self.chunk_size, self.kernel_size)])
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, int(mlp_ratio * dim))
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(int(mlp_ratio * dim), dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True) if layer_scale_init_value > 0 else None
        self.drop_path = nn.Identity()                 # DropPath(p): identity in eval mode; train mode: MS_MLP.forward (round 6)
        self.drop_path_rate = drop_path


class PatchEmbed(Block):
    """ms_mlp.py:229-253 (also the stage transition, with patch_size 2).  Inside an MS_MLP it runs on its own like the reference's (:255-262):
    (B, C_in, H, W) -> (B, embed_dim, H / p, W / p); round 5."""

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


class BasicLayer(Block):
    """ms_mlp.py:151-181.  Inside an MS_MLP a stage runs on its own like the reference's (:177-185): its blocks, then its PatchEmbed; round 5."""

    def __init__(self, dim, input_resolution, depth, shift_size, shift_dist, mix_size, mlp_ratio=4., drop=0., drop_path=0.,
                 norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            MixShiftBlock(dim=dim, input_resolution=input_resolution, shift_size=shift_size, shift_dist=shift_dist, mix_size=mix_size,
                          mlp_ratio=mlp_ratio, drop=drop, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                          norm_layer=norm_layer) for i in range(depth)])
        if downsample is not None:
            self.downsample = downsample(img_size=input_resolution, patch_size=2, in_chans=dim, embed_dim=2 * dim, norm_layer=norm_layer)
        else:
            self.downsample = None


class MS_MLP(StochasticDepth, E.EngineModule):
    """Same signature and defaults as the reference (ms_mlp.py:300-306).

    train() (round 6): the forward applies the blocks' stochastic depth (ms_mlp.py:46,77: x = input + drop_path(gamma * branch)) --
    see common.StochasticDepth; the LayerNorms have no batch statistics, Dropout has p = 0.  Forward only: the outputs carry no grad_fn."""
    _train_forward = True

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=[2, 2, 6, 2], shift_size=5,
                 shift_dist=[-2, -1, 0, 1, 2], mix_size=[[1, 1, 3, 5, 7], [1, 1, 3, 5, 5], [1, 1, 3, 3, 3], [1, 1, 1, 1, 3]], mlp_ratio=4.,
                 drop_rate=0., drop_path_rate=0.1, norm_layer=LayerNorm, patch_norm=True, use_checkpoint=False, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if self.patch_norm else None)
        patches_resolution = self.patch_embed.patches_resolution
        self.patches_resolution = patches_resolution
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i_layer in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i_layer),
                input_resolution=(patches_resolution[0] // (2 ** i_layer), patches_resolution[1] // (2 ** i_layer)),
                depth=depths[i_layer], shift_size=shift_size, shift_dist=shift_dist, mix_size=mix_size[i_layer], mlp_ratio=self.mlp_ratio,
                drop=drop_rate, drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])], norm_layer=norm_layer,
                downsample=PatchEmbed if (i_layer < self.num_layers - 1) else None, use_checkpoint=use_checkpoint))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        for li, layer in enumerate(self.layers):
            for bi, blk in enumerate(layer.blocks):
                blk.__dict__["_owner"] = (self, (li, bi))          # lets `model.layers[l].blocks[b](x)` run (common.Block)
            layer.__dict__["_owner"] = (self, (li, "layer"))       # ... `model.layers[l](x)`: the blocks, then the downsampling PatchEmbed
            if layer.downsample is not None:
                layer.downsample.__dict__["_owner"] = (self, (li, "down"))
        self.patch_embed.__dict__["_owner"] = (self, ("embed", None))
        self.apply(self._init_weights)

    def _init_weights(self, m):
        # ms_mlp.py:332-339
        if isinstance(m, nn.Linear):
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
        for li, layer in enumerate(self.layers):
            C = layer.dim
            for bi, blk in enumerate(layer.blocks):
                p = "l%d.b%d." % (li, bi)
                kmax = max(k for k, _ in blk.kernel_size)
                for tag, convs in (("lr", blk.dwconv_lr), ("td", blk.dwconv_td)):
                    wt = torch.zeros((kmax * kmax, C), dtype=torch.float32)
                    bs = torch.zeros((C,), dtype=torch.float32)
                    c0 = 0
                    for conv, cs, (k, _) in zip(convs, blk.chunk_size, blk.kernel_size):
                        wt[:k * k, c0:c0 + cs] = conv.weight.detach().float().reshape(cs, k * k).t().cpu()
                        bs[c0:c0 + cs] = conv.bias.detach().float().cpu()
                        c0 += cs
                    pk[p + tag + ".w"], pk[p + tag + ".b"] = wt.to(device).contiguous(), bs.to(device)
                pk[p + "ff.fc1.w"], pk[p + "ff.fc1.b"], pk[p + "ff.fc1.csum"] = E.pack_ln_folded(
                    blk.pwconv1.weight, blk.pwconv1.bias, blk.norm.weight, blk.norm.bias, dtype, device)
                pk[p + "ff.fc2.w"] = E.pack_matrix(blk.pwconv2.weight, dtype, device)
                pk[p + "ff.fc2.b"] = E.f32(blk.pwconv2.bias, device)
                pk[p + "gamma"] = E.f32(blk.gamma, device) if blk.gamma is not None else None
                w1 = blk.pwconv1.weight.reshape(blk.pwconv1.weight.shape[0], -1)
                if E.channel_mlp_fused_supported(dtype, w1.shape[1], w1.shape[0]):
                    # narrow stages: norm + pwconv1 + GELU + pwconv2 + layer scale + residual in one kernel (the scale folded into pwconv2)
                    pk[p + "ff.fused"] = E.pack_channel_mlp_fused(blk.pwconv1.weight, blk.pwconv1.bias, blk.pwconv2.weight, blk.pwconv2.bias,
                                                                 dtype, device, blk.norm.weight, blk.norm.bias, cscale=blk.gamma)
                    pk[p + "ff.fused.scaled"] = True
            if layer.downsample is not None:
                d = layer.downsample
                p = "l%d.down." % li
                # channel-last 2x2 patch gather: K order (i, j, ci)
                pk[p + "w"] = E.pack_matrix(d.proj.weight.detach().permute(0, 2, 3, 1).reshape(d.embed_dim, -1), dtype, device)
                pk[p + "b"] = E.f32(d.proj.bias, device)
                if d.norm is not None:
                    pk[p + "g"], pk[p + "be"] = E.f32(d.norm.weight, device), E.f32(d.norm.bias, device)
        pk["norm.g"], pk["norm.b"] = E.f32(self.norm.weight, device), E.f32(self.norm.bias, device)
        if isinstance(self.head, nn.Linear):
            pk["head.w"] = E.pack_matrix(self.head.weight, dtype, device)
            pk["head.b"] = E.f32(self.head.bias, device)
        return pk

    def _embed(self, ws, pk, x, B, cd):
        """the model's PatchEmbed (ms_mlp.py:255-262) on the NCHW image -> channel-last rows"""
        pe = self.patch_embed
        C = self.embed_dim
        cur, H, W = embed_patches(ws, "embed", x, pk["embed.w"], pk["embed.b"], cd, tuple(pe.patch_size),
                                  out=ws.get("l0.x", (B * pe.patches_resolution[0] * pe.patches_resolution[1], C)),
                                  ln=(pk["embed.g"], pk["embed.be"], MS_EPS) if pe.norm is not None else None)
        return cur, H, W

    def _down(self, ws, pk, li, cur, B, H, W, C):
        """a stage's downsampling PatchEmbed (patch 2) on channel-last rows: 2 x 2 gather + GEMM (+ LayerNorm)"""
        d = self.layers[li].downsample
        p = "l%d.down." % li
        assert H == d.img_size[0] and W == d.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({d.img_size[0]}*{d.img_size[1]})."
        H2, W2, C2 = H // 2, W // 2, d.embed_dim
        kp = pk[p + "w"].shape[1]
        nxt = ws.get("l%d.x" % (li + 1), (B * H2 * W2, C2))
        st = None
        if kp == 4 * C and H % 2 == 0 and W % 2 == 0 and E.conv_gemm_nhwc_supported(cur.dtype, C, 2, 2, 2, 0):
            # round 6: the 2 x 2 window is the product's operand loader (mlpk_conv_gemm_nhwc) -- no gathered copy -- and its epilogue delivers the
            # statistics of the LayerNorm that follows
            got = E.conv_gemm_nhwc(cur, pk[p + "w"], nxt, B, H, W, C, 2, 2, 2, 0, bias=pk[p + "b"], tag="ms_down",
                                   part=(ws, "l%d.down.part" % li) if d.norm is not None else None)
            st = finalize_stats(ws, got, B * H2 * W2, C2, tag="l%d.down.ln" % li, eps=MS_EPS) if d.norm is not None else None
        else:
            cols = ws.get("l%d.cols" % li, (B * H2 * W2, kp))
            E.patchify(cur, cols, B, C, H, W, 2, 2, 0, kp, layout=N.LAYOUT_NHWC, px_stride=C)
            E.gemm(cols, pk[p + "w"], nxt, B * H2 * W2, C2, kp, bias=pk[p + "b"], tag="ms_down")
        if d.norm is not None:
            mean, rstd = st if st is not None else layernorm_stats(ws, nxt, B * H2 * W2, C2, tag="l%d.down.ln" % li, eps=MS_EPS)
            E.norm_apply(nxt, B * H2 * W2, C2, C2, mean=mean, rstd=rstd, gamma=pk[p + "g"], beta=pk[p + "be"], out_rm=nxt, ld_rm=C2)
        return nxt, H2, W2, C2

    def _run_single(self, key, x):
        """An inner module alone on (B, C, H, W), as calling it does in the reference: `model.layers[l].blocks[b](x)` (ms_mlp.py:48-78),
        `model.layers[l](x)` (a stage: :177-185), `model.layers[l].downsample(x)` and `model.patch_embed(x)` (PatchEmbed: :255-262)"""
        li, bi = key
        E.require_gpu(x, "MS_MLP inner module")
        E.dtype_code(x.dtype)
        if li == "embed":
            pe = self.patch_embed
            if x.dim() != 4 or x.shape[1] != pe.in_chans:
                raise ValueError("expected a (B, %d, H, W) tensor" % pe.in_chans)
            B, _, H_in, W_in = x.shape
            assert H_in == pe.img_size[0] and W_in == pe.img_size[1], \
                f"Input image size ({H_in}*{W_in}) doesn't match model ({pe.img_size[0]}*{pe.img_size[1]})."          # ms_mlp.py:256-257
            with E.on_device(x):
                pk = self._get_pack(x.dtype, x.device)
                ws = self._get_space(("embed", B, H_in, W_in), x.dtype, x.device)
                cur, H, W = self._embed(ws, pk, x.contiguous(), B, x.dtype)
                return cur.reshape(B, H, W, self.embed_dim).permute(0, 3, 1, 2).contiguous()
        layer = self.layers[li]
        C = layer.dim
        if x.dim() != 4 or x.shape[1] != C:
            raise ValueError("expected a (B, %d, H, W) tensor" % C)
        B, _, H, W = x.shape
        rows = B * H * W
        with E.on_device(x):
            pk = self._get_pack(x.dtype, x.device)
            ws = self._get_space(("block", B, H, W, C), x.dtype, x.device)     # (C: blocks of different stages can meet at one map size)
            cur = ws.get("blk.x", (rows, C))
            cur.copy_(x.permute(0, 2, 3, 1).reshape(rows, C))                          # channel-last rows, as the stages keep them
            mix = ws.get("blk.mix", (rows, C))
            if bi != "down":
                for b_i, blk in enumerate(layer.blocks):
                    if bi != "layer" and b_i != bi:
                        continue
                    p = "l%d.b%d." % (li, b_i)
                    E.mixshift_nhwc(cur, mix, B, H, W, C, list(blk.shift_dist), [k for k, _ in blk.kernel_size], pk[p + "lr.w"], pk[p + "lr.b"],
                                    pk[p + "td.w"], pk[p + "td.b"])
                    channel_mlp(ws, mix, rows, C, pk, p + "ff.", int(self.mlp_ratio * C), cscale2=pk[p + "gamma"], res_src=cur, tag="blk.cm", eps=MS_EPS,
                                rscale=self._drop_scale(blk.drop_path_rate, B, H * W, x.dtype, x.device))
                    cur, mix = mix, cur
            if bi == "down" or (bi == "layer" and layer.downsample is not None):
                cur, H, W, C = self._down(ws, pk, li, cur, B, H, W, C)
            return cur.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()

    def _forward_train(self, x):
        """Train mode with autograd (round 6): ms_mlp.py:12-84,145-290,352-367 as autograd.Functions of `..autograd`, forward and backward
        through the C ABI.  The per-chunk torch.roll along W resp. H is one index table per direction (mlpk_index_gather at element
        granularity, built by running torch.roll on a tensor of positions; the inverse table is the gradient); the per-chunk depthwise
        convolutions of different sizes run as ONE depthwise convolution whose taps are the chunks' kernels zero-padded to the largest size
        (mlpk_dwconv_plain_nhwc; the padding and concatenation of the small weight tensors are torch views autograd maps back); layer scale =
        mlpk_ew_cols; stochastic depth on drop_path_uniform's draws; the stage transitions (Conv2d 2 x 2 stride 2 + LayerNorm) =
        mlpk_patch_rows_nhwc + mlpk_gemm_nt."""
        import torch.nn.functional as F
        from .. import autograd as AG
        E.require_gpu(x, "MS_MLP.forward")
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
        C = self.embed_dim
        for layer in self.layers:
            for blk in layer.blocks:
                sizes, dist = blk.chunk_size, blk.shift_dist

                def rolled(pos, axis, H=H, W=W, C=C, sizes=sizes, dist=dist):
                    g = pos.view(1, H, W, C).permute(0, 3, 1, 2)                                  # NCHW view of the channel-last positions
                    parts = [torch.roll(xc, sh, axis) for xc, sh in zip(torch.split(g, sizes, 1), dist)]       # ms_mlp.py:52-54
                    return torch.cat(parts, 1).permute(0, 2, 3, 1).contiguous()

                t_lr = AG.position_table(lambda p_: rolled(p_, 3), H * W * C, 1, dev, tables, ("roll", 3, H, W, C, tuple(sizes), tuple(dist)))
                t_td = AG.position_table(lambda p_: rolled(p_, 2), H * W * C, 1, dev, tables, ("roll", 2, H, W, C, tuple(sizes), tuple(dist)))
                kmax = max(ks[0] for ks in blk.kernel_size)

                def taps(convs):
                    ws_, bs_ = [], []
                    for cv in convs:
                        pd = (kmax - cv.weight.shape[-1]) // 2
                        ws_.append(F.pad(cv.weight, [pd, pd, pd, pd]))
                        bs_.append(cv.bias if cv.bias is not None else torch.zeros(cv.weight.shape[0], dtype=cv.weight.dtype, device=cv.weight.device))
                    return torch.cat(ws_, 0), torch.cat(bs_, 0)

                w_lr, b_lr = taps(blk.dwconv_lr)
                w_td, b_td = taps(blk.dwconv_td)
                x_lr = AG.DepthwiseConv.apply(AG.IndexMap.apply(t, t_lr, B, C), w_lr, b_lr, B, H, W)
                x_td = AG.DepthwiseConv.apply(AG.IndexMap.apply(t, t_td, B, C), w_td, b_td, B, H, W)
                n = ln(AG.ScaleAdd.apply(x_lr, x_td, None), blk.norm)
                z = AG.Linear.apply(AG.Gelu.apply(AG.Linear.apply(n, blk.pwconv1.weight, blk.pwconv1.bias, None)), blk.pwconv2.weight, blk.pwconv2.bias, None)
                if blk.gamma is not None:
                    z = AG.Affine.apply(z, blk.gamma, None)
                t = AG.drop_add(self, t, z, blk.drop_path_rate, B, H * W)
            if layer.downsample is not None:
                ds = layer.downsample
                t = AG.Linear.apply(AG.PatchRowsNHWC.apply(t, B, H, W, 2, 2), ds.proj.weight.permute(0, 2, 3, 1), ds.proj.bias, None)
                if ds.norm is not None:
                    t = ln(t, ds.norm)
                H, W, C = H // 2, W // 2, 2 * C
        pooled = ln(AG.TokenMean.apply(t, B, H * W), self.norm)                               # avgpool, THEN the norm (ms_mlp.py:359-361)
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
            f"Input image size ({H_in}*{W_in}) doesn't match model ({pe.img_size[0]}*{pe.img_size[1]})."          # ms_mlp.py:256-257
        pk = self._get_pack(cd, x.device)
        ws = self._get_space(B, cd, x.device)
        C = self.embed_dim
        cur, H, W = self._embed(ws, pk, x.contiguous(), B, cd)
        for li, layer in enumerate(self.layers):
            rows = B * H * W
            hid = int(self.mlp_ratio * C)
            mix = ws.get("l%d.mix" % li, (rows, C))
            for bi, blk in enumerate(layer.blocks):
                p = "l%d.b%d." % (li, bi)
                # round 6: the mix-shift kernel delivers the statistics planes of what it stores -- the block's LayerNorm needs no pass over mix
                got = E.mixshift_nhwc(cur, mix, B, H, W, C, list(blk.shift_dist), [k for k, _ in blk.kernel_size], pk[p + "lr.w"], pk[p + "lr.b"],
                                      pk[p + "td.w"], pk[p + "td.b"], part=(ws, "l%d.mix.part" % li))
                # mix <- cur + gamma * pwconv2(gelu(pwconv1(LN(mix))));  then the roles of the two buffers swap
                # (train mode: ... + drop_path(gamma * .), ms_mlp.py:77 -- a per-row scale in pwconv2's epilogue)
                channel_mlp(ws, mix, rows, C, pk, p + "ff.", hid, cscale2=pk[p + "gamma"], res_src=cur, tag="l%d.cm" % li, eps=MS_EPS,
                            rscale=self._drop_scale(blk.drop_path_rate, B, H * W, cd, x.device),
                            stats=finalize_stats(ws, got, rows, C, tag="l%d.cm.ln" % li, eps=MS_EPS))
                cur, mix = mix, cur
            if layer.downsample is not None:
                cur, H, W, C = self._down(ws, pk, li, cur, B, H, W, C)
        pooled = ws.get("pooled", (B, C))
        E.pool_mean(cur, B, H * W, C, C, pooled, C)
        mean, rstd = layernorm_stats(ws, pooled, B, C, tag="head.ln", eps=MS_EPS)                                   # LayerNorm after the pool
        E.norm_apply(pooled, B, C, C, mean=mean, rstd=rstd, gamma=pk["norm.g"], beta=pk["norm.b"], out_rm=pooled, ld_rm=C)
        if not isinstance(self.head, nn.Linear):
            out = pooled.clone()
            return out if out.dtype == x.dtype else out.to(x.dtype)
        return head_linear(ws, pooled, B, C, pk["head.w"], pk["head.b"], self.num_classes, x.dtype)
