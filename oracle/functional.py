"""CPU restatement of the reference forward path (TEST INFRASTRUCTURE ONLY).

Each function is an explicit-formula restatement of a
reference module; citations are `file:line` relative to /root/reference/.
All functions are pure: they take a `state_dict`-like mapping `sd` (key ->
tensor, the reference's own key names) and float tensors,
compute in the dtype of `x` (float32 or float64) on the CPU and mutate nothing.

Not a product path: see oracle/__init__.py.
"""
import math

import torch

LN_EPS = 1e-5  # nn.LayerNorm / nn.GroupNorm / nn.BatchNorm2d default eps


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def _p(sd, key, like):
    """Fetch a parameter as a CPU tensor of `like`'s dtype."""
    t = sd[key]
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.detach().to(device="cpu", dtype=like.dtype)


def gelu(z):
    """Exact erf GELU, `nn.GELU()` / `F.gelu` defaults (mlp_mixer.py:21, g_mlp.py:35)."""
    return 0.5 * z * (1.0 + torch.erf(z * (1.0 / math.sqrt(2.0))))


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """LayerNorm over the last axis, biased variance (mlp_mixer.py:10,13)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * gamma + beta


def group_norm1(x, gamma, beta, eps=LN_EPS):
    """GroupNorm(1, C) on NCHW: per-sample stats over (C,H,W), per-channel affine
    (as_mlp.py:343-344)."""
    n = x.shape[0]
    flat = x.reshape(n, -1)
    mu = flat.mean(dim=1).view(n, 1, 1, 1)
    xc = x - mu
    var = (xc * xc).reshape(n, -1).mean(dim=1).view(n, 1, 1, 1)
    return xc * torch.rsqrt(var + eps) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def batch_norm_eval(x, sd, prefix):
    """BatchNorm2d in eval mode on NCHW (conv_mixer.py:20,27,31)."""
    g = _p(sd, prefix + ".weight", x).view(1, -1, 1, 1)
    b = _p(sd, prefix + ".bias", x).view(1, -1, 1, 1)
    m = _p(sd, prefix + ".running_mean", x).view(1, -1, 1, 1)
    v = _p(sd, prefix + ".running_var", x).view(1, -1, 1, 1)
    return (x - m) * torch.rsqrt(v + LN_EPS) * g + b


def _pair(v):
    return (v, v) if not isinstance(v, (tuple, list)) else tuple(v)


def patch_embed(x, w, b, padding=0):
    """Conv2d(k = stride = patch) as a patch-gather GEMM; returns channel-last
    (B, Hp, Wp, Cout).  mlp_mixer.py:58-60,68-71 (padding=0) and
    conv_mixer.py:18 (padding = patch//2)."""
    cout, cin, ph, pw = w.shape
    if padding:
        x = torch.nn.functional.pad(x, (padding, padding, padding, padding))
    bsz, _, hh, ww = x.shape
    hp, wp = (hh - ph) // ph + 1, (ww - pw) // pw + 1
    x = x[:, :, : hp * ph, : wp * pw]
    # k index = ci*ph*pw + i*pw + j, matching w.reshape(cout, -1)
    pat = x.reshape(bsz, cin, hp, ph, wp, pw).permute(0, 2, 4, 1, 3, 5).reshape(bsz, hp, wp, cin * ph * pw)
    out = pat @ w.reshape(cout, -1).t()
    if b is not None:
        out = out + b
    return out


def linear(x, w, b=None):
    """nn.Linear: y = x W^T + b with W (out,in)."""
    y = x @ w.t()
    return y if b is None else y + b


def token_linear(x, w, b):
    """Conv1d(k=1) over the token axis of (B,S,C): y[b,t,c] = sum_s W[t,s] x[b,s,c] + b[t]
    (mlp_mixer.py:34,37; g_mlp.py:14,20; res_mlp.py:46,54)."""
    w2 = w.reshape(w.shape[0], w.shape[1])
    return torch.einsum("ts,bsc->btc", w2, x) + b.view(1, -1, 1)


def conv1x1(x, w, b=None):
    """1x1 Conv2d on NCHW (as_mlp.py:11,13,41-44; conv_mixer.py:29)."""
    w2 = w.reshape(w.shape[0], w.shape[1])
    y = torch.einsum("oc,nchw->nohw", w2, x)
    return y if b is None else y + b.view(1, -1, 1, 1)


# --------------------------------------------------------------------------
# MLP-Mixer  (mlp_mixer.py:6-76)
# --------------------------------------------------------------------------
def mixer_block(sd, x, pre):
    """One Mixer block on (B,S,C) (mlp_mixer.py:36-39; Appendix E)."""
    g, b = _p(sd, pre + "0.norm.weight", x), _p(sd, pre + "0.norm.bias", x)
    xh = layer_norm(x, g, b)
    h = gelu(token_linear(xh, _p(sd, pre + "0.fn.net.0.weight", x), _p(sd, pre + "0.fn.net.0.bias", x)))
    x = x + token_linear(h, _p(sd, pre + "0.fn.net.3.weight", x), _p(sd, pre + "0.fn.net.3.bias", x))
    g, b = _p(sd, pre + "1.norm.weight", x), _p(sd, pre + "1.norm.bias", x)
    xh = layer_norm(x, g, b)
    h = gelu(linear(xh, _p(sd, pre + "1.fn.net.0.weight", x), _p(sd, pre + "1.fn.net.0.bias", x)))
    x = x + linear(h, _p(sd, pre + "1.fn.net.3.weight", x), _p(sd, pre + "1.fn.net.3.bias", x))
    return x


def _depth(sd, fmt):
    i = 0
    while (fmt % i) in sd:
        i += 1
    return i


def mixer_forward(sd, x, hooks=None):
    """MLPMixerForImageClassification.forward (mlp_mixer.py:67-76)."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patcher.0.weight", x), _p(sd, "patcher.0.bias", x))
    bsz, hp, wp, c = t.shape
    t = t.reshape(bsz, hp * wp, c)
    for i in range(_depth(sd, "model.%d.0.norm.weight")):
        t = mixer_block(sd, t, "model.%d." % i)
        if hooks is not None:
            hooks("model.%d" % i, t)
    t = layer_norm(t, _p(sd, "active.weight", x), _p(sd, "active.bias", x))
    t = t.mean(dim=1)
    return linear(t, _p(sd, "mlp_head.0.weight", x), _p(sd, "mlp_head.0.bias", x))


# --------------------------------------------------------------------------
# gMLP  (g_mlp.py:10-81)
# --------------------------------------------------------------------------
def gmlp_block(sd, x, pre):
    """gMLPBlock.forward with SpatialGatingUnit (g_mlp.py:17-22, 32-39)."""
    xh = layer_norm(x, _p(sd, pre + "norm.weight", x), _p(sd, pre + "norm.bias", x))
    h = gelu(linear(xh, _p(sd, pre + "channel_proj1.weight", x), _p(sd, pre + "channel_proj1.bias", x)))
    f = h.shape[-1] // 2
    u, v = h[..., :f], h[..., f:]
    v = layer_norm(v, _p(sd, pre + "sgu.norm.weight", x), _p(sd, pre + "sgu.norm.bias", x))
    v = token_linear(v, _p(sd, pre + "sgu.spatial_proj.weight", x), _p(sd, pre + "sgu.spatial_proj.bias", x))
    y = linear(u * v, _p(sd, pre + "channel_proj2.weight", x), _p(sd, pre + "channel_proj2.bias", x))
    return y + x


def gmlp_forward(sd, x, hooks=None):
    """gMLPForImageClassification.forward (g_mlp.py:73-81); no final norm."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patcher.0.weight", x), _p(sd, "patcher.0.bias", x))
    bsz, hp, wp, c = t.shape
    t = t.reshape(bsz, hp * wp, c)
    for i in range(_depth(sd, "model.%d.norm.weight")):
        t = gmlp_block(sd, t, "model.%d." % i)
        if hooks is not None:
            hooks("model.%d" % i, t)
    t = t.mean(dim=1)
    return linear(t, _p(sd, "mlp_head.0.weight", x), _p(sd, "mlp_head.0.bias", x))


# --------------------------------------------------------------------------
# ResMLP  (res_mlp.py:11-99)
# --------------------------------------------------------------------------
def resmlp_block(sd, x, pre):
    """MLPblock.forward: the residual is taken on the affine'd tensor (res_mlp.py:52-57)."""
    x1 = x * _p(sd, pre + "pre_affine.alpha", x) + _p(sd, pre + "pre_affine.beta", x)
    x2 = x1 + _p(sd, pre + "gamma_1", x) * token_linear(
        x1, _p(sd, pre + "token_mix.weight", x), _p(sd, pre + "token_mix.bias", x))
    x3 = x2 * _p(sd, pre + "post_affine.alpha", x) + _p(sd, pre + "post_affine.beta", x)
    h = gelu(linear(x3, _p(sd, pre + "ff.net.0.weight", x), _p(sd, pre + "ff.net.0.bias", x)))
    y = linear(h, _p(sd, pre + "ff.net.3.weight", x), _p(sd, pre + "ff.net.3.bias", x))
    return x3 + _p(sd, pre + "gamma_2", x) * y


def resmlp_forward(sd, x, hooks=None):
    """ResMLPForImageClassification.forward (res_mlp.py:91-99); `affine.*` is unused (:86)."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patcher.0.weight", x), _p(sd, "patcher.0.bias", x))
    bsz, hp, wp, c = t.shape
    t = t.reshape(bsz, hp * wp, c)
    for i in range(_depth(sd, "model.%d.gamma_1")):
        t = resmlp_block(sd, t, "model.%d." % i)
        if hooks is not None:
            hooks("model.%d" % i, t)
    t = t.mean(dim=1)
    return linear(t, _p(sd, "mlp_head.0.weight", x), _p(sd, "mlp_head.0.bias", x))


# --------------------------------------------------------------------------
# split attention (vip.py:37-57 == s2_mlp_v2.py:31-51)
# --------------------------------------------------------------------------
def split_attention(x1, x2, x3, m1, m2):
    """SplitAttention.forward on three (B,H,W,C) tensors; mlp1/mlp2 are bias-free.
    hat_a flat index = k*C + c (vip.py:52)."""
    bsz, hh, ww, c = x1.shape
    a = (x1 + x2 + x3).reshape(bsz, -1, c).sum(dim=1)            # (B,C)  vip.py:50
    hat = linear(gelu(linear(a, m1)), m2).reshape(bsz, 3, c)     # vip.py:51-52
    bar = torch.softmax(hat, dim=1)                              # vip.py:53
    return (bar[:, 0].view(bsz, 1, 1, c) * x1 + bar[:, 1].view(bsz, 1, 1, c) * x2
            + bar[:, 2].view(bsz, 1, 1, c) * x3)


# --------------------------------------------------------------------------
# ViP  (vip.py:59-171)
# --------------------------------------------------------------------------
def vip_permute_h(x, s):
    """Rearrange('b h w (c s) -> b w c (h s)') (vip.py:69):
    y[b,w,g,h*s+j] = x[b,h,w,g*s+j]."""
    b, h, w, c = x.shape
    g = c // s
    return x.reshape(b, h, w, g, s).permute(0, 2, 3, 1, 4).reshape(b, w, g, h * s)


def vip_unpermute_h(y, s):
    """Rearrange('b w c (h s) -> b h w (c s)') (vip.py:71)."""
    b, w, g, hs = y.shape
    h = hs // s
    return y.reshape(b, w, g, h, s).permute(0, 3, 1, 2, 4).reshape(b, h, w, g * s)


def vip_permute_w(x, s):
    """Rearrange('b h w (c s) -> b h c (w s)') (vip.py:74):
    y[b,h,g,w*s+j] = x[b,h,w,g*s+j]."""
    b, h, w, c = x.shape
    g = c // s
    return x.reshape(b, h, w, g, s).permute(0, 1, 3, 2, 4).reshape(b, h, g, w * s)


def vip_unpermute_w(y, s):
    """Rearrange('b h c (w s) -> b h w (c s)') (vip.py:76)."""
    b, h, g, ws = y.shape
    w = ws // s
    return y.reshape(b, h, g, w, s).permute(0, 1, 3, 2, 4).reshape(b, h, w, g * s)


def vip_block(sd, x, pre, segments, weighted):
    """One (Weighted)Permutator block on (B,H,W,C) (vip.py:65-90 / 100-125)."""
    p0 = pre + "0.fn.0."
    xh = layer_norm(x, _p(sd, pre + "0.norm.weight", x), _p(sd, pre + "0.norm.bias", x))
    zh = linear(vip_permute_h(xh, segments), _p(sd, p0 + "fns.0.1.weight", x), _p(sd, p0 + "fns.0.1.bias", x))
    x_h = vip_unpermute_h(zh, segments)
    zw = linear(vip_permute_w(xh, segments), _p(sd, p0 + "fns.1.1.weight", x), _p(sd, p0 + "fns.1.1.bias", x))
    x_w = vip_unpermute_w(zw, segments)
    x_c = linear(xh, _p(sd, p0 + "fns.2.weight", x), _p(sd, p0 + "fns.2.bias", x))
    if weighted:
        m = split_attention(x_h, x_w, x_c, _p(sd, p0 + "split_attention.mlp1.weight", x),
                            _p(sd, p0 + "split_attention.mlp2.weight", x))
    else:
        m = x_h + x_w + x_c                                       # ParallelSum vip.py:16-22
    x = x + linear(m, _p(sd, pre + "0.fn.1.weight", x), _p(sd, pre + "0.fn.1.bias", x))
    xh = layer_norm(x, _p(sd, pre + "1.norm.weight", x), _p(sd, pre + "1.norm.bias", x))
    h = gelu(linear(xh, _p(sd, pre + "1.fn.0.weight", x), _p(sd, pre + "1.fn.0.bias", x)))
    return x + linear(h, _p(sd, pre + "1.fn.3.weight", x), _p(sd, pre + "1.fn.3.bias", x))


def vip_forward(sd, x, segments, hooks=None):
    """ViP.forward (vip.py:166-171); head = LN -> mean(h,w) -> Linear (vip.py:160-164)."""
    x = x.detach().cpu()
    weighted = "blocks.model.0.0.fn.0.split_attention.mlp1.weight" in sd
    t = patch_embed(x, _p(sd, "patcher.0.weight", x), _p(sd, "patcher.0.bias", x))   # (B,H,W,C)
    for i in range(_depth(sd, "blocks.model.%d.0.norm.weight")):
        t = vip_block(sd, t, "blocks.model.%d." % i, segments, weighted)
        if hooks is not None:
            hooks("blocks.model.%d" % i, t)
    t = layer_norm(t, _p(sd, "mlp_head.0.weight", x), _p(sd, "mlp_head.0.bias", x))
    t = t.mean(dim=(1, 2))
    return linear(t, _p(sd, "mlp_head.2.weight", x), _p(sd, "mlp_head.2.bias", x))


# --------------------------------------------------------------------------
# S2-MLP shifts  (s2_mlp_v2.py:15-29; s2_mlp_v1.py:19-25)
# --------------------------------------------------------------------------
def _shift_axis(x, axis, direction, mode):
    """Shift one channel group of a (b, d1, d2, c') tensor by one along `axis`.

    direction=+1 is the reference's `x[:,1:] = x[:,:n-1]`, -1 its `x[:,:n-1] = x[:,1:]`.
    The reference assigns IN PLACE on overlapping views.  mode:
      'shift'              -- the intended semantics (paper Algorithm 1, Jittor twin):
                              +1: y[i] = x[i-1] (y[0] = x[0]);  -1: y[i] = x[i+1] (y[n-1] = x[n-1]).
      'reference_inplace'  -- what the PyTorch reference deterministically computes with
                              one CPU thread: +1 groups smear,
                              y[i] = x[0] for all i;  -1 groups equal the clean shift.
    """
    n = x.shape[axis]
    idx = torch.arange(n)
    if direction < 0:
        src = torch.clamp(idx + 1, max=n - 1)
    elif mode == "shift":
        src = torch.clamp(idx - 1, min=0)
    elif mode == "reference_inplace":
        src = torch.zeros_like(idx)
    else:
        raise ValueError("unknown S2 shift mode %r" % (mode,))
    return x.index_select(axis, src)


def _s2_groups(c):
    """Channel group boundaries exactly as the slices of s2_mlp_v2.py:17-20."""
    return [(0, c // 4), (c // 4, c // 2), (c // 2, c * 3 // 4), (3 * c // 4, c)]


def spatial_shift1(x, mode="reference_inplace"):
    """s2_mlp_v2.py:15-21 / s2_mlp_v1.py:19-25 (out of place): groups -> (dim1,+1),(dim1,-1),(dim2,+1),(dim2,-1)."""
    c = x.shape[-1]
    plan = [(1, +1), (1, -1), (2, +1), (2, -1)]
    out = x.clone()
    for (lo, hi), (axis, d) in zip(_s2_groups(c), plan):
        if hi > lo:
            out[..., lo:hi] = _shift_axis(x[..., lo:hi], axis, d, mode)
    return out


def spatial_shift2(x, mode="reference_inplace"):
    """s2_mlp_v2.py:23-29: groups -> (dim2,+1),(dim2,-1),(dim1,+1),(dim1,-1)."""
    c = x.shape[-1]
    plan = [(2, +1), (2, -1), (1, +1), (1, -1)]
    out = x.clone()
    for (lo, hi), (axis, d) in zip(_s2_groups(c), plan):
        if hi > lo:
            out[..., lo:hi] = _shift_axis(x[..., lo:hi], axis, d, mode)
    return out


# --------------------------------------------------------------------------
# S2-MLPv2  (s2_mlp_v2.py:53-132)
# --------------------------------------------------------------------------
def s2v2_block(sd, x, pre, mode):
    """One S2Block layer on (B,H,W,C): S2Attention + MLP (s2_mlp_v2.py:60-69, 76-85)."""
    c = x.shape[-1]
    p0 = pre + "0.fn."
    xh = layer_norm(x, _p(sd, pre + "0.norm.weight", x), _p(sd, pre + "0.norm.bias", x))
    t = linear(xh, _p(sd, p0 + "mlp1.weight", x), _p(sd, p0 + "mlp1.bias", x))          # (B,H,W,3C)
    x1 = spatial_shift1(t[..., :c], mode)
    x2 = spatial_shift2(t[..., c:2 * c], mode)
    x3 = t[..., 2 * c:]
    a = split_attention(x1, x2, x3, _p(sd, p0 + "split_attention.mlp1.weight", x),
                        _p(sd, p0 + "split_attention.mlp2.weight", x))
    x = x + linear(a, _p(sd, p0 + "mlp2.weight", x), _p(sd, p0 + "mlp2.bias", x))
    xh = layer_norm(x, _p(sd, pre + "1.norm.weight", x), _p(sd, pre + "1.norm.bias", x))
    h = gelu(linear(xh, _p(sd, pre + "1.fn.0.weight", x), _p(sd, pre + "1.fn.0.bias", x)))
    return x + linear(h, _p(sd, pre + "1.fn.3.weight", x), _p(sd, pre + "1.fn.3.bias", x))


def _s2_stages(sd, x, block_fn, hooks):
    t = x                                                       # NCHW into each stage conv
    s = 0
    while ("stages.%d.0.weight" % s) in sd:
        t = patch_embed(t, _p(sd, "stages.%d.0.weight" % s, x), _p(sd, "stages.%d.0.bias" % s, x))  # -> NHWC
        for i in range(_depth(sd, "stages.%d" % s + ".1.model.%d.0.norm.weight")):
            t = block_fn(t, "stages.%d.1.model.%d." % (s, i))
            if hooks is not None:
                hooks("stages.%d.1.model.%d" % (s, i), t)
        t = t.permute(0, 3, 1, 2)                               # back to NCHW (s2_mlp_v2.py:91)
        s += 1
    t = t.mean(dim=(2, 3))                                      # Reduce('b c h w -> b c','mean') :125
    return linear(t, _p(sd, "mlp_head.1.weight", x), _p(sd, "mlp_head.1.bias", x))


def s2mlpv2_forward(sd, x, mode="reference_inplace", hooks=None):
    """S2MLPv2.forward (s2_mlp_v2.py:129-132); no final LayerNorm."""
    x = x.detach().cpu()
    return _s2_stages(sd, x, lambda t, pre: s2v2_block(sd, t, pre, mode), hooks)


# --------------------------------------------------------------------------
# S2-MLPv1  (s2_mlp_v1.py:15-93)
# --------------------------------------------------------------------------
def s2v1_block(sd, x, pre, mode):
    """Linear -> GELU -> Spatial_Shift -> Linear, then MLP (s2_mlp_v1.py:33-46)."""
    xh = layer_norm(x, _p(sd, pre + "0.norm.weight", x), _p(sd, pre + "0.norm.bias", x))
    t = gelu(linear(xh, _p(sd, pre + "0.fn.0.weight", x), _p(sd, pre + "0.fn.0.bias", x)))
    t = spatial_shift1(t, mode)
    x = x + linear(t, _p(sd, pre + "0.fn.3.weight", x), _p(sd, pre + "0.fn.3.bias", x))
    xh = layer_norm(x, _p(sd, pre + "1.norm.weight", x), _p(sd, pre + "1.norm.bias", x))
    h = gelu(linear(xh, _p(sd, pre + "1.fn.0.weight", x), _p(sd, pre + "1.fn.0.bias", x)))
    return x + linear(h, _p(sd, pre + "1.fn.3.weight", x), _p(sd, pre + "1.fn.3.bias", x))


def s2mlpv1_forward(sd, x, mode="reference_inplace", hooks=None):
    """S2MLPv1.forward (s2_mlp_v1.py:90-93)."""
    x = x.detach().cpu()
    return _s2_stages(sd, x, lambda t, pre: s2v1_block(sd, t, pre, mode), hooks)


# --------------------------------------------------------------------------
# AS-MLP  (as_mlp.py; utils/shift_cuda.py)
# --------------------------------------------------------------------------
def axial_shift_nchw(x, kernel_size, dim):
    """The `Shift` op's forward kernel formula (shift_cuda.py:44-72, group :119):
    group = ceil(C/k); g = c // group; s = k//2 - g;
    out[n,c,h,w] = in[n,c,h+s,w] (dim 2) or in[n,c,h,w+s] (dim 3), zero out of range.
    kernel_size == 1 is the identity (shift_cuda.py:188-189)."""
    assert dim in (2, 3)
    assert kernel_size % 2 == 1
    if kernel_size == 1:
        return x
    n, c, h, w = x.shape
    group = int(math.ceil(c / kernel_size))
    out = torch.zeros_like(x)
    for g in range((c + group - 1) // group):
        lo, hi = g * group, min(c, (g + 1) * group)
        s = kernel_size // 2 - g
        length = h if dim == 2 else w
        # destination index i reads source i+s, valid when 0 <= i+s < length
        d0, d1 = max(0, -s), min(length, length - s)
        if d1 <= d0:
            continue
        if dim == 2:
            out[:, lo:hi, d0:d1, :] = x[:, lo:hi, d0 + s:d1 + s, :]
        else:
            out[:, lo:hi, :, d0:d1] = x[:, lo:hi, :, d0 + s:d1 + s]
    return out


def axial_shift_nchw_backward(grad_out, kernel_size, dim):
    """The `Shift` op's backward kernel formula (shift_cuda.py:75-103, launched from _shift.backward :131-162):
    grad_in[n,c,h,w] = grad_out[n,c,h-s,w] (dim 2) or grad_out[n,c,h,w-s] (dim 3), zero out of range, with the forward's
    group = ceil(C/k), g = c // group, s = k//2 - g -- the adjoint of the forward gather."""
    assert dim in (2, 3)
    assert kernel_size % 2 == 1
    if kernel_size == 1:
        return grad_out
    n, c, h, w = grad_out.shape
    group = int(math.ceil(c / kernel_size))
    out = torch.zeros_like(grad_out)
    for g in range((c + group - 1) // group):
        lo, hi = g * group, min(c, (g + 1) * group)
        s = kernel_size // 2 - g
        length = h if dim == 2 else w
        # destination index i reads source i - s, valid when 0 <= i - s < length
        d0, d1 = max(0, s), min(length, length + s)
        if d1 <= d0:
            continue
        if dim == 2:
            out[:, lo:hi, d0:d1, :] = grad_out[:, lo:hi, d0 - s:d1 - s, :]
        else:
            out[:, lo:hi, :, d0:d1] = grad_out[:, lo:hi, :, d0 - s:d1 - s]
    return out


def asmlp_axial_shift(sd, x, pre, shift_size):
    """AxialShift.forward (as_mlp.py:55-95)."""
    t = conv1x1(x, _p(sd, pre + "conv1.weight", x), _opt(sd, pre + "conv1.bias", x))
    t = gelu(group_norm1(t, _p(sd, pre + "norm1.weight", x), _p(sd, pre + "norm1.bias", x)))
    lr = axial_shift_nchw(t, shift_size, 3)                      # shift_dim3 :81
    td = axial_shift_nchw(t, shift_size, 2)                      # shift_dim2 :82
    lr = gelu(conv1x1(lr, _p(sd, pre + "conv2_1.weight", x), _opt(sd, pre + "conv2_1.bias", x)))
    td = gelu(conv1x1(td, _p(sd, pre + "conv2_2.weight", x), _opt(sd, pre + "conv2_2.bias", x)))
    t = group_norm1(lr + td, _p(sd, pre + "norm2.weight", x), _p(sd, pre + "norm2.bias", x))
    return conv1x1(t, _p(sd, pre + "conv3.weight", x), _opt(sd, pre + "conv3.bias", x))


def _opt(sd, key, like):
    return _p(sd, key, like) if key in sd else None


def drop_path(x, rate, u):
    """Stochastic depth per sample, train mode (as_mlp.py:144 DropPath = timm's drop_path, restated by the reference repository in
    conv_mlp.py:17-34): x / keep * floor(keep + u), u = one uniform draw in [0, 1) per sample."""
    keep = 1.0 - rate
    return x.div(keep) * torch.floor(keep + u.reshape((-1,) + (1,) * (x.dim() - 1)).to(x.dtype))


def asmlp_block(sd, x, pre, shift_size, drop=None):
    """AxialShiftedBlock.forward (as_mlp.py:149-162).  Eval mode (drop None): DropPath is the identity; train mode: drop = (rate, u1, u2),
    the block's rate and the uniform draws of its two DropPath calls."""
    t = group_norm1(x, _p(sd, pre + "norm1.weight", x), _p(sd, pre + "norm1.bias", x))
    t = asmlp_axial_shift(sd, t, pre + "axial_shift.", shift_size)
    x = x + (t if drop is None else drop_path(t, drop[0], drop[1]))
    t = group_norm1(x, _p(sd, pre + "norm2.weight", x), _p(sd, pre + "norm2.bias", x))
    h = gelu(conv1x1(t, _p(sd, pre + "mlp.fc1.weight", x), _p(sd, pre + "mlp.fc1.bias", x)))
    t = conv1x1(h, _p(sd, pre + "mlp.fc2.weight", x), _p(sd, pre + "mlp.fc2.bias", x))
    return x + (t if drop is None else drop_path(t, drop[0], drop[2]))


def asmlp_patch_merging(sd, x, pre):
    """PatchMerging.forward (as_mlp.py:197-216): order (0,0),(1,0),(0,1),(1,1) on (h,w) parity."""
    x0 = x[:, :, 0::2, 0::2]
    x1 = x[:, :, 1::2, 0::2]
    x2 = x[:, :, 0::2, 1::2]
    x3 = x[:, :, 1::2, 1::2]
    t = torch.cat([x0, x1, x2, x3], dim=1)
    t = group_norm1(t, _p(sd, pre + "norm.weight", x), _p(sd, pre + "norm.bias", x))
    return conv1x1(t, _p(sd, pre + "reduction.weight", x), None)


def asmlp_forward(sd, x, shift_size=5, hooks=None, drop_path_rate=None, draws=None):
    """AS_MLP.forward (as_mlp.py:428-443), NCHW throughout.  Eval mode by default; train mode: drop_path_rate = the constructor's rate (the
    blocks get torch.linspace(0, rate, total depth), as_mlp.py:394; a block with rate 0 holds an Identity and draws nothing) and draws =
    the uniform draws of the DropPath calls in forward order, each (B,)."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patch_embed.proj.weight", x), _p(sd, "patch_embed.proj.bias", x))
    t = t.permute(0, 3, 1, 2)
    if "patch_embed.norm.weight" in sd:
        t = group_norm1(t, _p(sd, "patch_embed.norm.weight", x), _p(sd, "patch_embed.norm.bias", x))
    layer = 0
    rates, nb, nd = None, 0, 0
    if drop_path_rate is not None:
        total = 0
        while ("layers.%d.blocks.0.norm1.weight" % total) in sd:
            total += 1
        total = sum(_depth(sd, "layers.%d" % l + ".blocks.%d.norm1.weight") for l in range(total))
        rates = [float(v) for v in torch.linspace(0, drop_path_rate, total)]
    while ("layers.%d.blocks.0.norm1.weight" % layer) in sd:
        for i in range(_depth(sd, "layers.%d" % layer + ".blocks.%d.norm1.weight")):
            drop = None
            if rates is not None and rates[nb] > 0.0:
                drop = (rates[nb], torch.as_tensor(draws[nd]), torch.as_tensor(draws[nd + 1]))
                nd += 2
            nb += 1
            t = asmlp_block(sd, t, "layers.%d.blocks.%d." % (layer, i), shift_size, drop)
            if hooks is not None:
                hooks("layers.%d.blocks.%d" % (layer, i), t)
        if ("layers.%d.downsample.reduction.weight" % layer) in sd:
            t = asmlp_patch_merging(sd, t, "layers.%d.downsample." % layer)
        layer += 1
    t = group_norm1(t, _p(sd, "norm.weight", x), _p(sd, "norm.bias", x))
    t = t.mean(dim=(2, 3))
    return linear(t, _p(sd, "head.weight", x), _p(sd, "head.bias", x))


# --------------------------------------------------------------------------
# ConvMixer  (conv_mixer.py:13-45)
# --------------------------------------------------------------------------
def depthwise_conv_same(x, w, b):
    """Depthwise k x k conv, padding='same' (odd k -> (k-1)/2 each side) (conv_mixer.py:25):
    out[n,c,y,x] = b[c] + sum_{dy,dx} w[c,0,dy,dx] * in[n,c,y+dy-p,x+dx-p]."""
    k = w.shape[-1]
    p = (k - 1) // 2
    n, c, h, ww = x.shape
    xp = torch.nn.functional.pad(x, (p, k - 1 - p, p, k - 1 - p))
    out = torch.zeros_like(x)
    for dy in range(k):
        for dx in range(k):
            out += w[:, 0, dy, dx].view(1, c, 1, 1) * xp[:, :, dy:dy + h, dx:dx + ww]
    return out + b.view(1, c, 1, 1)


def convmixer_forward(sd, x, hooks=None):
    """ConvMixer.forward in eval mode: GELU *then* BN (conv_mixer.py:17-21, 23-32, 35-45)."""
    x = x.detach().cpu()
    w = _p(sd, "embedding.0.weight", x)
    t = patch_embed(x, w, _p(sd, "embedding.0.bias", x), padding=w.shape[-1] // 2).permute(0, 3, 1, 2)
    t = batch_norm_eval(gelu(t), sd, "embedding.2")
    for i in range(_depth(sd, "blocks.%d.0.fn.0.weight")):
        pre = "blocks.%d." % i
        d = depthwise_conv_same(t, _p(sd, pre + "0.fn.0.weight", x), _p(sd, pre + "0.fn.0.bias", x))
        t = t + batch_norm_eval(gelu(d), sd, pre + "0.fn.2")     # Residual :5-11
        t = conv1x1(t, _p(sd, pre + "1.weight", x), _p(sd, pre + "1.bias", x))
        t = batch_norm_eval(gelu(t), sd, pre + "3")
        if hooks is not None:
            hooks("blocks.%d" % i, t)
    t = t.mean(dim=(2, 3))
    return linear(t, _p(sd, "classifier.2.weight", x), _p(sd, "classifier.2.bias", x))


# --------------------------------------------------------------------------
# Sparse-MLP  (sparse_mlp.py:17-167)  -- SURVEY.md 8(f) rank 2
# --------------------------------------------------------------------------
def sparsemlp_block(sd, x, pre):
    """One entry of sMLPStage.model on NCHW x (sparse_mlp.py:84-104):
    x + dwconv3x3(BN(x)); x + fuse(cat[proj_h(BN x), proj_w(BN x), BN x]); x + FF(LN(x)) channel-last."""
    t = batch_norm_eval(x, sd, pre + "0.norm")
    x = x + depthwise_conv_same(t, _p(sd, pre + "0.fn.0.weight", x), _p(sd, pre + "0.fn.0.bias", x))      # :85-87, padding=1
    t = batch_norm_eval(x, sd, pre + "1.norm")
    # sMLPBlock.forward (:68-74): proj_h mixes along H (via the (0,1,3,2) permutes), proj_w along W
    xh = torch.einsum("gh,nchw->ncgw", _p(sd, pre + "1.fn.0.proj_h.weight", x), t) + _p(sd, pre + "1.fn.0.proj_h.bias", x).view(1, 1, -1, 1)
    xw = torch.einsum("gw,nchw->nchg", _p(sd, pre + "1.fn.0.proj_w.weight", x), t) + _p(sd, pre + "1.fn.0.proj_w.bias", x).view(1, 1, 1, -1)
    x = x + conv1x1(torch.cat([xh, xw, t], dim=1), _p(sd, pre + "1.fn.0.fuse.weight", x), _p(sd, pre + "1.fn.0.fuse.bias", x))
    tl = x.permute(0, 2, 3, 1)                                                                          # :93
    n = layer_norm(tl, _p(sd, pre + "3.norm.weight", x), _p(sd, pre + "3.norm.bias", x))
    h = gelu(linear(n, _p(sd, pre + "3.fn.0.weight", x), _p(sd, pre + "3.fn.0.bias", x)))
    tl = tl + linear(h, _p(sd, pre + "3.fn.3.weight", x), _p(sd, pre + "3.fn.3.bias", x))
    return tl.permute(0, 3, 1, 2)                                                                       # :101


def sparsemlp_patch_merging(sd, x, pre):
    """PatchMerging.forward on channel-last x (sparse_mlp.py:34-52): (h,w) parities (0,0),(1,0),(0,1),(1,1) on C,
    LayerNorm(4C), bias-free Linear(4C, 2C)."""
    t = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], dim=-1)
    t = layer_norm(t, _p(sd, pre + "norm.weight", x), _p(sd, pre + "norm.bias", x))
    return linear(t, _p(sd, pre + "reduction.weight", x), None)


def sparsemlp_forward(sd, x, hooks=None):
    """SparseMLP.forward (sparse_mlp.py:159-166) in eval mode."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patcher.0.weight", x), _p(sd, "patcher.0.bias", x))                      # :124
    if "patcher.1.1.weight" in sd:                                                                      # patcher_norm (:126-130)
        t = layer_norm(t, _p(sd, "patcher.1.1.weight", x), _p(sd, "patcher.1.1.bias", x))
    t = t.permute(0, 3, 1, 2)
    layer = 0
    while ("layers.%d.model.0.0.fn.0.weight" % layer) in sd:
        for i in range(_depth(sd, "layers.%d" % layer + ".model.%d.0.fn.0.weight")):
            t = sparsemlp_block(sd, t, "layers.%d.model.%d." % (layer, i))
            if hooks is not None:
                hooks("layers.%d.model.%d" % (layer, i), t)
        if ("layers.%d.model.0.0.fn.0.weight" % (layer + 1)) in sd:                                      # pooling = not the last stage (:138)
            t = sparsemlp_patch_merging(sd, t.permute(0, 2, 3, 1), "layers.%d.patch_merge.1." % layer).permute(0, 3, 1, 2)
        layer += 1
    t = layer_norm(t.permute(0, 2, 3, 1), _p(sd, "mlp_head.1.weight", x), _p(sd, "mlp_head.1.bias", x))  # :152-157
    t = t.mean(dim=(1, 2))
    return linear(t, _p(sd, "mlp_head.3.weight", x), _p(sd, "mlp_head.3.bias", x))


# --------------------------------------------------------------------------
# Hire-MLP  (hire_mlp.py:8-229)  -- SURVEY.md 8(f) rank 2
# --------------------------------------------------------------------------
def conv2d_im2col(x, w, b, stride, padding):
    """Conv2d on NCHW with square kernel k, stride s, zero padding p, as an explicit window gather + matmul
    (hire_mlp.py:21 patcher k=7 s=patch p=3; :161 patch_merge k=3 s=2 p=1).  Returns channel-last (B, Ho, Wo, Cout)."""
    cout, cin, kh, kw = w.shape
    xp = torch.nn.functional.pad(x, (padding, padding, padding, padding))
    bsz, _, hh, ww = xp.shape
    ho, wo = (hh - kh) // stride + 1, (ww - kw) // stride + 1
    cols = []
    for i in range(kh):
        for j in range(kw):
            cols.append(xp[:, :, i:i + (ho - 1) * stride + 1:stride, j:j + (wo - 1) * stride + 1:stride])       # (B, Cin, Ho, Wo)
    pat = torch.stack(cols, dim=2).permute(0, 3, 4, 1, 2).reshape(bsz, ho, wo, cin * kh * kw)                  # k = ci*kh*kw + i*kw + j
    out = pat @ w.reshape(cout, -1).t()
    return out if b is None else out + b


def _circular_pad_hw(x, pad_h, pad_w):
    """F.pad(x, (0, pad_w, 0, pad_h), 'circular') on NCHW (hire_mlp.py:133): the appended rows / columns repeat the first ones."""
    if pad_w:
        x = torch.cat([x, x[:, :, :, :pad_w]], dim=3)
    if pad_h:
        x = torch.cat([x, x[:, :, :pad_h, :]], dim=2)
    return x


def hiremlp_block(sd, x, pre, h, w, step, cross):
    """HireMLPBlock.forward on channel-last x = LayerNorm output (hire_mlp.py:127-152), padding_type 'circular'.
    NB the pad is h - H % h (w - W % w): a whole extra region when the size already divides (:131-133)."""
    t = x.permute(0, 3, 1, 2)
    bsz, c, hh, ww = t.shape
    t = _circular_pad_hw(t, h - hh % h, w - ww % w)
    hp, wp = t.shape[2], t.shape[3]
    th = torch.roll(t, step, 2) if cross else t                                                   # CrossRegion dim 2 (:44-51, 109)
    tw = torch.roll(t, step, 3) if cross else t
    gh, gw = hp // h, wp // w
    # InnerRegionH 'b c (h group) w -> b (c h) group w' (:68-70); InnerRegionW 'b c h (w group) -> b (c w) h group' (:57-59)
    xh = th.reshape(bsz, c, h, gh, wp).reshape(bsz, c * h, gh, wp)
    xw = tw.reshape(bsz, c, hp, w, gw).permute(0, 1, 3, 2, 4).reshape(bsz, c * w, hp, gw)

    def ff(z, p):                                                                                 # FeedForward (:33-42)
        z = gelu(conv1x1(z, _p(sd, p + "net.0.weight", x), _p(sd, p + "net.0.bias", x)))
        return conv1x1(z, _p(sd, p + "net.2.weight", x), _p(sd, p + "net.2.bias", x))

    xh = ff(xh, pre + "proj_h.")
    xw = ff(xw, pre + "proj_w.")
    xc = conv1x1(t, _p(sd, pre + "proj_c.weight", x), _p(sd, pre + "proj_c.bias", x))
    xh = xh.reshape(bsz, c, h, gh, wp).reshape(bsz, c, hp, wp)                                    # restore (:90-92)
    xw = xw.reshape(bsz, c, w, hp, gw).permute(0, 1, 3, 2, 4).reshape(bsz, c, hp, wp)             # restore (:79-81)
    if cross:
        xh = torch.roll(xh, -step, 2)
        xw = torch.roll(xw, -step, 3)
    out = (xc + xh + xw)[:, :, :hh, :ww]
    return out.permute(0, 2, 3, 1)


def hiremlp_forward(sd, x, h, w, cross_region_step, cross_region_interval=2, patch_size=4, hooks=None):
    """HireMLP.forward (hire_mlp.py:222-228) in eval mode; h, w, cross_region_step are the per-stage lists of the constructor."""
    x = x.detach().cpu()
    t = conv2d_im2col(x, _p(sd, "patcher.reduction.0.weight", x), _p(sd, "patcher.reduction.0.bias", x), patch_size, 3)     # :203
    if "patcher.reduction.1.1.weight" in sd:
        t = layer_norm(t, _p(sd, "patcher.reduction.1.1.weight", x), _p(sd, "patcher.reduction.1.1.bias", x))
    layer = 0
    while ("layers.%d.model.0.0.norm.weight" % layer) in sd:
        for i in range(_depth(sd, "layers.%d" % layer + ".model.%d.0.norm.weight")):
            pre = "layers.%d.model.%d." % (layer, i)
            cross = ((i + 1) % cross_region_interval == 0)                                       # cross_region_id = i_depth + 1 (:168, 104)
            n = layer_norm(t, _p(sd, pre + "0.norm.weight", x), _p(sd, pre + "0.norm.bias", x))
            t = t + hiremlp_block(sd, n, pre + "0.fn.0.", h[layer], w[layer], cross_region_step[layer], cross)
            n = layer_norm(t, _p(sd, pre + "1.norm.weight", x), _p(sd, pre + "1.norm.bias", x))
            hdn = gelu(linear(n, _p(sd, pre + "1.fn.0.weight", x), _p(sd, pre + "1.fn.0.bias", x)))
            t = t + linear(hdn, _p(sd, pre + "1.fn.3.weight", x), _p(sd, pre + "1.fn.3.bias", x))
            if hooks is not None:
                hooks("layers.%d.model.%d" % (layer, i), t)
        if ("layers.%d.model.0.0.norm.weight" % (layer + 1)) in sd:                               # pooling (:213)
            pm = "layers.%d.patch_merge.1.reduction.0." % layer
            t = conv2d_im2col(t.permute(0, 3, 1, 2), _p(sd, pm + "weight", x), _p(sd, pm + "bias", x), 2, 1)
        layer += 1
    t = layer_norm(t, _p(sd, "mlp_head.0.weight", x), _p(sd, "mlp_head.0.bias", x))
    t = t.mean(dim=(1, 2))
    return linear(t, _p(sd, "mlp_head.2.weight", x), _p(sd, "mlp_head.2.bias", x))


# --------------------------------------------------------------------------
# MS-MLP  (ms_mlp.py:11-359)  -- SURVEY.md 8(f) rank 3
# --------------------------------------------------------------------------
MS_EPS = 1e-6          # ms_mlp.py:279: its own LayerNorm class defaults to eps = 1e-6


def msmlp_block(sd, x, pre, shift_dist, mix_size):
    """MixShiftBlock.forward on NCHW x in eval mode (ms_mlp.py:48-78): channel chunks (torch.chunk sizes) rolled by their
    relative distance along W / along H, each mixed by its own k x k depthwise conv (zero padding k//2), the two sums added,
    LayerNorm(eps 1e-6) -> Linear -> GELU -> Linear -> layer scale gamma, residual onto the block input."""
    n, c, hh, ww = x.shape
    groups = len(shift_dist)
    sizes = [t.shape[0] for t in torch.chunk(torch.zeros(c), groups)]                   # :33
    lr, td, c0 = [], [], 0
    for i, cs in enumerate(sizes):
        xc = x[:, c0:c0 + cs]
        wl, bl = _p(sd, pre + "dwconv_lr.%d.weight" % i, x), _p(sd, pre + "dwconv_lr.%d.bias" % i, x)
        wt, bt = _p(sd, pre + "dwconv_td.%d.weight" % i, x), _p(sd, pre + "dwconv_td.%d.bias" % i, x)
        assert wl.shape[-1] == mix_size[i]
        lr.append(depthwise_conv_same(torch.roll(xc, shift_dist[i], 3), wl, bl))         # :56, :61
        td.append(depthwise_conv_same(torch.roll(xc, shift_dist[i], 2), wt, bt))         # :57, :62
        c0 += cs
    t = (torch.cat(lr, 1) + torch.cat(td, 1)).permute(0, 2, 3, 1)
    t = layer_norm(t, _p(sd, pre + "norm.weight", x), _p(sd, pre + "norm.bias", x), eps=MS_EPS)
    t = linear(gelu(linear(t, _p(sd, pre + "pwconv1.weight", x), _p(sd, pre + "pwconv1.bias", x))),
               _p(sd, pre + "pwconv2.weight", x), _p(sd, pre + "pwconv2.bias", x))
    if (pre + "gamma") in sd:
        t = t * _p(sd, pre + "gamma", x)
    return x + t.permute(0, 3, 1, 2)


def msmlp_forward(sd, x, shift_dist=(-2, -1, 0, 1, 2), mix_size=((1, 1, 3, 5, 7), (1, 1, 3, 5, 5), (1, 1, 3, 3, 3), (1, 1, 1, 1, 3)), hooks=None):
    """MS_MLP.forward (ms_mlp.py:341-359), eval mode: patch embed + LN, stages of MixShiftBlocks with a 2x2-conv + LN
    downsample (a PatchEmbed, :178-180), global average pool, LayerNorm AFTER the pool (:352-354), head."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patch_embed.proj.weight", x), _p(sd, "patch_embed.proj.bias", x))
    if "patch_embed.norm.weight" in sd:
        t = layer_norm(t, _p(sd, "patch_embed.norm.weight", x), _p(sd, "patch_embed.norm.bias", x), eps=MS_EPS)
    t = t.permute(0, 3, 1, 2)
    layer = 0
    while ("layers.%d.blocks.0.norm.weight" % layer) in sd:
        for i in range(_depth(sd, "layers.%d" % layer + ".blocks.%d.norm.weight")):
            t = msmlp_block(sd, t, "layers.%d.blocks.%d." % (layer, i), shift_dist, mix_size[layer])
            if hooks is not None:
                hooks("layers.%d.blocks.%d" % (layer, i), t)
        pre = "layers.%d.downsample." % layer
        if (pre + "proj.weight") in sd:
            t = patch_embed(t, _p(sd, pre + "proj.weight", x), _p(sd, pre + "proj.bias", x))
            if (pre + "norm.weight") in sd:
                t = layer_norm(t, _p(sd, pre + "norm.weight", x), _p(sd, pre + "norm.bias", x), eps=MS_EPS)
            t = t.permute(0, 3, 1, 2)
        layer += 1
    t = t.mean(dim=(2, 3))
    t = layer_norm(t, _p(sd, "norm.weight", x), _p(sd, "norm.bias", x), eps=MS_EPS)
    return linear(t, _p(sd, "head.weight", x), _p(sd, "head.bias", x))


# --------------------------------------------------------------------------
# Swin-MLP  (swin_mlp.py:12-460)  -- SURVEY.md 8(f) rank 3
# --------------------------------------------------------------------------
def swinmlp_block(sd, x, pre, hh, ww, num_heads, window_size, shift_size):
    """SwinMLPBlock.forward on x (B, H*W, C) in eval mode (swin_mlp.py:113-157): LayerNorm, zero-pad by
    (ws - shift, shift) on the left/top and right/bottom when shifted (:101-102, 122-124), ws x ws windows, per head a
    (ws^2 x ws^2) token mix with bias -- the grouped Conv1d of :105-108 -- window merge, crop, residual; then the channel MLP."""
    bsz, _, c = x.shape
    ws = window_size
    if min(hh, ww) <= ws:                                                                       # :94-97
        shift_size, ws = 0, min(hh, ww)
    t = layer_norm(x, _p(sd, pre + "norm1.weight", x), _p(sd, pre + "norm1.bias", x)).reshape(bsz, hh, ww, c)
    if shift_size > 0:
        pl, pr, pt, pb = ws - shift_size, shift_size, ws - shift_size, shift_size
        t = torch.nn.functional.pad(t, (0, 0, pl, pr, pt, pb))
    hp, wp = t.shape[1], t.shape[2]
    win = t.reshape(bsz, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, num_heads, c // num_heads)
    wgt = _p(sd, pre + "spatial_mlp.weight", x).reshape(num_heads, ws * ws, ws * ws)            # Conv1d(groups = heads): [h][t_out][t_in]
    bia = _p(sd, pre + "spatial_mlp.bias", x).reshape(num_heads, ws * ws)
    mixed = torch.einsum("hts,nshd->nthd", wgt, win) + bia.t().reshape(1, ws * ws, num_heads, 1)
    t = mixed.reshape(bsz, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(bsz, hp, wp, c)
    if shift_size > 0:
        t = t[:, pt:hp - pb, pl:wp - pr, :]
    x = x + t.reshape(bsz, hh * ww, c)
    n = layer_norm(x, _p(sd, pre + "norm2.weight", x), _p(sd, pre + "norm2.bias", x))
    hdn = gelu(linear(n, _p(sd, pre + "mlp.fc1.weight", x), _p(sd, pre + "mlp.fc1.bias", x)))
    return x + linear(hdn, _p(sd, pre + "mlp.fc2.weight", x), _p(sd, pre + "mlp.fc2.bias", x))


def swinmlp_forward(sd, x, num_heads=(3, 6, 12, 24), window_size=7, hooks=None):
    """SwinMLP.forward (swin_mlp.py:433-446) in eval mode; ape = True iff the state dict holds `absolute_pos_embed` (:386-388, :437-438)."""
    x = x.detach().cpu()
    t = patch_embed(x, _p(sd, "patch_embed.proj.weight", x), _p(sd, "patch_embed.proj.bias", x))
    bsz, hh, ww, c = t.shape
    t = t.reshape(bsz, hh * ww, c)
    if "patch_embed.norm.weight" in sd:
        t = layer_norm(t, _p(sd, "patch_embed.norm.weight", x), _p(sd, "patch_embed.norm.bias", x))
    if "absolute_pos_embed" in sd:
        t = t + _p(sd, "absolute_pos_embed", x)
    layer = 0
    while ("layers.%d.blocks.0.norm1.weight" % layer) in sd:
        for i in range(_depth(sd, "layers.%d" % layer + ".blocks.%d.norm1.weight")):
            t = swinmlp_block(sd, t, "layers.%d.blocks.%d." % (layer, i), hh, ww, num_heads[layer], window_size,
                              0 if i % 2 == 0 else window_size // 2)                            # :246
            if hooks is not None:
                hooks("layers.%d.blocks.%d" % (layer, i), t)
        pre = "layers.%d.downsample." % layer
        if (pre + "reduction.weight") in sd:                                                    # PatchMerging (:193-212)
            c = t.shape[-1]
            g = t.reshape(bsz, hh, ww, c)
            g = torch.cat([g[:, 0::2, 0::2, :], g[:, 1::2, 0::2, :], g[:, 0::2, 1::2, :], g[:, 1::2, 1::2, :]], dim=-1)
            hh, ww = hh // 2, ww // 2
            g = layer_norm(g.reshape(bsz, hh * ww, 4 * c), _p(sd, pre + "norm.weight", x), _p(sd, pre + "norm.bias", x))
            t = linear(g, _p(sd, pre + "reduction.weight", x), None)
        layer += 1
    t = layer_norm(t, _p(sd, "norm.weight", x), _p(sd, "norm.bias", x)).mean(dim=1)
    return linear(t, _p(sd, "head.weight", x), _p(sd, "head.bias", x))


# --------------------------------------------------------------------------
# CycleMLP  (cycle_mlp.py:54-350)  -- SURVEY.md 8(f) rank 3
# --------------------------------------------------------------------------
# cycle_mlp.py:14 takes `deform_conv2d` from torchvision.ops.deform_conv, a dependency that is ABSENT from this image
# and not pinned by the reference (no requirements file; README.md:182 names only the import).  Its published algorithm
# (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp, deformable_im2col + bilinear_interpolate; docs of
# torchvision.ops.deform_conv2d): with weight (Cout, Cin/groups, kh, kw), offset (B, 2 * G_off * kh * kw, Ho, Wo) holding
# (dy, dx) pairs per offset group and kernel point, out[b,o,y,x] = bias[o] + sum_{c,i,j} W[o,c,i,j] *
# bilinear(in[b,c], y*s - p + i*d + dy, x*s - p + j*d + dx), where bilinear() is zero for a sample point at or beyond
# one pixel outside the map (h <= -1 or h >= H, likewise w) and takes each of the four corners only when it lies inside.
# `deform_conv2d_pointwise_loop` below restates exactly that for the only case CycleFC uses (1 x 1 kernel, stride 1, no
# padding, one offset group per input channel) with an explicit per-element loop; `cycle_fc` is the vectorised form for
# CycleFC's integer offsets (a per-channel shifted gather with zero fill followed by a 1 x 1 convolution), and
# tests/test_oracle_golden.py checks the two against each other.  CycleFC is therefore "parity unpinned" against
# torchvision itself; everything around it is pinned by running the reference's own module code (make_golden.py).
def _bilinear_zero(img, h, w):
    """torchvision's bilinear_interpolate on one (H, W) plane at a real-valued point."""
    hh, ww = img.shape
    if h <= -1 or h >= hh or w <= -1 or w >= ww:
        return img.new_zeros(())
    h_low, w_low = int(math.floor(h)), int(math.floor(w))
    h_high, w_high = h_low + 1, w_low + 1
    lh, lw = h - h_low, w - w_low
    hhg, hw = 1 - lh, 1 - lw
    v1 = img[h_low, w_low] if (h_low >= 0 and w_low >= 0) else 0.0
    v2 = img[h_low, w_high] if (h_low >= 0 and w_high <= ww - 1) else 0.0
    v3 = img[h_high, w_low] if (h_high <= hh - 1 and w_low >= 0) else 0.0
    v4 = img[h_high, w_high] if (h_high <= hh - 1 and w_high <= ww - 1) else 0.0
    return hhg * hw * v1 + hhg * lw * v2 + lh * hw * v3 + lh * lw * v4


def deform_conv2d_pointwise_loop(inp, offset, weight, bias=None):
    """deform_conv2d for a 1 x 1 kernel, stride 1, padding 0, dilation 1, groups 1 and one offset group per input channel
    (offset: (B or 1, 2*Cin, H or 1, W or 1), (dy, dx) interleaved), as an explicit loop over every output element
    (small inputs only).  cycle_mlp.py:126-131."""
    bsz, cin, hh, ww = inp.shape
    cout = weight.shape[0]
    off = offset.expand(bsz, 2 * cin, hh, ww)
    wmat = weight.reshape(cout, cin)
    cols = inp.new_zeros((bsz, cin, hh, ww))
    for b in range(bsz):
        for c in range(cin):
            for y in range(hh):
                for x in range(ww):
                    dy, dx = float(off[b, 2 * c, y, x]), float(off[b, 2 * c + 1, y, x])
                    cols[b, c, y, x] = _bilinear_zero(inp[b, c], y + dy, x + dx)
    out = torch.einsum("oc,bchw->bohw", wmat, cols)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def cycle_offsets(channels, kernel_size):
    """CycleFC.gen_offset (cycle_mlp.py:104-120): per input channel i the integer (dy, dx)."""
    kh, kw = kernel_size
    assert kh == 1 or kw == 1
    start = (kh * kw) // 2
    dy = [0 if kh == 1 else (i + start) % kh - kh // 2 for i in range(channels)]
    dx = [(i + start) % kw - kw // 2 if kh == 1 else 0 for i in range(channels)]
    return dy, dx


def cycle_fc(x, weight, bias, kernel_size):
    """CycleFC.forward on NCHW x (cycle_mlp.py:122-131): channel i is read at (y + dy_i, x + dx_i), zero outside the map,
    then a 1 x 1 convolution."""
    bsz, cin, hh, ww = x.shape
    dy, dx = cycle_offsets(cin, kernel_size)
    g = torch.zeros_like(x)
    for i in range(cin):
        ys0, ys1 = max(0, -dy[i]), min(hh, hh - dy[i])
        xs0, xs1 = max(0, -dx[i]), min(ww, ww - dx[i])
        if ys1 > ys0 and xs1 > xs0:
            g[:, i, ys0:ys1, xs0:xs1] = x[:, i, ys0 + dy[i]:ys1 + dy[i], xs0 + dx[i]:xs1 + dx[i]]
    return conv1x1(g, weight.reshape(weight.shape[0], cin), bias)


def cyclemlp_attn(sd, x, pre):
    """CycleMLP.forward on channel-last x (B,H,W,C) (cycle_mlp.py:160-175)."""
    bsz, hh, ww, c = x.shape
    xc = x.permute(0, 3, 1, 2)
    h = cycle_fc(xc, _p(sd, pre + "sfc_h.weight", x), _p(sd, pre + "sfc_h.bias", x), (1, 3)).permute(0, 2, 3, 1)
    w = cycle_fc(xc, _p(sd, pre + "sfc_w.weight", x), _p(sd, pre + "sfc_w.bias", x), (3, 1)).permute(0, 2, 3, 1)
    cc = linear(x, _p(sd, pre + "mlp_c.weight", x), _opt(sd, pre + "mlp_c.bias", x))
    a = (h + w + cc).mean(dim=(1, 2))                                                              # (B, C)
    a = linear(gelu(linear(a, _p(sd, pre + "reweight.fc1.weight", x), _p(sd, pre + "reweight.fc1.bias", x))),
               _p(sd, pre + "reweight.fc2.weight", x), _p(sd, pre + "reweight.fc2.bias", x))       # (B, 3C), index c*3 + k
    a = torch.softmax(a.reshape(bsz, c, 3), dim=2)                                                 # softmax over k (:170)
    t = h * a[:, :, 0].view(bsz, 1, 1, c) + w * a[:, :, 1].view(bsz, 1, 1, c) + cc * a[:, :, 2].view(bsz, 1, 1, c)
    return linear(t, _p(sd, pre + "proj.weight", x), _p(sd, pre + "proj.bias", x))


def flatten_outputs(out):
    """A list of feature maps (fork_feat) as ONE (B, sum of C H W) matrix -- how the fixtures and the tests hold such an output."""
    return torch.cat([t.reshape(t.shape[0], -1) for t in out], dim=1) if isinstance(out, (list, tuple)) else out


def cyclemlp_forward(sd, x, hooks=None):
    """CycleNet.forward (cycle_mlp.py:322-350) in eval mode, skip_lam 1.  Classification head, or -- fork_feat = True, recognised by the
    `norm0` ... `norm6` layers the constructor adds instead (:274-287) -- the list of the four normalised stage outputs as (B, C, H, W)
    (:326-334; a `norm{i}` that is an Identity, FORK_LAST3, has no parameters: the stage output goes out as it is)."""
    x = x.detach().cpu()
    fork = not ("norm.weight" in sd)
    outs = []
    t = conv2d_im2col(x, _p(sd, "patch_embed.proj.weight", x), _p(sd, "patch_embed.proj.bias", x), 4, 2)   # 7x7 s4 p2 (:261)
    idx = 0
    while True:
        if ("network.%d.0.norm1.weight" % idx) in sd:                                             # a stage of CycleBlocks
            for i in range(_depth(sd, "network.%d" % idx + ".%d.norm1.weight")):
                pre = "network.%d.%d." % (idx, i)
                n = layer_norm(t, _p(sd, pre + "norm1.weight", x), _p(sd, pre + "norm1.bias", x))
                t = t + cyclemlp_attn(sd, n, pre + "attn.")
                n = layer_norm(t, _p(sd, pre + "norm2.weight", x), _p(sd, pre + "norm2.bias", x))
                hdn = gelu(linear(n, _p(sd, pre + "mlp.fc1.weight", x), _p(sd, pre + "mlp.fc1.bias", x)))
                t = t + linear(hdn, _p(sd, pre + "mlp.fc2.weight", x), _p(sd, pre + "mlp.fc2.bias", x))
                if hooks is not None:
                    hooks("network.%d.%d" % (idx, i), t)
        elif ("network.%d.proj.weight" % idx) in sd:                                              # Downsample 3x3 s2 p1 (:220-231)
            t = conv2d_im2col(t.permute(0, 3, 1, 2), _p(sd, "network.%d.proj.weight" % idx, x), _p(sd, "network.%d.proj.bias" % idx, x), 2, 1)
        else:
            break
        if fork and idx in (0, 2, 4, 6):
            o = t
            if ("norm%d.weight" % idx) in sd:
                o = layer_norm(t, _p(sd, "norm%d.weight" % idx, x), _p(sd, "norm%d.bias" % idx, x))
            outs.append(o.permute(0, 3, 1, 2).contiguous())
        idx += 1
    if fork:
        return outs
    t = layer_norm(t, _p(sd, "norm.weight", x), _p(sd, "norm.bias", x))
    t = t.reshape(t.shape[0], -1, t.shape[-1]).mean(dim=1)
    return linear(t, _p(sd, "head.weight", x), _p(sd, "head.bias", x))
