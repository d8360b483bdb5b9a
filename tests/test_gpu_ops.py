"""-m gpu: every C-ABI kernel against a float64 CPU restatement of its header contract
(include/mlpk.h) on seeded inputs, including ragged tails and the edge cases the reference's
ops have (zero-fill borders, non-divisible channel groups, K padding)."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import load_pkg

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16, torch.bfloat16]
EPS = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}


def dev():
    return torch.device("cuda:0")


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def gemm_ref(A, B, M, N, K, bias=None, act=0, cscale=None, cshift=None, rscale=None, rperiod=1, R=None, res=0,
             out_mode=0, t_rows=0, t_tokens=0, ldc=None):
    """float64 restatement of the mlpk_gemm_nt contract; returns the dense expected C tensor(s)."""
    acc = A[:M, :K].double() @ B[:N, :K].double().t()
    v = acc
    if bias is not None:
        v = v + bias.double()[None, :N]
    if act:
        v = oracle.gelu(v)
    if cscale is not None:
        v = v * cscale.double()[None, :N]
    if cshift is not None:
        v = v + cshift.double()[None, :N]
    if rscale is not None:
        idx = torch.arange(M) % rperiod
        v = v * rscale.double()[idx][:, None]
    if out_mode == 0:
        if res:
            r = R[:M, :N].double()
            v = v + r if res == 1 else v * r
        return v
    nimg = M // t_rows
    v = v.reshape(nimg, t_rows, N).permute(0, 2, 1)             # (img, token n, channel c)
    if res:
        r = R.reshape(nimg, t_tokens, -1)[:, :N, :t_rows].double()
        v = v + r if res == 1 else v * r
    return v


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 5])
def test_gemm_rowmajor_epilogues(dtype, algo):
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    cases = [  # M, N, K, flags
        (300, 200, 72, dict(bias=True)),
        (257, 129, 200, dict(bias=True, act=1)),
        (512, 256, 64, dict(bias=True, res=1)),
        (130, 70, 136, dict(bias=True, act=1, cscale=True, cshift=True)),
        (96, 1000, 64, dict(bias=True, cscale=True, res=1)),
        (64, 10, 32, dict(bias=True)),                        # ldc = 10: scalar-store path
    ]
    for ci, (M, Nn, K, fl) in enumerate(cases):
        A = rnd((M, K), dtype, 10 + ci).to(dev())
        B = rnd((Nn, K), dtype, 20 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 30 + ci).to(dev()) if fl.get("bias") else None
        cs = (rnd((Nn,), torch.float32, 40 + ci) * 0.2 + 1).to(dev()) if fl.get("cscale") else None
        ch = rnd((Nn,), torch.float32, 50 + ci).to(dev()) if fl.get("cshift") else None
        R = rnd((M, Nn), dtype, 60 + ci).to(dev()) if fl.get("res") else None
        C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
        E.gemm(A, B, C, M, Nn, K, bias=bias, act=fl.get("act", 0), cscale=cs, cshift=ch, R=R, res=fl.get("res", 0), algo=algo)
        torch.cuda.synchronize()
        ref = gemm_ref(A.cpu(), B.cpu(), M, Nn, K, bias=None if bias is None else bias.cpu(), act=fl.get("act", 0),
                       cscale=None if cs is None else cs.cpu(), cshift=None if ch is None else ch.cpu(),
                       R=None if R is None else R.cpu(), res=fl.get("res", 0))
        got = C.cpu().double()
        assert torch.isfinite(got).all(), (ci, "non-finite output")
        err = (got - ref).abs().max().item()
        tol = EPS[dtype] * max(1.0, ref.abs().max().item()) * 4
        assert err < tol, (str(dtype), algo, ci, err, tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("algo", [0, 1, 4, 5])
def test_gemm_token_transposed(dtype, algo):
    """OUT_TOKEN_T: the token-mixing form (Mixer fc2 residual, ResMLP gamma_1 row scale, gMLP gate)."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (nimg, t_rows, S, K, res, rscale, ldr_extra) in enumerate([
            (3, 64, 49, 56, 1, False, 0), (2, 40, 196, 200, 1, True, 0), (2, 128, 16, 16, 2, False, 128), (5, 8, 20, 24, 0, False, 0)]):
        M = nimg * t_rows
        A = rnd((M, K), dtype, 100 + ci).to(dev())
        B = rnd((S, K), dtype, 110 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((S,), torch.float32, 120 + ci).to(dev())
        rs = (rnd((t_rows,), torch.float32, 130 + ci) * 0.3 + 1).to(dev()) if rscale else None
        ldr = t_rows + ldr_extra
        R = rnd((nimg * S, ldr), dtype, 140 + ci).to(dev()) if res else None
        C = torch.full((nimg * S, t_rows), float("nan"), dtype=dtype, device=dev())
        E.gemm(A, B, C, M, S, K, ldc=t_rows, bias=bias, rscale=rs, rperiod=t_rows if rscale else 0, R=R, ldr=ldr if res else None,
               res=res, out_mode=N.OUT_TOKEN_T, t_rows=t_rows, t_tokens=S, algo=algo)
        torch.cuda.synchronize()
        ref = gemm_ref(A.cpu(), B.cpu(), M, S, K, bias=bias.cpu(), rscale=None if rs is None else rs.cpu(), rperiod=t_rows,
                       R=None if R is None else R.cpu(), res=res, out_mode=1, t_rows=t_rows, t_tokens=S)
        got = C.cpu().double().reshape(nimg, S, t_rows)
        assert torch.isfinite(got).all()
        err = (got - ref).abs().max().item()
        tol = EPS[dtype] * max(1.0, ref.abs().max().item()) * 4
        assert err < tol, (str(dtype), algo, ci, err, tol)


def test_gemm_inplace_residual_and_identity():
    """A = I with an ASYMMETRIC B catches a transposed accumulator layout; R aliasing C is allowed."""
    pkg = load_pkg()
    E = pkg.engine
    for dtype in DTYPES:
        n = 96
        A = torch.eye(n, dtype=dtype, device=dev())
        B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 7 - 3).to(dtype).to(dev())
        C = torch.ones((n, n), dtype=dtype, device=dev())
        E.gemm(A, B, C, n, n, n, R=C, res=1)
        torch.cuda.synchronize()
        assert torch.equal(C.cpu().float(), B.cpu().float().t() + 1.0)


def test_gemm_argument_errors():
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    A = torch.zeros((16, 12), dtype=torch.bfloat16, device=dev())
    with pytest.raises(N.MlpkError):
        E.gemm(A, A, A, 16, 16, 12)                             # K not a multiple of 8 elements
    with pytest.raises(TypeError):
        E.gemm(A.to(torch.float64), A, A, 16, 16, 8)


@pytest.mark.parametrize("dtype", DTYPES)
def test_row_stats_and_norm_apply(dtype):
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    B_, S, C = 3, 50, 72
    rows = B_ * S
    x = (rnd((rows, C), torch.float32, 1) * 2 + 0.5).to(dtype).to(dev())
    mean = torch.empty(rows, device=dev())
    rstd = torch.empty(rows, device=dev())
    E.row_stats(x, rows, C, C, mean, rstd)
    xd = x.cpu().double()
    mu = xd.mean(1)
    var = ((xd - mu[:, None]) ** 2).mean(1)
    assert (mean.cpu().double() - mu).abs().max() < 1e-5
    assert (rstd.cpu().double() - 1 / torch.sqrt(var + 1e-5)).abs().max() < 1e-4
    g = (rnd((C,), torch.float32, 2) * 0.2 + 1).to(dev())
    b = rnd((C,), torch.float32, 3).to(dev())
    sp = 56
    out_rm = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
    out_tt = torch.full((B_ * C, sp), float("nan"), dtype=dtype, device=dev())
    E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=g, beta=b, out_rm=out_rm, ld_rm=C, out_tt=out_tt, S=S, ld_tt=sp)
    torch.cuda.synchronize()
    ref = oracle.layer_norm(xd, g.cpu().double(), b.cpu().double())
    tol = EPS[dtype] * 8
    assert (out_rm.cpu().double() - ref).abs().max() < tol
    tt = out_tt.cpu().double().reshape(B_, C, sp)
    assert (tt[:, :, :S] - ref.reshape(B_, S, C).permute(0, 2, 1)).abs().max() < tol
    assert (tt[:, :, S:] == 0).all()                          # K padding of the token GEMM
    # GELU + long-row (GroupNorm-like) statistics: one stat per sample covering S rows
    gm = torch.empty(B_, device=dev())
    gr = torch.empty(B_, device=dev())
    E.row_stats(x, B_, S * C, S * C, gm, gr)
    out = torch.empty_like(x)
    E.norm_apply(x, rows, C, C, mean=gm, rstd=gr, gamma=g, beta=b, act=N.ACT_GELU, stat_group=S, out_rm=out, ld_rm=C)
    torch.cuda.synchronize()
    xs = xd.reshape(B_, S * C)
    m2 = xs.mean(1)
    v2 = ((xs - m2[:, None]) ** 2).mean(1)
    ref2 = oracle.gelu(((xs - m2[:, None]) / torch.sqrt(v2 + 1e-5)[:, None]).reshape(rows, C) * g.cpu().double() + b.cpu().double())
    assert (out.cpu().double() - ref2).abs().max() < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_vip_permutes(dtype):
    pkg = load_pkg()
    E = pkg.engine
    B_, H, W, C, seg = 2, 4, 6, 24, 6
    x = rnd((B_, H, W, C), dtype, 5).to(dev())
    G = C // seg
    ldh, ldw = E.round_up(H * seg, 8), E.round_up(W * seg, 8)
    ph = torch.full((B_ * W * G, ldh), float("nan"), dtype=dtype, device=dev())
    pw = torch.full((B_ * H * G, ldw), float("nan"), dtype=dtype, device=dev())
    E.norm_apply(x, B_ * H * W, C, C, out_ph=ph, H=H, W=W, seg=seg, ld_p=ldh)
    E.norm_apply(x, B_ * H * W, C, C, out_pw=pw, H=H, W=W, seg=seg, ld_p=ldw)
    torch.cuda.synchronize()
    xh = oracle.vip_permute_h(x.cpu().float(), seg).reshape(-1, H * seg)
    xw = oracle.vip_permute_w(x.cpu().float(), seg).reshape(-1, W * seg)
    assert torch.equal(ph.cpu().float()[:, :H * seg], xh) and (ph.cpu().float()[:, H * seg:] == 0).all()
    assert torch.equal(pw.cpu().float()[:, :W * seg], xw) and (pw.cpu().float()[:, W * seg:] == 0).all()
    back_h = torch.empty_like(x)
    back_w = torch.empty_like(x)
    E.vip_unpermute(0, ph, back_h, B_, H, W, C, seg, ldh)
    E.vip_unpermute(1, pw, back_w, B_, H, W, C, seg, ldw)
    torch.cuda.synchronize()
    assert torch.equal(back_h.cpu(), x.cpu()) and torch.equal(back_w.cpu(), x.cpu())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vip_permute_byproduct_sums(dtype):
    """The optional by-products of the ViP rearrange passes: sums of the ROUNDED normalised values over the walked axis, laid
    out as the reduced operand of the other branch -- and the linearity they are used for (vip.py:49 on the branch outputs
    == tiny GEMMs on these sums): sum over all pixels of Linear_h applied in the 'b w c (h s)' layout."""
    pkg = load_pkg()
    E = pkg.engine
    B_, H, W, C, seg = 3, 8, 6, 32, 4
    G = C // seg
    rows = B_ * H * W
    x = rnd((B_, H, W, C), dtype, 15).to(dev())
    g = (rnd((C,), torch.float32, 16) * 0.2 + 1).to(dev())
    b = rnd((C,), torch.float32, 17).to(dev())
    mean = torch.empty(rows, device=dev())
    rstd = torch.empty(rows, device=dev())
    E.row_stats(x, rows, C, C, mean, rstd)
    ldh, ldw = E.round_up(H * seg, 32), E.round_up(W * seg, 32)
    ph = torch.zeros((B_ * W * G, ldh), dtype=dtype, device=dev())
    pw = torch.zeros((B_ * H * G, ldw), dtype=dtype, device=dev())
    Ah = torch.zeros((B_ * G, ldh), device=dev())
    Aw = torch.zeros((B_ * G, ldw), device=dev())
    E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=g, beta=b, out_ph=ph, H=H, W=W, seg=seg, ld_p=ldh, sum_ph=Aw, ld_sum=ldw)
    E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=g, beta=b, out_pw=pw, H=H, W=W, seg=seg, ld_p=ldw, sum_pw=Ah, ld_sum=ldh)
    torch.cuda.synchronize()
    y = ph.cpu().double()[:, :H * seg].reshape(B_, W, G, H, seg)                   # rounded normalised values, (b, w, g, h, j)
    ref_Ah = y.sum(1).reshape(B_ * G, H * seg)                                     # sum over w: rows (b, g), columns (h, j)
    ref_Aw = y.sum(3).permute(0, 2, 1, 3).reshape(B_ * G, W * seg)                 # sum over h: rows (b, g), columns (w, j)
    assert (Ah.cpu().double()[:, :H * seg] - ref_Ah).abs().max() < 1e-5 * max(1.0, ref_Ah.abs().max().item())
    assert (Aw.cpu().double()[:, :W * seg] - ref_Aw).abs().max() < 1e-5 * max(1.0, ref_Aw.abs().max().item())
    assert (Ah.cpu()[:, H * seg:] == 0).all() and (Aw.cpu()[:, W * seg:] == 0).all()
    # linearity: sum over (h, w) of Linear_h's output at channel (g, j) == Ah . (sum_h' W[(h', j), :]) + W * sum_h' bias[(h', j)]
    wgt = rnd((H * seg, H * seg), torch.float64, 18, 0.3)
    bias = rnd((H * seg,), torch.float64, 19)
    z = y.reshape(B_ * W * G, H * seg) @ wgt.t() + bias                            # rows (b, w, g), columns (h', j)
    direct = z.reshape(B_, W, G, H, seg).sum((1, 3)).reshape(B_, C)               # sum over w and h' -> (b, (g, j))
    via = ref_Ah @ wgt.reshape(H, seg, H * seg).sum(0).t() + W * bias.reshape(H, seg).sum(0)
    assert (via.reshape(B_, C) - direct).abs().max() < 1e-9 * max(1.0, direct.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_mean(dtype):
    pkg = load_pkg()
    E = pkg.engine
    B_, S, C = 3, 37, 72
    x = rnd((B_ * S, C), dtype, 7).to(dev())
    out = torch.empty((B_, C), dtype=dtype, device=dev())
    E.pool_mean(x, B_, S, C, C, out, C)
    torch.cuda.synchronize()
    ref = x.cpu().double().reshape(B_, S, C).mean(1)
    assert (out.cpu().double() - ref).abs().max() < EPS[dtype] * 4
    mean = torch.empty(B_ * S, device=dev())
    rstd = torch.empty(B_ * S, device=dev())
    E.row_stats(x, B_ * S, C, C, mean, rstd)
    g = (rnd((C,), torch.float32, 8) * 0.2 + 1).to(dev())
    b = rnd((C,), torch.float32, 9).to(dev())
    E.pool_mean(x, B_, S, C, C, out, C, mean=mean, rstd=rstd, gamma=g, beta=b)
    torch.cuda.synchronize()
    ref = oracle.layer_norm(x.cpu().double(), g.cpu().double(), b.cpu().double()).reshape(B_, S, C).mean(1)
    assert (out.cpu().double() - ref).abs().max() < EPS[dtype] * 4


@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
def test_patchify(src_dtype):
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for (B_, cin, H, W, ph, pw, pad) in [(2, 3, 32, 32, 8, 8, 0), (2, 3, 32, 24, 8, 4, 0), (1, 3, 28, 28, 7, 7, 3), (2, 3, 16, 16, 4, 4, 0)]:
        x = rnd((B_, cin, H, W), src_dtype, 11).to(dev())
        hp, wp = (H + 2 * pad - ph) // ph + 1, (W + 2 * pad - pw) // pw + 1
        K = cin * ph * pw
        kp = E.round_up(K, 8)
        out = torch.full((B_ * hp * wp, kp), float("nan"), dtype=torch.float32, device=dev())
        E.patchify(x, out, B_, cin, H, W, ph, pw, pad, kp)
        torch.cuda.synchronize()
        w_eye = torch.eye(K).reshape(K, cin, ph, pw)
        ref = oracle.patch_embed(x.cpu().float(), w_eye, None, padding=pad).reshape(-1, K)
        assert torch.equal(out.cpu()[:, :K], ref)
        assert (out.cpu()[:, K:] == 0).all()
    # channel-last source (S2 stage-2 embedding) and the PatchMerging order (as_mlp.py:207-211)
    B_, H, W, cin = 2, 8, 8, 16
    x = rnd((B_, H, W, cin), torch.float32, 12).to(dev())
    out = torch.empty((B_ * 16, 4 * cin), dtype=torch.float32, device=dev())
    E.patchify(x, out, B_, cin, H, W, 2, 2, 0, 4 * cin, layout=N.LAYOUT_NHWC, px_stride=cin, order=0)
    torch.cuda.synchronize()
    xc = x.cpu()
    ref = xc.reshape(B_, 4, 2, 4, 2, cin).permute(0, 1, 3, 2, 4, 5).reshape(B_ * 16, 4 * cin)   # k = (i*2+j)*cin + ci
    assert torch.equal(out.cpu(), ref)
    E.patchify(x, out, B_, cin, H, W, 2, 2, 0, 4 * cin, layout=N.LAYOUT_NHWC, px_stride=cin, order=1)
    torch.cuda.synchronize()
    nchw = xc.permute(0, 3, 1, 2)
    merged = torch.cat([nchw[:, :, 0::2, 0::2], nchw[:, :, 1::2, 0::2], nchw[:, :, 0::2, 1::2], nchw[:, :, 1::2, 1::2]], 1)
    assert torch.equal(out.cpu(), merged.permute(0, 2, 3, 1).reshape(B_ * 16, 4 * cin))


def test_axial_shift_against_golden(golden_dir):
    """The reference's one native op: bit-exact against the pinned outputs of its own torch_shift."""
    pkg = load_pkg()
    E = pkg.engine
    z = np.load(golden_dir + "/ops.npz")
    i = 0
    while "shift%d/x" % i in z.files:
        x = torch.from_numpy(z["shift%d/x" % i]).to(dev())
        k = int(z["shift%d/k" % i])
        for dim in (2, 3):
            out = torch.full_like(x, float("nan"))
            E.shift_nchw(x, out, k, dim)
            torch.cuda.synchronize()
            assert torch.equal(out.cpu(), torch.from_numpy(z["shift%d/dim%d" % (i, dim)]))
            n, c, h, w = x.shape
            xl = x.permute(0, 2, 3, 1).contiguous()
            ol = torch.full_like(xl, float("nan"))
            E.shift_nhwc(xl, ol, n, h, w, c, k, dim)
            torch.cuda.synchronize()
            assert torch.equal(ol.permute(0, 3, 1, 2).cpu(), torch.from_numpy(z["shift%d/dim%d" % (i, dim)]))
        i += 1
    for dt in (torch.float16, torch.bfloat16):
        x = rnd((2, 10, 5, 6), dt, 3).to(dev())
        out = torch.empty_like(x)
        E.shift_nchw(x, out, 3, 3)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu().float(), oracle.axial_shift_nchw(x.cpu().float(), 3, 3))
    with pytest.raises(pkg._native.MlpkError):
        E.shift_nchw(x, out, 4, 2)                              # even kernel (shift_cuda.py:167)
    with pytest.raises(pkg._native.MlpkError):
        E.shift_nchw(x, out, 3, 1)                              # bad dim (shift_cuda.py:168)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_shift_backward_matches_reference_autograd(dtype):
    """mlpk_shift_nchw_backward and the `Shift` module's autograd path (shift_cuda.py:75-103,131-162): bit-equal to the grads the
    reference's autograd produced through its own torch_shift (tests/golden/ops.npz), for both axes, ragged channel groups,
    non-contiguous grad_output; grads flow through the module like through the reference's `_shift.apply`."""
    pkg = load_pkg()
    E = pkg.engine
    from importlib import import_module
    ut = import_module("jittor-mlp_amd.models_pytorch.utils")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops.npz"))
    i = 0
    while "shift%d/x" % i in z.files:
        k = int(z["shift%d/k" % i])
        x = torch.from_numpy(z["shift%d/x" % i]).to(dtype)
        for dim in (2, 3):
            go = torch.from_numpy(z["shift%d/gout_dim%d" % (i, dim)]).to(dtype)
            want = oracle.axial_shift_nchw_backward(go, k, dim)                       # (== the reference's grads in fp32, test_oracle_golden)
            if dtype == torch.float32:
                assert torch.equal(want, torch.from_numpy(z["shift%d/gin_dim%d" % (i, dim)]))
            gi = torch.empty_like(go, device=dev())
            E.shift_nchw_backward(go.to(dev()), gi, k, dim)
            assert torch.equal(gi.cpu(), want)
            xr = x.to(dev()).requires_grad_(True)
            y = ut.Shift(k, dim)(xr)
            assert y.requires_grad
            y.backward(go.to(dev()).permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2))   # a non-contiguous grad_output
            assert torch.equal(xr.grad.cpu(), want)
        i += 1
    assert i >= 4
    x = torch.zeros(1, 4, 4, 4, dtype=dtype, device=dev())
    with pytest.raises(RuntimeError):
        E.shift_nchw_backward(x, torch.empty_like(x), 4, 2)



def test_s2_shift_and_split_attention(golden_dir):
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    z = np.load(golden_dir + "/ops.npz")
    i = 0
    while "s2shift%d/x" % i in z.files:
        x = torch.from_numpy(z["s2shift%d/x" % i]).to(dev())
        b, d1, d2, c = x.shape
        out = torch.full_like(x, float("nan"))
        E.s2_shift(x, out, b, d1, d2, c, c, c, N.SHIFT_S2_REF)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), torch.from_numpy(z["s2shift%d/ref1" % i]))
        E.s2_shift(x, out, b, d1, d2, c, c, c, N.SHIFT_S2)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), oracle.spatial_shift1(x.cpu(), mode="shift"))
        i += 1
    for mode, omode in ((N.SHIFT_NONE, None), (N.SHIFT_S2, "shift"), (N.SHIFT_S2_REF, "reference_inplace")):
        for dtype in DTYPES:
            B_, H, W, C = 2, 5, 6, 24
            t = rnd((B_, H, W, 3 * C), dtype, 21).to(dev())
            x0, x1, x2 = t[..., :C], t[..., C:2 * C], t[..., 2 * C:]
            a = torch.empty((B_, C), device=dev())
            E.split_sum(x0, x1, x2, 3 * C, 3 * C, 3 * C, B_, H, W, C, mode, a)
            tc = t.cpu().double()
            r0, r1, r2 = tc[..., :C], tc[..., C:2 * C], tc[..., 2 * C:]
            if omode:
                r0, r1 = oracle.spatial_shift1(r0, omode), oracle.spatial_shift2(r1, omode)
            torch.cuda.synchronize()
            assert (a.cpu().double() - (r0 + r1 + r2).sum((1, 2))).abs().max() < 1e-3 * (1 if dtype == torch.float32 else 30)
            hat = rnd((B_, 3 * C), torch.float32, 22).to(dev())
            bar = torch.empty((B_, 3 * C), device=dev())
            E.split_softmax(hat, bar, B_, C)
            out = torch.empty((B_, H, W, C), dtype=dtype, device=dev())
            E.split_apply(x0, x1, x2, 3 * C, 3 * C, 3 * C, B_, H, W, C, mode, bar, out, C)
            torch.cuda.synchronize()
            sm = torch.softmax(hat.cpu().double().reshape(B_, 3, C), 1)
            assert (bar.cpu().double().reshape(B_, 3, C) - sm).abs().max() < 1e-6
            ref = sm[:, 0, None, None] * r0 + sm[:, 1, None, None] * r1 + sm[:, 2, None, None] * r2
            assert (out.cpu().double() - ref).abs().max() < EPS[dtype] * 8


@pytest.mark.parametrize("dtype", DTYPES)
def test_dwconv(dtype):
    pkg = load_pkg()
    E = pkg.engine
    for (B_, H, W, C, k) in [(2, 8, 8, 32, 5), (1, 32, 32, 64, 9), (2, 7, 9, 16, 3)]:
        x = rnd((B_, H, W, C), dtype, 31).to(dev())
        w = rnd((C, 1, k, k), torch.float32, 32, 1.0 / k)
        bias = rnd((C,), torch.float32, 33)
        bns = rnd((C,), torch.float32, 34) * 0.2 + 1
        bnh = rnd((C,), torch.float32, 35)
        wt = w.reshape(C, k * k).t().contiguous().to(dev())        # tap-major [k*k][C]
        out = torch.empty_like(x)
        E.dwconv_nhwc(x, out, B_, H, W, C, k, wt, bias.to(dev()), bns.to(dev()), bnh.to(dev()))
        torch.cuda.synchronize()
        xn = x.cpu().double().permute(0, 3, 1, 2)
        d = oracle.functional.depthwise_conv_same(xn, w.double(), bias.double())
        ref = xn + oracle.gelu(d) * bns.double().view(1, C, 1, 1) + bnh.double().view(1, C, 1, 1)
        assert (out.cpu().double().permute(0, 3, 1, 2) - ref).abs().max() < EPS[dtype] * 8


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_p8_pingpong_tile(dtype):
    """algo 14 (persistent ping-pong tile, whole tiles only): every tile height (64 .. 256 rows, alone and mixed in one
    call), 2 .. 12 K slabs (peeled tail slabs), column groups, several persistent rounds; what it does not take
    (ragged M / N, token-transposed output, fp32, a single or ragged K slab) is refused, not computed wrongly."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (M, Nn, nslab) in enumerate([(64, 256, 2), (128, 512, 3), (192, 256, 5), (320, 256, 4), (1024, 768, 12), (256 * 9 + 192, 1024, 6),
                                         (64 * 301, 256 * 5, 2), (64 * 1000, 256 * 3, 3)]):
        K = nslab * 64
        A = rnd((M, K), dtype, 300 + ci).to(dev())
        B = rnd((Nn, K), dtype, 310 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 320 + ci).to(dev())
        R = rnd((M, Nn), dtype, 330 + ci).to(dev())
        # GELU + residual is served by the LDS-staged epilogue (256-row tiles): other heights are exercised without the residual
        res = M % 256 == 0
        ref = gemm_ref(A.cpu(), B.cpu(), M, Nn, K, bias=bias.cpu(), act=1, R=R.cpu() if res else None, res=1 if res else 0)
        for dbg in (0, 16, 64, 128):           # mixed heights | 256-row tiles + one short panel | staged epilogue | one column group
            if dbg == 64 and M % 256:
                with pytest.raises(N.MlpkError):
                    E.gemm(A, B, torch.empty((M, Nn), dtype=dtype, device=dev()), M, Nn, K, bias=bias, act=1, algo=14, dbg=dbg)
                continue
            C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=1, R=R if res else None, res=1 if res else 0, algo=14, dbg=dbg)
            torch.cuda.synchronize()
            got = C.cpu().double()
            assert torch.isfinite(got).all(), (ci, dbg, "non-finite")
            err = (got - ref).abs().max().item()
            tol = EPS[dtype] * max(1.0, ref.abs().max().item()) * 4
            assert err < tol, (str(dtype), ci, dbg, err, tol)
    # bias + residual without activation = the direct epilogue's residual form (channel fc2), every tile height
    for ci, (M, Nn, nslab) in enumerate([(64, 256, 2), (192, 512, 3), (320, 256, 4), (256 * 9 + 128, 768, 6)]):
        K = nslab * 64
        A = rnd((M, K), dtype, 340 + ci).to(dev())
        B = rnd((Nn, K), dtype, 350 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 360 + ci).to(dev())
        R = rnd((M, Nn), dtype, 370 + ci).to(dev())
        C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
        E.gemm(A, B, C, M, Nn, K, bias=bias, R=R, res=1, algo=14)
        torch.cuda.synchronize()
        ref = gemm_ref(A.cpu(), B.cpu(), M, Nn, K, bias=bias.cpu(), R=R.cpu(), res=1)
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < EPS[dtype] * max(1.0, ref.abs().max().item()) * 4, (str(dtype), "res", ci, err)
    z = lambda *sh: torch.zeros(sh, dtype=dtype, device=dev())
    for (M, Nn, K) in ((64, 256, 64), (64, 256, 160), (100, 256, 128), (64, 200, 128)):      # one slab / ragged slab / ragged M / ragged N
        with pytest.raises(N.MlpkError):
            E.gemm(z(M, K), z(Nn, K), z(M, Nn), M, Nn, K, algo=14)
    with pytest.raises(N.MlpkError):                                                           # token-transposed output
        E.gemm(z(128, 128), z(256, 128), z(256, 64), 128, 256, 128, ldc=64, out_mode=N.OUT_TOKEN_T, t_rows=64, t_tokens=256, algo=14)
    with pytest.raises(N.MlpkError):                                                           # fp32
        E.gemm(torch.zeros((64, 128), device=dev()), torch.zeros((256, 128), device=dev()), torch.zeros((64, 256), device=dev()), 64, 256, 128, algo=14)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_p8_fast_epilogue_classes(dtype):
    """The eight classes of the persistent tile's overlapped epilogue (GELU x folded LayerNorm x column scale/shift), with
    and without a residual, on whole 256 x 256 tiles over several persistent rounds (tiles > CUs), against gemm_ref."""
    pkg = load_pkg()
    E = pkg.engine
    M, Nn, K = 256 * 70, 256 * 4, 192                       # 280 tiles: two rounds on 256 CUs, ragged last round
    A = rnd((M, K), dtype, 500).to(dev())
    B = rnd((Nn, K), dtype, 501, 1.0 / math.sqrt(K)).to(dev())
    bias = rnd((Nn,), torch.float32, 502).to(dev())
    cs = (rnd((Nn,), torch.float32, 503) * 0.2 + 1).to(dev())
    ch = rnd((Nn,), torch.float32, 504).to(dev())
    R = rnd((M, Nn), dtype, 505).to(dev())
    mean = rnd((M,), torch.float32, 506, 0.1).to(dev())
    rstd = (rnd((M,), torch.float32, 507, 0.1) + 1.0).to(dev())
    csum = B.float().sum(dim=1).contiguous()
    for cls in range(8):
        for res in (0, 1):
            act = cls & 1
            ln = (mean, rstd, csum) if cls & 2 else None
            kw = dict(cscale=cs, cshift=ch) if cls & 4 else {}
            C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=act, R=R if res else None, res=res, ln=ln, algo=14, **kw)
            torch.cuda.synchronize()
            acc = A.cpu().double() @ B.cpu().double().T
            if ln is not None:
                acc = (acc - mean.cpu().double()[:, None] * csum.cpu().double()[None, :]) * rstd.cpu().double()[:, None]
            v = acc + bias.cpu().double()[None, :]
            if act:
                v = oracle.gelu(v)
            if cls & 4:
                v = v * cs.cpu().double()[None, :] + ch.cpu().double()[None, :]
            if res:
                v = v + R.cpu().double()
            got = C.cpu().double()
            assert torch.isfinite(got).all(), (str(dtype), cls, res)
            err = (got - v).abs().max().item()
            tol = EPS[dtype] * max(1.0, v.abs().max().item()) * 4
            assert err < tol, (str(dtype), cls, res, err, tol)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("algo", [4, 12, 13])
def test_gemm_token_transposed_staged(dtype, algo):
    """Token-transposed outputs whose tiles lie inside one image take the LDS-staged store path (16-byte stores and
    gate / residual loads in whole channel runs): gate-multiply (gMLP SGU) and residual-add (ResMLP) forms."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (nimg, t_rows, Nn, K, res) in enumerate([(3, 512, 196, 256, 2), (2, 768, 200, 128, 1), (2, 256, 49, 128, 1)]):
        M = nimg * t_rows
        A = rnd((M, K), dtype, 600 + ci).to(dev())
        B = rnd((Nn, K), dtype, 610 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 620 + ci).to(dev())
        R = rnd((nimg * Nn, t_rows), dtype, 630 + ci).to(dev())
        C = torch.full((nimg * Nn, t_rows), float("nan"), dtype=dtype, device=dev())
        E.gemm(A, B, C, M, Nn, K, ldc=t_rows, bias=bias, R=R, ldr=t_rows, res=res, out_mode=N.OUT_TOKEN_T, t_rows=t_rows,
               t_tokens=Nn, algo=algo)
        torch.cuda.synchronize()
        ref = gemm_ref(A.cpu(), B.cpu(), M, Nn, K, bias=bias.cpu(), R=R.cpu(), res=res, out_mode=1, t_rows=t_rows, t_tokens=Nn)
        got = C.cpu().double().reshape(nimg, Nn, t_rows)
        assert torch.isfinite(got).all(), (ci, "non-finite")
        err = (got - ref).abs().max().item()
        tol = EPS[dtype] * max(1.0, ref.abs().max().item()) * 6     # the staged path rounds before the residual / gate
        assert err < tol, (str(dtype), algo, ci, err, tol)


def test_gemm_p8_race_screen_bit_equal_to_s3():
    """The hand-scheduled LDS-DMA / barrier pipeline of algo 14 against the independent s3 tile (algo 13) on the
    channel-MLP shapes: both accumulate k in the same order, so every run must be BIT-equal."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for (M, Nn, K) in ((50176 // 4, 3072, 768), (50176 // 4, 768, 3072)):
        A = rnd((M, K), torch.bfloat16, 400).to(dev())
        B = rnd((Nn, K), torch.bfloat16, 401, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 402).to(dev())
        ref = torch.empty((M, Nn), dtype=torch.bfloat16, device=dev())
        E.gemm(A, B, ref, M, Nn, K, bias=bias, act=1, algo=13)
        for run in range(6):
            C = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=1, algo=14)
            torch.cuda.synchronize()
            assert torch.equal(C.view(torch.int16), ref.view(torch.int16)), (M, Nn, K, run, (C.float() - ref.float()).abs().max().item())
        # every scheduling variant (256-row tiles only, LDS-staged epilogue, a single column group, all three) produces the
        # same bits: tile heights, tile order and the epilogue's store path never change an element's K order
        for dbg in (16, 64, 128, 16 | 64 | 128):
            C = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=1, algo=14, dbg=dbg)
            torch.cuda.synchronize()
            assert torch.equal(C.view(torch.int16), ref.view(torch.int16)), (M, Nn, K, dbg)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("algo", [6, 7, 8, 9, 10, 11, 12, 13])
def test_gemm_direct_to_lds_tiles(dtype, algo):
    """global_load_lds staging: ragged M/N (clamped source rows), K = whole and half slabs, both outputs."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    es = 4 if dtype == torch.float32 else 8
    for ci, (M, Nn, kmul, trans) in enumerate([(300, 200, 12, False), (513, 129, 4, False), (256, 256, 20, False),
                                               (3 * 64, 49, 8, True), (2 * 200, 196, 28, True)]):
        K = kmul * es * 2                                      # multiples of half a 128-byte slab
        A = rnd((M, K), dtype, 200 + ci).to(dev())
        B = rnd((Nn, K), dtype, 210 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 220 + ci).to(dev())
        if not trans:
            R = rnd((M, Nn), dtype, 230 + ci).to(dev())
            C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=1, R=R, res=1, algo=algo)
            ref = gemm_ref(A.cpu(), B.cpu(), M, Nn, K, bias=bias.cpu(), act=1, R=R.cpu(), res=1)
            got = C.cpu().double()
        else:
            t_rows = 64 if ci == 3 else 200
            nimg = M // t_rows
            R = rnd((nimg * Nn, t_rows), dtype, 230 + ci).to(dev())
            C = torch.full((nimg * Nn, t_rows), float("nan"), dtype=dtype, device=dev())
            E.gemm(A, B, C, M, Nn, K, ldc=t_rows, bias=bias, R=R, ldr=t_rows, res=1, out_mode=N.OUT_TOKEN_T, t_rows=t_rows,
                   t_tokens=Nn, algo=algo)
            ref = gemm_ref(A.cpu(), B.cpu(), M, Nn, K, bias=bias.cpu(), R=R.cpu(), res=1, out_mode=1, t_rows=t_rows, t_tokens=Nn)
            got = C.cpu().double().reshape(nimg, Nn, t_rows)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all(), (ci, "non-finite")
        err = (got - ref).abs().max().item()
        tol = EPS[dtype] * max(1.0, ref.abs().max().item()) * 4
        assert err < tol, (str(dtype), algo, ci, err, tol)
    with pytest.raises(N.MlpkError):                          # K not a multiple of half a slab -> refused, not wrong
        A = torch.zeros((64, es * 3), dtype=dtype, device=dev())
        E.gemm(A, A, torch.zeros((64, 64), dtype=dtype, device=dev()), 64, 64, es * 3, algo=algo)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_folded_layernorm(dtype):
    """LayerNorm folded into the consuming GEMM (ln_mean/ln_rstd/ln_csum) == LN then Linear."""
    pkg = load_pkg()
    E = pkg.engine
    M, C, Nn = 300, 96, 200
    x = (rnd((M, C), torch.float32, 301) * 1.5 + 0.7).to(dtype).to(dev())
    w = rnd((Nn, C), torch.float32, 302, 1.0 / math.sqrt(C))
    b = rnd((Nn,), torch.float32, 303)
    g = rnd((C,), torch.float32, 304) * 0.2 + 1
    be = rnd((C,), torch.float32, 305) * 0.3
    wp, bp, cs = E.pack_ln_folded(w, b, g, be, dtype, dev())
    mean = torch.empty(M, device=dev())
    rstd = torch.empty(M, device=dev())
    E.row_stats(x, M, C, C, mean, rstd)
    for algo in (0, 5, 12):
        out = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
        E.gemm(x, wp, out, M, Nn, C, bias=bp, act=1, ln=(mean, rstd, cs), algo=algo)
        torch.cuda.synchronize()
        ref = oracle.gelu(oracle.layer_norm(x.cpu().double(), g.double(), be.double()) @ w.double().t() + b.double())
        err = (out.cpu().double() - ref).abs().max().item()
        assert err < EPS[dtype] * 8 * max(1.0, ref.abs().max().item()), (str(dtype), algo, err)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_token_mlp(dtype, layout):
    """mlpk_token_mlp == the two token-mixing GEMMs + GELU + residual (mlp_mixer.py:16-27 with Conv1d k=1),
    incl. ragged tokens (S not a multiple of 16/32), several hidden chunks, rows spanning several images; both weight
    layouts (0: 128-row tiles, hidden through LDS; 1: 256-row tiles, hidden in registers, W2 columns permuted)."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B_, C, S, T) in enumerate([(2, 32, 16, 64), (3, 40, 49, 196), (2, 128, 196, 784), (5, 8, 20, 40), (1, 32, 32, 32), (3, 256, 50, 96), (2, 64, 208, 1024)]):
        sp = E.round_up(S, 32)
        xn = rnd((B_, S, C), dtype, 400 + ci)                                 # LN output (token-major)
        x = rnd((B_ * S, C), dtype, 410 + ci).to(dev())                       # residual stream
        w1 = rnd((T, S), torch.float32, 420 + ci, 1.0 / math.sqrt(S))
        b1 = rnd((T,), torch.float32, 430 + ci)
        w2 = rnd((S, T), torch.float32, 440 + ci, 1.0 / math.sqrt(T))
        b2 = rnd((S,), torch.float32, 450 + ci)
        xt = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
        xt[:, :S] = xn.permute(0, 2, 1).reshape(B_ * C, S).to(dev())
        w1p, b1p, w2p, b2p, nch, lay = E.pack_token_mlp(w1, b1, w2, b2, dtype, dev(), sp, layout=layout)
        assert lay == layout
        x0 = x.clone()
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C, layout=lay)
        torch.cuda.synchronize()
        w1r, w2r = w1.to(dtype).double(), w2.to(dtype).double()
        h = oracle.gelu(torch.einsum("ts,bsc->btc", w1r, xn.double()) + b1.double().view(1, -1, 1))
        h = h.to(dtype).double()                                              # the kernel rounds H to the storage dtype
        ref = x0.cpu().double().reshape(B_, S, C) + torch.einsum("st,btc->bsc", w2r, h) + b2.double().view(1, -1, 1)
        got = x.cpu().double().reshape(B_, S, C)
        assert torch.isfinite(got).all()
        err = (got - ref).abs().max().item()
        assert err < EPS[dtype] * 6 * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)
        if C % 128 == 0:
            # epilogue statistics: same x bit for bit, and mean / rstd of the rows of the ROUNDED x (what the next LayerNorm reads)
            x2 = x0.clone()
            part = torch.full((C // 128, B_ * S, 2), float("nan"), dtype=torch.float32, device=dev())
            E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x2, C, C, stats=part, layout=lay)
            mean = torch.empty(B_ * S, dtype=torch.float32, device=dev())
            rstd = torch.empty_like(mean)
            E.stats_finalize_planar(part, B_ * S, C, mean, rstd, eps=1e-5)
            torch.cuda.synchronize()
            assert torch.equal(x2, x) and not torch.isnan(part).any()
            xd = x.cpu().double()
            mu = xd.mean(1)
            rs = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5)
            assert (mean.cpu().double() - mu).abs().max().item() < 2e-6 * max(1.0, xd.abs().max().item())
            assert ((rstd.cpu().double() - rs).abs() / rs).max().item() < 2e-5
    with pytest.raises(RuntimeError):                                         # partial tiles per image: refused, not mis-summed
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C, stats=torch.zeros(B_ * S * 2, device=dev()), layout=lay)


def test_gemm_two_tile_heights_in_one_launch():
    """The persistent tile's mixed-height plan for Mixer-B/16 fc2 at 256 images (50176 x 768 x 3072: one round of 256-row + two
    rounds of 192-row tiles) goes out as ONE launch (gemm_nt_p8_pair_kernel): same tiles, same K order -- bit-equal to the two
    launches (MLPK_P8_PAIR=0), with and without the by-product statistics; the fp64 check of the shape's class is test_gemm_*'s.
    (The other pairs -- 256-row panels + a 128- or 64-row tail -- are exercised by the bs=256 model tests, which are bit-compared
    with small batches whose plans have no such pair.)"""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    M, Nn, K = 50176, 768, 3072
    dtype = torch.bfloat16
    g = torch.Generator(device=dev()).manual_seed(5)
    A = (torch.rand((M, K), device=dev(), generator=g) - 0.5).to(dtype)
    B = ((torch.rand((Nn, K), device=dev(), generator=g) - 0.5) / 16).to(dtype)
    R = (torch.rand((M, Nn), device=dev(), generator=g) - 0.5).to(dtype)
    bias = torch.rand(Nn, device=dev(), generator=g)
    outs = {}
    for pair in ("1", "0"):
        os.environ["MLPK_P8_PAIR"] = pair
        try:
            for stats in (False, True):
                C = torch.zeros((M, Nn), dtype=dtype, device=dev())
                ws = E.Workspace(dev(), dtype) if stats else None
                got = E.gemm(A, B, C, M, Nn, K, bias=bias, R=R, res=N.RES_ADD, algo=14, part=(ws, "p") if stats else None)
                torch.cuda.synchronize()
                outs[(pair, stats)] = (C, got[0].clone() if stats and got is not None else None)
        finally:
            del os.environ["MLPK_P8_PAIR"]
    for stats in (False, True):
        assert torch.equal(outs[("1", stats)][0], outs[("0", stats)][0])
    assert torch.equal(outs[("1", True)][0], outs[("1", False)][0])
    assert outs[("1", True)][1] is not None and torch.equal(outs[("1", True)][1], outs[("0", True)][1])
    ref = (A[:64].double() @ B.double().t() + bias.double() + R[:64].double())
    assert (outs[("1", False)][0][:64].double() - ref).abs().max().item() < EPS[dtype] * 4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_token_mlp_generated_kernel(dtype):
    """layout 2 of mlpk_token_mlp (the generated one-wave-per-SIMD kernel, csrc/gen/t4gen.py): against the fp64 restatement of
    mlp_mixer.py:16-27, bit-equal to layout 1 in x (same operation sequence), statistics over 64-channel planes; several tiles
    per workgroup (more tiles than CUs), odd and even group counts, a ragged hidden size; shapes it does not take are refused.
    bf16 (round 5): the library answers layout 3 -- the same kernel with the GELU in packed f16 and the hidden KEPT in f16 (W2 packed as f16) --
    held to the fp64 restatement with the hidden and W2 rounded to f16, at HALF the tolerance; the all-bf16 layout 2 stays selectable and is run too."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    S, sp = 196, 224
    cases = [(1, 256, 784), (3, 512, 512), (2, 768, 100), (5, 256, 64), (300, 256, 96)]
    for ci, (B_, C, T, want) in enumerate([c + (None,) for c in cases] + ([c + (2,) for c in cases[:3]] if dtype == torch.bfloat16 else [])):
        xn = rnd((B_, S, C), dtype, 900 + ci)
        x = rnd((B_ * S, C), dtype, 910 + ci).to(dev())
        w1 = rnd((T, S), torch.float32, 920 + ci, 1.0 / math.sqrt(S))
        b1 = rnd((T,), torch.float32, 930 + ci)
        w2 = rnd((S, T), torch.float32, 940 + ci, 1.0 / math.sqrt(T))
        b2 = rnd((S,), torch.float32, 950 + ci)
        xt = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
        xt[:, :S] = xn.permute(0, 2, 1).reshape(B_ * C, S).to(dev())
        w1p, b1p, w2p, b2p, nch, lay = E.pack_token_mlp(w1, b1, w2, b2, dtype, dev(), sp, t_rows=C, layout=want)
        assert lay == (want or (3 if dtype == torch.bfloat16 else 2)), "the generated kernel must take S = 196 with whole 256-channel tiles"
        assert w2p.dtype == (torch.float16 if lay == 3 else dtype)
        hdt = torch.float16 if lay == 3 else dtype                 # storage type of the hidden and of W2
        x0 = x.clone()
        part = torch.full((E.token_mlp_stat_planes(C, lay), B_ * S, 2), float("nan"), dtype=torch.float32, device=dev())
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C, stats=part, layout=lay)
        x_nostats = x0.clone()
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x_nostats, C, C, layout=lay)
        p1 = E.pack_token_mlp(w1, b1, w2, b2, dtype, dev(), sp, layout=1)
        x1 = x0.clone()
        E.token_mlp(xt, sp, B_ * C, S, p1[0], p1[1], p1[2], p1[3], p1[4], x1, C, C, layout=1)
        mean = torch.empty(B_ * S, dtype=torch.float32, device=dev())
        rstd = torch.empty_like(mean)
        E.stats_finalize_planar(part, B_ * S, C, mean, rstd, eps=1e-5)
        # the generic form of the kernel (every iteration with all three stages, dummy groups against the zero W2 group) computes
        # the same sums in the same order as the shipped one (fill / drain iterations without the dummy stages): bit-equal
        os.environ["MLPK_T4_SHAPE"] = "0"
        try:
            x_generic = x0.clone()
            E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x_generic, C, C, layout=lay)
        finally:
            del os.environ["MLPK_T4_SHAPE"]
        torch.cuda.synchronize()
        assert torch.equal(x, x_nostats) and not torch.isnan(part).any()
        assert torch.equal(x, x_generic)
        if B_ <= 8:
            w1r, w2r = w1.to(dtype).double(), w2.to(hdt).double()
            h = oracle.gelu(torch.einsum("ts,bsc->btc", w1r, xn.double()) + b1.double().view(1, -1, 1)).to(hdt).double()
            ref = x0.cpu().double().reshape(B_, S, C) + torch.einsum("st,btc->bsc", w2r, h) + b2.double().view(1, -1, 1)
            got = x.cpu().double().reshape(B_, S, C)
            assert torch.isfinite(got).all()
            err = (got - ref).abs().max().item()
            print("t4 layout %d %s case %d: max err vs fp64 %.3e (max|ref| %.2f)" % (lay, str(dtype)[6:], ci, err, ref.abs().max().item()))
            assert err < EPS[dtype] * (1.5 if lay == 3 else 6) * max(1.0, ref.abs().max().item()), (str(dtype), ci, lay, err)
        d = (x.float() - x1.float()).abs()
        if lay == 3:
            # against the all-bf16 256-row kernel: one noise level apart (its hidden and W2 carry 8 bits)
            assert d.max().item() <= EPS[dtype] * 6 * max(1.0, x1.float().abs().max().item()), (str(dtype), ci, d.max().item())
        else:
            # the same operation sequence as the 256-row kernel: differences only where fp32 sums were formed in another order
            assert d.max().item() <= EPS[dtype] * 2 * max(1.0, x1.float().abs().max().item()), (str(dtype), ci, d.max().item())
            assert (d > 0).float().mean().item() < 0.02, (str(dtype), ci, (d > 0).float().mean().item())
        xd = x.cpu().double()
        mu = xd.mean(1)
        rs = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5)
        assert (mean.cpu().double() - mu).abs().max().item() < 2e-6 * max(1.0, xd.abs().max().item())
        assert ((rstd.cpu().double() - rs).abs() / rs).max().item() < 2e-5
    # not whole 256-channel tiles / another token count: refused, never mis-tiled
    assert N.lib().mlpk_token_mlp_layout_for(E.dtype_code(dtype), 196, 25, 384) not in (2, 3)
    assert N.lib().mlpk_token_mlp_layout_for(E.dtype_code(dtype), 49, 7, 512) not in (2, 3)
    with pytest.raises(RuntimeError):
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C - 128, layout=lay)
    if dtype == torch.float16:
        with pytest.raises(RuntimeError):                           # the f16-hidden form exists for bf16 storage only
            E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C, layout=3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_token_mixing_prenorm_residual_in_one_kernel(dtype):
    """mlpk_token_mlp_ln == x + FeedForward(LayerNorm(x) transposed) (mlp_mixer.py:6-13 with :16-27, :34): against the fp64 oracle with the
    hidden and the normalised operand rounded where the kernel rounds them, and against the two-kernel path (mlpk_layernorm_transpose +
    mlpk_token_mlp layout 2), from which it differs only by the rounding of the LayerNorm's fp32 expression; by-product statistics;
    several tiles per workgroup, both parities of the group count."""
    pkg = load_pkg()
    E = pkg.engine
    S, sp = 196, 224
    for ci, (B_, C, T) in enumerate([(1, 256, 784), (3, 512, 512), (2, 768, 100), (300, 256, 96)]):
        x = (rnd((B_ * S, C), dtype, 1500 + ci) * 2 + 0.3).to(dtype).to(dev())
        g = rnd((C,), torch.float32, 1510 + ci) + 1.1
        be = rnd((C,), torch.float32, 1520 + ci)
        w1 = rnd((T, S), torch.float32, 1530 + ci, 1.0 / math.sqrt(S))
        b1 = rnd((T,), torch.float32, 1540 + ci)
        w2 = rnd((S, T), torch.float32, 1550 + ci, 1.0 / math.sqrt(T))
        b2 = rnd((S,), torch.float32, 1560 + ci)
        w1p, b1p, w2p, b2p, nch, lay = E.pack_token_mlp(w1, b1, w2, b2, dtype, dev(), sp, t_rows=C)
        assert lay == (3 if dtype == torch.bfloat16 else 2)
        hdt = torch.float16 if lay == 3 else dtype
        mean = torch.empty(B_ * S, dtype=torch.float32, device=dev())
        rstd = torch.empty_like(mean)
        E.row_stats(x, B_ * S, C, C, mean, rstd)
        x0 = x.clone()
        part = torch.full((E.token_mlp_stat_planes(C, lay), B_ * S, 2), float("nan"), dtype=torch.float32, device=dev())
        E.token_mlp_ln(x, C, B_ * C, S, mean, rstd, g.to(dev()), be.to(dev()), w1p, b1p, w2p, b2p, nch, C, stats=part, layout=lay)
        # the two-kernel path on the same input
        xt = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
        E.layernorm_transpose(x0, B_, S, C, g.to(dev()), be.to(dev()), xt, sp)
        x2 = x0.clone()
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x2, C, C, layout=lay)
        m2 = torch.empty_like(mean)
        r2 = torch.empty_like(mean)
        E.stats_finalize_planar(part, B_ * S, C, m2, r2, eps=1e-5)
        torch.cuda.synchronize()
        assert torch.isfinite(x.float()).all() and not torch.isnan(part).any()
        d = (x.float() - x2.float()).abs()
        assert d.max().item() <= EPS[dtype] * 4 * max(1.0, x2.float().abs().max().item()), (str(dtype), ci, d.max().item())
        assert (d > 0).float().mean().item() < 0.05, (str(dtype), ci, (d > 0).float().mean().item())
        if B_ <= 8:
            xn = oracle.layer_norm(x0.cpu().double().reshape(B_, S, C), g.double(), be.double()).to(dtype).double()      # (B, S, C), rounded as the operand is
            w1r, w2r = w1.to(dtype).double(), w2.to(hdt).double()
            h = oracle.gelu(torch.einsum("ts,bsc->btc", w1r, xn) + b1.double().view(1, -1, 1)).to(hdt).double()
            ref = x0.cpu().double().reshape(B_, S, C) + torch.einsum("st,btc->bsc", w2r, h) + b2.double().view(1, -1, 1)
            err = (x.cpu().double().reshape(B_, S, C) - ref).abs().max().item()
            print("t4 ln layout %d %s case %d: max err vs fp64 %.3e (max|ref| %.2f)" % (lay, str(dtype)[6:], ci, err, ref.abs().max().item()))
            assert err < EPS[dtype] * (1.5 if lay == 3 else 6) * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)
        xd = x.cpu().double()
        mu = xd.mean(1)
        rs = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5)
        assert (m2.cpu().double() - mu).abs().max().item() < 2e-6 * max(1.0, xd.abs().max().item())
        assert ((r2.cpu().double() - rs).abs() / rs).max().item() < 2e-5
    with pytest.raises(RuntimeError):                               # one hidden group: the kernel without fill / drain shaping has no LayerNorm loader
        p1 = E.pack_token_mlp(w1[:32], b1[:32], w2[:, :32], b2, dtype, dev(), sp, t_rows=C)
        E.token_mlp_ln(x, C, B_ * C, S, mean, rstd, g.to(dev()), be.to(dev()), p1[0], p1[1], p1[2], p1[3], p1[4], C, layout=p1[5])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_layernorm_transpose_one_pass(dtype):
    """mlpk_layernorm_transpose == nn.LayerNorm over channels followed by the per-image transpose (mlp_mixer.py:34, :6-13),
    zero K-padding columns, ragged last token tile; against the fp64 oracle and bit-compared with the two-kernel path's layout."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B_, S, C) in enumerate([(2, 196, 768), (3, 49, 128), (1, 33, 1024), (2, 64, 256), (5, 7, 512), (2, 40, 1280), (1, 70, 2048)]):
        sp = E.round_up(S, 32)
        x = (rnd((B_ * S, C), dtype, 1300 + ci) * 3 + 0.7).to(dtype).to(dev())
        g = rnd((C,), torch.float32, 1310 + ci) + 1.2
        be = rnd((C,), torch.float32, 1320 + ci)
        xt = torch.full((B_ * C, sp), float("nan"), dtype=dtype, device=dev())
        E.layernorm_transpose(x, B_, S, C, g.to(dev()), be.to(dev()), xt, sp)
        torch.cuda.synchronize()
        ref = oracle.layer_norm(x.cpu().double().reshape(B_, S, C), g.double(), be.double()).permute(0, 2, 1)   # (B, C, S)
        got = xt.cpu().double().reshape(B_, C, sp)
        assert torch.isfinite(got).all()
        assert (got[:, :, S:] == 0).all()
        err = (got[:, :, :S] - ref).abs().max().item()
        assert err < EPS[dtype] * 1.01 * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)
        # the two-kernel path (row statistics, then normalise + transpose) rounds the same values
        mean = torch.empty(B_ * S, dtype=torch.float32, device=dev())
        rstd = torch.empty_like(mean)
        E.row_stats(x, B_ * S, C, C, mean, rstd)
        xt2 = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
        E.norm_apply(x, B_ * S, C, C, mean=mean, rstd=rstd, gamma=g.to(dev()), beta=be.to(dev()), out_tt=xt2, S=S, ld_tt=sp)
        torch.cuda.synchronize()
        assert ((xt.float() - xt2.float()).abs() <= EPS[dtype] * 2 * xt2.float().abs().clamp(min=1.0)).all()
    with pytest.raises(RuntimeError):
        E.layernorm_transpose(x[:, :40], B_, S, 40, g.to(dev()), be.to(dev()), xt, sp)
    # wide rows (gMLP's SGU: 1536 channels, the second half of a 3072-wide tensor -> row stride 3072)
    B_, S, C = 2, 50, 1536
    sp = E.round_up(S, 32)
    wide = (rnd((B_ * S, 2 * C), dtype, 1340) * 2 - 0.3).to(dtype).to(dev())
    v = wide[:, C:]
    g = rnd((C,), torch.float32, 1341) + 1.2
    be = rnd((C,), torch.float32, 1342)
    xt = torch.full((B_ * C, sp), float("nan"), dtype=dtype, device=dev())
    E.layernorm_transpose(v, B_, S, C, g.to(dev()), be.to(dev()), xt, sp)
    torch.cuda.synchronize()
    ref = oracle.layer_norm(v.cpu().double().reshape(B_, S, C), g.double(), be.double()).permute(0, 2, 1)
    got = xt.cpu().double().reshape(B_, C, sp)
    assert (got[:, :, S:] == 0).all()
    assert (got[:, :, :S] - ref).abs().max().item() < EPS[dtype] * 1.01 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_token_gemm(dtype):
    """mlpk_token_gemm: out[b,t,c] = R[b,t,c] (+|*) rscale[c] * (sum_s W[t,s] xt[b*C+c, s] + bias[t]) -- gMLP's gate (g_mlp.py:17-22,
    R = u inside a wider tensor, MUL) and ResMLP's cross-patch sublayer (res_mlp.py:52-55, in place, ADD, gamma_1) -- against fp64,
    incl. ragged tokens, tiles spanning images, a partial last tile, and the same operation through mlpk_gemm_nt's transposed epilogue.
    The last two cases have more 256-row tiles than the chip has CUs: workgroups then walk SEVERAL tiles, and the weight ring must
    run on across them -- with an odd number of 32-token groups (196 tokens -> 7) the round-2 kernel multiplied the first group of every
    later tile by the wrong weights (found at bs = 256 by test_batch_256_rows_match_small_batch, invisible at the golden batch sizes)."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (B_, C, S, mode) in enumerate([(2, 256, 196, "mul"), (3, 384, 196, "add"), (5, 40, 49, "add"), (1, 64, 32, "none"), (2, 1536, 50, "mul"), (4, 96, 224, "add"),
                                           (90, 768, 196, "mul"), (150, 512, 100, "add")]):
        sp = E.round_up(S, 32)
        xn = rnd((B_, S, C), dtype, 2000 + ci)
        xt = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
        xt[:, :S] = xn.permute(0, 2, 1).reshape(B_ * C, S).to(dev())
        w = rnd((S, S), torch.float32, 2010 + ci, 1.0 / math.sqrt(S))
        bias = rnd((S,), torch.float32, 2020 + ci)
        wp, bp, ng = E.pack_token_gemm(w, bias, dtype, dev())
        wr = w.to(dtype).double()
        core = torch.einsum("ts,bsc->btc", wr, xn.double()) + bias.double().view(1, -1, 1)
        if mode == "mul":
            wide = rnd((B_ * S, 2 * C), dtype, 2030 + ci).to(dev())              # R = first half of a wider tensor (row stride 2C)
            out = torch.full((B_ * S, C), float("nan"), dtype=dtype, device=dev())
            E.token_gemm(xt, sp, B_ * C, S, wp, bp, ng, out, C, C, R=wide, ldr=2 * C, res=N.RES_MUL)
            ref = core * wide.cpu().double()[:, :C].reshape(B_, S, C)
        elif mode == "add":
            x = rnd((B_ * S, C), dtype, 2030 + ci).to(dev())
            g1 = (rnd((C,), torch.float32, 2040 + ci) * 0.3 + 0.5).to(dev())
            ref = x.cpu().double().reshape(B_, S, C) + core * g1.cpu().double().view(1, 1, -1)
            out = x
            E.token_gemm(xt, sp, B_ * C, S, wp, bp, ng, out, C, C, R=x, ldr=C, res=N.RES_ADD, rscale=g1, rperiod=C)   # in place
        else:
            out = torch.full((B_ * S, C), float("nan"), dtype=dtype, device=dev())
            E.token_gemm(xt, sp, B_ * C, S, wp, bp, ng, out, C, C)
            ref = core
        torch.cuda.synchronize()
        got = out.cpu().double().reshape(B_, S, C)
        assert torch.isfinite(got).all(), (str(dtype), ci)
        err = (got - ref).abs().max().item()
        assert err < EPS[dtype] * 4 * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)
    with pytest.raises(RuntimeError):
        E.token_gemm(xt, sp, B_ * C, 300, wp, bp, ng, out, C, C)                # more tokens than the packed groups hold


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_axial_shift_core_in_one_kernel(dtype):
    """mlpk_as_conv2 (round 4; as_mlp.py:64-66,84-93, utils/shift_cuda.py:49-69): GroupNorm affine + GELU, the two axial shifts, conv2_1,
    conv2_2, their GELUs and the sum in one kernel, the shifts applied as LDS read addresses of the MFMA operands -- BIT-EQUAL to the
    three kernels it replaces (mlpk_norm_shift_nhwc writing both shifted copies, then two mlpk_gemm_nt), and within rounding of an
    fp64 restatement through the oracle's shift.  Maps that are not multiples of anything (bands with a short tail, pixel blocks that
    wrap rows), both widths the kernel is built for, several bands per image."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    # (round 6: the kernel is persistent over (image, row segment) units and walks a ring of staged rows -- one image split into several
    #  segments, more units than CUs, a single step per unit, steps with a short tail)
    for ci, (B, H, W, C) in enumerate([(2, 56, 56, 96), (3, 28, 28, 192), (2, 7, 9, 96), (1, 33, 5, 192), (5, 14, 14, 96), (2, 1, 1, 192),
                                       (1, 56, 56, 96), (7, 28, 28, 192), (300, 7, 9, 96), (1, 61, 12, 192)]):
        rows = B * H * W
        t = (rnd((rows, C), dtype, 4000 + ci) * 1.5 + 0.3).to(dev())
        mean = (rnd((B,), torch.float32, 4010 + ci) * 0.2).to(dev())
        rstd = (rnd((B,), torch.float32, 4020 + ci).abs() * 0.3 + 0.6).to(dev())
        gamma = (rnd((C,), torch.float32, 4030 + ci) * 0.3 + 1.0).to(dev())
        beta = (rnd((C,), torch.float32, 4040 + ci) * 0.2).to(dev())
        w1 = rnd((C, C), dtype, 4050 + ci, 1.0 / math.sqrt(C)).to(dev())
        w2 = rnd((C, C), dtype, 4060 + ci, 1.0 / math.sqrt(C)).to(dev())
        b1 = rnd((C,), torch.float32, 4070 + ci).to(dev())
        b2 = rnd((C,), torch.float32, 4080 + ci).to(dev())
        assert E.as_conv2_supported(dtype, H, W, C, 5)
        y = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
        E.as_conv2(t, y, B, H, W, C, 5, mean, rstd, gamma, beta, w1, b1, w2, b2)
        # the sequence it replaces
        uw = torch.empty_like(t)
        uh = torch.empty_like(t)
        E.norm_shift_nhwc(t, uw, uh, B, H, W, C, 5, mean, rstd, gamma, beta, N.ACT_GELU)
        ref = torch.empty_like(t)
        E.gemm(uw, w1, ref, rows, C, C, bias=b1, act=N.ACT_GELU)
        E.gemm(uh, w2, ref, rows, C, C, bias=b2, act=N.ACT_GELU, R=ref, res=N.RES_ADD)
        torch.cuda.synchronize()
        assert torch.isfinite(y.float()).all(), (str(dtype), ci)
        nd = int((y.view(torch.int16) != ref.view(torch.int16)).sum())
        assert nd == 0, (str(dtype), ci, (B, H, W, C), nd, (y.float() - ref.float()).abs().max().item())
        # fp64 restatement on the rounded operand u (NCHW through the oracle's shift, shift_cuda.py:49-69)
        u = torch.nn.functional.gelu((t.double().cpu().reshape(B, H * W, C) - mean.double().cpu().view(B, 1, 1)) * rstd.double().cpu().view(B, 1, 1)
                                     * gamma.double().cpu() + beta.double().cpu()).to(dtype).double()
        un = u.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        sw = oracle.axial_shift_nchw(un, 5, 3).permute(0, 2, 3, 1).reshape(rows, C)
        sh = oracle.axial_shift_nchw(un, 5, 2).permute(0, 2, 3, 1).reshape(rows, C)
        want = torch.nn.functional.gelu(sw @ w1.double().cpu().t() + b1.double().cpu()) + torch.nn.functional.gelu(sh @ w2.double().cpu().t() + b2.double().cpu())
        err = (y.double().cpu() - want).abs().max().item()
        assert err < EPS[dtype] * 4 * max(1.0, want.abs().max().item()), (str(dtype), ci, err)
        # round 6: the same kernel finishing the GroupNorm(1, C) statistics of what it stored (as_mlp.py:52,94) -- same y, the statistics
        # of the ROUNDED values, and an image's statistics independent of the batch it is in (two calls on the workspace: the
        # counters of the segmented small-batch path are left zeroed)
        ws = E.Workspace(dev(), dtype)
        for _ in range(2):
            y2 = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
            mo, ro = E.as_conv2(t, y2, B, H, W, C, 5, mean, rstd, gamma, beta, w1, b1, w2, b2, stats=(ws, "asc"))
            torch.cuda.synchronize()
            assert int((y2.view(torch.int16) != y.view(torch.int16)).sum()) == 0, (str(dtype), ci)
            yd = y.double().cpu().reshape(B, -1)
            mu = yd.mean(dim=1)
            rs = 1.0 / torch.sqrt(yd.var(dim=1, unbiased=False) + 1e-5)
            assert (mo.double().cpu() - mu).abs().max().item() < 5e-6 * max(1.0, mu.abs().max().item()), (str(dtype), ci)
            assert ((ro.double().cpu() - rs).abs() / rs).max().item() < 5e-5, (str(dtype), ci)
        if B > 1:
            last = slice((B - 1) * H * W, rows)
            y1 = torch.empty((H * W, C), dtype=dtype, device=dev())
            m1, r1 = E.as_conv2(t[last].contiguous(), y1, 1, H, W, C, 5, mean[B - 1:].contiguous(), rstd[B - 1:].contiguous(), gamma, beta,
                                w1, b1, w2, b2, stats=(E.Workspace(dev(), dtype), "one"))
            torch.cuda.synchronize()
            assert m1.view(torch.int32)[0].item() == mo.view(torch.int32)[B - 1].item(), (str(dtype), ci)
            assert r1.view(torch.int32)[0].item() == ro.view(torch.int32)[B - 1].item(), (str(dtype), ci)
    assert not E.as_conv2_supported(dtype, 14, 14, 384, 5) and not E.as_conv2_supported(torch.float32, 56, 56, 96, 5)
    with pytest.raises(RuntimeError):
        E.as_conv2(t, t, B, H, W, C, 5, mean, rstd, gamma, beta, w1, b1, w2, b2)       # in place: a band reads its neighbours' rows


def test_gemm_skinny_fp32_kernel():
    """algo 16 (round 4): the small fp32 products of the SplitAttention / re-weighting MLPs (vip.py:42-53; s2_mlp_v2.py:36-47) on the
    whole chip without MFMA -- lanes own output columns (or rows, when there are few columns), four waves split K.  Against fp64, and a
    row's bits do not depend on how many rows the call has (the batch)."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (M, Nn, K, act, has_bias) in enumerate([(256, 384, 768, 1, True), (256, 1152, 384, 0, False), (8192, 24, 768, 0, True), (7, 100, 64, 1, True),
                                                    (64, 24, 768, 0, True), (300, 1000, 1536, 0, True), (2, 48, 16, 1, False)]):
        A = rnd((M, K), torch.float32, 3000 + ci).to(dev())
        B = rnd((Nn, K), torch.float32, 3010 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = rnd((Nn,), torch.float32, 3020 + ci).to(dev()) if has_bias else None
        ref = A.double().cpu() @ B.double().cpu().t()
        if has_bias:
            ref = ref + bias.double().cpu()
        if act:
            ref = torch.nn.functional.gelu(ref)
        outs = {}
        for algo in (16, 0, 5):
            C = torch.full((M, Nn), float("nan"), dtype=torch.float32, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=act, algo=algo)
            torch.cuda.synchronize()
            err = (C.double().cpu() - ref).abs().max().item()
            assert err < 2e-5 * max(1.0, ref.abs().max().item()), (ci, algo, err)
            outs[algo] = C
        if M >= 8:
            k = M // 2
            Ch = torch.full((k, Nn), float("nan"), dtype=torch.float32, device=dev())
            E.gemm(A[:k].contiguous(), B, Ch, k, Nn, K, bias=bias, act=act, algo=16)
            torch.cuda.synchronize()
            assert torch.equal(Ch, outs[16][:k]), ci                       # same rows, half the batch: same bits
    # opt-in (MLPK_GEMM_SKINNY=1): measured neutral on its own and worse beside a persistent GEMM (mlpk_gemm.hip, gemm_prepare)
    assert ctypes_name(pkg, 256, 384, 768) != "gemm_skinny_f32_kernel"
    with pytest.raises(RuntimeError):
        E.gemm(A, B, C, M, Nn, K, algo=16, R=C, res=N.RES_ADD)               # no residual class


def ctypes_name(pkg, M, Nn, K):
    import ctypes
    N = pkg._native
    d = N.GemmDesc()
    d.dtype, d.M, d.N, d.K, d.lda, d.ldb, d.ldc = N.F32, M, Nn, K, K, K, Nn
    d.A = d.B = d.C = 1 << 20
    buf = ctypes.create_string_buffer(96)
    assert N.lib().mlpk_gemm_kernel_name(ctypes.byref(d), buf, 96) == 0
    return buf.value.decode()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_token_gemm_with_the_layernorm_as_its_operand_loader(dtype):
    """mlpk_token_gemm_ln (round 4): out[b,t,c] = R[b,t,c] (+|*) rscale[c] * (sum_s W[t,s] LN(x)[b,s,c] + bias[t]) with LN(x) built inside the
    kernel from the token-major x -- gMLP's spatial gating unit (g_mlp.py:17-22: x = the v half of a wider tensor, LayerNorm over its
    channels, gate u = the other half) and ResMLP's cross-patch sublayer (res_mlp.py:17-19,52-55: Aff, no statistics, residual = the
    affine output).  Against fp64 on the SAME rounded operand, and against the two-kernel path it replaces (normalise + transpose pass,
    then mlpk_token_gemm): the two differ only by the rounding of the LayerNorm expression.  Cases: ragged tokens (196 = 6.1 groups,
    50, 224), tiles spanning images (384 channels), more tiles than CUs (workgroups walk several), a partial last tile."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (B_, C, S, mode) in enumerate([(2, 256, 196, "sgu"), (3, 384, 196, "aff"), (5, 64, 49, "sgu"), (2, 1536, 50, "sgu"), (4, 96, 224, "aff"),
                                           (90, 768, 196, "sgu"), (150, 384, 100, "aff"), (3, 64, 97, "sgu"), (40, 1536, 196, "sgu"),
                                           (4, 32, 66, "aff"), (2, 2048, 128, "sgu"), (2, 2080, 96, "aff"), (300, 32, 224, "sgu")]):
        # (round 5: >= 3 groups of 32 tokens and an even token count run as token_gemm_pipe_kernel -- cases 0, 1, 4, 5, 6, 8; two groups or
        #  an odd count -- cases 2, 3, 7 -- stay on the round-4 kernel; 9 - 12: three groups exactly, the widest image the LDS tables hold
        #  (2048 channels), one more than that (round-4 kernel), seven groups with 32-channel images)
        rows = B_ * S
        w = rnd((S, S), torch.float32, 2110 + ci, 1.0 / math.sqrt(S))
        bias = rnd((S,), torch.float32, 2120 + ci)
        wp, bp, ng = E.pack_token_gemm(w, bias, dtype, dev())
        wr = w.to(dtype).double()
        gamma = (rnd((C,), torch.float32, 2130 + ci) * 0.3 + 1.0).to(dev())
        beta = (rnd((C,), torch.float32, 2140 + ci) * 0.2).to(dev())
        if mode == "sgu":
            wide = (rnd((rows, 2 * C), dtype, 2150 + ci) * 1.5 + 0.25).to(dev())   # h = (u | v), row stride 2C
            v = wide[:, C:]
            mean = torch.empty((rows,), dtype=torch.float32, device=dev())
            rstd = torch.empty((rows,), dtype=torch.float32, device=dev())
            E.row_stats(v, rows, C, 2 * C, mean, rstd)
            out = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
            E.token_gemm_ln(v, 2 * C, B_ * C, S, mean, rstd, gamma, beta, wp, bp, ng, out, C, C, R=wide, ldr=2 * C, res=N.RES_MUL)
            torch.cuda.synchronize()
            vn = ((v.double().cpu() - mean.double().cpu()[:, None]) * rstd.double().cpu()[:, None] * gamma.double().cpu() + beta.double().cpu())
            vn = vn.to(dtype).double().reshape(B_, S, C)                            # the operand is rounded once
            ref = (torch.einsum("ts,bsc->btc", wr, vn) + bias.double().view(1, -1, 1)) * wide.cpu().double()[:, :C].reshape(B_, S, C)
            # the path it replaces
            sp = E.round_up(S, 32)
            vt = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
            E.norm_apply(v, rows, C, 2 * C, mean=mean, rstd=rstd, gamma=gamma, beta=beta, out_tt=vt, S=S, ld_tt=sp)
            two = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
            E.token_gemm(vt, sp, B_ * C, S, wp, bp, ng, two, C, C, R=wide, ldr=2 * C, res=N.RES_MUL)
        else:
            x = rnd((rows, C), dtype, 2150 + ci).to(dev())
            g1 = (rnd((C,), torch.float32, 2160 + ci) * 0.3 + 0.5).to(dev())
            x1 = torch.empty_like(x)                                                  # Aff(x), the residual (res_mlp.py:53-55)
            E.norm_apply(x, rows, C, C, gamma=gamma, beta=beta, out_rm=x1, ld_rm=C)
            out = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
            E.token_gemm_ln(x, C, B_ * C, S, None, None, gamma, beta, wp, bp, ng, out, C, C, R=x1, ldr=C, res=N.RES_ADD, rscale=g1, rperiod=C)
            torch.cuda.synchronize()
            xa = (x.double().cpu() * gamma.double().cpu() + beta.double().cpu()).to(dtype).double().reshape(B_, S, C)
            ref = x1.cpu().double().reshape(B_, S, C) + (torch.einsum("ts,bsc->btc", wr, xa) + bias.double().view(1, -1, 1)) * g1.cpu().double().view(1, 1, -1)
            sp = E.round_up(S, 32)
            xt = torch.zeros((B_ * C, sp), dtype=dtype, device=dev())
            E.norm_apply(x, rows, C, C, gamma=gamma, beta=beta, out_tt=xt, S=S, ld_tt=sp)
            two = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
            E.token_gemm(xt, sp, B_ * C, S, wp, bp, ng, two, C, C, R=x1, ldr=C, res=N.RES_ADD, rscale=g1, rperiod=C)
            # ... and with the residual rebuilt from x itself (R == x), in place: bit-equal to the run with the stored Aff output
            xin = x.clone()
            E.token_gemm_ln(xin, C, B_ * C, S, None, None, gamma, beta, wp, bp, ng, xin, C, C, R=xin, ldr=C, res=N.RES_ADD_AFFINE, rscale=g1, rperiod=C)
            with pytest.raises(RuntimeError):              # plain ADD with R aliasing x no longer selects the affine residual silently
                E.token_gemm_ln(xin, C, B_ * C, S, None, None, gamma, beta, wp, bp, ng, xin, C, C, R=xin, ldr=C, res=N.RES_ADD, rscale=g1, rperiod=C)
            torch.cuda.synchronize()
            assert torch.equal(xin.view(torch.int16), out.view(torch.int16)), (str(dtype), ci, "in place, residual rebuilt")
            # round 5: the affine that FOLLOWS the sublayer (ResMLP's post_affine, res_mlp.py:56) applied where the result is stored -- bit-equal
            # to mlpk_norm_apply on the stored result; shapes outside the pipelined kernel are refused
            pa, pb = (rnd((C,), torch.float32, 2170 + ci) * 0.3 + 1.0).to(dev()), (rnd((C,), torch.float32, 2180 + ci) * 0.2).to(dev())
            if E.token_gemm_ln_post_supported(dtype, S, C, C):
                xp = x.clone()
                E.token_gemm_ln(xp, C, B_ * C, S, None, None, gamma, beta, wp, bp, ng, xp, C, C, R=xp, ldr=C, res=N.RES_ADD_AFFINE, rscale=g1, rperiod=C,
                                post=(pa, pb))
                E.norm_apply(xin, rows, C, C, gamma=pa, beta=pb, out_rm=xin, ld_rm=C)
                torch.cuda.synchronize()
                assert torch.equal(xp.view(torch.int16), xin.view(torch.int16)), (str(dtype), ci, "post affine")
            else:
                with pytest.raises(RuntimeError):
                    E.token_gemm_ln(x.clone(), C, B_ * C, S, None, None, gamma, beta, wp, bp, ng, torch.empty_like(x), C, C, R=x1, ldr=C, res=N.RES_ADD,
                                    rscale=g1, rperiod=C, post=(pa, pb))
        torch.cuda.synchronize()
        got = out.cpu().double().reshape(B_, S, C)
        assert torch.isfinite(got).all(), (str(dtype), ci)
        scale = max(1.0, ref.abs().max().item())
        err = (got - ref).abs().max().item()
        assert err < EPS[dtype] * 4 * scale, (str(dtype), ci, mode, err)
        d2 = (got - two.cpu().double().reshape(B_, S, C)).abs().max().item()
        assert d2 < EPS[dtype] * 4 * scale, (str(dtype), ci, mode, d2)
    with pytest.raises(RuntimeError):
        E.token_gemm_ln(x, C, B_ * C, S, None, None, gamma, beta, wp, bp, ng, out, C, 40)      # t_rows must be whole groups of 32 channels


@pytest.mark.parametrize("dtype", DTYPES)
def test_dwconv_affine_nhwc(dtype):
    """Sparse-MLP depthwise step: x + dwconv_same(scale * x + shift) + bias with zero padding AFTER the affine
    (sparse_mlp.py:84-87), against torch's BatchNorm-affine -> conv2d(groups=C, padding=k//2) in fp32."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B, H, W, C, k) in enumerate(((2, 8, 12, 16, 3), (3, 7, 5, 24, 3), (1, 4, 4, 8, 5), (2, 1, 1, 8, 3))):
        x = rnd((B, H, W, C), dtype, 900 + ci).to(dev())
        w = rnd((C, 1, k, k), torch.float32, 910 + ci, 0.5)
        bias = rnd((C,), torch.float32, 920 + ci)
        sc = rnd((C,), torch.float32, 930 + ci) + 1.5
        sh = rnd((C,), torch.float32, 940 + ci)
        xf = x.float().cpu().permute(0, 3, 1, 2)
        ref = xf + torch.nn.functional.conv2d(xf * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), w, bias, padding=k // 2, groups=C)
        out = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        E.dwconv_affine_nhwc(x, out, B, H, W, C, k, w.reshape(C, k * k).t().contiguous().to(dev()), bias.to(dev()), sc.to(dev()), sh.to(dev()))
        torch.cuda.synchronize()
        err = (out.float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        tol = 2e-5 if dtype == torch.float32 else (4e-3 if dtype == torch.float16 else 3e-2)
        assert err < tol * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)


@pytest.mark.parametrize("dtype", DTYPES)
def test_im2col_strided(dtype):
    """mlpk_im2col: overlapping windows (kernel != stride) with zero padding, both source layouts, against
    torch.nn.functional.unfold (hire_mlp.py:21 7x7 s4 p3 on NCHW; :161 3x3 s2 p1 on channel-last)."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (B, C, H, W, k, s_, p, layout) in enumerate(((2, 3, 20, 28, 7, 4, 3, 0), (2, 8, 9, 7, 3, 2, 1, 1), (1, 16, 8, 8, 3, 2, 1, 1),
                                                         (1, 3, 16, 16, 5, 3, 2, 0))):
        x = rnd((B, C, H, W), dtype, 1000 + ci)
        Ho, Wo = (H + 2 * p - k) // s_ + 1, (W + 2 * p - k) // s_ + 1
        cols = torch.nn.functional.unfold(x.float(), k, padding=p, stride=s_)                    # (B, C*k*k, Ho*Wo), k index ci*k*k + i*k + j
        ref = cols.transpose(1, 2).reshape(B * Ho * Wo, C, k * k)
        K = C * k * k
        kp = (K + 7) // 8 * 8
        out = torch.full((B * Ho * Wo, kp), float("nan"), dtype=dtype, device=dev())
        if layout == 0:
            E.im2col(x.to(dev()), out, B, C, H, W, k, k, s_, s_, p, kp)
            want = ref.reshape(B * Ho * Wo, K)
        else:
            xl = x.permute(0, 2, 3, 1).contiguous().to(dev())
            E.im2col(xl, out, B, C, H, W, k, k, s_, s_, p, kp, layout=N.LAYOUT_NHWC, px_stride=C)
            want = ref.permute(0, 2, 1).reshape(B * Ho * Wo, K)                                    # (i*k + j)*C + ci
        torch.cuda.synchronize()
        got = out.float().cpu()
        assert torch.equal(got[:, :K], want.to(dtype).float()), (str(dtype), ci)
        assert (got[:, K:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_hire_gather_combine(dtype):
    """Hire-MLP region remaps against the reference formulation (circular pad -> roll -> einops fold, hire_mlp.py:127-150)
    written with torch ops: gather builds the branch operands, combine is the inverse map (+ crop) added onto x.  Bit-exact moves."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B, H, W, C, h, w, step) in enumerate(((2, 8, 12, 16, 3, 2, 1), (1, 16, 16, 8, 4, 4, 2), (2, 4, 6, 8, 2, 3, 0), (1, 7, 5, 8, 3, 2, 2))):
        xn = rnd((B, H, W, C), dtype, 1100 + ci)
        Hp, Wp = H + (h - H % h), W + (w - W % w)
        gh, gw = Hp // h, Wp // w
        t = xn.permute(0, 3, 1, 2)
        t = torch.cat([t, t[:, :, :, :Wp - W]], dim=3)
        t = torch.cat([t, t[:, :, :Hp - H, :]], dim=2)
        th, tw = torch.roll(t, step, 2), torch.roll(t, step, 3)
        # 'b c (h group) w -> b (c h) group w' read back as rows (b, group, w) x columns (hh, c); only the columns < W are needed
        ref_h = th.reshape(B, C, h, gh, Wp)[..., :W].permute(0, 3, 4, 2, 1).reshape(B * gh * W, h * C)
        ref_w = tw.reshape(B, C, Hp, w, gw)[:, :, :H].permute(0, 2, 4, 3, 1).reshape(B * H * gw, w * C)
        a_h = torch.full((B * gh * W, h * C), float("nan"), dtype=dtype, device=dev())
        a_w = torch.full((B * H * gw, w * C), float("nan"), dtype=dtype, device=dev())
        E.hire_gather(xn.to(dev()), a_h, a_w, B, H, W, C, h, w, step, h * C, w * C)
        torch.cuda.synchronize()
        assert torch.equal(a_h.cpu(), ref_h) and torch.equal(a_w.cpu(), ref_w), (str(dtype), ci)
        # combine: the inverse maps applied to arbitrary branch outputs
        y_h = rnd((B * gh * W, h * C), dtype, 1110 + ci)
        y_w = rnd((B * H * gw, w * C), dtype, 1120 + ci)
        full_h = torch.zeros((B, C, Hp, Wp), dtype=torch.float32)
        full_h[..., :W] = y_h.float().reshape(B, gh, W, h, C).permute(0, 4, 3, 1, 2).reshape(B, C, Hp, W)
        full_w = torch.zeros((B, C, Hp, Wp), dtype=torch.float32)
        full_w[:, :, :H] = y_w.float().reshape(B, H, gw, w, C).permute(0, 4, 1, 3, 2).reshape(B, C, H, Wp)
        back = (torch.roll(full_h, -step, 2) + torch.roll(full_w, -step, 3))[:, :, :H, :W].permute(0, 2, 3, 1)
        x0 = rnd((B, H, W, C), dtype, 1130 + ci)
        want = (x0.float() + back).to(dtype)
        xg = x0.clone().to(dev())
        E.hire_combine(xg, y_h.to(dev()), y_w.to(dev()), B, H, W, C, h, w, step, h * C, w * C)
        torch.cuda.synchronize()
        err = (xg.float().cpu() - want.float()).abs().max().item()
        assert err <= (1e-6 if dtype == torch.float32 else 2e-2), (str(dtype), ci, err)
        # round 5: the same two kernels without a stored LayerNorm output -- gather_ln normalises what it moves (bit-equal to mlpk_norm_apply
        # followed by the gather), combine_from adds the branch results onto another tensor (bit-equal to the in-place form on a copy)
        rows = B * H * W
        xr = (rnd((rows, C), dtype, 1140 + ci) * 1.5 + 0.2).to(dev())
        gamma, beta = (rnd((C,), torch.float32, 1150 + ci) * 0.3 + 1.0).to(dev()), (rnd((C,), torch.float32, 1160 + ci) * 0.2).to(dev())
        mean, rstd = torch.empty((rows,), dtype=torch.float32, device=dev()), torch.empty((rows,), dtype=torch.float32, device=dev())
        E.row_stats(xr, rows, C, C, mean, rstd)
        xnn = torch.empty_like(xr)
        E.norm_apply(xr, rows, C, C, mean=mean, rstd=rstd, gamma=gamma, beta=beta, out_rm=xnn, ld_rm=C)
        b_h, b_w = torch.full_like(a_h, float("nan")), torch.full_like(a_w, float("nan"))
        E.hire_gather(xnn, b_h, b_w, B, H, W, C, h, w, step, h * C, w * C)
        c_h, c_w = torch.full_like(a_h, float("nan")), torch.full_like(a_w, float("nan"))
        E.hire_gather_ln(xr, mean, rstd, gamma, beta, c_h, c_w, B, H, W, C, h, w, step, h * C, w * C)
        src = rnd((B, H, W, C), dtype, 1170 + ci).to(dev())
        inplace = src.clone()
        E.hire_combine(inplace, y_h.to(dev()), y_w.to(dev()), B, H, W, C, h, w, step, h * C, w * C)
        other = torch.full_like(src, float("nan"))
        E.hire_combine_from(other, src, y_h.to(dev()), y_w.to(dev()), B, H, W, C, h, w, step, h * C, w * C)
        torch.cuda.synchronize()
        assert torch.equal(c_h, b_h) and torch.equal(c_w, b_w), (str(dtype), ci)
        assert torch.equal(other, inplace), (str(dtype), ci)
        # round 6 (ABI 12): the combine that also delivers the LayerNorm statistics of the rows it writes -- the same bits, statistics of exactly them
        third = torch.full_like(src, float("nan"))
        m3 = torch.full((B * H * W,), float("nan"), dtype=torch.float32, device=dev())
        r3 = torch.full((B * H * W,), float("nan"), dtype=torch.float32, device=dev())
        E.hire_combine_stats(third, src, y_h.to(dev()), y_w.to(dev()), B, H, W, C, h, w, step, h * C, w * C, m3, r3, eps=1e-5)
        torch.cuda.synchronize()
        assert torch.equal(third, inplace), (str(dtype), ci)
        od = third.double().reshape(B * H * W, C).cpu()
        assert (m3.cpu().double() - od.mean(1)).abs().max().item() < 1e-5 * max(1.0, od.abs().max().item()), (str(dtype), ci)
        want_r = 1.0 / torch.sqrt(od.var(1, unbiased=False) + 1e-5)
        assert ((r3.cpu().double() - want_r).abs() / want_r).max().item() < 1e-4, (str(dtype), ci)


@pytest.mark.parametrize("dtype", DTYPES)
def test_mixshift_nhwc(dtype):
    """MS-MLP mix-shift (ms_mlp.py:52-67): per-chunk roll along W / H, per-chunk depthwise k x k conv with zero padding, the two
    branches added -- against torch.chunk / torch.roll / conv2d in fp32; chunk sizes that are not whole vectors included."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B, H, W, C, shift, ks) in enumerate(((2, 8, 10, 24, [-2, -1, 0, 1, 2], [1, 1, 3, 5, 7]), (1, 6, 5, 16, [-1, 0, 3], [3, 1, 5]),
                                                  (2, 4, 4, 8, [7], [3]), (1, 7, 9, 12, [2, -3], [5, 3]),
                                                  # outside the LDS-tiled kernel (kernel size 9; a map wider than 56): the gather kernels
                                                  (1, 6, 6, 16, [1, -1], [9, 3]), (1, 4, 64, 8, [2], [3]), (1, 5, 6, 12, [1, 2], [9, 1]))):
        x = rnd((B, H, W, C), dtype, 1200 + ci)
        xf = x.float().permute(0, 3, 1, 2)
        chunks = torch.chunk(xf, len(shift), 1)
        kmax = max(ks)
        w_lr = torch.zeros((kmax * kmax, C)); w_td = torch.zeros((kmax * kmax, C)); b_lr = torch.zeros(C); b_td = torch.zeros(C)
        lr, td, c0 = [], [], 0
        for gi, (xc, s_, k) in enumerate(zip(chunks, shift, ks)):
            cs = xc.shape[1]
            wl, wt = rnd((cs, 1, k, k), torch.float32, 1210 + 10 * ci + gi, 0.5), rnd((cs, 1, k, k), torch.float32, 1250 + 10 * ci + gi, 0.5)
            bl, bt = rnd((cs,), torch.float32, 1290 + 10 * ci + gi), rnd((cs,), torch.float32, 1330 + 10 * ci + gi)
            lr.append(torch.nn.functional.conv2d(torch.roll(xc, s_, 3), wl, bl, padding=k // 2, groups=cs))
            td.append(torch.nn.functional.conv2d(torch.roll(xc, s_, 2), wt, bt, padding=k // 2, groups=cs))
            w_lr[:k * k, c0:c0 + cs] = wl.reshape(cs, k * k).t(); w_td[:k * k, c0:c0 + cs] = wt.reshape(cs, k * k).t()
            b_lr[c0:c0 + cs] = bl; b_td[c0:c0 + cs] = bt
            c0 += cs
        ref = (torch.cat(lr, 1) + torch.cat(td, 1)).permute(0, 2, 3, 1)
        out = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        E.mixshift_nhwc(x.to(dev()), out, B, H, W, C, shift, ks, w_lr.to(dev()), b_lr.to(dev()), w_td.to(dev()), b_td.to(dev()))
        torch.cuda.synchronize()
        err = (out.float().cpu() - ref).abs().max().item()
        tol = 2e-5 if dtype == torch.float32 else (4e-3 if dtype == torch.float16 else 3e-2)
        assert err < tol * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_patch_embed4_conv_and_layernorm_in_one_kernel(dtype):
    """mlpk_patch_embed4 (ABI 12): Conv2d(3 -> C, k = stride = 4) on the NCHW image, flatten, transpose (+ LayerNorm) of swin_mlp.py:324-333 /
    ms_mlp.py:255-262 / as_mlp.py:319,330 -- against F.conv2d + F.layer_norm in fp64 on the rounded operands (the product rounded to the storage type in
    between, as the GEMM it replaces stored it); 16-bit and fp32 images, C = 32 .. 128, maps that are not square, a token count that is not a multiple of a
    workgroup's 128."""
    pkg = load_pkg()
    E = pkg.engine
    F = torch.nn.functional
    for ci, (B, H, W, C, with_ln, src) in enumerate([(2, 32, 32, 96, True, dtype), (3, 20, 12, 32, True, torch.float32), (1, 64, 36, 128, False, dtype),
                                                      (5, 8, 8, 64, True, dtype), (2, 224, 224, 96, True, dtype), (1, 12, 4, 96, False, torch.float32)]):
        assert E.patch_embed4_supported(src, dtype, 3, H, W, C)
        x = rnd((B, 3, H, W), src, 1700 + ci)
        wconv = rnd((C, 3, 4, 4), torch.float32, 1710 + ci, 1.0 / math.sqrt(48))
        bias = rnd((C,), torch.float32, 1720 + ci, 0.3)
        gamma, beta = rnd((C,), torch.float32, 1730 + ci) * 0.3 + 1.0, rnd((C,), torch.float32, 1740 + ci) * 0.2
        wp = E.pack_matrix(wconv, dtype, dev())
        out = torch.full((B * (H // 4) * (W // 4), C), float("nan"), dtype=dtype, device=dev())
        E.patch_embed4(x.to(dev()), wp, bias.to(dev()), out, B, H, W, C, gamma=gamma.to(dev()) if with_ln else None, beta=beta.to(dev()) if with_ln else None, eps=1e-5)
        torch.cuda.synchronize()
        xd = x.to(dtype).double()
        y = F.conv2d(xd, wconv.to(dtype).double(), bias.double(), stride=4).flatten(2).transpose(1, 2).reshape(-1, C)
        y = y.to(dtype).double()
        if with_ln:
            y = F.layer_norm(y, (C,), gamma.double(), beta.double(), 1e-5)
        err = (out.double().cpu() - y).abs().max().item()
        assert torch.isfinite(out.float()).all() and err < EPS[dtype] * 4 * max(1.0, y.abs().max().item()), (str(dtype), ci, err)
    assert not E.patch_embed4_supported(dtype, dtype, 4, 32, 32, 96) and not E.patch_embed4_supported(dtype, dtype, 3, 30, 32, 96)
    assert not E.patch_embed4_supported(dtype, dtype, 3, 32, 32, 160) and not E.patch_embed4_supported(dtype, torch.float32, 3, 32, 32, 96)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_stem7_direct_convolution(dtype):
    """mlpk_stem7 (ABI 12): Conv2d(3 -> C, k = 7, stride = 4, pad 3 / 2) of hire_mlp.py:21 / cycle_mlp.py:261 as a direct convolution -- against F.conv2d in
    fp64 on the rounded operands; 16-bit and fp32 images, an odd number of output rows, heights that are not a multiple of the stride, and the
    LayerNorm statistics of the written rows."""
    pkg = load_pkg()
    E = pkg.engine
    F = torch.nn.functional
    for ci, (B, H, W, C, pad, src) in enumerate([(2, 32, 32, 64, 3, dtype), (3, 30, 40, 64, 2, dtype), (1, 224, 224, 64, 3, dtype), (2, 21, 16, 32, 2, torch.float32),
                                                  (2, 36, 24, 128, 3, torch.float32), (1, 9, 8, 96, 0, dtype)]):
        assert E.stem7_supported(src, dtype, 3, H, W, pad, C)
        x = rnd((B, 3, H, W), src, 1800 + ci)
        wconv = rnd((C, 3, 7, 7), torch.float32, 1810 + ci, 1.0 / math.sqrt(147))
        bias = rnd((C,), torch.float32, 1820 + ci, 0.3)
        w7 = E.pack_stem7(wconv, dtype, dev())
        Ho, Wo = (H + 2 * pad - 7) // 4 + 1, (W + 2 * pad - 7) // 4 + 1
        out = torch.full((B * Ho * Wo, C), float("nan"), dtype=dtype, device=dev())
        mean = torch.full((B * Ho * Wo,), float("nan"), dtype=torch.float32, device=dev())
        rstd = torch.full_like(mean, float("nan"))
        E.stem7(x.to(dev()), w7, bias.to(dev()), out, B, H, W, pad, C, out_stats=(mean, rstd), eps=1e-5)
        plain = torch.full_like(out, float("nan"))
        E.stem7(x.to(dev()), w7, bias.to(dev()), plain, B, H, W, pad, C)
        torch.cuda.synchronize()
        y = F.conv2d(x.to(dtype).double(), wconv.to(dtype).double(), bias.double(), stride=4, padding=pad).permute(0, 2, 3, 1).reshape(-1, C)
        err = (out.double().cpu() - y).abs().max().item()
        assert torch.isfinite(out.float()).all() and err < EPS[dtype] * 4 * max(1.0, y.abs().max().item()), (str(dtype), ci, err)
        assert torch.equal(plain, out), (str(dtype), ci)
        od = out.double().cpu()
        assert (mean.cpu().double() - od.mean(1)).abs().max().item() < 1e-5 * max(1.0, od.abs().max().item()), (str(dtype), ci)
        want_r = 1.0 / torch.sqrt(od.var(1, unbiased=False) + 1e-5)
        assert ((rstd.cpu().double() - want_r).abs() / want_r).max().item() < 1e-4, (str(dtype), ci)
    assert not E.stem7_supported(dtype, dtype, 4, 32, 32, 3, 64) and not E.stem7_supported(dtype, dtype, 3, 32, 30, 3, 64) and not E.stem7_supported(dtype, dtype, 3, 32, 512, 3, 64)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_gemm_nhwc_is_im2col_plus_gemm(dtype):
    """mlpk_conv_gemm_nhwc (ABI 12): the strided 3 x 3 transitions of hire_mlp.py:161 / cycle_mlp.py:220-231 as one product whose operand loader is the
    window (zero slabs outside the map) -- BIT-EQUAL to mlpk_im2col + mlpk_gemm_nt on the same tile (algo 12), which the model goldens hold to the
    reference; Hire-MLP's / CycleMLP's three transitions at small batch, odd maps, a 2 x 2 stride-2 window without padding, by-product statistics."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native

    class WS:
        def get(self, name, shape, dt):
            return torch.full(shape, float("nan"), dtype=dt, device=dev())
    for ci, (B, H, W, Cin, Cout, k, st, pad) in enumerate([(3, 56, 56, 64, 128, 3, 2, 1), (2, 28, 28, 128, 320, 3, 2, 1), (2, 14, 14, 320, 512, 3, 2, 1), (2, 9, 7, 32, 72, 3, 2, 1),
                                                            (2, 8, 12, 96, 192, 2, 2, 0), (1, 5, 5, 64, 64, 3, 1, 1)]):
        assert E.conv_gemm_nhwc_supported(dtype, Cin, k, k, st, pad)
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        K = k * k * Cin
        x = rnd((B * H * W, Cin), dtype, 1900 + ci).to(dev())
        w = (rnd((Cout, K), dtype, 1910 + ci) / math.sqrt(K)).to(dtype).to(dev())
        bias = rnd((Cout,), torch.float32, 1920 + ci).to(dev())
        cols = torch.full((B * Ho * Wo, K), float("nan"), dtype=dtype, device=dev())
        E.im2col(x, cols, B, Cin, H, W, k, k, st, st, pad, K, layout=N.LAYOUT_NHWC, px_stride=Cin)
        want = torch.full((B * Ho * Wo, Cout), float("nan"), dtype=dtype, device=dev())
        part_w = E.gemm(cols, w, want, B * Ho * Wo, Cout, K, bias=bias, algo=12, part=(WS(), "a"))
        got = torch.full_like(want, float("nan"))
        part_g = E.conv_gemm_nhwc(x, w, got, B, H, W, Cin, k, k, st, pad, bias=bias, part=(WS(), "b"))
        torch.cuda.synchronize()
        assert not torch.isnan(got.float()).any() and torch.equal(got, want), (str(dtype), ci, (got.float() - want.float()).abs().max().item())
        assert (part_w is None) == (part_g is None)
        if part_g is not None:
            assert part_g[1] == part_w[1] and torch.equal(part_g[0], part_w[0]), (str(dtype), ci)
    assert not E.conv_gemm_nhwc_supported(dtype, 48, 3, 3, 2, 1) and not E.conv_gemm_nhwc_supported(torch.float32, 64, 3, 3, 2, 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_patch_merging_without_the_merged_tensor(dtype):
    """PatchMerging (swin_mlp.py:193-212; sparse_mlp.py:33-50) without the concatenated tensor: mlpk_merge2x2_row_stats gives the LayerNorm statistics of the
    4 C-wide rows, mlpk_conv_gemm_nhwc multiplies through the 2 x 2 window with the weight's column blocks in tap order (engine.merge_taps) and the LayerNorm
    folded in -- against the reference's own lines (strided slices, cat, LayerNorm, Linear) in fp64 on the rounded operands."""
    pkg = load_pkg()
    E = pkg.engine
    F = torch.nn.functional
    for ci, (B, H, W, C) in enumerate([(2, 8, 8, 96), (3, 14, 6, 32), (1, 28, 28, 192), (2, 4, 6, 384)]):
        x = (rnd((B, H, W, C), dtype, 2000 + ci) * 1.3 + 0.2).to(dtype)
        wlin = rnd((2 * C, 4 * C), torch.float32, 2010 + ci, 1.0 / math.sqrt(4 * C))
        gamma, beta = rnd((4 * C,), torch.float32, 2020 + ci) * 0.3 + 1.0, rnd((4 * C,), torch.float32, 2030 + ci) * 0.2
        rows = B * (H // 2) * (W // 2)
        mean = torch.full((rows,), float("nan"), dtype=torch.float32, device=dev()); rstd = torch.full_like(mean, float("nan"))
        xg = x.reshape(B * H * W, C).to(dev())
        E.merge2x2_row_stats(xg, B, H, W, C, mean, rstd, eps=1e-5)
        xd = x.double()
        cat = torch.cat([xd[:, 0::2, 0::2], xd[:, 1::2, 0::2], xd[:, 0::2, 1::2], xd[:, 1::2, 1::2]], -1).reshape(rows, 4 * C)      # swin_mlp.py:203-208
        torch.cuda.synchronize()
        assert (mean.cpu().double() - cat.mean(1)).abs().max().item() < 1e-5 * max(1.0, cat.abs().max().item()), (str(dtype), ci)
        want_r = 1.0 / torch.sqrt(cat.var(1, unbiased=False) + 1e-5)
        assert ((rstd.cpu().double() - want_r).abs() / want_r).max().item() < 1e-4, (str(dtype), ci)
        # ... and the same statistics combined from the per-pixel LayerNorm statistics a producer delivered (mlpk_merge2x2_stats_combine)
        pm = torch.empty((B * H * W,), dtype=torch.float32, device=dev()); pr = torch.empty_like(pm)
        E.row_stats(xg, B * H * W, C, C, pm, pr, eps=1e-5)
        m2 = torch.full((rows,), float("nan"), dtype=torch.float32, device=dev()); r2 = torch.full_like(m2, float("nan"))
        E.merge2x2_stats_combine(pm, pr, B, H, W, m2, r2, eps_in=1e-5, eps_out=1e-5)
        torch.cuda.synchronize()
        assert (m2 - mean).abs().max().item() < 1e-5 * max(1.0, cat.abs().max().item()), (str(dtype), ci)
        assert ((r2 - rstd).abs() / rstd).max().item() < 1e-4, (str(dtype), ci)
        wp, bp, csum = E.pack_ln_folded(wlin, None, gamma, beta, dtype, dev())
        out = torch.full((rows, 2 * C), float("nan"), dtype=dtype, device=dev())
        E.conv_gemm_nhwc(xg, E.merge_taps(wp, C), out, B, H, W, C, 2, 2, 2, 0, bias=bp, ln=(mean, rstd, csum))
        torch.cuda.synchronize()
        ref = F.linear(F.layer_norm(cat, (4 * C,), gamma.double(), beta.double(), 1e-5), wlin.double())
        err = (out.double().cpu() - ref).abs().max().item()
        assert torch.isfinite(out.float()).all() and err < EPS[dtype] * 12 * max(1.0, ref.abs().max().item()), (str(dtype), ci, err)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_pair_gives_the_bits_of_two_calls(dtype):
    """mlpk_gemm_nt_pair (ABI 12): two independent products in one launch where the dispatch gives both the same "s3" tile -- Hire-MLP's proj_h / proj_w
    pairs (hire_mlp.py:139-143) at their stage-3 and stage-1 sizes, with GELU and without -- and the fall-back (different tile families, fp32): in every case
    exactly the bits of two mlpk_gemm_nt calls."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    cases = [((17920, 160, 960), (17920, 160, 960), N.ACT_GELU), ((17920, 960, 160), (12544, 960, 160), N.ACT_NONE), ((3000, 32, 256), (2816, 32, 256), N.ACT_GELU),
             ((1024, 256, 512), (50176, 3072, 768), N.ACT_NONE), ((300, 96, 64), (333, 72, 40), N.ACT_NONE)]
    for ci, (s0, s1, act) in enumerate(cases):
        ops = []
        for si, (M, Nn, K) in enumerate((s0, s1)):
            A = rnd((M, K), dtype, 1600 + 10 * ci + si).to(dev())
            B = (rnd((Nn, K), dtype, 1605 + 10 * ci + si) / math.sqrt(K)).to(dtype).to(dev())
            bias = rnd((Nn,), torch.float32, 1608 + 10 * ci + si).to(dev())
            ops.append((A, B, bias, M, Nn, K))
        want, got = [], []
        for A, B, bias, M, Nn, K in ops:
            C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=act)
            want.append(C)
            got.append(torch.full((M, Nn), float("nan"), dtype=dtype, device=dev()))
        E.gemm_pair(((ops[0][0], ops[0][1], got[0]) + ops[0][3:], dict(bias=ops[0][2], act=act)),
                    ((ops[1][0], ops[1][1], got[1]) + ops[1][3:], dict(bias=ops[1][2], act=act)))
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (str(dtype), ci)
        assert not torch.isnan(got[0].float()).any() and not torch.isnan(got[1].float()).any()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mixshift_tile_kernel_is_bit_equal_to_the_per_chunk_kernels(dtype):
    """mixshift_tile_kernel (round 6): the whole mix-shift of ms_mlp.py:52-67 as one launch over aligned 32-channel blocks -- the same fused
    multiply-adds in the same order as the per-chunk band / k = 1 kernels it replaces (MLPK_MIXSHIFT_TILE=0), so BIT-EQUAL to them, which
    test_mixshift_nhwc and the model goldens hold to the reference.  MS-MLP-T's four stages (chunks of 20 / 39 / 77 / 154 channels: no chunk
    boundary on a 16-byte vector), bands that do not divide the map, a last block of fewer than 32 channels, one-channel chunks."""
    pkg = load_pkg()
    E = pkg.engine
    cases = ((2, 56, 56, 96, [-2, -1, 0, 1, 2], [1, 1, 3, 5, 7]), (2, 28, 28, 192, [-2, -1, 0, 1, 2], [1, 1, 3, 5, 5]),
             (3, 14, 14, 384, [-2, -1, 0, 1, 2], [1, 1, 3, 3, 3]), (2, 7, 7, 768, [-2, -1, 0, 1, 2], [1, 1, 1, 1, 3]),
             (2, 13, 11, 40, [5, -7, 1], [7, 3, 5]), (1, 9, 50, 8, [1, 2, 3, 4, 5, 6, 7, 8], [1, 3, 5, 7, 7, 5, 3, 1]),
             (2, 10, 6, 72, [0, 30], [5, 7]))
    for ci, (B, H, W, C, shift, ks) in enumerate(cases):
        kmax = max(ks)
        x = rnd((B, H, W, C), dtype, 1500 + ci).to(dev())
        w_lr, w_td = rnd((kmax * kmax, C), torch.float32, 1510 + ci, 0.5).to(dev()), rnd((kmax * kmax, C), torch.float32, 1520 + ci, 0.5).to(dev())
        b_lr, b_td = rnd((C,), torch.float32, 1530 + ci).to(dev()), rnd((C,), torch.float32, 1540 + ci).to(dev())
        outs = []
        for mode in ("1", "0"):
            os.environ["MLPK_MIXSHIFT_TILE"] = mode
            try:
                out = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
                E.mixshift_nhwc(x, out, B, H, W, C, shift, ks, w_lr, b_lr, w_td, b_td)
                torch.cuda.synchronize()
            finally:
                os.environ.pop("MLPK_MIXSHIFT_TILE", None)
            outs.append(out)
        assert not torch.isnan(outs[0].float()).any(), (str(dtype), ci)
        if dtype == torch.float16:
            # the k = 1 kernel's last multiply-add and the conversion are ONE instruction for f16 (v_fma_mixlo_f16: one rounding), here the sum is
            # rounded to fp32 first: a handful of results per million differ by one f16 ulp on the k = 1 chunks -- and only there
            chunk0 = (C + len(ks) - 1) // len(ks)
            k_of = torch.tensor([ks[c // chunk0] for c in range(C)], device=dev())
            a, b_ = outs[0].float(), outs[1].float()
            assert torch.equal(a[..., k_of > 1], b_[..., k_of > 1]), (str(dtype), ci)
            d = (a - b_).abs()
            assert (d <= b_.abs() * 2.0 ** -10 + 1e-7).all() and (d > 0).float().mean().item() < 1e-4, (str(dtype), ci, d.max().item())
            continue
        assert torch.equal(outs[0], outs[1]), (str(dtype), ci, (outs[0].float() - outs[1].float()).abs().max().item())
    # ABI 12: the same kernel delivering the statistics planes of what it stores (one (sum, sum of squares) pair per pixel and 32 channels)
    class WS:
        def get(self, name, shape, dt):
            return torch.full(shape, float("nan"), dtype=dt, device=dev())
    for ci, (B, H, W, C, shift, ks) in enumerate(cases):
        kmax = max(ks)
        x = rnd((B, H, W, C), dtype, 1500 + ci).to(dev())
        w_lr, w_td = rnd((kmax * kmax, C), torch.float32, 1510 + ci, 0.5).to(dev()), rnd((kmax * kmax, C), torch.float32, 1520 + ci, 0.5).to(dev())
        b_lr, b_td = rnd((C,), torch.float32, 1530 + ci).to(dev()), rnd((C,), torch.float32, 1540 + ci).to(dev())
        plain = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        assert E.mixshift_nhwc(x, plain, B, H, W, C, shift, ks, w_lr, b_lr, w_td, b_td) is None
        out = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        got = E.mixshift_nhwc(x, out, B, H, W, C, shift, ks, w_lr, b_lr, w_td, b_td, part=(WS(), "p"))
        torch.cuda.synchronize()
        assert (got is not None) == (C % 32 == 0), (str(dtype), ci)
        assert torch.equal(out, plain), (str(dtype), ci)
        if got is not None:
            buf, nq = got
            assert nq == C // 32 and tuple(buf.shape) == (nq, B * H * W, 2)
            od = out.double().reshape(B * H * W, nq, 32)
            s1, s2 = od.sum(2).t(), (od * od).sum(2).t()
            assert (buf[..., 0].double() - s1).abs().max().item() < 1e-4 * max(1.0, s1.abs().max().item()), (str(dtype), ci)
            assert ((buf[..., 1].double() - s2).abs() / s2.clamp_min(1.0)).max().item() < 1e-5, (str(dtype), ci)
            mean = torch.empty((B * H * W,), dtype=torch.float32, device=dev()); rstd = torch.empty_like(mean)
            E.stats_finalize_planar(buf, B * H * W, C, mean, rstd, eps=1e-6)
            torch.cuda.synchronize()
            o2 = out.double().reshape(B * H * W, C)
            assert (mean.double() - o2.mean(1)).abs().max().item() < 1e-5 * max(1.0, o2.abs().max().item())
            want_r = 1.0 / torch.sqrt(o2.var(1, unbiased=False) + 1e-6)
            assert ((rstd.double() - want_r).abs() / want_r).max().item() < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_window_gather_scatter(dtype):
    """Swin-MLP window partition with the shifted blocks' zero padding and the inverse merge + crop + residual
    (swin_mlp.py:29-60, 122-151) against F.pad / view / permute.  Bit-exact moves."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B, H, W, C, ws, shift) in enumerate(((2, 8, 8, 16, 4, 2), (1, 14, 14, 8, 7, 3), (2, 6, 6, 8, 6, 0), (1, 8, 12, 8, 4, 1))):
        x = rnd((B, H, W, C), dtype, 1400 + ci)
        pl, pr, pt, pb = (ws - shift, shift, ws - shift, shift) if shift > 0 else (0, 0, 0, 0)
        xp = torch.nn.functional.pad(x, (0, 0, pl, pr, pt, pb))
        Hp, Wp = xp.shape[1], xp.shape[2]
        ref = xp.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)
        win = torch.full((B * Hp * Wp, C), float("nan"), dtype=dtype, device=dev())
        E.window_gather(x.to(dev()), win, B, H, W, C, ws, pt, pl, Hp, Wp)
        torch.cuda.synchronize()
        assert torch.equal(win.cpu(), ref), (str(dtype), ci)
        yw = rnd((B * Hp * Wp, C), dtype, 1410 + ci)
        back = yw.float().view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, pt:pt + H, pl:pl + W, :]
        x0 = rnd((B, H, W, C), dtype, 1420 + ci)
        want = (x0.float() + back).to(dtype)
        xg = x0.clone().to(dev())
        E.window_scatter_add(xg, yw.to(dev()), B, H, W, C, ws, pt, pl, Hp, Wp)
        torch.cuda.synchronize()
        assert torch.equal(xg.cpu(), want), (str(dtype), ci)


@pytest.mark.parametrize("dtype", DTYPES)
def test_row_stats_short_rows(dtype):
    """Rows of <= 128 elements take the 16-lanes-per-row kernel: lengths 8..128, row counts that are not multiples of 16,
    a row stride larger than the row; fp64 statistics as the reference."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (rows, C, ld) in enumerate(((1, 8, 8), (37, 96, 96), (16, 128, 128), (50, 64, 192), (3, 24, 40))):
        x = (rnd((rows, ld), torch.float32, 1500 + ci) * 3 + 0.7).to(dtype).to(dev())
        mean = torch.full((rows,), float("nan"), device=dev())
        rstd = torch.full((rows,), float("nan"), device=dev())
        E.row_stats(x, rows, C, ld, mean, rstd, eps=1e-6)
        torch.cuda.synchronize()
        xd = x.cpu().double()[:, :C]
        mu = xd.mean(1)
        var = ((xd - mu[:, None]) ** 2).mean(1)
        assert (mean.cpu().double() - mu).abs().max() < 1e-5, (str(dtype), ci)
        assert ((rstd.cpu().double() - 1 / torch.sqrt(var + 1e-6)).abs() * torch.sqrt(var + 1e-6)).max() < 1e-5, (str(dtype), ci)


@pytest.mark.parametrize("dtype", DTYPES)
def test_cycle_shift_bit_exact(dtype):
    """mlpk_cycle_shift = the sampling half of CycleFC (cycle_mlp.py:104-131): a pure gather, bit-exact against the oracle's
    per-channel shifted copy (itself checked against the explicit deform_conv2d loop); vector and scalar kernels, k = 3 / 5 / 7,
    channel counts that are no multiple of the cycle, maps narrower than the cycle."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B, H, W, C, k) in enumerate(((2, 5, 7, 16, 3), (1, 14, 14, 64, 3), (2, 3, 4, 40, 5), (1, 2, 9, 24, 7), (2, 6, 5, 10, 3), (1, 1, 1, 8, 3))):
        x = rnd((B, H, W, C), dtype, 1500 + ci)
        xg = x.to(dev())
        oh = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        ow = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        E.cycle_shift(xg, oh, ow, B, H, W, C, k, C, C)
        torch.cuda.synchronize()
        xc = x.permute(0, 3, 1, 2)
        eye = torch.eye(C, dtype=x.dtype).reshape(C, C, 1, 1)
        ref_h = oracle.cycle_fc(xc.float(), eye.float(), None, (1, k)).permute(0, 2, 3, 1)
        ref_w = oracle.cycle_fc(xc.float(), eye.float(), None, (k, 1)).permute(0, 2, 3, 1)
        assert torch.equal(oh.float().cpu(), ref_h), (str(dtype), ci)
        assert torch.equal(ow.float().cpu(), ref_w), (str(dtype), ci)
        only = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=dev())
        E.cycle_shift(xg, None, only, B, H, W, C, k, C, C)                      # one output only
        assert torch.equal(only.float().cpu(), ref_w)
        if dtype != torch.float32 and C % 8 == 0:
            # round 5: the same on LayerNorm(x) without storing it -- bit-equal to mlpk_norm_apply followed by the plain shift
            rows = B * H * W
            gamma, beta = (rnd((C,), torch.float32, 1510 + ci) * 0.3 + 1.0).to(dev()), (rnd((C,), torch.float32, 1520 + ci) * 0.2).to(dev())
            mean, rstd = torch.empty((rows,), dtype=torch.float32, device=dev()), torch.empty((rows,), dtype=torch.float32, device=dev())
            E.row_stats(xg.view(rows, C), rows, C, C, mean, rstd)
            xn = torch.empty_like(xg)
            E.norm_apply(xg.view(rows, C), rows, C, C, mean=mean, rstd=rstd, gamma=gamma, beta=beta, out_rm=xn.view(rows, C), ld_rm=C)
            ah, aw = torch.full_like(oh, float("nan")), torch.full_like(ow, float("nan"))
            E.cycle_shift(xn, ah, aw, B, H, W, C, k, C, C)
            bh, bw = torch.full_like(oh, float("nan")), torch.full_like(ow, float("nan"))
            E.cycle_shift_ln(xg, mean, rstd, gamma, beta, bh, bw, B, H, W, C, k, C, C)
            torch.cuda.synchronize()
            assert torch.equal(bh, ah) and torch.equal(bw, aw), (str(dtype), ci)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vip_split_attention_on_permuted_layout(dtype):
    """mlpk_vip_split_apply reads the H- and W-branch GEMM outputs where they lie (inverse rearranges of vip.py:71,76 as load
    addresses; 8 x 8 pixel tiles staged in LDS where the map allows, element gathers otherwise): BIT-equal to unpermute + the
    plain split kernel, and the unpermute itself is pinned to the reference's einops patterns by test_vip_permutes."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (B, H, W, C, seg) in enumerate(((2, 4, 6, 32, 8), (1, 32, 32, 384, 12), (3, 5, 3, 64, 4), (2, 8, 8, 48, 24))):
        G = C // seg
        ldh, ldw = (H * seg + 7) // 8 * 8, (W * seg + 7) // 8 * 8
        zh = rnd((B * W * G, ldh), dtype, 1600 + ci).to(dev())
        zw = rnd((B * H * G, ldw), dtype, 1610 + ci).to(dev())
        xc = rnd((B * H * W, C), dtype, 1620 + ci).to(dev())
        bar = torch.softmax(rnd((B, 3, C), torch.float32, 1630 + ci), dim=1).reshape(B, 3 * C).contiguous().to(dev())
        xh = torch.empty((B * H * W, C), dtype=dtype, device=dev())
        xw = torch.empty((B * H * W, C), dtype=dtype, device=dev())
        E.vip_unpermute(0, zh, xh, B, H, W, C, seg, ldh)
        E.vip_unpermute(1, zw, xw, B, H, W, C, seg, ldw)
        m_ref = torch.empty((B * H * W, C), dtype=dtype, device=dev())
        m = torch.full((B * H * W, C), float("nan"), dtype=dtype, device=dev())
        E.split_apply(xh, xw, xc, C, C, C, B, H, W, C, N.SHIFT_NONE, bar, m_ref, C)
        E.vip_split_apply(zh, zw, xc, ldh, ldw, C, B, H, W, C, seg, bar, m, C)
        torch.cuda.synchronize()
        assert torch.equal(m.view(torch.int16), m_ref.view(torch.int16)), (str(dtype), ci)
    with pytest.raises(N.MlpkError):                                            # seg % 4 != 0 is refused (the host keeps the unfused path)
        E.vip_split_apply(zh, zw, xc, ldh, ldw, C, 1, 2, 2, 12, 6, bar, m, C)


class _Space:
    """the two-argument slice of engine.Workspace that engine.gemm(part=...) uses"""

    def __init__(self):
        self.bufs = {}

    def get(self, name, shape, dtype):
        self.bufs[name] = torch.full(shape, float("nan"), dtype=dtype, device=dev())
        return self.bufs[name]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 7, 11, 12, 13, 14, 15])
def test_gemm_byproduct_row_statistics(dtype, algo):
    """mlpk.h row_part: (sum, sum of squares) of the STORED values per row and column block, from every epilogue that
    delivers them; C itself must not change by a bit, and mlpk_stats_finalize_planar must reproduce LayerNorm / per-sample
    GroupNorm statistics of C (vip.py:66,82; as_mlp.py:90)."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    shapes = [(1000, 384, 96, True, 0), (512, 96, 64, False, 1), (777, 200, 128, True, 1), (1024, 512, 256, True, 0)]
    if algo == 14:
        shapes = [(1024, 512, 256, True, 0), (832, 256, 128, True, 0)]       # whole tiles, bias + residual
    if algo == 15:
        shapes = [(1024, 512, 256, True, 0), (768, 384, 192, True, 0), (2048, 384, 1152, True, 0)]     # the generated tile: bias + residual
    for (M, Nn, K, has_res, act) in shapes:
        A = rnd((M, K), dtype, 500).to(dev())
        B = rnd((Nn, K), dtype, 501, 1.0 / math.sqrt(K)).to(dev())
        bias = (rnd((Nn,), torch.float32, 502) + 0.5).to(dev())
        R = rnd((M, Nn), dtype, 503).to(dev()) if has_res else None
        res = N.RES_ADD if has_res else N.RES_NONE
        ref = torch.empty((M, Nn), dtype=dtype, device=dev())
        E.gemm(A, B, ref, M, Nn, K, bias=bias, act=act, R=R, res=res, algo=algo)
        C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
        sp = _Space()
        got = E.gemm(A, B, C, M, Nn, K, bias=bias, act=act, R=R, res=res, algo=algo, part=(sp, "p"))
        assert got is not None, (M, Nn, K, algo)
        part, nparts = got
        torch.cuda.synchronize()
        assert torch.equal(C.view(torch.int16), ref.view(torch.int16)), (M, Nn, K, algo)
        width = 32                     # every tile writes planes of 32 columns (round 4: one reduction order library-wide, mlpk.h row_part)
        assert nparts == -(-Nn // width)
        c64 = C.double().cpu()
        p64 = part.double().cpu()
        assert not torch.isnan(p64).any()
        for q in range(nparts):
            blk = c64[:, q * width:min(Nn, (q + 1) * width)]
            tol = 1e-5 * blk.abs().sum(1) + 1e-6
            assert ((p64[q, :, 0] - blk.sum(1)).abs() <= tol).all(), (M, Nn, K, algo, q)
            assert ((p64[q, :, 1] - (blk * blk).sum(1)).abs() <= 1e-5 * (blk * blk).sum(1) + 1e-6).all(), (M, Nn, K, algo, q)
        # LayerNorm statistics of the rows
        mean = torch.empty((M,), dtype=torch.float32, device=dev())
        rstd = torch.empty((M,), dtype=torch.float32, device=dev())
        assert tuple(part.shape) == (nparts, M, 2)
        E.stats_finalize_planar(part, M, Nn, mean, rstd, eps=1e-5)
        mu = c64.mean(1)
        rs = 1.0 / torch.sqrt(c64.var(1, unbiased=False) + 1e-5)
        assert (mean.double().cpu() - mu).abs().max().item() <= 1e-5
        assert ((rstd.double().cpu() - rs).abs() / rs).max().item() <= 1e-4
        # per-sample GroupNorm(1, C): groups of `hw` consecutive rows share one statistic
        for hw in ([8] if M % 8 == 0 else []) + ([M // 4] if M % 4 == 0 else []):
            Bn = M // hw
            gm = torch.empty((Bn,), dtype=torch.float32, device=dev())
            gr = torch.empty((Bn,), dtype=torch.float32, device=dev())
            E.stats_finalize_planar(part, Bn, hw * Nn, gm, gr, eps=1e-5, group=hw)
            g64 = c64.reshape(Bn, hw * Nn)
            mu = g64.mean(1)
            rs = 1.0 / torch.sqrt(g64.var(1, unbiased=False) + 1e-5)
            assert (gm.double().cpu() - mu).abs().max().item() <= 1e-5, (hw, nparts)
            assert ((gr.double().cpu() - rs).abs() / rs).max().item() <= 1e-4, (hw, nparts)


def test_gemm_row_parts_refusals():
    """what cannot deliver statistics says so (the caller then runs mlpk_row_stats): fp32, 64-column tiles, token-transposed
    outputs, the persistent tile with an epilogue class it does not instantiate them for."""
    import ctypes
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    sp = _Space()
    A = rnd((256, 64), torch.float32, 1).to(dev())
    B = rnd((128, 64), torch.float32, 2).to(dev())
    C = torch.empty((256, 128), dtype=torch.float32, device=dev())
    assert E.gemm(A, B, C, 256, 128, 64, part=(sp, "p")) is None and not sp.bufs
    A, B, C = A.bfloat16(), B.bfloat16(), C.bfloat16()
    assert E.gemm(A, B, C, 256, 128, 64, algo=5, part=(sp, "p")) is None
    assert E.gemm(A, B, C, 256, 128, 64, part=(sp, "p")) is not None
    d = N.GemmDesc()
    d.dtype, d.M, d.N, d.K, d.lda, d.ldb, d.ldc = N.BF16, 256, 128, 64, 64, 64, 128
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.row_part, d.row_part_ld = sp.bufs["p.4"].data_ptr(), 255      # (N = 128: four planes of 32 columns) a plane shorter than M
    assert N.lib().mlpk_gemm_nt(ctypes.byref(d), None) != 0
    d.row_part_ld, d.algo = 256, 14                                  # persistent tile without a residual: no statistics class
    assert N.lib().mlpk_gemm_nt(ctypes.byref(d), None) != 0


@pytest.mark.gpu
def test_byproduct_statistics_with_a_large_row_mean():
    """Rows whose mean is two orders of magnitude above their spread (a deep residual stream): the by-product (sum, sum of squares)
    pairs are reduced in fp64 by mlpk_stats_finalize_planar, so the variance is not lost to the S2 / n - mean^2 cancellation; against
    the two-pass mlpk_row_stats on the same stored values."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    dtype = torch.bfloat16
    M, Nn, K = 1024, 512, 256
    A = rnd((M, K), dtype, 700).to(dev())
    B = rnd((Nn, K), dtype, 701, 1.0 / math.sqrt(K)).to(dev())
    bias = torch.full((Nn,), 100.0, device=dev())
    R = rnd((M, Nn), dtype, 703).to(dev())
    for algo in (0, 11, 14, 15):
        C = torch.empty((M, Nn), dtype=dtype, device=dev())
        got = E.gemm(A, B, C, M, Nn, K, bias=bias, R=R, res=N.RES_ADD, algo=algo, part=(_Space(), "p"))
        assert got is not None
        mean = torch.empty((M,), dtype=torch.float32, device=dev())
        rstd = torch.empty((M,), dtype=torch.float32, device=dev())
        E.stats_finalize_planar(got[0], M, Nn, mean, rstd, eps=1e-5)
        m2 = torch.empty_like(mean)
        r2 = torch.empty_like(rstd)
        E.row_stats(C, M, Nn, Nn, m2, r2)
        torch.cuda.synchronize()
        assert (mean - m2).abs().max().item() <= 1e-3, algo
        assert ((rstd - r2).abs() / r2).max().item() <= 2e-3, algo


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_q4_generated_tile(dtype):
    """algo 15, the generated one-wave-per-SIMD kernels (csrc/gen/q4gen.py): every epilogue class (bias | + folded LayerNorm | + GELU |
    + both | + residual) on shapes with several tiles per workgroup, rolled iterations, one and several column groups -- against
    fp64, and BIT-EQUAL to the independent s3 tile (same K order, same epilogue operation sequence): the race screen of the
    hand-counted LDS-DMA / barrier protocol (a wrong count shows as differing bits on some tiles)."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (M, Nn, K) in enumerate([(256, 128, 192), (1024, 384, 384), (2048, 768, 768), (4096, 768, 3072), (16384, 3072, 768), (50176, 384, 384)]):
        A = rnd((M, K), dtype, 900 + ci).to(dev())
        B = rnd((Nn, K), dtype, 910 + ci, 1.0 / math.sqrt(K)).to(dev())
        bias = (rnd((Nn,), torch.float32, 920 + ci) * 0.5).to(dev())
        R = rnd((M, Nn), dtype, 930 + ci).to(dev())
        ln3 = ((rnd((M,), torch.float32, 940 + ci) * 0.1).to(dev()), (rnd((M,), torch.float32, 950 + ci).abs() + 0.5).to(dev()), B.float().sum(dim=1).contiguous())
        for gelu, ln, res in ((0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1)):
            kw = dict(R=R, res=N.RES_ADD) if res else {}
            if ln:
                kw["ln"] = ln3
            outs = []
            for algo in (15, 11):
                C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
                E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, **kw)
                outs.append(C)
            torch.cuda.synchronize()
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (str(dtype), M, Nn, K, gelu, ln, res)
            if M * Nn <= 2048 * 768:
                acc = A.double() @ B.double().t()
                v = (acc - ln3[0].double()[:, None] * ln3[2].double()[None, :]) * ln3[1].double()[:, None] + bias.double() if ln else acc + bias.double()
                if gelu:
                    v = torch.nn.functional.gelu(v)
                if res:
                    v = v.to(dtype).double() + R.double()
                err = (outs[0].double() - v).abs().max().item()
                assert err < EPS[dtype] * 4 * max(1.0, v.abs().max().item()), (str(dtype), M, Nn, K, gelu, ln, res, err)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_byproduct_statistics_do_not_depend_on_the_tile(dtype):
    """Round 4 (SURVEY.md section 4 tier 6: a sharded forward == the single forward on the concatenated batch, row for row): which tile
    stores a row follows from the batch size, so the by-product LayerNorm statistics must not depend on the tile.  Every tile family
    reduces planes of 32 columns in ONE order (mlpk.h row_part): the planes of the register-staged, direct-to-LDS, s3, persistent and
    generated tiles are BIT-equal on the same product, and so are the finalized mean / rstd."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for (M, Nn, K) in ((1024, 512, 256), (768, 768, 384), (2048, 256, 1152)):
        A = rnd((M, K), dtype, 520).to(dev())
        B = rnd((Nn, K), dtype, 521, 1.0 / math.sqrt(K)).to(dev())
        bias = (rnd((Nn,), torch.float32, 522) + 0.5).to(dev())
        R = (rnd((M, Nn), dtype, 523) * 3.0 + 1.0).to(dev())
        ref = None
        for algo in (13, 1, 3, 7, 11, 12, 14, 15, 0):
            C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
            sp = _Space()
            got = E.gemm(A, B, C, M, Nn, K, bias=bias, R=R, res=N.RES_ADD, algo=algo, part=(sp, "p"))
            assert got is not None, (M, Nn, K, algo)
            part, nparts = got
            mean = torch.empty((M,), dtype=torch.float32, device=dev())
            rstd = torch.empty((M,), dtype=torch.float32, device=dev())
            E.stats_finalize_planar(part, M, Nn, mean, rstd, eps=1e-5)
            torch.cuda.synchronize()
            assert nparts == Nn // 32
            cur = (C.clone(), part.clone(), mean, rstd)
            if ref is None:
                ref = cur
                continue
            assert torch.equal(cur[0].view(torch.int16), ref[0].view(torch.int16)), (M, Nn, K, algo)
            assert torch.equal(cur[1].view(torch.int32), ref[1].view(torch.int32)), (M, Nn, K, algo, int((cur[1] != ref[1]).sum()))
            assert torch.equal(cur[2], ref[2]) and torch.equal(cur[3], ref[3]), (M, Nn, K, algo)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_channel_mlp_of_a_narrow_stage_in_one_kernel(dtype):
    """mlpk_channel_mlp (round 4): out = R + fc2(gelu(fc1(norm(x)))) on channel-last rows with C <= 192 -- both products, the GELU and
    the residual in one kernel, the hidden never written (as_mlp.py:36-52 with the GroupNorm(1, C) of :343-344 folded; the FeedForward /
    Mlp of every hierarchical family's narrow stages).  Against fp64 on the SAME rounded operands (hidden rounded once to the storage
    type, as the two-GEMM path stores it), and against that two-GEMM path.  Cases: every supported width, a hidden that is not a
    multiple of 32, ragged row counts (partial last tile, fewer tiles than workgroups, more tiles than CUs), LayerNorm (one statistic
    per row) and GroupNorm (one per sample), no norm, no residual, residual from another tensor, in place."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    cases = [(64, 256, 1000, "ln", "x"), (96, 384, 256 * 9 + 40, "gn", "x"), (128, 512, 300, None, "x"), (160, 640, 2048, "ln", None),
             (192, 768, 256 * 5, "gn", "other"), (96, 200, 777, "ln", "x"), (192, 576, 256 * 300 + 8, "gn", "x"), (64, 1024, 512, None, None)]
    for ci, (C, hid, M, norm, res) in enumerate(cases):
        assert E.channel_mlp_fused_supported(dtype, C, hid)
        w1 = rnd((hid, C), torch.float32, 3100 + ci, 1.0 / math.sqrt(C))
        b1 = rnd((hid,), torch.float32, 3110 + ci, 0.3)
        w2 = rnd((C, hid), torch.float32, 3120 + ci, 1.0 / math.sqrt(hid))
        b2 = rnd((C,), torch.float32, 3130 + ci, 0.3)
        gamma = rnd((C,), torch.float32, 3140 + ci) * 0.3 + 1.0
        beta = rnd((C,), torch.float32, 3150 + ci) * 0.2
        x = (rnd((M, C), dtype, 3160 + ci) * 1.5 + 0.25).to(dev())
        other = rnd((M, C), dtype, 3170 + ci).to(dev())
        group = 1
        ln = None
        if norm:
            group = 1 if norm == "ln" else 64
            ns = (M + group - 1) // group
            mean = torch.empty((ns,), dtype=torch.float32, device=dev())
            rstd = torch.empty((ns,), dtype=torch.float32, device=dev())
            if norm == "ln":
                E.row_stats(x, M, C, C, mean, rstd)
            else:
                xs = x.float()
                pad = ns * group - M
                xp = torch.cat([xs, xs[-1:].expand(pad, C)]) if pad else xs      # (the last, partial sample: any finite statistic will do)
                mean.copy_(xp.view(ns, -1).mean(1))
                rstd.copy_(1.0 / torch.sqrt(xp.view(ns, -1).var(1, unbiased=False) + 1e-5))
            ln = (mean, rstd)
        pack = E.pack_channel_mlp_fused(w1, b1, w2, b2, dtype, dev(), gamma if norm else None, beta if norm else None)
        R = x if res == "x" else (other if res == "other" else None)
        out = torch.full((M, C), float("nan"), dtype=dtype, device=dev())
        ws = E.Workspace(dev(), dtype)
        got_part = E.channel_mlp_fused(x, M, C, pack, out, R=R, ln=ln, ln_group=group, part=(ws, "cm.part"))
        torch.cuda.synchronize()
        # the by-product: (sum, sum of squares) of the values written to each row, one plane
        assert got_part is not None and got_part[1] == 1 and tuple(got_part[0].shape) == (1, M, 2)
        od = out.cpu().double()
        sums = got_part[0][0].cpu().double()
        assert (sums[:, 0] - od.sum(1)).abs().max().item() < 1e-4 * max(1.0, od.abs().sum(1).max().item())
        assert (sums[:, 1] - (od * od).sum(1)).abs().max().item() < 1e-4 * max(1.0, (od * od).sum(1).max().item())
        # fp64 on the rounded operands: the folded W1, the hidden rounded once
        xd = x.cpu().double()
        w1f = (w1 * gamma.view(1, -1) if norm else w1).to(dtype).double()
        b1f = (b1 + w1 @ beta if norm else b1).double()
        acc = xd @ w1f.t()
        if norm:
            idx = torch.arange(M) // group
            mu, rs = mean.cpu().double()[idx], rstd.cpu().double()[idx]
            acc = (acc - mu[:, None] * w1f.sum(1)[None, :]) * rs[:, None]
        h = oracle.gelu(acc + b1f[None, :]).to(dtype).double()
        ref = h @ w2.to(dtype).double().t() + b2.double()[None, :]
        if R is not None:
            ref = ref + R.cpu().double()
        got = out.cpu().double()
        assert torch.isfinite(got).all(), (str(dtype), ci)
        scale = max(1.0, ref.abs().max().item())
        err = (got - ref).abs().max().item()
        assert err < EPS[dtype] * 4 * scale, (str(dtype), ci, C, hid, M, err)
        # the two GEMMs it replaces (same folded weights)
        if norm:
            wq, bq, csum = E.pack_ln_folded(w1, b1, gamma, beta, dtype, dev())
        else:
            wq, bq, csum = E.pack_matrix(w1, dtype, dev()), b1.to(dev()), None
        hb = torch.empty((M, E.round_up(hid, 8)), dtype=dtype, device=dev())
        two = torch.empty((M, C), dtype=dtype, device=dev())
        E.gemm(x, wq, hb, M, hid, C, bias=bq, act=N.ACT_GELU, ln=(mean, rstd, csum) if norm else None, ln_group=group)
        E.gemm(hb, E.pack_matrix(w2, dtype, dev()), two, M, C, hid, bias=b2.to(dev()), R=R, res=N.RES_ADD if R is not None else N.RES_NONE)
        torch.cuda.synchronize()
        d2 = (got - two.cpu().double()).abs().max().item()
        assert d2 < EPS[dtype] * 4 * scale, (str(dtype), ci, d2)
        if res == "x":                                                            # in place: bit-equal to the run into a fresh tensor
            xin = x.clone()
            E.channel_mlp_fused(xin, M, C, pack, xin, R=xin, ln=ln, ln_group=group)
            torch.cuda.synchronize()
            assert torch.equal(xin.view(torch.int16), out.view(torch.int16)), (str(dtype), ci, "in place")
    assert not E.channel_mlp_fused_supported(dtype, 224, 896) and not E.channel_mlp_fused_supported(dtype, 80, 320)
    assert not E.channel_mlp_fused_supported(torch.float32, 96, 384) and not E.channel_mlp_fused_supported(dtype, 96, 2048)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_swin_spatial_mlp_half_of_a_block_in_one_kernel(dtype):
    """mlpk_swin_spatial (round 4): x += crop(merge(spatial_mlp(partition(pad(LayerNorm(x)))))) of swin_mlp.py:97-151 in one kernel --
    against an fp64 restatement of those lines (zero padding AFTER the norm, window partition, grouped Conv1d over the window positions
    with one matrix per head of 32 channels, merge, crop, residual) on the rounded operands.  Cases: plain and shifted windows (padding on
    both axes), maps that are not a multiple of the window, 1 .. 24 heads, window sizes 7 and 4."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B, H, W, heads, ws, shift) in enumerate([(2, 14, 14, 3, 7, 0), (3, 14, 14, 3, 7, 3), (2, 7, 7, 24, 7, 0), (1, 28, 21, 6, 7, 3), (2, 8, 8, 1, 4, 2),
                                                      (5, 56, 56, 3, 7, 3), (2, 14, 14, 12, 7, 3), (2, 28, 28, 6, 7, 3), (2, 12, 10, 2, 5, 2)]):
        C, t = heads * 32, ws * ws
        assert E.swin_spatial_supported(dtype, C, heads, ws)
        x = (rnd((B * H * W, C), dtype, 4100 + ci) * 1.3 + 0.2).to(dev())
        gamma = rnd((C,), torch.float32, 4110 + ci) * 0.3 + 1.0
        beta = rnd((C,), torch.float32, 4120 + ci) * 0.2
        wgt = rnd((heads * t, t, 1), torch.float32, 4130 + ci, 1.0 / math.sqrt(t))
        bias = rnd((heads * t,), torch.float32, 4140 + ci, 0.3)
        mean = torch.empty((B * H * W,), dtype=torch.float32, device=dev())
        rstd = torch.empty((B * H * W,), dtype=torch.float32, device=dev())
        E.row_stats(x, B * H * W, C, C, mean, rstd)
        # swin_mlp.py:79-81: padding of a shifted block = [ws - shift, shift] on both axes (left / top first)
        pad_l = pad_t = (ws - shift) if shift else 0
        pad_r = pad_b = shift if shift else 0
        Hp, Wp = H + pad_t + pad_b, W + pad_l + pad_r
        Hp, Wp = -(-Hp // ws) * ws, -(-Wp // ws) * ws                                     # (maps that are not whole windows: more padding behind)
        wp, bp = E.pack_swin_spatial(wgt, bias, heads, ws, dtype, dev())
        got = x.clone()
        E.swin_spatial(got, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, mean, rstd, gamma.to(dev()), beta.to(dev()), wp, bp)
        torch.cuda.synchronize()
        xd = x.cpu().double().reshape(B, H, W, C)
        xn = ((xd - mean.cpu().double().reshape(B, H, W, 1)) * rstd.cpu().double().reshape(B, H, W, 1) * gamma.double() + beta.double()).to(dtype).double()
        xp = torch.zeros((B, Hp, Wp, C), dtype=torch.float64)
        xp[:, pad_t:pad_t + H, pad_l:pad_l + W] = xn
        win = xp.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, t, heads, 32)      # (window, token, head, channel)
        wd = wgt.to(dtype).double().reshape(heads, t, t)
        y = torch.einsum("hts,wshc->wthc", wd, win) + bias.double().reshape(heads, t).t().reshape(1, t, heads, 1)
        y = y.to(dtype).double().reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        ref = xd + y[:, pad_t:pad_t + H, pad_l:pad_l + W]
        g = got.cpu().double().reshape(B, H, W, C)
        assert torch.isfinite(g).all(), (str(dtype), ci)
        scale = max(1.0, ref.abs().max().item())
        err = (g - ref).abs().max().item()
        assert err < EPS[dtype] * 4 * scale, (str(dtype), ci, (B, H, W, heads, ws, shift), err)
        # round 5: the same call delivering the LayerNorm statistics of the rows it wrote -- the same output bits, statistics of exactly them
        got2 = x.clone()
        m2 = torch.full((B * H * W,), float("nan"), dtype=torch.float32, device=dev())
        r2 = torch.full((B * H * W,), float("nan"), dtype=torch.float32, device=dev())
        E.swin_spatial(got2, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, mean, rstd, gamma.to(dev()), beta.to(dev()), wp, bp, out_stats=(m2, r2), eps=1e-5)
        torch.cuda.synchronize()
        assert torch.equal(got2, got), (str(dtype), ci)
        # round 6: the kernel above is the quad-token form (8-byte LDS writes, transposed product); the round-4 form gives the same bits
        got3 = x.clone()
        os.environ["MLPK_SWIN_SPATIAL_Q"] = "0"
        try:
            E.swin_spatial(got3, B, H, W, C, ws, pad_t, pad_l, Hp, Wp, heads, mean, rstd, gamma.to(dev()), beta.to(dev()), wp, bp)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("MLPK_SWIN_SPATIAL_Q", None)
        assert torch.equal(got3, got), (str(dtype), ci, (got3.float() - got.float()).abs().max().item())
        gd = got.cpu().double().reshape(B * H * W, C)
        assert (m2.cpu().double() - gd.mean(1)).abs().max().item() < 1e-5 * scale, (str(dtype), ci)
        want_r = 1.0 / torch.sqrt(gd.var(1, unbiased=False) + 1e-5)
        assert ((r2.cpu().double() - want_r).abs() / want_r).max().item() < 1e-4, (str(dtype), ci)
    assert not E.swin_spatial_supported(dtype, 96, 4, 7) and not E.swin_spatial_supported(dtype, 96, 3, 9) and not E.swin_spatial_supported(torch.float32, 96, 3, 7)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_short_k_linear_gelu_with_resident_rows(dtype):
    """mlpk_linear_gelu (round 4): out = gelu(norm(x) W^T + b) for K <= 512 with the rows resident in registers and no epilogue -- gMLP's
    channel_proj1 (g_mlp.py:28,35) and the fc1 of the K = 384 channel MLPs.  Against fp64 on the rounded operands, against mlpk_gemm_nt with
    the same fold, and the by-product statistics planes (32 columns each, canonical order) against the sums of what was stored -- and
    bit-equal to the planes mlpk_gemm_nt delivers for the same output where it delivers them."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    os.environ["MLPK_LINEAR_GELU"] = "1"                                         # (opt-in: the GEMM tiles are faster on the models' shapes)
    for ci, (M, K, Nn, norm) in enumerate([(256, 256, 1536, "ln"), (512, 384, 1152, "ln"), (768, 128, 96, None), (256 * 5, 192, 768, "gn"), (256, 512, 2048, "ln"),
                                            (256 * 270, 256, 512, "ln"), (1024, 384, 1536, None)]):
        assert E.linear_gelu_supported(dtype, M, K, Nn)
        w = rnd((Nn, K), torch.float32, 5100 + ci, 1.0 / math.sqrt(K))
        b = rnd((Nn,), torch.float32, 5110 + ci, 0.3)
        gamma = rnd((K,), torch.float32, 5120 + ci) * 0.3 + 1.0
        beta = rnd((K,), torch.float32, 5130 + ci) * 0.2
        x = (rnd((M, K), dtype, 5140 + ci) * 1.5 + 0.25).to(dev())
        group, ln = 1, None
        if norm:
            group = 1 if norm == "ln" else 128
            ns = M // group
            mean = torch.empty((ns,), dtype=torch.float32, device=dev())
            rstd = torch.empty((ns,), dtype=torch.float32, device=dev())
            if norm == "ln":
                E.row_stats(x, M, K, K, mean, rstd)
            else:
                xs = x.float().view(ns, -1)
                mean.copy_(xs.mean(1))
                rstd.copy_(1.0 / torch.sqrt(xs.var(1, unbiased=False) + 1e-5))
            ln = (mean, rstd)
        pack = E.pack_linear_gelu(w, b, dtype, dev(), gamma if norm else None, beta if norm else None)
        out = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev())
        ws = E.Workspace(dev(), dtype)
        got_part = E.linear_gelu(x, M, K, pack, out, ln=ln, ln_group=group, part=(ws, "lg.part"))
        torch.cuda.synchronize()
        wf = (w * gamma.view(1, -1) if norm else w).to(dtype).double()
        bf = (b + w @ beta if norm else b).double()
        acc = x.cpu().double() @ wf.t()
        if norm:
            idx = torch.arange(M) // group
            acc = (acc - mean.cpu().double()[idx][:, None] * wf.sum(1)[None, :]) * rstd.cpu().double()[idx][:, None]
        ref = oracle.gelu(acc + bf[None, :])
        g = out.cpu().double()
        assert torch.isfinite(g).all(), (str(dtype), ci)
        scale = max(1.0, ref.abs().max().item())
        err = (g - ref).abs().max().item()
        assert err < EPS[dtype] * 4 * scale, (str(dtype), ci, (M, K, Nn), err)
        # statistics planes: sums over each group of 32 stored columns
        assert got_part is not None and got_part[1] == Nn // 32 and tuple(got_part[0].shape) == (Nn // 32, M, 2)
        planes = got_part[0].cpu().double()
        gs = g.view(M, Nn // 32, 32)
        assert (planes[:, :, 0].t() - gs.sum(2)).abs().max().item() < 1e-4 * max(1.0, gs.abs().sum(2).max().item())
        assert (planes[:, :, 1].t() - (gs * gs).sum(2)).abs().max().item() < 1e-4 * max(1.0, (gs * gs).sum(2).max().item())
        # the GEMM tiles on the same fold
        if norm:
            wq, bq, csum = E.pack_ln_folded(w, b, gamma, beta, dtype, dev())
        else:
            wq, bq, csum = E.pack_matrix(w, dtype, dev()), b.to(dev()), None
        two = torch.empty((M, Nn), dtype=dtype, device=dev())
        gp = E.gemm(x, wq, two, M, Nn, K, bias=bq, act=N.ACT_GELU, ln=(mean, rstd, csum) if norm else None, ln_group=group, part=(ws, "gemm.part"))
        torch.cuda.synchronize()
        d2 = (g - two.cpu().double()).abs().max().item()
        assert d2 < EPS[dtype] * 4 * scale, (str(dtype), ci, d2)
        if gp is not None and torch.equal(out.view(torch.int16), two.view(torch.int16)):
            assert gp[1] == got_part[1] and torch.equal(gp[0], got_part[0]), (str(dtype), ci, "planes of equal outputs differ")
    assert not E.linear_gelu_supported(dtype, 250, 256, 512) and not E.linear_gelu_supported(dtype, 256, 320, 512) and not E.linear_gelu_supported(dtype, 256, 256, 8192)
    del os.environ["MLPK_LINEAR_GELU"]
    assert not E.linear_gelu_supported(dtype, 256, 256, 512)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vip_branch_in_one_kernel_is_bit_equal_to_rearrange_plus_gemm(dtype):
    """mlpk_vip_branch (round 5): LayerNorm + einops rearrange + the branch Linear of ViP's WeightedPermuteMLP (vip.py:66-76) in ONE kernel,
    the rearrange as the LDS staging order.  BIT-EQUAL -- outputs and the by-product sums -- to the two-kernel path it replaces
    (mlpk_norm_apply writing the rearranged operand + mlpk_gemm_nt), which the einops pins of ops.npz and the model goldens hold to the
    reference; ViP-S7's geometry (32 x 32 pixels, 384 channels in 32 groups of 12) with a ragged number of slabs per workgroup, and a
    rectangular map with K = 128 / 256."""
    pkg = load_pkg()
    E = pkg.engine
    for ci, (B_, H, W, C, seg) in enumerate([(3, 32, 32, 384, 12), (2, 16, 32, 256, 8), (5, 32, 32, 384, 12)]):
        G = C // seg
        rows = B_ * H * W
        x = (rnd((rows, C), dtype, 2100 + ci) * 1.7 + 0.2).to(dtype).to(dev())
        gamma = (rnd((C,), torch.float32, 2110 + ci) * 0.3 + 1.0).to(dev())
        beta = (rnd((C,), torch.float32, 2120 + ci) * 0.2).to(dev())
        mean = torch.empty((rows,), dtype=torch.float32, device=dev())
        rstd = torch.empty_like(mean)
        E.row_stats(x, rows, C, C, mean, rstd)
        assert E.vip_branch_supported(dtype, H, W, C, seg)
        for which, L, O in ((0, H, W), (1, W, H)):
            K = L * seg
            w = rnd((K, K), dtype, 2130 + 2 * ci + which, 1.0 / math.sqrt(K)).to(dev())
            bias = (rnd((K,), torch.float32, 2140 + 2 * ci + which) * 0.3).to(dev())
            ldp = E.round_up(K, 32)
            assert ldp == K
            perm = torch.full((B_ * O * G, ldp), float("nan"), dtype=dtype, device=dev())
            s_old = torch.zeros((B_ * G, O * seg), dtype=torch.float32, device=dev())
            kw = dict(out_ph=perm, sum_ph=s_old) if which == 0 else dict(out_pw=perm, sum_pw=s_old)
            E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=gamma, beta=beta, H=H, W=W, seg=seg, ld_p=ldp, ld_sum=O * seg, **kw)
            z_old = torch.full((B_ * O * G, K), float("nan"), dtype=dtype, device=dev())
            E.gemm(perm, w, z_old, B_ * O * G, K, K, bias=bias)
            z_new = torch.full((B_ * O * G, K), float("nan"), dtype=dtype, device=dev())
            s_new = torch.full((B_ * G, O * seg), float("nan"), dtype=torch.float32, device=dev())
            E.vip_branch(which, x, C, B_, H, W, C, seg, mean, rstd, gamma, beta, w, bias, z_new, K, sums=s_new, ld_sum=O * seg)
            z_nosum = torch.full((B_ * O * G, K), float("nan"), dtype=dtype, device=dev())
            E.vip_branch(which, x, C, B_, H, W, C, seg, mean, rstd, gamma, beta, w, bias, z_nosum, K)
            torch.cuda.synchronize()
            assert torch.isfinite(z_new.float()).all()
            assert torch.equal(z_new.view(torch.int16), z_old.view(torch.int16)), (str(dtype), ci, which, (z_new.float() - z_old.float()).abs().max().item())
            assert torch.equal(z_new, z_nosum)
            assert torch.equal(s_new, s_old), (str(dtype), ci, which)
    # shapes it does not take: refused by the query, and by the call
    assert not E.vip_branch_supported(dtype, 4, 4, 32, 8) and not E.vip_branch_supported(dtype, 32, 32, 384, 16)
    with pytest.raises(RuntimeError):
        E.vip_branch(0, x, C, B_, 7, 32, C, seg, mean, rstd, gamma, beta, w, bias, z_new, K)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_smlp_mix_in_one_kernel(dtype):
    """mlpk_smlp_mix (round 5): Sparse-MLP's sMLP block up to the concatenation behind its eval-mode BatchNorm (sparse_mlp.py:61-72,92) -- x^ =
    s x + h, proj_h along H, proj_w along W, out = [x_h | x_w | x^] -- against fp64 on the same rounded x^, incl. rectangular maps, maps that
    need two 16-position blocks, more units than the chip holds at once (workgroups walk several), the smallest map."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (B_, H, W, C) in enumerate([(2, 14, 14, 64), (3, 28, 28, 32), (2, 7, 7, 96), (2, 9, 20, 64), (70, 14, 14, 384), (1, 32, 32, 32), (2, 1, 5, 32)]):
        assert E.smlp_mix_supported(dtype, H, W, C)
        rows = B_ * H * W
        x = (rnd((rows, C), dtype, 2300 + ci) * 1.5).to(dev())
        s = (rnd((C,), torch.float32, 2310 + ci) * 0.3 + 1.0).to(dev())
        h = (rnd((C,), torch.float32, 2320 + ci) * 0.5).to(dev())
        wh = rnd((H, H), torch.float32, 2330 + ci, 1.0 / math.sqrt(H))
        ww = rnd((W, W), torch.float32, 2340 + ci, 1.0 / math.sqrt(W))
        bh, bw = rnd((H,), torch.float32, 2350 + ci), rnd((W,), torch.float32, 2360 + ci)
        whp, bhp = E.pack_smlp_mix(wh, bh, dtype, dev())
        wwp, bwp = E.pack_smlp_mix(ww, bw, dtype, dev())
        out = torch.full((rows, 3 * C), float("nan"), dtype=dtype, device=dev())
        E.smlp_mix(x, C, B_, H, W, C, s, h, whp, bhp, wwp, bwp, out, 3 * C)
        torch.cuda.synchronize()
        got = out.cpu().double().reshape(B_, H, W, 3 * C)
        assert torch.isfinite(got).all(), (str(dtype), ci)
        xh = out[:, 2 * C:].cpu()                                                       # the kernel's x^: one rounding of the fp32 fma ...
        want = x.cpu().double() * s.cpu().double() + h.cpu().double()
        assert (xh.double() - want).abs().max().item() < EPS[dtype] * max(1.0, want.abs().max().item()), (str(dtype), ci)
        xd = xh.double().reshape(B_, H, W, C)                                           # ... and the operand of both mixes
        ref_h = torch.einsum("gh,bhwc->bgwc", wh.to(dtype).double(), xd) + bh.double().view(1, H, 1, 1)
        ref_w = torch.einsum("vw,bhwc->bhvc", ww.to(dtype).double(), xd) + bw.double().view(1, 1, W, 1)
        for name, ref, sl in (("h", ref_h, slice(0, C)), ("w", ref_w, slice(C, 2 * C))):
            err = (got[..., sl] - ref).abs().max().item()
            assert err < EPS[dtype] * 2 * max(1.0, ref.abs().max().item()), (str(dtype), ci, name, err)
    assert not E.smlp_mix_supported(dtype, 56, 56, 96) and not E.smlp_mix_supported(dtype, 14, 14, 48) and not E.smlp_mix_supported(torch.float32, 14, 14, 64)
    with pytest.raises(RuntimeError):
        E.smlp_mix(x, C, B_, 56, 56, C, s, h, whp, bhp, wwp, bwp, out, 3 * C)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_smlp_mix_with_the_depthwise_sublayer_in_front(dtype):
    """mlpk_smlp_mix_dw (round 5): x' = x + dwconv3x3(BN(x)) + b (sparse_mlp.py:88-91) and the sMLP mixing of x' in one kernel, for maps up to
    15 x 15 -- bit-equal to mlpk_dwconv_affine_nhwc followed by mlpk_smlp_mix, for x' and for all 3 C output columns."""
    pkg = load_pkg()
    E, N = pkg.engine, pkg._native
    for ci, (B_, H, W, C) in enumerate([(2, 14, 14, 64), (3, 7, 7, 96), (70, 14, 14, 384), (2, 15, 13, 32), (2, 9, 12, 64), (1, 1, 1, 32)]):
        assert E.smlp_mix_dw_supported(dtype, H, W, C)
        rows = B_ * H * W
        x = (rnd((rows, C), dtype, 2400 + ci) * 1.5).to(dev())
        dw_w = (rnd((9, C), torch.float32, 2410 + ci) * 0.3).to(dev())
        dw_b = (rnd((C,), torch.float32, 2420 + ci) * 0.2).to(dev())
        dw_s = (rnd((C,), torch.float32, 2430 + ci) * 0.3 + 1.0).to(dev())
        dw_h = (rnd((C,), torch.float32, 2440 + ci) * 0.5).to(dev())
        s = (rnd((C,), torch.float32, 2450 + ci) * 0.3 + 1.0).to(dev())
        h = (rnd((C,), torch.float32, 2460 + ci) * 0.5).to(dev())
        whp, bhp = E.pack_smlp_mix(rnd((H, H), torch.float32, 2470 + ci, 1.0 / math.sqrt(H)), rnd((H,), torch.float32, 2480 + ci), dtype, dev())
        wwp, bwp = E.pack_smlp_mix(rnd((W, W), torch.float32, 2490 + ci, 1.0 / math.sqrt(W)), rnd((W,), torch.float32, 2500 + ci), dtype, dev())
        x1 = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
        E.dwconv_affine_nhwc(x, x1, B_, H, W, C, 3, dw_w, dw_b, dw_s, dw_h)
        two = torch.full((rows, 3 * C), float("nan"), dtype=dtype, device=dev())
        E.smlp_mix(x1, C, B_, H, W, C, s, h, whp, bhp, wwp, bwp, two, 3 * C)
        x2 = torch.full((rows, C), float("nan"), dtype=dtype, device=dev())
        one = torch.full((rows, 3 * C), float("nan"), dtype=dtype, device=dev())
        E.smlp_mix_dw(x, C, B_, H, W, C, dw_w, dw_b, dw_s, dw_h, x2, C, s, h, whp, bhp, wwp, bwp, one, 3 * C)
        torch.cuda.synchronize()
        assert torch.isfinite(one.float()).all() and torch.isfinite(x2.float()).all()
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), (str(dtype), ci, (x2.float() - x1.float()).abs().max().item())
        assert torch.equal(one.view(torch.int16), two.view(torch.int16)), (str(dtype), ci, (one.float() - two.float()).abs().max().item())
    assert not E.smlp_mix_dw_supported(dtype, 28, 28, 192)
    with pytest.raises(RuntimeError):
        E.smlp_mix_dw(x, C, B_, H, W, C, dw_w, dw_b, dw_s, dw_h, x, C, s, h, whp, bhp, wwp, bwp, one, 3 * C)      # not in place
