#!/usr/bin/env python3
"""Where does mlpk_token_gemm differ from the token-transposed NT GEMM at full-batch shapes?  (round-3 finding: gMLP at bs=256)"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = "cuda:0"
dt = torch.bfloat16
for (B, F, S, res, strided) in ((256, 768, 196, N.RES_MUL, True), (64, 768, 196, N.RES_MUL, True), (256, 768, 196, N.RES_MUL, False), (256, 384, 196, N.RES_ADD, False), (256, 768, 196, N.RES_NONE, False)):
    sp = E.round_up(S, 32)
    g = torch.Generator(device=dev).manual_seed(1)
    vt = torch.zeros((B * F, sp), device=dev, dtype=dt)
    vt[:, :S] = (torch.rand((B * F, S), device=dev, generator=g) * 2 - 1).to(dt)
    w = ((torch.rand((S, S), device=dev, generator=g) * 2 - 1) / S ** 0.5)
    b = torch.rand(S, device=dev, generator=g)
    rows = B * S
    ldr = 2 * F if strided else F
    h = (torch.rand((rows, ldr), device=dev, generator=g) * 2 - 1).to(dt)
    tg = E.pack_token_gemm(w, b, dt, dev)
    wp = E.pack_matrix(w, dt, dev, kpad=32)
    o1 = torch.full((rows, F), float("nan"), device=dev, dtype=dt)
    o2 = torch.full((rows, F), float("nan"), device=dev, dtype=dt)
    kw = dict(R=h, ldr=ldr, res=res) if res != N.RES_NONE else {}
    E.token_gemm(vt, sp, B * F, S, tg[0], tg[1], tg[2], o1, F, F, **kw)
    E.gemm(vt, wp, o2, B * F, S, sp, ldc=F, bias=E.f32(b, dev), out_mode=N.OUT_TOKEN_T, t_rows=F, t_tokens=S, **kw)
    torch.cuda.synchronize()
    d = (o1.float() - o2.float()).abs()
    bad = (d > 0.02 * o2.float().abs().clamp(min=0.5)).nonzero()
    print("B=%d F=%d S=%d res=%d strided=%s: max diff %.4f, NaN %d, bad %d" % (B, F, S, res, strided, d.nan_to_num(9).max().item(), int(torch.isnan(o1.float()).sum()), bad.shape[0]))
    if bad.shape[0]:
        r, c = bad[:, 0], bad[:, 1]
        img, tok = r // S, r % S
        print("   images:", sorted(set(img.tolist()))[:20], " n tokens:", len(set(tok.tolist())), sorted(set(tok.tolist()))[:12], " channels:", sorted(set(c.tolist()))[:16], len(set(c.tolist())))
        print("   tiles (b*F + c) // 256:", sorted(set(((img * F + c) // 256).tolist()))[:24])
