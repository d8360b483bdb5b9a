"""Tuning/debug aid: SplitAttention's `a` of one ViP block computed (1) by summing the 16-bit branch outputs (what the reference's
formulation does on rounded tensors) and (2) from the by-product input sums + tiny GEMMs; prints both against an fp64 evaluation of
the same operands (measured: fp16 4.5e-2 vs 2.5e-4, bf16 3.8e-1 vs 3.0e-4 on |a| ~ 500)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
torch.manual_seed(0)
for dtype in (torch.float16, torch.bfloat16):
    B, H, W, C, seg = 4, 32, 32, 384, 12
    G = C // seg
    rows = B * H * W
    x = torch.randn(rows, C).to(dtype).cuda()
    g = (torch.randn(C) * 0.2 + 1).cuda(); b = torch.randn(C).cuda() * 0.1
    Wh = (torch.randn(H * seg, H * seg) / (H * seg) ** 0.5); bh = torch.randn(H * seg) * 0.1
    Ww = (torch.randn(W * seg, W * seg) / (W * seg) ** 0.5); bw = torch.randn(W * seg) * 0.1
    Wc = (torch.randn(C, C) / C ** 0.5); bc = torch.randn(C) * 0.1
    ldh, ldw = E.round_up(H * seg, 32), E.round_up(W * seg, 32)
    whp = E.pack_matrix(Wh, dtype, "cuda", kpad=32); wwp = E.pack_matrix(Ww, dtype, "cuda", kpad=32); wcp = E.pack_matrix(Wc, dtype, "cuda")
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    E.row_stats(x, rows, C, C, mean, rstd)
    xn = torch.zeros((rows, C), dtype=dtype, device="cuda")
    ph = torch.zeros((B * W * G, ldh), dtype=dtype, device="cuda"); pw = torch.zeros((B * H * G, ldw), dtype=dtype, device="cuda")
    Ah = torch.zeros((B * G, ldh), device="cuda"); Aw = torch.zeros((B * G, ldw), device="cuda")
    E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=g, beta=b, out_rm=xn, ld_rm=C, out_ph=ph, H=H, W=W, seg=seg, ld_p=ldh, sum_ph=Aw, ld_sum=ldw)
    E.norm_apply(x, rows, C, C, mean=mean, rstd=rstd, gamma=g, beta=b, out_pw=pw, H=H, W=W, seg=seg, ld_p=ldw, sum_pw=Ah, ld_sum=ldh)
    zh = torch.zeros((B * W * G, H * seg), dtype=dtype, device="cuda"); zw = torch.zeros((B * H * G, W * seg), dtype=dtype, device="cuda"); xc = torch.zeros((rows, C), dtype=dtype, device="cuda")
    E.gemm(ph, whp, zh, B * W * G, H * seg, ldh, bias=bh.cuda()); E.gemm(pw, wwp, zw, B * H * G, W * seg, ldw, bias=bw.cuda()); E.gemm(xn, wcp, xc, rows, C, C, bias=bc.cuda())
    a_old = torch.zeros((B, C), device="cuda")
    a_old = (zh.float().reshape(B, W, G, H, seg).sum((1, 3)).reshape(B, C) + zw.float().reshape(B, H, G, W, seg).sum((1, 3)).reshape(B, C)
             + xc.float().reshape(B, H * W, C).sum(1))
    wh, ww = whp.float(), wwp.float()
    sa_wh = wh.view(H, seg, -1).sum(0).contiguous(); sa_ww = ww.view(W, seg, -1).sum(0).contiguous()
    sa_bh = (bh.cuda().view(H, seg).sum(0) * W).contiguous(); sa_bw = (bw.cuda().view(W, seg).sum(0) * H).contiguous()
    sel = torch.zeros((seg, ldh), device="cuda")
    for j in range(seg):
        sel[j, j:H * seg:seg] = 1
    a12 = torch.zeros((B * G, seg), device="cuda"); xb = torch.zeros((B * G, seg), device="cuda"); a_new = torch.zeros((B, C), device="cuda")
    E.gemm(Ah, sa_wh, a12, B * G, seg, ldh, bias=sa_bh)
    E.gemm(Aw, sa_ww, a12, B * G, seg, ldw, bias=sa_bw, R=a12, res=N.RES_ADD)
    E.gemm(Ah, sel, xb, B * G, seg, ldh)
    E.gemm(xb.view(B, C), wcp.float().contiguous(), a_new, B, C, C, bias=(bc.cuda() * H * W).contiguous(), R=a12.view(B, C), res=N.RES_ADD)
    torch.cuda.synchronize()
    # fp64 from the rounded operands
    z64 = (ph.double()[:, :H * seg] @ wh.double()[:, :H * seg].t() + bh.cuda().double()).reshape(B, W, G, H, seg).sum((1, 3)).reshape(B, C)
    z64 += (pw.double()[:, :W * seg] @ ww.double()[:, :W * seg].t() + bw.cuda().double()).reshape(B, H, G, W, seg).sum((1, 3)).reshape(B, C)
    z64 += (xn.double() @ wcp.double().t() + bc.cuda().double()).reshape(B, H * W, C).sum(1)
    print(str(dtype), "max|a| %.3f  old-vs-fp64 %.3e  new-vs-fp64 %.3e  old-vs-new %.3e" % (z64.abs().max().item(), (a_old.double() - z64).abs().max().item(), (a_new.double() - z64).abs().max().item(), (a_old - a_new).abs().max().item()))
