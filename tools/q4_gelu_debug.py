#!/usr/bin/env python3
"""debug: where do q4 (algo 15) and s3 (algo 11) differ for the gelu + ln class?  (operands of tests/test_gpu_ops.py::test_gemm_q4_generated_tile)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import q4_probe as qp
E, N, dev = qp.E, qp.N, qp.dev


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


dtype = torch.bfloat16
for ci, (M, Nn, K) in enumerate([(256, 128, 192), (1024, 384, 384), (2048, 768, 768), (4096, 768, 3072), (16384, 3072, 768), (50176, 384, 384)]):
    A = rnd((M, K), dtype, 900 + ci).to(dev)
    B = rnd((Nn, K), dtype, 910 + ci, 1.0 / math.sqrt(K)).to(dev)
    bias = (rnd((Nn,), torch.float32, 920 + ci) * 0.5).to(dev)
    R = rnd((M, Nn), dtype, 930 + ci).to(dev)
    ln3 = ((rnd((M,), torch.float32, 940 + ci) * 0.1).to(dev), (rnd((M,), torch.float32, 950 + ci).abs() + 0.5).to(dev), B.float().sum(dim=1).contiguous())
    for gelu, ln, res in ((1, 0, 0), (1, 1, 0)):
        kw = dict(ln=ln3) if ln else {}
        outs = []
        for algo in (15, 11, 13):
            C = torch.full((M, Nn), float("nan"), dtype=dtype, device=dev)
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, **kw)
            outs.append(C)
        torch.cuda.synchronize()
        d = outs[0].view(torch.int16) != outs[1].view(torch.int16)
        d2 = outs[2].view(torch.int16) != outs[1].view(torch.int16)
        print("M=%d N=%d K=%d gelu=%d ln=%d: q4 vs s3 differ %d of %d; s3(13) vs s3(11) differ %d" % (M, Nn, K, gelu, ln, int(d.sum()), d.numel(), int(d2.sum())))
        acc = None
        for r, c in d.nonzero()[:8].tolist():
            a = (A[r].double() * B[c].double()).sum().item()
            v = (a - ln3[0][r].item() * ln3[2][c].item()) * ln3[1][r].item() + bias[c].item() if ln else a + bias[c].item()
            print("    [%d, %d] pre-activation %.6f  q4 %r  s3 %r" % (r, c, v, float(outs[0][r, c]), float(outs[1][r, c])))
