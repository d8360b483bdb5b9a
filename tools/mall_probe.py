#!/usr/bin/env python3
"""GPU experiment (round 4): would fc2 run faster if its A operand (the fc1 hidden) came out of the 256 MB Infinity Cache
instead of HBM?  Same kernel, same shape, same tile plan; only the residency of A differs:
  warm: one A buffer of ~100 MB re-read by every launch (it stays in the Infinity Cache between launches),
  cold: eight A / C buffers in rotation (800 MB: every launch finds its operand evicted).
If the two rates agree, banding fc1 -> fc2 through the cache (VERDICT r3 item 1 iii) cannot pay for its dependency tracking.
usage: python tools/mall_probe.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev, dt = "cuda:0", torch.bfloat16


def run(title, M, Nn, K, nbuf, algo, res=True, gelu=False, reps=24, rounds=5):
    As = [((torch.rand((M, K), device=dev) * 2 - 1)).to(dt) for _ in range(nbuf)]
    Cs = [torch.zeros((M, Nn), dtype=dt, device=dev) for _ in range(nbuf)]
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)

    def f(i):
        kw = dict(R=Cs[i % nbuf], res=N.RES_ADD) if res else {}
        E.gemm(As[i % nbuf], B, Cs[i % nbuf], M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, **kw)
    for i in range(2 * nbuf):
        f(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts.sort()
    med = ts[len(ts) // 2]
    print("   %-34s A %4.0f MB x %d   median %.4f ms   %7.1f TFLOP/s" % (title, M * K * 2 / 1e6, nbuf, med, 2.0 * M * Nn * K / med / 1e9), flush=True)
    return med


for name, M, Nn, K, algo, res, gelu in (("fc2 p8", 16384, 768, 3072, 14, True, False), ("fc2 p8", 21504, 768, 3072, 14, True, False),
                                        ("fc2 q4", 16384, 768, 3072, 15, True, False), ("fc1 q4", 16384, 3072, 768, 15, False, True)):
    print("== %s M=%d N=%d K=%d" % (name, M, Nn, K))
    w = run("warm (Infinity Cache resident)", M, Nn, K, 1, algo, res, gelu)
    c = run("cold (8 buffers in rotation)", M, Nn, K, 8, algo, res, gelu)
    print("   cold / warm = %.3f" % (c / w))
