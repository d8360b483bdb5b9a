#!/usr/bin/env python3
"""Tuning aid: effect of de-phasing co-resident workgroups (dbg bit 16, delay in bits 8..15)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dt = torch.bfloat16
for name, M, Nn, K in [("fc1", 50176, 3072, 768), ("fc2", 50176, 768, 3072)]:
    A = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device="cuda") * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device="cuda")
    C = torch.zeros((M, Nn), dtype=dt, device="cuda")
    kw = dict(bias=bias, act=1) if name == "fc1" else dict(bias=bias, R=C, res=1)
    for algo in (11, 13):
        for delay in (0, 1, 2, 3, 4, 6):
            dbg = (16 | (delay << 8)) if delay else 0
            for _ in range(2):
                E.gemm(A, B, C, M, Nn, K, algo=algo, dbg=dbg, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                E.gemm(A, B, C, M, Nn, K, algo=algo, dbg=dbg, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("%s algo=%d delay=%d  %.3f ms  %.1f TF" % (name, algo, delay, ms, 2.0 * M * Nn * K / ms / 1e9), flush=True)
