#!/usr/bin/env python3
"""Tuning aid: epilogue/main-loop ablations of one GEMM shape (desc.reserved debug bits)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = "cuda:0"
dt = torch.bfloat16
algos = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "6").split(",")]
for name, M, Nn, K in [("fc1", 50176, 3072, 768), ("fc2", 50176, 768, 3072), ("gproj1", 50176, 3072, 256)]:
    A = (torch.rand((M, K), device=dev) * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    for algo in algos:
        for label, kw in [("full gelu", dict(bias=bias, act=1)), ("bias only", dict(bias=bias)), ("plain", dict()),
                          ("residual", dict(bias=bias, R=C, res=1)),
                          ("gelu nostore", dict(bias=bias, act=1, dbg=2)), ("no epilogue", dict(dbg=4)),
                          ("epilogue only gelu", dict(bias=bias, act=1, dbg=1)), ("epilogue only plain", dict(dbg=1)),
                          ("nothing", dict(dbg=5))]:
            for _ in range(2):
                E.gemm(A, B, C, M, Nn, K, algo=algo, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                E.gemm(A, B, C, M, Nn, K, algo=algo, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("%-7s algo=%d %-20s %8.3f ms  %7.1f TF" % (name, algo, label, ms, 2.0 * M * Nn * K / ms / 1e9), flush=True)
