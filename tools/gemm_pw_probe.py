#!/usr/bin/env python3
"""Tuning aid: ConvMixer-1536/20's pointwise GEMM shape (262144 x 1536 x 1536, bf16) on the persistent tile with its real epilogue (GELU + BatchNorm
scale / shift, algo 14) against the generated q4 tile with GELU only (algo 15: a timing proxy for a q4 class that would carry the scale / shift),
alternating, HIP-event timed, the output evicted from the caches by a 600 MB fill between calls.  usage: gemm_pw_probe.py [reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
M, Nn, K = 262144, 1536, 1536
dt = torch.bfloat16
A = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
B = ((torch.rand((Nn, K), device="cuda") * 2 - 1) / K ** 0.5).to(dt)
bias, cs, ch = torch.rand(Nn, device="cuda"), torch.rand(Nn, device="cuda") + 0.5, torch.rand(Nn, device="cuda")
C = torch.zeros((M, Nn), dtype=dt, device="cuda")
big = torch.empty(600 << 20, dtype=torch.uint8, device="cuda")
cases = {"p8 gelu+bn (algo 14)": dict(algo=14, bias=bias, act=1, cscale=cs, cshift=ch), "p8 gelu (algo 14)": dict(algo=14, bias=bias, act=1),
         "q4 gelu (algo 15)": dict(algo=15, bias=bias, act=1), "auto gelu+bn": dict(bias=bias, act=1, cscale=cs, cshift=ch)}
tot = {k: 0.0 for k in cases}
for r in range(reps + 1):
    for k, kw in cases.items():
        big.fill_(r & 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        E.gemm(A, B, C, M, Nn, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        if r:
            tot[k] += e0.elapsed_time(e1)
for k, v in tot.items():
    ms = v / reps
    print("%-24s %8.3f ms  %7.1f TFLOP/s" % (k, ms, 2.0 * M * Nn * K / ms / 1e9))
