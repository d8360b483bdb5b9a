#!/usr/bin/env python3
"""GPU tuning aid: interleaved A/B of GEMM variants in one process (rounds x variants, median and min per variant;
run-to-run noise of a single timing is +-3 %, cdna_hip_programming.md 5.4 rule 24).
usage: python tools/gemm_ab.py [rounds]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda:0"
dt = torch.bfloat16


def case(M, Nn, K, gelu, ln, res, algo, dbg):
    A = (torch.rand((M, K), device=dev) * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    kw = dict(R=C, res=N.RES_ADD) if res else {}
    if ln:
        kw["ln"] = (torch.rand(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5, B.float().sum(dim=1).contiguous())
    return lambda: E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, dbg=dbg, **kw)


def ab(title, M, Nn, K, variants, reps=10):
    fns = [(name, case(M, Nn, K, *v)) for name, v in variants]
    for _, f in fns:
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    times = {name: [] for name, _ in fns}
    for r in range(rounds):
        for name, f in (fns if r % 2 == 0 else fns[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / reps)
    print("== %s  M=%d N=%d K=%d" % (title, M, Nn, K))
    for name, _ in fns:
        t = sorted(times[name])
        med = t[len(t) // 2]
        print("   %-44s median %8.4f ms  min %8.4f  %7.1f TFLOP/s (median)" % (name, med, t[0], 2.0 * M * Nn * K / med / 1e9), flush=True)


# (gelu, ln, res, algo, dbg)
fc1 = [("p8 default", (1, 1, 0, 14, 0)), ("p8 256-row tiles only (16)", (1, 1, 0, 14, 16)), ("p8 staged epilogue (64)", (1, 1, 0, 14, 64)),
       ("p8 one column group (128)", (1, 1, 0, 14, 128)), ("p8 round-1 equivalent (208)", (1, 1, 0, 14, 208)),
       ("p8 default, no LN fold", (1, 0, 0, 14, 0)), ("p8 default, no GELU no LN", (0, 0, 0, 14, 0)), ("p8 no stores (2)", (1, 1, 0, 14, 2)),
       ("s3 128x256 (algo 13)", (1, 1, 0, 13, 0))]
ab("channel fc1", 50176, 3072, 768, fc1)
fc2 = [("p8 default", (0, 0, 1, 14, 0)), ("p8 256-row tiles only (16)", (0, 0, 1, 14, 16)), ("p8 staged epilogue (64)", (0, 0, 1, 14, 64)),
       ("p8 round-1 equivalent (208)", (0, 0, 1, 14, 208)), ("p8 default, no residual", (0, 0, 0, 14, 0)), ("p8 no stores (2)", (0, 0, 1, 14, 2)),
       ("s3 128x256 (algo 13)", (0, 0, 1, 13, 0))]
ab("channel fc2", 50176, 768, 3072, fc2)
ab("mixer-L fc1", 50176, 4096, 1024, [("p8 default", (1, 1, 0, 14, 0)), ("p8 round-1 equivalent (208)", (1, 1, 0, 14, 208)), ("p8 one column group (128)", (1, 1, 0, 14, 128))])
ab("mixer-L fc2", 50176, 1024, 4096, [("p8 default", (0, 0, 1, 14, 0)), ("p8 round-1 equivalent (208)", (0, 0, 1, 14, 208)), ("p8 one column group (128)", (0, 0, 1, 14, 128))])
ab("convmixer pw", 262144, 1536, 1536, [("p8 default", (1, 0, 0, 14, 0)), ("p8 round-1 equivalent (208)", (1, 0, 0, 14, 208)), ("p8 one column group (128)", (1, 0, 0, 14, 128))], reps=4)
