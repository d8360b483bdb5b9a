#!/usr/bin/env python3
"""GPU tuning aid: time of one persistent round of each tile height (run once per MLPK_P8_FORCE_NI = 1..4) and the
scheduling variants of the persistent tile on the channel-MLP shapes.  usage: python tools/gemm_p8_calib.py [reps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
dt = torch.bfloat16
ni = int(os.environ.get("MLPK_P8_FORCE_NI", "0"))


def run(name, M, Nn, K, gelu, dbg, ln=False):
    A = (torch.rand((M, K), device=dev) * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    kw = dict(R=C, res=N.RES_ADD) if not gelu else {}
    if ln:
        kw["ln"] = (torch.rand(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5, B.float().sum(dim=1).contiguous())
    for _ in range(3):
        E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=14, dbg=dbg, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=14, dbg=dbg, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ms = sorted(ts)[1]
    print("%-26s ni=%d M=%6d N=%4d K=%4d dbg=%3d  %8.4f ms  %7.1f TFLOP/s" % (name, ni, M, Nn, K, dbg, ms, 2.0 * M * Nn * K / ms / 1e9), flush=True)


if ni:
    # exactly one round (255 or 252 tiles on 256 CUs) and three rounds of tiles of this height
    h = 64 * ni
    run("fc2 1 round", 85 * h, 768, 3072, False, 0)
    run("fc2 3 rounds", 255 * h, 768, 3072, False, 0)
    run("fc1 1 round", 21 * h, 3072, 768, True, 0, ln=True)
    run("fc1 3 rounds", 64 * h, 3072, 768, True, 0, ln=True)
else:
    for dbg in (0, 16, 64, 128, 16 | 128, 16 | 64 | 128):
        run("channel_fc1 (gelu+ln)", 50176, 3072, 768, True, dbg, ln=True)
        run("channel_fc2 (res)", 50176, 768, 3072, False, dbg)
    for dbg in (0, 16 | 64 | 128):
        run("mixer_l_fc1", 50176, 4096, 1024, True, dbg, ln=True)
        run("mixer_l_fc2", 50176, 1024, 4096, False, dbg)
        run("mixer_s_fc1", 50176, 2048, 512, True, dbg, ln=True)
        run("mixer_s_fc2", 50176, 512, 2048, False, dbg)
        run("convmixer_pw", 262144, 1536, 1536, True, dbg)
        run("vip_fc1", 262144, 1152, 384, True, dbg, ln=True)
