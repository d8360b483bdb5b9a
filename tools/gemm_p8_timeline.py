#!/usr/bin/env python3
"""Tuning aid: per-workgroup [prologue | main loop | epilogue] s_memtime spans of the p8 GEMM (dbg bit 8).
usage: python tools/gemm_p8_timeline.py [fc1|fc2] [gelu 0/1]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
which = sys.argv[1] if len(sys.argv) > 1 else "fc1"
gelu = int(sys.argv[2]) if len(sys.argv) > 2 else (1 if which == "fc1" else 0)
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M, Nn, K = (50176, 3072, 768) if which == "fc1" else (50176, 768, 3072)
dt = torch.bfloat16
A = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
B = ((torch.rand((Nn, K), device="cuda") * 2 - 1) / K ** 0.5).to(dt)
bias = torch.rand(Nn, device="cuda")
C = torch.zeros((M, Nn), dtype=dt, device="cuda")
ntiles = (M // 256) * (Nn // 256)
grid = min(ntiles, 256)
dbg = torch.zeros((grid, 64), dtype=torch.int64, device="cuda")
for _ in range(3):
    E.gemm(A, B, C, M, Nn, K, bias=bias, act=gelu, algo=14, R=dbg, res=0, dbg=8 | extra)
torch.cuda.synchronize()
t = dbg.cpu().numpy().astype(np.float64)
tot = t[:, :3].sum(axis=1)
print("%s gelu=%d dbg+%d tiles %d: per-workgroup totals (shader clocks): wait %.0f  main %.0f  epilogue(+hand-over) %.0f  sum mean %.0f max %.0f" % (
    which, gelu, extra, ntiles, t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), tot.mean(), tot.max()))
if which == "fc2" and not (extra & 64):
    w = np.arange(grid)
    for name, m in (("part 0", ((w >> 3) < 30) & ((w >> 3) % 3 == 0)), ("part 1-2", ((w >> 3) < 30) & ((w >> 3) % 3 != 0)), ("no tail slice", (w >> 3) >= 30)):
        print("  %-14s n=%3d  wait %.0f  main %.0f  epi+hand-over %.0f  total %.0f   [poll %.0f, partial reads %.0f]" % (name, m.sum(), t[m, 0].mean(), t[m, 1].mean(), t[m, 2].mean(), tot[m].mean(), t[m, 4].mean(), t[m, 5].mean()))
