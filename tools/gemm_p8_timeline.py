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
per = t[:, :3] / t[:, 3:4]
print("%s gelu=%d dbg+%d tiles %d, per-tile shader clocks (mean over workgroups of per-workgroup means)" % (which, gelu, extra, ntiles))
for k, name in enumerate(("first-slab wait", "main loop", "epilogue")):
    v = per[:, k]
    print("  %-15s mean %8.1f  min %8.1f  max %8.1f" % (name, v.mean(), v.min(), v.max()))
names = ["1a math+ds_write", "1a barrier", "2a lds->global", "2a barrier", "1b math+ds_write", "1b barrier", "2b lds->global", "2b barrier"]
for k in range(8):
    v = t[:, 4 + k] / t[:, 3]
    print("    epilogue %-18s mean %8.1f" % (names[k], v.mean()))
