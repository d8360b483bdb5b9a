"""Tuning aid: mlpk_token_gemm_ln alone at gMLP-S / ResMLP shapes (256 images), for rocprofv3 runs: python tools/tg_probe.py [sgu|aff] [reps]"""
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
mode = sys.argv[1] if len(sys.argv) > 1 else "sgu"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, dt = "cuda:0", torch.bfloat16
B, S = 256, 196
C = 768 if mode == "sgu" else 384
rows = B * S
torch.manual_seed(0)
w = torch.randn(S, S) / math.sqrt(S)
wp, bp, ng = E.pack_token_gemm(w, torch.randn(S), dt, dev)
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
if mode == "sgu":
    wide = torch.randn(rows, 2 * C, device=dev, dtype=dt)
    v = wide[:, C:]
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    E.row_stats(v, rows, C, 2 * C, mean, rstd)
    out = torch.empty(rows, C, device=dev, dtype=dt)
    big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
    for _ in range(reps):
        big.fill_(1)                                        # push the previous call's lines out of the caches
        E.token_gemm_ln(v, 2 * C, B * C, S, mean, rstd, gamma, beta, wp, bp, ng, out, C, C, R=wide, ldr=2 * C, res=N.RES_MUL)
else:
    x = torch.randn(rows, C, device=dev, dtype=dt)
    g1 = torch.rand(C, device=dev)
    big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
    for _ in range(reps):
        big.fill_(1)
        E.token_gemm_ln(x, C, B * C, S, None, None, gamma, beta, wp, bp, ng, x, C, C, R=x, ldr=C, res=N.RES_ADD_AFFINE, rscale=g1, rperiod=C)
torch.cuda.synchronize()
print("done", mode)
