#!/usr/bin/env python3
"""Tuning aid: per-workgroup s_memtime timeline of the GEMM main loop (dbg bit 8)."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
algo = int(sys.argv[1]) if len(sys.argv) > 1 else 6
M, Nn, K = 50176, 3072, 768
dt = torch.bfloat16
A = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
B = ((torch.rand((Nn, K), device="cuda") * 2 - 1) / K ** 0.5).to(dt)
C = torch.zeros((M, Nn), dtype=dt, device="cuda")
bm, bn = {6: (256, 256), 11: (256, 128), 12: (128, 128), 13: (128, 256), 9: (128, 128)}[algo]
ntiles = (M // bm) * (Nn // bn)
dbg = torch.zeros((ntiles, 64), dtype=torch.int64, device="cuda")
for _ in range(2):
    E.gemm(A, B, C, M, Nn, K, algo=algo, R=dbg, res=0, dbg=8 | 4)
torch.cuda.synchronize()
t = dbg.cpu().numpy().astype(np.int64)
t0 = t[:, 0].min()
rel = t - t0
print("algo", algo, "tiles", ntiles, "kernel span (cycles @100MHz?)", (t.max() - t0))
# s_memtime ticks: constant-rate counter; print ratios only
nk = K // (64 if algo in (6, 9) else 32)
for wg in (0, 1, 255, 256, 700, 1500, ntiles - 1):
    r = rel[wg]
    print("wg %5d start %8d prologue %6d" % (wg, r[0], r[1] - r[0]), end=" | ")
    if algo in (6, 9):
        for kt in range(min(nk, 12)):
            a, b, c, d = r[2 + 4 * kt:6 + 4 * kt]
            print("[issue %5d wait %4d bar %4d]" % (b - a, c - b, d - c), end="")
    else:
        prev = r[1]
        for kt in range(min(nk, 24)):
            a, b = r[2 + 2 * kt], r[3 + 2 * kt]
            print("[comp %5d wait+bar %4d]" % (a - prev, b - a), end="")
            prev = b
    print(" end %d" % (r[60] - r[0]))
dur = rel[:, 60] - rel[:, 0]
print("per-WG main-loop duration: mean %.0f min %d max %d ticks" % (dur.mean(), dur.min(), dur.max()))
starts = np.sort(rel[:, 0])
print("start times percentiles", [int(np.percentile(starts, q)) for q in (0, 10, 25, 50, 75, 90, 100)])
