"""Round 6 probe: does the gMLP spatial product (mlpk_token_gemm_ln, SGU form) run at a higher HBM rate when an image's operand is one contiguous
region?  The real layout reads 512-byte pieces (256 channels of one token) at a 6 KB stride (h is (rows, 3072)); with d_ffn = 256 and 6 x the
images the same bytes lie in 196 KB contiguous regions per image -- what a channel-blocked h ([panel][row][256]) would give the real model.
python tools/sgu_layout_probe.py"""
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev, dt = "cuda:0", torch.bfloat16
S = 196
torch.manual_seed(0)
w = torch.randn(S, S) / math.sqrt(S)
wp, bp, ng = E.pack_token_gemm(w, torch.randn(S), dt, dev)
big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
for F, B in ((1536, 256), (768, 512), (256, 1536)):
    rows = B * S
    gamma, beta = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.1
    wide = torch.randn(rows, 2 * F, device=dev, dtype=dt)
    v = wide[:, F:]
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    E.row_stats(v, rows, F, 2 * F, mean, rstd)
    out = torch.empty(rows, F, device=dev, dtype=dt)
    nbytes = rows * 3 * F * 2
    for cold in (False, True):
        ts = []
        for _ in range(8):
            if cold:
                big.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            E.token_gemm_ln(v, 2 * F, B * F, S, mean, rstd, gamma, beta, wp, bp, ng, out, F, F, R=wide, ldr=2 * F, res=N.RES_MUL)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts)[1:-1]
        t = sum(ts) / len(ts)
        print("d_ffn %5d x %5d images (%s): %7.1f us  %5.2f TB/s  (%d MB)" % (F, B, "cold" if cold else "warm", t, nbytes / t / 1e6, nbytes >> 20))
