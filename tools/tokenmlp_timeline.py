#!/usr/bin/env python3
"""Tuning aid: time the fused token-MLP kernel at Mixer-B/16 bs=256 (and the unfused pair for comparison)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
B_, C, S, T = 256, 768, 196, 784
dt = torch.bfloat16
sp = 224
xt = torch.zeros((B_ * C, sp), dtype=dt, device="cuda"); xt[:, :S] = torch.randn((B_ * C, S), device="cuda").to(dt)
x = torch.randn((B_ * S, C), device="cuda").to(dt)
w1p, b1p, w2p, b2p, nch, lay = E.pack_token_mlp(torch.randn(T, S) / 14, torch.randn(T), torch.randn(S, T) / 28, torch.randn(S), dt, "cuda", sp)
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C, layout=lay)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("fused token mlp: %.4f ms  (%.1f TFLOP/s algorithmic)" % (ms, 2 * 2.0 * B_ * C * S * T / ms / 1e9))
import ctypes
import numpy as np
fn = ctypes.CDLL(N.LIB_PATH).mlpk_token_mlp_debug
fn.argtypes = [ctypes.c_void_p]
grid = 256
dbg = torch.zeros((grid, 4), dtype=torch.int64, device="cuda")
fn(dbg.data_ptr())
E.token_mlp(xt, sp, B_ * C, S, w1p, b1p, w2p, b2p, nch, x, C, C, layout=lay)
torch.cuda.synchronize()
fn(None)
t = dbg.cpu().numpy().astype(np.float64)
per = t[:, :2] / t[:, 2:3]
print("layout %d; per tile (shader clocks, matrix wave 0): main loop mean %.0f (%.0f per iteration), epilogue mean %.0f" % (lay, per[:, 0].mean(), per[:, 0].mean() / (nch + (0 if lay else 2)), per[:, 1].mean()))
