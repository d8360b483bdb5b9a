#!/usr/bin/env python3
"""Run one GEMM configuration a few times (for rocprofv3 --pmc).  usage: gemm_one.py name algo [mode]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
name, algo = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "gelu"
M, Nn, K = {"fc1": (50176, 3072, 768), "fc2": (50176, 768, 3072)}[name]
dt = torch.bfloat16
A = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
B = ((torch.rand((Nn, K), device="cuda") * 2 - 1) / K ** 0.5).to(dt)
bias = torch.rand(Nn, device="cuda")
C = torch.zeros((M, Nn), dtype=dt, device="cuda")
kw = {"gelu": dict(bias=bias, act=1), "res": dict(bias=bias, R=C, res=1), "noepi": dict(dbg=4), "plain": dict()}[mode]
for _ in range(5):
    E.gemm(A, B, C, M, Nn, K, algo=algo, **kw)
torch.cuda.synchronize()
