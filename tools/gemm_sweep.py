#!/usr/bin/env python3
"""GPU tuning aid: time every GEMM tile configuration on the Mixer-B/16 bs=256 shapes (HIP events,
random bf16 operands) and print TFLOP/s.  usage: python tools/gemm_sweep.py [dtype] [algos]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
algos = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(1, N.lib().mlpk_gemm_algo_count() + 1))
dbgs = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
dev = "cuda:0"
SHAPES = [  # name, M, N, K, token_t(t_rows, t_tokens) or None, gelu
    ("channel_fc1", 50176, 3072, 768, None, True),
    ("channel_fc2", 50176, 768, 3072, None, False),
    ("token_fc1", 196608, 784, 200, None, True),
    ("token_fc2", 196608, 196, 784, (768, 196), False),
    ("gmlp_proj1", 50176, 3072, 256, None, True),
    ("vip_branch", 262144, 384, 384, None, False),
    ("mixer_l_fc1", 50176, 4096, 1024, None, True),
    ("mixer_l_fc2", 50176, 1024, 4096, None, False),
    ("mixer_s_fc1", 50176, 2048, 512, None, True),
    ("mixer_s_fc2", 50176, 512, 2048, None, False),
    ("resmlp_fc1", 50176, 1536, 384, None, True),
    ("resmlp_fc2", 50176, 384, 1536, None, False),
    ("convmixer_pw", 262144, 1536, 1536, None, True),
    ("s2_mlp1", 262144, 576, 192, None, False),
    ("s2_fc1", 65536, 1152, 384, None, True),
    ("gmlp_proj2", 50176, 256, 1536, None, False),
    ("vip_fc1", 262144, 1152, 384, None, True),
    ("vip_fc2", 262144, 384, 1152, None, False),
    ("s2_mlp2", 262144, 192, 192, None, False),
    ("s2_fc2", 65536, 384, 1152, None, False),
    ("k128", 50176, 1024, 128, None, True),        # synthetic short-K shapes: calibration of the tile choice
    ("k192", 50176, 1536, 192, None, True),
    ("k128_big", 200704, 512, 128, None, False),
    ("k256_small", 12544, 1024, 256, None, True),
    ("k320", 50176, 1280, 320, None, True),
]
for name, M, Nn, K, tt, gelu in SHAPES:
    if only and name not in only:
        continue
    A = (torch.rand((M, K), device=dev) * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)
    if tt:
        C = torch.zeros((M // tt[0] * tt[1], tt[0]), dtype=dt, device=dev)
        kw = dict(ldc=tt[0], R=C, ldr=tt[0], res=N.RES_ADD, out_mode=N.OUT_TOKEN_T, t_rows=tt[0], t_tokens=tt[1])
    else:
        C = torch.zeros((M, Nn), dtype=dt, device=dev)
        kw = dict(R=C, res=N.RES_ADD) if not gelu else {}
    for algo, dbgv in [(a, d) for a in algos for d in dbgs]:
        kw["dbg"] = dbgv
        try:
            for _ in range(2):
                E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print("%-12s M=%6d N=%4d K=%4d algo=%d dbg=%d  %8.3f ms  %7.1f TFLOP/s" % (name, M, Nn, K, algo, dbgv, ms, 2.0 * M * Nn * K / ms / 1e9), flush=True)
        except Exception as ex:  # noqa
            print(name, algo, "ERR", str(ex)[:80])
    del A, B, C
