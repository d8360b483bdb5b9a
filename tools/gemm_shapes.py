#!/usr/bin/env python3
"""GPU tuning aid: the distinct GEMM calls of a model's forward at bs=256 bf16, each timed with the generated q4 tile switched off / on
(MLPK_GEMM_Q4 is read per process, so the two settings run as explicit algo choices: auto-without-q4 is emulated by algo from a first
pass with the library's own choice logged).  usage: python tools/gemm_shapes.py model [model ...]"""
import importlib
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--time":
    # child: time the shapes given on stdin under the current MLPK_GEMM_Q4 setting
    pkg = importlib.import_module("jittor-mlp_amd")
    E, N = pkg.engine, pkg._native
    dev = "cuda:0"
    import ast
    for line in sys.stdin:
        dt, M, Nn, K, act, res, ln, part, aff, rsc, om, hb = ast.literal_eval(line)
        if om != 0 or aff or rsc or dt != "torch.bfloat16":
            continue
        A = (torch.rand((M, K), device=dev) * 2 - 1).bfloat16()
        B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).bfloat16()
        C = torch.zeros((M, Nn), device=dev).bfloat16()
        kw = {}
        if res:
            kw.update(R=(torch.rand((M, Nn), device=dev)).bfloat16(), res=res)
        if ln:
            kw["ln"] = (torch.rand(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5, B.float().sum(1).contiguous())
        if hb:
            kw["bias"] = torch.rand(Nn, device=dev)

        class WS:
            def get(self, name, shape, dtype):
                return torch.zeros(shape, dtype=dtype, device=dev)
        if part:
            kw["part"] = (WS(), "p")
        f = lambda: E.gemm(A, B, C, M, Nn, K, act=act, **kw)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        print("%s\t%.5f" % (line.strip(), sorted(ts)[2]), flush=True)
    sys.exit(0)

bench = importlib.import_module("bench")
pkg = importlib.import_module("jittor-mlp_amd")
E = pkg.engine
for name in sys.argv[1:]:
    ctor, kw, _ = bench.MODELS[name]
    model = getattr(pkg.models_pytorch, ctor)(**kw).eval().to("cuda:0")
    x = torch.rand((256, 3, 224, 224), device="cuda:0").bfloat16()
    with torch.no_grad():
        model(x)
        E.GEMM_LOG = set()
        model(x)
    shapes = sorted(E.GEMM_LOG)
    E.GEMM_LOG = None
    del model
    inp = "\n".join(repr(s) for s in shapes) + "\n"
    res = {}
    for q4 in ("0", "2"):
        env = dict(os.environ, MLPK_GEMM_Q4=q4)
        out = subprocess.run([sys.executable, __file__, "--time", "-"], input=inp, capture_output=True, text=True, env=env).stdout
        for l in out.strip().split("\n"):
            if "\t" in l:
                k, t = l.split("\t")
                res.setdefault(k, {})[q4] = float(t)
    print("== %s: M N K act res ln stats | ms without q4 | ms with q4 wherever it applies | ratio" % name)
    for k, v in res.items():
        dt, M, Nn, K, act, r, ln, part, aff, rsc, om, hb = eval(k)
        if "0" in v and "2" in v:
            print("   %7d %5d %5d  act %d res %d ln %d st %d   %8.4f  %8.4f   %5.2f %s" % (M, Nn, K, act, r, ln, part, v["0"], v["2"], v["2"] / v["0"],
                  "" if abs(v["2"] / v["0"] - 1) < 0.02 else ("q4 ahead" if v["2"] < v["0"] else "q4 BEHIND")), flush=True)
