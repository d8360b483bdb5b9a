#!/usr/bin/env python3
"""GPU check + timing of the generated q4 GEMM (algo 15) against the persistent tile (14) and the s3 tile (11).
usage: python tools/q4_probe.py [check|time|all]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 and __name__ == "__main__" else "none"


def operands(M, Nn, K, dt, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device=dev, generator=g) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev, generator=g) - 0.5
    R = (torch.rand((M, Nn), device=dev, generator=g) * 2 - 1).to(dt)
    mean = torch.rand(M, device=dev, generator=g) * 0.1
    rstd = torch.rand(M, device=dev, generator=g) + 0.5
    csum = B.float().sum(dim=1).contiguous()
    return A, B, bias, R, (mean, rstd, csum)


def run(A, B, bias, R, ln3, M, Nn, K, gelu, ln, res, algo, dbg=0):
    C = torch.full((M, Nn), float("nan"), dtype=A.dtype, device=dev)
    kw = dict(R=R, res=N.RES_ADD) if res else {}
    if ln:
        kw["ln"] = ln3
    E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, dbg=dbg, **kw)
    return C


def reference(A, B, bias, R, ln3, gelu, ln, res):
    acc = A.double() @ B.double().t()
    if ln:
        v = (acc - ln3[0].double()[:, None] * ln3[2].double()[None, :]) * ln3[1].double()[:, None] + bias.double()[None, :]
    else:
        v = acc + bias.double()[None, :]
    if gelu:
        v = torch.nn.functional.gelu(v)
    if res:
        v = v.to(A.dtype).double() + R.double()
    return v


def check():
    bad = 0
    for dt in (torch.bfloat16, torch.float16):
        for (M, Nn, K) in ((256, 128, 192), (1024, 384, 384), (2048, 768, 768), (4096, 768, 3072), (50176, 384, 384)):
            ops = operands(M, Nn, K, dt)
            for name, (gelu, ln, res) in (("p", (0, 0, 0)), ("l", (0, 1, 0)), ("g", (1, 0, 0)), ("gl", (1, 1, 0)), ("r", (0, 0, 1))):
                c15 = run(*ops, M, Nn, K, gelu, ln, res, 15)
                c11 = run(*ops, M, Nn, K, gelu, ln, res, 11)
                torch.cuda.synchronize()
                eq = torch.equal(c15.view(torch.int16), c11.view(torch.int16))
                ndiff = int((c15.view(torch.int16) != c11.view(torch.int16)).sum())
                msg = ""
                if M * Nn <= 4096 * 3072:
                    ref = reference(*ops, gelu, ln, res)
                    tol = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * ref.abs().clamp(min=1.0)
                    e15 = ((c15.double() - ref).abs() / tol).max().item()
                    e11 = ((c11.double() - ref).abs() / tol).max().item()
                    msg = "err/tol q4 %.3f s3 %.3f" % (e15, e11)
                    if not (e15 <= 1.0):
                        bad += 1
                if not eq and ndiff > 0:
                    # where?
                    d = (c15.view(torch.int16) != c11.view(torch.int16)).nonzero()
                    msg += "  first diffs %s  maxabs %.3g" % (d[:4].tolist(), (c15.float() - c11.float()).abs().max().item())
                print("%s M=%d N=%d K=%d %-3s bit-equal to s3: %s (%d differ)  %s" % (str(dt)[6:], M, Nn, K, name, eq, ndiff, msg), flush=True)
                if torch.isnan(c15.float()).any():
                    bad += 1
                    print("   NaN in q4 output: %d" % int(torch.isnan(c15.float()).sum()))
    print("CHECK", "FAILED %d" % bad if bad else "OK")
    return bad


def timeit(fns, rounds=7, reps=10):
    for _, f in fns:
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    times = {name: [] for name, _ in fns}
    for r in range(rounds):
        for name, f in (fns if r % 2 == 0 else fns[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / reps)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in times.items()}


def bench(title, M, Nn, K, variants, dt=torch.bfloat16, reps=10):
    ops = operands(M, Nn, K, dt)
    A, B, bias, R, ln3 = ops
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    fns = []
    for name, (gelu, ln, res, algo, dbg, nkf) in variants:
        kw = dict(R=R, res=N.RES_ADD) if res else {}
        if ln:
            kw["ln"] = ln3

        def f(gelu=gelu, algo=algo, dbg=dbg, kw=kw, nkf=nkf):
            if nkf:
                os.environ["MLPK_Q4_NKF"] = str(nkf)
            else:
                os.environ.pop("MLPK_Q4_NKF", None)
            E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, dbg=dbg, **kw)
        fns.append((name, f))
    res = timeit(fns, reps=reps)
    print("== %s  M=%d N=%d K=%d" % (title, M, Nn, K))
    for name, _ in fns:
        med, mn = res[name]
        print("   %-40s median %8.4f ms  min %8.4f  %7.1f TFLOP/s" % (name, med, mn, 2.0 * M * Nn * K / med / 1e9), flush=True)


def ablate():
    """wall time AND cycles of the ablation variants (the chip is power-capped: a stall raises the clock, so cycles alone mislead)"""
    for title, (M, Nn, K), cls, nkf in (("channel fc1", (50176, 3072, 768), (1, 1, 0), 12), ("channel fc2", (50176, 768, 3072), (0, 0, 1), 12)):
        A, B, bias, R, ln3 = operands(M, Nn, K, torch.bfloat16)
        C = torch.zeros((M, Nn), dtype=torch.bfloat16, device=dev)
        gelu, ln, res = cls
        kw = dict(R=R, res=N.RES_ADD) if res else {}
        if ln:
            kw["ln"] = ln3
        os.environ["MLPK_Q4_NKF"] = str(nkf)
        print("== %s M=%d N=%d K=%d" % (title, M, Nn, K))
        rows = []
        for name, dbg in (("full", 0), ("no stores", 2), ("no fillers", 4), ("no dma", 1), ("no dma, no stores", 3), ("stores over one tile", 64), ("burst nt stores", 1 << 16), ("spread plain stores", 2 << 16), ("burst plain stores", 3 << 16), ("no dma no fillers", 5), ("+ no reads", 13), ("mfma + barrier", 29)):
            buf = torch.zeros(256 * 2, dtype=torch.int32, device=dev)

            def f(dbg=dbg, buf=buf):
                E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=15, dbg=dbg | 32, prof=buf, **kw)
            rows.append((name, f, buf))
        res_t = timeit([(n, f) for n, f, _ in rows], rounds=5, reps=10)
        for name, f, buf in rows:
            h = buf.cpu().view(256, 2).double()
            cyc = h[:, 0].max().item()
            med = res_t[name][0]
            print("   %-22s %8.4f ms  %7.1f TFLOP/s   max WG cycles %8d  -> %.2f GHz   cycles/MFMA %.1f" %
                  (name, med, 2.0 * M * Nn * K / med / 1e9, cyc, cyc / (med * 1e6), (h[:, 0] / h[:, 1].clamp(min=1)).mean().item() / (8 * K / 16)), flush=True)
    os.environ.pop("MLPK_Q4_NKF", None)


def times():
    # (gelu, ln, res, algo, dbg, nkf)
    bench("channel fc1", 50176, 3072, 768, [("p8", (1, 1, 0, 14, 0, 0)), ("q4 f12", (1, 1, 0, 15, 0, 12)), ("q4 f6", (1, 1, 0, 15, 0, 6)), ("q4 f4", (1, 1, 0, 15, 0, 4)),
                                            ("q4 f12 no fillers", (1, 1, 0, 15, 4, 12)), ("q4 f12 no dma", (1, 1, 0, 15, 1, 12)), ("q4 f12 neither", (1, 1, 0, 15, 5, 12)),
                                            ("q4 no gelu f6", (0, 1, 0, 15, 0, 6)), ("q4 plain f4", (0, 0, 0, 15, 0, 4)), ("q4 one group f12", (1, 1, 0, 15, 128, 12)),
                                            ("s3 256x128", (1, 1, 0, 11, 0, 0))])
    bench("channel fc2", 50176, 768, 3072, [("p8", (0, 0, 1, 14, 0, 0)), ("q4 f12", (0, 0, 1, 15, 0, 12)), ("q4 f6", (0, 0, 1, 15, 0, 6)), ("q4 f4", (0, 0, 1, 15, 0, 4)),
                                            ("q4 f12 no fillers", (0, 0, 1, 15, 4, 12)), ("q4 f12 no dma", (0, 0, 1, 15, 1, 12)), ("q4 f12 neither", (0, 0, 1, 15, 5, 12)),
                                            ("q4 plain f4", (0, 0, 0, 15, 0, 4)), ("s3 256x128", (0, 0, 1, 11, 0, 0))])
    bench("vip K=N=384", 50176, 384, 384, [("auto", (0, 0, 0, 0, 0, 0)), ("q4 plain f4", (0, 0, 0, 15, 0, 4)), ("q4 plain f3", (0, 0, 0, 15, 0, 3)), ("q4 ln f6", (0, 1, 0, 15, 0, 6)),
                                           ("q4 res f6", (0, 0, 1, 15, 0, 6)), ("q4 res f4", (0, 0, 1, 15, 0, 4)), ("s3 res", (0, 0, 1, 11, 0, 0))])
    bench("vip fc1", 50176, 1152, 384, [("auto", (1, 1, 0, 0, 0, 0)), ("q4 gl f6", (1, 1, 0, 15, 0, 6)), ("q4 gl f4", (1, 1, 0, 15, 0, 4))])
    bench("vip fc2", 50176, 384, 1152, [("auto", (0, 0, 1, 0, 0, 0)), ("q4 res f6", (0, 0, 1, 15, 0, 6)), ("q4 res f12", (0, 0, 1, 15, 0, 12))])
    bench("gmlp proj1", 50176, 3072, 256, [("auto", (1, 1, 0, 0, 0, 0)), ("q4 gl f4", (1, 1, 0, 15, 0, 4))])
    bench("mixer-L fc1", 50176, 4096, 1024, [("p8", (1, 1, 0, 14, 0, 0)), ("q4 f12", (1, 1, 0, 15, 0, 12))])
    bench("mixer-L fc2", 50176, 1024, 4096, [("p8", (0, 0, 1, 14, 0, 0)), ("q4 f12", (0, 0, 1, 15, 0, 12))])
    bench("square 8192", 8192, 8192, 8192, [("p8", (0, 0, 0, 14, 0, 0)), ("q4 plain f4", (0, 0, 0, 15, 0, 4))], reps=4)


def prof():
    """cycles per tile / per MFMA from the in-kernel counters (clock independent)"""
    for title, (M, Nn, K), variants in (
            ("channel fc1", (50176, 3072, 768), [("full f12", (1, 1, 0, 0, 12)), ("full f6", (1, 1, 0, 0, 6)), ("no stores", (1, 1, 0, 2, 12)), ("no fillers", (1, 1, 0, 4, 12)),
                                                ("no dma", (1, 1, 0, 1, 12)), ("no dma no fillers", (1, 1, 0, 5, 12)), ("+ no reads", (1, 1, 0, 13, 12)),
                                                ("no dma/fillers, min tail", (1, 1, 0, 21, 12)), ("mfma + barrier only", (1, 1, 0, 29, 12))]),
            ("channel fc2", (50176, 768, 3072), [("full f12", (0, 0, 1, 0, 12)), ("full f6", (0, 0, 1, 0, 6)), ("no stores", (0, 0, 1, 2, 12)), ("no fillers", (0, 0, 1, 4, 12)), ("no dma", (0, 0, 1, 1, 12)),
                                                ("no dma no fillers", (0, 0, 1, 5, 12)), ("+ no reads", (0, 0, 1, 13, 12)), ("no dma/fillers, min tail", (0, 0, 1, 21, 12)),
                                                ("mfma + barrier only", (0, 0, 1, 29, 12))])):
        A, B, bias, R, ln3 = operands(M, Nn, K, torch.bfloat16)
        C = torch.zeros((M, Nn), dtype=torch.bfloat16, device=dev)
        print("== %s M=%d N=%d K=%d: cycles per 256x128 tile (incl. the draining block), per MFMA (4 x 8 x K/16 per wave and tile)" % (title, M, Nn, K))
        for name, (gelu, ln, res, dbg, nkf) in variants:
            buf = torch.zeros(256 * 2, dtype=torch.int32, device=dev)
            os.environ["MLPK_Q4_NKF"] = str(nkf)
            kw = dict(R=R, res=N.RES_ADD) if res else {}
            if ln:
                kw["ln"] = ln3
            for _ in range(2):
                E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=15, dbg=dbg | 32, prof=buf, **kw)
            torch.cuda.synchronize()
            h = buf.cpu().view(256, 2).double()
            cyc, nt = h[:, 0], h[:, 1]
            per_tile = (cyc / nt.clamp(min=1)).mean().item()
            print("   %-28s tiles/WG %4.1f  cycles/tile %8.0f  cycles/MFMA %6.2f   (max WG cycles %d)" % (name, nt.mean().item(), per_tile, per_tile / (8 * K / 16), int(cyc.max().item())), flush=True)
    os.environ.pop("MLPK_Q4_NKF", None)


if __name__ == "__main__":
    if what == "prof":
        prof()
        sys.exit(0)
    if what == "ablate":
        ablate()
        sys.exit(0)
    rc = 0
    if what in ("check", "all"):
        rc = check()
    if what in ("time", "all"):
        times()
    if what in ("all",):
        prof()
    sys.exit(1 if rc else 0)
