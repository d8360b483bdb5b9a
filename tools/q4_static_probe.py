#!/usr/bin/env python3
"""GPU check + A/B of the round-4 "static" q4 kernels (built for one K: q4gen.Q4.static) against the general q4 kernels of round 3.
   check: bit-equality with the s3 tile (same K order, same epilogue formulas) on K = 384 / 768 / 1152 shapes, all classes;
   time : wall time, in-kernel cycles and the implied clock, static vs general (MLPK_Q4_NKF forces the general kernel).
usage: python tools/q4_static_probe.py [check|time|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import q4_probe as qp  # noqa: E402

E, N, dev = qp.E, qp.N, qp.dev


def check():
    bad = 0
    for dt in (torch.bfloat16, torch.float16):
        for (M, Nn, K) in ((256, 128, 384), (2048, 384, 384), (4096, 768, 768), (12544, 3072, 768), (2048, 384, 1152), (50176, 384, 384)):
            ops = qp.operands(M, Nn, K, dt)
            for name, (gelu, ln, res) in (("p", (0, 0, 0)), ("l", (0, 1, 0)), ("g", (1, 0, 0)), ("gl", (1, 1, 0)), ("r", (0, 0, 1))):
                os.environ.pop("MLPK_Q4_NKF", None)
                c15 = qp.run(*ops, M, Nn, K, gelu, ln, res, 15)
                c11 = qp.run(*ops, M, Nn, K, gelu, ln, res, 11)
                torch.cuda.synchronize()
                ndiff = int((c15.view(torch.int16) != c11.view(torch.int16)).sum())
                nan = int(torch.isnan(c15.float()).sum())
                if ndiff or nan:
                    bad += 1
                print("%s M=%d N=%d K=%d %-3s static q4 bit-equal to s3: %s (%d differ, %d NaN)" % (str(dt)[6:], M, Nn, K, name, ndiff == 0, ndiff, nan), flush=True)
    print("CHECK", "FAILED %d" % bad if bad else "OK")
    return bad


def times():
    for title, (M, Nn, K), (gelu, ln, res), nkf in (("channel fc1", (50176, 3072, 768), (1, 1, 0), 12), ("vip fc1", (50176, 1152, 384), (1, 1, 0), 6),
                                                   ("vip K=N=384 res", (50176, 384, 384), (0, 0, 1), 6), ("vip fc2", (50176, 384, 1152), (0, 0, 1), 12),
                                                   ("resmlp ff1", (50176, 1536, 384), (1, 1, 0), 6)):
        A, B, bias, R, ln3 = qp.operands(M, Nn, K, torch.bfloat16)
        C = torch.zeros((M, Nn), dtype=torch.bfloat16, device=dev)
        kw = dict(R=R, res=N.RES_ADD) if res else {}
        if ln:
            kw["ln"] = ln3
        rows = []
        for name, force in (("static s%d" % (K // 64), 0), ("general f%d" % nkf, nkf)):
            buf = torch.zeros(256 * 2, dtype=torch.int32, device=dev)

            def f(force=force, buf=buf):
                if force:
                    os.environ["MLPK_Q4_NKF"] = str(force)
                else:
                    os.environ.pop("MLPK_Q4_NKF", None)
                E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=15, dbg=32, prof=buf, **kw)
            rows.append((name, f, buf))
        res_t = qp.timeit([(n, f) for n, f, _ in rows], rounds=7, reps=10)
        print("== %s M=%d N=%d K=%d" % (title, M, Nn, K))
        for name, f, buf in rows:
            h = buf.cpu().view(256, 2).double()
            cyc = h[:, 0].max().item()
            med = res_t[name][0]
            print("   %-14s %8.4f ms  %7.1f TFLOP/s   max WG cycles %8d  -> %.2f GHz   cycles/MFMA %.1f" %
                  (name, med, 2.0 * M * Nn * K / med / 1e9, cyc, cyc / (med * 1e6), (h[:, 0] / h[:, 1].clamp(min=1)).mean().item() / (8 * K / 16)), flush=True)
    os.environ.pop("MLPK_Q4_NKF", None)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = check() if what in ("check", "all") else 0
    if what in ("time", "all"):
        times()
    sys.exit(1 if rc else 0)
