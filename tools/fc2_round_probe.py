#!/usr/bin/env python3
"""Round-6 probe for the review's question "where do fc2's 2.9 round-times for 2.3 rounds of work go?".
The persistent tile (algo 14) on the channel-MLP fc2 shape (rows x 768 x 3072, bias + residual + by-product statistics,
the epilogue the model runs), timed in 1.5-second loops with the shader clock / power sampled beside it:
  * the real 50176-row plan, one round of 255 tiles, two rounds, three rounds (256-row tiles only);
  * each of them with the A operand WARM (one buffer, re-read every launch: 133 MB of a single round sits in the 256 MB
    Infinity Cache) and COLD (launches rotate over enough distinct A buffers to exceed it) -- the round-2 numbers that
    contradicted each other (profiles/r02_gemm_power_probe.txt vs r02_gemm_tile_height_calib_v1.txt) differed in exactly that.
With the -DMLPK_P8_PROF build (MLPK_LIB_PATH=.../libmlpk_p8prof.so) `timeline` adds the per-workgroup cycle ledger:
[first-slab wait | K loop | epilogue + hand-over] per body of the pair launch, and the start / end skew between CUs on the
100 MHz wall clock.
usage: python tools/fc2_round_probe.py [loops|timeline|all]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = "cuda:0"
dt = torch.bfloat16
Nn, K = 768, 3072


class WS:
    def __init__(self):
        self.t = {}

    def get(self, name, shape, dtype=None, fill=0.0):
        if name not in self.t:
            self.t[name] = torch.zeros(shape, dtype=dtype, device=dev)
        return self.t[name]


def case(M, dbg, nbuf, stats=True, prof=None, Nn=Nn, K=K, gelu=False, algo=14, ln=False):
    As = [(torch.rand((M, K), device=dev) * 2 - 1).to(dt) for _ in range(nbuf)]
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    R = (torch.rand((M, Nn), device=dev) * 2 - 1).to(dt)
    ws = WS()
    kw = {}
    if not gelu:
        kw.update(R=R, res=N.RES_ADD)
        if stats:
            kw["part"] = (ws, "p")
    if ln:
        kw["ln"] = (torch.rand(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5, B.float().sum(dim=1).contiguous())
    state = {"i": 0}

    def f():
        A = As[state["i"] % nbuf]
        state["i"] += 1
        E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=algo, dbg=dbg, prof=prof, **kw)
    return f


def loops():
    import gemm_power_probe as gp     # (prints its own table first: the sensors, fc1 / fc2 / 8192^3 as in round 2)
    print("---- fc2 rounds, warm vs cold A (rows x 768 x 3072, bias + residual + statistics) ----", flush=True)
    for name, M, dbg in (("real plan 50176 rows (588 tiles)", 50176, 0), ("256-row tiles only, 50176 rows", 50176, 16),
                         ("1 round: 255 tiles", 21760, 16), ("2 rounds: 510 tiles", 43520, 16), ("3 rounds: 765 tiles", 65280, 16),
                         ("1 round of 192-row tiles", 85 * 192, 0)):
        os.environ.pop("MLPK_P8_FORCE_NI", None)
        for nbuf, label in ((1, "warm"), (max(2, int(np.ceil(700e6 / (M * K * 2)))), "cold")):
            f = case(M, dbg, nbuf)
            gp.loop("%-34s %s x%d" % (name, label, nbuf), f, 2.0 * M * Nn * K)
            del f
            torch.cuda.empty_cache()
    print("---- fc1 on the generated tile (algo 15) and on the persistent tile (14), warm vs cold ----", flush=True)
    for algo in (15, 14):
        for nbuf, label in ((1, "warm"), (4, "cold")):
            f = case(50176, 0, nbuf, Nn=3072, K=768, gelu=True, algo=algo, ln=True)
            gp.loop("fc1 algo %d %s x%d" % (algo, label, nbuf), f, 2.0 * 50176 * 3072 * 768)
            del f
            torch.cuda.empty_cache()


def timeline():
    for name, M, dbg, nbuf in (("real plan", 50176, 0, 1), ("256-row only", 50176, 16, 1), ("1 round warm", 21760, 16, 1), ("1 round cold", 21760, 16, 6),
                               ("3 rounds", 65280, 16, 1)):
        prof = torch.zeros((256, 64), dtype=torch.int64, device=dev)
        f = case(M, dbg | 8, nbuf, prof=prof)
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prof.zero_()
        torch.cuda.synchronize()
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        t = prof.cpu().numpy().astype(np.float64)
        print("== %s: M=%d, one launch %.1f us (with stamps)" % (name, M, us))
        starts, ends = [], []
        for b in (0, 16):
            o = t[:, b:b + 16]
            live = o[:, 3] > 0
            if not live.any():
                continue
            o = o[live]
            tot = o[:, 0] + o[:, 1] + o[:, 2]
            for nt in sorted(set(o[:, 3].astype(int))):
                m = o[:, 3] == nt
                print("   body %d: %3d workgroups with %d tiles: per tile  wait %7.0f  loop %7.0f  epi+hand-over %7.0f  = %7.0f cycles;  body total %7.0f cycles = %.1f us wall, clock %.2f GHz"
                      % (b // 16, m.sum(), nt, o[m, 0].mean() / nt, o[m, 1].mean() / nt, o[m, 2].mean() / nt, tot[m].mean() / nt, o[m, 14].mean(),
                         (o[m, 13] - o[m, 12]).mean() / 100.0, o[m, 14].mean() / ((o[m, 13] - o[m, 12]).mean() * 10.0)))
            starts.append(o[:, 12])
            ends.append(o[:, 13])
        s0 = np.concatenate(starts).min()
        first = starts[0] - s0
        last = np.concatenate(ends) - s0
        lastb = ends[-1] - s0
        print("   wall (us from the first workgroup's start): starts  p50 %.2f  max %.2f | ends (last body)  min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f"
              % (np.median(first) / 100, first.max() / 100, lastb.min() / 100, np.percentile(lastb, 10) / 100, np.median(lastb) / 100, np.percentile(lastb, 90) / 100, last.max() / 100))
        if len(ends) == 2:
            e0b = ends[0] - s0
            print("   first body ends: min %.1f p50 %.1f max %.1f us" % (e0b.min() / 100, np.median(e0b) / 100, e0b.max() / 100))
        del f
        torch.cuda.empty_cache()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("loops", "all"):
        loops()
    if what in ("timeline", "all"):
        timeline()
