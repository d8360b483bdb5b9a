"""Run a generated q4 kernel on the numpy emulator of isa.py against an fp64 restatement of the operation."""
import math
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa  # noqa: E402
import q4gen  # noqa: E402


def bf16_round(x):
    u = np.asarray(x, np.float32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def to16(x, dtype):
    return bf16_round(x) if dtype == "bf16" else np.asarray(x, np.float32).astype(np.float16).view(np.uint16)


def from16(u, dtype):
    return bf16_to_f32(u) if dtype == "bf16" else u.view(np.float16).astype(np.float32)


def gelu_ref(x):
    return 0.5 * x * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))


def plan(M, N, K, grid, cgroups=1, m_base=0):
    """the host-side launch arithmetic (mirrors launch_q4 in mlpk_gemm_q4.hip)"""
    tiles_n = N // 128
    X = 8 // cgroups
    cg = tiles_n // cgroups
    panels = M // 256
    U = panels * cg
    Q = (U + X - 1) // X
    magic = ((1 << 31) + cg - 1) // cg
    return dict(nk=K // 64, cg=cg, cg_magic=magic & 0xFFFFFFFF, U=U, Q=Q, log2X=int(math.log2(X)), m_base=m_base, grid=grid)


class Q4Emu(isa.Emu):
    def __init__(self, gen, mem, karg_addr, bid, **kw):
        asm = isa.Asm()
        asm.ins = list(gen.a.ins) + [isa.Ins("s_endpgm", (), {})]
        super().__init__(asm, mem, lds_bytes=163840, nwaves=4, **kw)
        self.karg_addr, self.bid = karg_addr, bid

    def init(self, waves):
        for w in waves:
            w.s[0] = self.karg_addr & 0xFFFFFFFF
            w.s[1] = self.karg_addr >> 32
            w.s[2] = self.bid
            w.v[0] = (np.arange(64) + 64 * w.wid).astype(np.uint32)
            # everything else starts as garbage on the hardware
            w.v[1:] = 0x7FC0BEEF
            w.a[:] = 0x7FC0BEEF
        self.lds[:] = 0xEE


def run_case(gen, M, N, K, grid, cgroups=1, seed=0, dma_mode="late", order=None, verbose=False):
    rng = np.random.default_rng(seed)
    dt = gen.dtype
    A = to16(rng.uniform(-1, 1, (M, K)), dt)
    B = to16(rng.uniform(-1, 1, (N, K)) * 0.1, dt)
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    R = to16(rng.uniform(-1, 1, (M, N)), dt)
    mean = rng.uniform(-0.5, 0.5, M).astype(np.float32)
    rstd = rng.uniform(0.5, 2.0, M).astype(np.float32)
    csum = rng.uniform(-1, 1, N).astype(np.float32)
    C = np.full((M, N), 0x7FC1, np.uint16)
    mem = isa.Mem()
    aA, aB, aC, aR = mem.add(A), mem.add(B), mem.add(C), mem.add(R)
    aBias, aMean, aRstd, aCsum = mem.add(bias), mem.add(mean), mem.add(rstd), mem.add(csum)
    pl = plan(M, N, K, grid, cgroups)
    ka = bytearray(144)
    for name, val in (("A", aA), ("B", aB), ("C", aC), ("R", aR), ("bias", aBias), ("ln_mean", aMean), ("ln_rstd", aRstd), ("ln_csum", aCsum)):
        struct.pack_into("<Q", ka, q4gen.KA[name], val)
    ints = dict(lda=K, ldb=K, ldc=N, ldr=N, **pl)
    for name, val in ints.items():
        struct.pack_into("<I", ka, q4gen.KA[name], val)
    part = np.full((N // 32, M, 2), np.nan, np.float32)
    aPart = mem.add(part)
    struct.pack_into("<Q", ka, q4gen.KA["row_part"], aPart)
    struct.pack_into("<I", ka, q4gen.KA["row_part_ld"], M)
    prof = np.zeros(grid * 2, np.uint32)
    aProf = mem.add(prof)
    struct.pack_into("<Q", ka, q4gen.KA["prof"], aProf)
    karg = mem.add(np.frombuffer(bytes(ka), np.uint8))
    Cbuf = mem.get(aC)
    nins = 0
    for bid in range(grid):
        e = Q4Emu(gen, mem, karg, bid, dma_mode=dma_mode, order=order)
        waves = e.run()
        nins += sum(w.nissued for w in waves)
    if verbose:
        print("prof:", mem.get(aProf).view(np.uint32).reshape(grid, 2).tolist())
    out = from16(Cbuf.view(np.uint16).reshape(M, N), dt).astype(np.float64)
    # reference
    acc = from16(A, dt).astype(np.float64) @ from16(B, dt).astype(np.float64).T
    if gen.ln:
        v = (acc - mean[:, None].astype(np.float64) * csum[None, :]) * rstd[:, None] + bias[None, :]
    else:
        v = acc + bias[None, :]
    if gen.gelu:
        v = gelu_ref(v)
    if gen.res:
        v = from16(to16(v, dt), dt).astype(np.float64) + from16(R, dt)
    ref = from16(to16(v, dt), dt).astype(np.float64)
    written = Cbuf.view(np.uint16).reshape(M, N) != 0x7FC1
    err = np.abs(out - ref)
    tol = (2.0 ** -7 if dt == "bf16" else 2.0 ** -10) * np.maximum(1.0, np.abs(ref)) * 1.01
    bad = ~(err <= tol)
    if verbose or bad.any() or not written.all():
        print("case M=%d N=%d K=%d grid=%d cg=%d mode=%s: unwritten %d, bad %d, max err %.3g, instructions %d" %
              (M, N, K, grid, cgroups, dma_mode, (~written).sum(), bad.sum(), np.nanmax(err) if np.isfinite(err).any() else float("nan"), nins))
        if bad.any():
            idx = np.argwhere(bad)
            print("  first bad:", idx[:8].tolist())
            rows = sorted(set(idx[:, 0] // 32))
            cols = sorted(set(idx[:, 1] // 32))
            print("  bad row blocks of 32:", rows[:40], " col blocks:", cols[:40])
    ok = bool(written.all() and not bad.any())
    if getattr(gen, "stats", False):
        got = mem.get(aPart).view(np.float32).reshape(N // 32, M, 2).astype(np.float64)
        outq = out.reshape(M, N // 32, 32)
        want = np.stack([outq.sum(axis=2).T, (outq * outq).sum(axis=2).T], axis=2)
        perr = np.abs(got - want).max() if np.isfinite(got).all() else float("nan")
        if not (perr < 1e-3):
            print("row statistics: max error %s" % perr)
            ok = False
    return ok


if __name__ == "__main__":
    import time
    t0 = time.time()
    g = q4gen.Q4(gelu=False, ln=False, nkf=2)
    print("generated", len(g.a.ins), "instructions in %.1fs" % (time.time() - t0))
    pr = isa.lint(g.a, verbose=True)
    print("lint problems:", len(pr))
    t0 = time.time()
    ok = run_case(g, 256, 128, 192, grid=8, verbose=True)
    print("ok" if ok else "FAILED", "%.1fs" % (time.time() - t0))
