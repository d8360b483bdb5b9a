#!/usr/bin/env python3
"""Static check of the hand-scheduled GEMM main loops in the gfx950 ISA hipcc produced (no GPU needed).

The persistent tile's K loop counts its LDS-DMA prefetches by hand (`s_waitcnt vmcnt(8)` inside volatile asm).  hipcc
knows nothing about those asm loads: if register pressure makes it spill inside the loop, the `scratch_load` it adds
is a VMEM operation of its own and the waits it inserts for it (`s_waitcnt vmcnt(0..4)`) drain the whole prefetch
queue every slab -- the kernel still passes every test and loses a third of its speed.  This lint disassembles the
kernel, finds the steady-state loop (the backward branch that encloses 64 MFMAs at NI = 4) and fails on
  * any scratch_ / buffer_ spill traffic inside it,
  * any s_waitcnt on vmcnt other than the hand-counted ones (vmcnt(8) together with lgkmcnt(0)).
usage: python tools/isa_lint.py <file.s>      (hipcc -save-temps output)  -> exit status 0 / 1
"""
import re
import sys


def kernels(lines, pattern):
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN4mlpk\w*%s\w*):" % pattern, l)
        if m:
            j = i
            while j < len(lines) and ".end_amdhsa_kernel" not in lines[j] and not lines[j].startswith("\t.section\t.rodata"):
                j += 1
            out.append((m.group(1), i, j))
    return out


def steady_loop(body):
    """(start, end) of the innermost loop containing MFMAs: a conditional backward branch to a label."""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            s = labels[m.group(1)]
            n = sum("v_mfma" in x for x in body[s:i])
            if n >= 8 and (best is None or i - s < best[1] - best[0]):
                best = (s, i, n)
    return best


def lint(path, pattern="gemm_nt_p8_kernel"):
    lines = open(path).read().split("\n")
    bad = 0
    for name, a, b in kernels(lines, pattern):
        body = lines[a:b]
        lp = steady_loop(body)
        if lp is None:
            print("%s: no MFMA loop found" % name)
            bad += 1
            continue
        s, e, n = lp
        loop = [x.strip() for x in body[s:e + 1]]
        spills = [x for x in loop if x.startswith(("scratch_", "buffer_load", "buffer_store"))]
        waits = [x for x in loop if x.startswith("s_waitcnt") and "vmcnt" in x]
        foreign = [x for x in waits if not re.match(r"s_waitcnt vmcnt\(8\) lgkmcnt\(0\)$", x)]
        status = "ok" if not spills and not foreign else "BAD"
        print("%-70s loop %4d instr, %3d mfma, %d glds, %d spill ops, vmcnt waits: %s  -> %s"
              % (name[9:], len([x for x in loop if x and not x.startswith((";", "."))]), n,
                 sum("global_load_lds" in x for x in loop), len(spills), sorted(set(waits)), status))
        bad += status != "ok"
    return bad


if __name__ == "__main__":
    sys.exit(1 if lint(sys.argv[1], *(sys.argv[2:3])) else 0)
