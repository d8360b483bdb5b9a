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


def mfma_loops(body):
    """Innermost loops containing >= 8 MFMAs: (start, end, n_mfma) of every conditional backward branch to a label whose
    body holds no other such loop."""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            s = labels[m.group(1)]
            n = sum("v_mfma" in x for x in body[s:i])
            if n >= 8:
                loops.append((s, i, n))
    return [a for a in loops if not any(b is not a and a[0] <= b[0] and b[1] <= a[1] for b in loops)]


def steady_loop(body):
    """the smallest of them (the persistent GEMM tile has one)"""
    loops = mfma_loops(body)
    return min(loops, key=lambda t: t[1] - t[0]) if loops else None


def lint(path, pattern="gemm_nt_p8_kernel", hand_wait=r"s_waitcnt vmcnt\(8\) lgkmcnt\(0\)$", every_loop=False):
    lines = open(path).read().split("\n")
    bad = 0
    for name, a, b in kernels(lines, pattern):
        body = lines[a:b]
        found = mfma_loops(body) if every_loop else [steady_loop(body)]
        if not found or found[0] is None:
            print("%s: no MFMA loop found" % name)
            bad += 1
            continue
        for s, e, n in found:
            loop = [x.strip() for x in body[s:e + 1]]
            spills = [x for x in loop if x.startswith(("scratch_", "buffer_load", "buffer_store"))]
            waits = [x for x in loop if x.startswith("s_waitcnt") and "vmcnt" in x]
            foreign = [x for x in waits if not re.match(hand_wait, x)]
            status = "ok" if not spills and not foreign else "BAD"
            print("%-70s loop %4d instr, %3d mfma, %d glds, %d spill ops, vmcnt waits: %s  -> %s"
                  % (name[9:], len([x for x in loop if x and not x.startswith((";", "."))]), n,
                     sum("global_load_lds" in x for x in loop), len(spills), sorted(set(waits)), status))
            bad += status != "ok"
    return bad


# the fused token-mixing kernels: one hand-counted wait per iteration each
TOKEN_KERNELS = (("token_mlp_kernel", r"s_waitcnt vmcnt\(7\) lgkmcnt\(0\)$"), ("token_mlp_rr_kernel", r"s_waitcnt vmcnt\(4\) lgkmcnt\(0\)$"))


def lint_token(path):
    return sum(lint(path, pat, wait, every_loop=True) for pat, wait in TOKEN_KERNELS)


def _regs(tok):
    """VGPR numbers named by one operand (v7, v[4:7])"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def lint_inflight(path, pattern="token_gemm_pipe_kernel"):
    """Kernels that issue global loads from `asm volatile` and wait for them with hand-counted s_waitcnt: the compiler does not know
    that the destination of such a load is in flight.  Walk every kernel in layout order: a register written by an asm global_load is
    in flight until the next asm `s_waitcnt vmcnt` (a conservative reading: in the loop the covering wait is the one at the top of the
    iteration after next, which in layout order comes first); any compiler instruction that names it before that -- a copy, a spill, a
    use -- reads data that has not arrived.  Also: no scratch traffic, and no vmcnt wait the compiler added, anywhere in the loop."""
    lines = open(path).read().split("\n")
    bad = 0
    for name, a, b in kernels(lines, pattern):
        body = lines[a:b]
        inflight, in_asm, hits, loads = set(), False, [], 0
        for l in body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".")):
                continue
            ops = [x.strip() for x in t.split(None, 1)[1].split(",")] if len(t.split(None, 1)) > 1 else []
            if in_asm:
                if t.startswith("global_load_dword"):
                    inflight |= _regs(ops[0])
                    loads += 1
                elif t.startswith("s_waitcnt") and "vmcnt" in t:
                    inflight = set()
                continue
            named = set()
            for o in ops:
                named |= _regs(o.split()[0] if o else o)
            if named & inflight:
                hits.append(t)
        loop = [x.strip() for x in body]
        spills = [x for x in loop if x.startswith("scratch_")]
        print("%-70s %3d asm loads, %d compiler instructions on in-flight registers, %d scratch ops -> %s"
              % (name[9:], loads, len(hits), len(spills), "ok" if not hits and not spills and loads else "BAD"))
        for h in hits[:5]:
            print("    ", h)
        bad += bool(hits or spills or not loads)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "inflight":
        sys.exit(1 if lint_inflight(sys.argv[1]) else 0)
    if len(sys.argv) > 2 and sys.argv[2] == "token":
        sys.exit(1 if lint_token(sys.argv[1]) else 0)
    sys.exit(1 if lint(sys.argv[1], *(sys.argv[2:3])) else 0)
