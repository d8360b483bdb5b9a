"""A small gfx950 instruction IR: emit as assembler text, run functionally on numpy, lint for software hazards.

Used by q4gen.py (the hand-scheduled one-wave-per-SIMD GEMM).  The kernel is written as a Python program that appends
instructions to an `Asm`; the SAME instruction list is
  * printed as the body of one `asm volatile` block (hipcc only allocates nothing inside it: every register is named here),
  * executed by `Emu` -- one workgroup of 64-lane waves on numpy arrays, with the asynchronous parts modelled ADVERSARIALLY:
    LDS-DMA data lands only when the issuing wave's own `s_waitcnt vmcnt` forces it to ("late" mode; "early" mode lands it at
    issue), loaded registers hold poison until the covering wait, waves run one after the other between barriers -- so a
    missing wait / barrier / wrong count shows up as a wrong result on the CPU instead of as a rare wrong tile on the GPU,
  * checked by `lint` for the software-visible hazards hipcc would normally pad (nothing pads inline asm).
Only the instructions the generator uses are implemented; an unknown mnemonic raises.
"""
import struct

import os

import numpy as np

POISON = 0x7FC0DEAD


# ------------------------------------------------------------------ operands
class Reg:
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        self.kind, self.idx, self.n = kind, idx, n

    def __getitem__(self, k):
        if isinstance(k, slice):
            a, b, _ = k.indices(self.n)
            return Reg(self.kind, self.idx + a, b - a)
        assert 0 <= k < self.n, (self, k)
        return Reg(self.kind, self.idx + k, 1)

    def text(self):
        if self.n == 1:
            return "%s%d" % (self.kind, self.idx)
        return "%s[%d:%d]" % (self.kind, self.idx, self.idx + self.n - 1)

    def __repr__(self):
        return self.text()


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


class Neg:
    """source modifier -x (VOP3 float operands)"""

    def __init__(self, r):
        self.r = r


class Abs:
    """source modifier |x| (VOP3 float operands)"""

    def __init__(self, r):
        self.r = r


class F:
    """float literal / inline constant"""
    INLINE = {0.0: "0", 0.5: "0.5", -0.5: "-0.5", 1.0: "1.0", -1.0: "-1.0", 2.0: "2.0", -2.0: "-2.0", 4.0: "4.0", -4.0: "-4.0"}

    def __init__(self, x):
        self.x = float(np.float32(x))

    def bits(self):
        return struct.unpack("<I", struct.pack("<f", self.x))[0]

    def text(self):
        return self.INLINE.get(self.x, "0x%08x" % self.bits())


class H:
    """inline constant of a PACKED f16 instruction (VOP3P): the assembler puts the f16 value in the low half of the operand and zero in
    the high half, so the instruction must name it with op_sel_hi 0 for that source to feed both lanes (tools/ubench/h2_semantics.hip)"""
    INLINE = F.INLINE

    def __init__(self, x):
        self.x = float(x)
        assert self.x in self.INLINE, "not an inline constant: %r" % x

    def bits(self):
        return int(np.float16(self.x).view(np.uint16))

    def text(self):
        return self.INLINE[self.x]


def h2bits(x):
    """the same f16 value in both halves of a 32-bit word (a packed constant held in a register)"""
    b = int(np.float16(x).view(np.uint16))
    return b | (b << 16)


def _optext(o):
    if isinstance(o, H):
        return o.text()
    if isinstance(o, Reg):
        return o.text()
    if isinstance(o, Neg):
        return "-" + _optext(o.r)
    if isinstance(o, Abs):
        return "|" + _optext(o.r) + "|"
    if isinstance(o, F):
        return o.text()
    if isinstance(o, bool):
        raise TypeError(o)
    if isinstance(o, int):
        return str(o) if -16 <= o <= 64 else "0x%x" % (o & 0xFFFFFFFF)
    if isinstance(o, str):
        return o
    raise TypeError(o)


class Ins:
    __slots__ = ("op", "args", "mods", "tag")

    def __init__(self, op, args, mods, tag=None):
        self.op, self.args, self.mods, self.tag = op, args, mods, tag

    def text(self):
        if self.op == "label":
            return "%s:" % self.args[0]
        if self.op == "s_waitcnt":
            parts = []
            if "vmcnt" in self.mods:
                parts.append("vmcnt(%d)" % self.mods["vmcnt"])
            if "lgkmcnt" in self.mods:
                parts.append("lgkmcnt(%d)" % self.mods["lgkmcnt"])
            return "s_waitcnt " + " ".join(parts)
        s = self.op
        if self.args:
            s += " " + ", ".join(_optext(a) for a in self.args)
        for k, v in self.mods.items():
            if k == "offset":
                if v:
                    s += " offset:%d" % v
            elif v is True:
                s += " " + k
            else:
                s += " %s:%s" % (k, v)
        return s


class Asm:
    def __init__(self):
        self.ins = []
        self._lab = 0

    def __call__(self, op, *args, **mods):
        tag = mods.pop("tag", None)
        i = Ins(op, args, mods, tag)
        self.ins.append(i)
        return i

    def label(self, name):
        self.ins.append(Ins("label", (name,), {}))

    def newlabel(self, stem="L"):
        self._lab += 1
        return "%s_%d_%%=" % (stem, self._lab)

    def text(self):
        return "\n".join(("" if i.op == "label" else "  ") + i.text() for i in self.ins)

    def c_string(self):
        """the body as a C string literal for one asm volatile block ('%=' keeps labels unique per instantiation)"""
        out = []
        for i in self.ins:
            t = i.text()
            out.append('"%s\\n\\t"' % t)
        return "\n".join(out)


# ------------------------------------------------------------------ emulator
def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _f32_to_bf16_rne(f):
    u = f.astype(np.float32).view(np.uint32)
    r = u + 0x7FFF + ((u >> 16) & 1)
    out = (r >> 16).astype(np.uint32)
    nan = np.isnan(f)
    out[nan] = 0x7FC0
    return out


class Mem:
    """flat global memory made of numpy byte buffers at fixed base addresses"""

    def __init__(self):
        self.bufs = []
        self.next = 0x10000000

    def add(self, arr, writable=False):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        if writable:
            b = b.copy()
        base = self.next
        self.next += (len(b) + 0xFFFF) & ~0xFFFF
        self.next += 0x10000
        self.bufs.append((base, b))
        return base

    def get(self, base):
        for b0, b in self.bufs:
            if b0 == base:
                return b
        raise KeyError(base)

    def _find(self, addr, nbytes):
        for b0, b in self.bufs:
            if b0 <= addr and addr + nbytes <= b0 + len(b):
                return b0, b
        raise RuntimeError("global access out of bounds: 0x%x (+%d)" % (addr, nbytes))

    def gather(self, addrs, nbytes):
        """addrs: int64[L]; returns uint8[L, nbytes]"""
        lo, hi = int(addrs.min()), int(addrs.max())
        b0, b = self._find(lo, 1)
        if hi + nbytes > b0 + len(b):
            raise RuntimeError("global gather out of bounds: 0x%x..0x%x buffer 0x%x+%d" % (lo, hi + nbytes, b0, len(b)))
        idx = (addrs - b0)[:, None] + np.arange(nbytes)[None, :]
        return b[idx]

    def scatter(self, addrs, data):
        nbytes = data.shape[1]
        lo, hi = int(addrs.min()), int(addrs.max())
        b0, b = self._find(lo, 1)
        if hi + nbytes > b0 + len(b):
            raise RuntimeError("global scatter out of bounds")
        idx = (addrs - b0)[:, None] + np.arange(nbytes)[None, :]
        b[idx] = data


class Wave:
    def __init__(self, wid):
        self.wid = wid
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(128, np.uint64)          # 32-bit values (kept in 64-bit cells to dodge overflow warnings)
        self.m0 = 0
        self.exec = np.ones(64, bool)
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.vm = []       # outstanding vector-memory operations, issue order: ("load"|"store"|"dma", completion closure)
        self.lgkm = []     # outstanding LDS reads / scalar loads: completion closures
        self.spend = set() # SGPRs with a scalar load pending
        self.nissued = 0


class Emu:
    def __init__(self, asm, mem, lds_bytes=163840, nwaves=4, dma_mode="late", order=None, max_steps=50_000_000):
        self.ins = asm.ins
        self.labels = {i.args[0]: k for k, i in enumerate(self.ins) if i.op == "label"}
        self.mem = mem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.nwaves = nwaves
        self.dma_mode = dma_mode
        self.order = order or list(range(nwaves))
        self.max_steps = max_steps
        self.trace = None

    # ---- register access
    def rd(self, w, o, n=None):
        """vector-shaped (uint32[64]) value of a 32-bit operand"""
        if isinstance(o, Reg):
            assert o.n == 1, o
            if o.kind == "v":
                return w.v[o.idx]
            if o.kind == "a":
                return w.a[o.idx]
            if o.kind == "s":
                if o.idx in w.spend:
                    raise RuntimeError("read of s%d before its s_load was waited for (pc %d)" % (o.idx, w.pc))
                return np.full(64, w.s[o.idx], np.uint32)
        if isinstance(o, (F, H)):
            return np.full(64, o.bits(), np.uint32)
        if isinstance(o, Neg):
            return self.rd(w, o.r) ^ np.uint32(0x80000000)
        if isinstance(o, Abs):
            return self.rd(w, o.r) & np.uint32(0x7FFFFFFF)
        if isinstance(o, int):
            return np.full(64, o & 0xFFFFFFFF, np.uint32)
        if o == "m0":
            return np.full(64, w.m0, np.uint32)
        raise TypeError(o)

    def rds(self, w, o):
        """scalar value"""
        if isinstance(o, Reg):
            assert o.kind == "s" and o.n == 1, o
            if o.idx in w.spend:
                raise RuntimeError("read of s%d before its s_load was waited for (pc %d)" % (o.idx, w.pc))
            return int(w.s[o.idx])
        if isinstance(o, F):
            return o.bits()
        if isinstance(o, int):
            return o & 0xFFFFFFFF
        if o == "m0":
            return w.m0
        raise TypeError(o)

    def rds64(self, w, o):
        assert o.kind == "s" and o.n == 2
        return self.rds(w, o[0]) | (self.rds(w, o[1]) << 32)

    def wrs(self, w, o, val):
        val &= 0xFFFFFFFF
        if o == "m0":
            w.m0 = val
        else:
            assert o.kind == "s" and o.n == 1
            w.s[o.idx] = val

    def wrv(self, w, o, val):
        assert o.n == 1
        (w.v if o.kind == "v" else w.a)[o.idx] = val.astype(np.uint32, copy=False)

    @staticmethod
    def f(x):
        return x.view(np.float32)

    @staticmethod
    def u(x):
        return np.asarray(x, np.float32).view(np.uint32)

    # ---- asynchronous completion
    # Stores are modelled as the operations that complete FIRST (out of issue order): adversarial for any wait that would count them.
    # The family's documented behaviour is in-order completion for loads and stores alike (LLVM AMDGPUUsage, memory model GFX6-GFX9 /
    # GFX90A / GFX942: "completion is reported to a wavefront in execution order"); STORES_IN_ORDER = True models that -- needed only by
    # kernels generated with MLPK_Q4_COUNT_STORES=1 (a round-6 experiment that measured no gain).
    STORES_IN_ORDER = os.environ.get("MLPK_Q4_COUNT_STORES", "0") == "1"

    def wait_vm(self, w, n):
        """in issue order, as late as allowed (STORES_IN_ORDER = False: adversarial for counted stores -- stores complete first)"""
        while len(w.vm) > n:
            k = None if self.STORES_IN_ORDER else next((i for i, (kind, _) in enumerate(w.vm) if kind == "store"), None)
            if k is None:
                k = 0
            _, fn = w.vm.pop(k)
            if fn:
                fn()

    def wait_lgkm(self, w, n):
        while len(w.lgkm) > n:
            fn = w.lgkm.pop(0)
            fn()

    # ---- execution
    def run(self):
        waves = [Wave(i) for i in range(self.nwaves)]
        self.waves = waves
        self.init(waves)
        steps = 0
        while not all(w.done for w in waves):
            progressed = False
            for wi in self.order:
                w = waves[wi]
                if w.done or w.at_barrier:
                    continue
                while not w.done and not w.at_barrier:
                    self.step(w)
                    steps += 1
                    if steps > self.max_steps:
                        raise RuntimeError("emulator step limit")
                progressed = True
            live = [w for w in waves if not w.done]
            if live and all(w.at_barrier for w in live):
                if len(live) != len(waves):
                    raise RuntimeError("barrier reached by %d of %d waves (others ended)" % (len(live), len(waves)))
                for w in live:
                    w.at_barrier = False
                progressed = True
            if not progressed:
                raise RuntimeError("deadlock")
        return waves

    def init(self, waves):
        pass

    def step(self, w):
        i = self.ins[w.pc]
        w.pc += 1
        if i.op == "label":
            return
        w.nissued += 1
        fn = getattr(self, "x_" + i.op, None)
        if fn is None:
            raise NotImplementedError(i.op)
        fn(w, i)

    # ---- SALU
    def x_s_mov_b32(self, w, i):
        d = i.args[0]
        if d in ("exec_lo", "exec_hi"):
            val = self.rds(w, i.args[1])
            lo = 0 if d == "exec_lo" else 32
            for k in range(32):
                w.exec[lo + k] = bool((val >> k) & 1)
            return
        self.wrs(w, d, self.rds(w, i.args[1]))

    def x_s_mov_b64(self, w, i):
        d, s = i.args
        if d == "exec":
            val = 0xFFFFFFFFFFFFFFFF if s == -1 else self.rds64(w, s)
            w.exec = np.array([(val >> k) & 1 for k in range(64)], bool)
            return
        self.wrs(w, d[0], self.rds(w, s[0]))
        self.wrs(w, d[1], self.rds(w, s[1]))

    def x_s_add_u32(self, w, i):
        r = self.rds(w, i.args[1]) + self.rds(w, i.args[2])
        w.scc = 1 if r > 0xFFFFFFFF else 0
        self.wrs(w, i.args[0], r)

    def x_s_addc_u32(self, w, i):
        r = self.rds(w, i.args[1]) + self.rds(w, i.args[2]) + w.scc
        w.scc = 1 if r > 0xFFFFFFFF else 0
        self.wrs(w, i.args[0], r)

    def x_s_sub_u32(self, w, i):
        a, b = self.rds(w, i.args[1]), self.rds(w, i.args[2])
        w.scc = 1 if b > a else 0
        self.wrs(w, i.args[0], a - b)

    def x_s_subb_u32(self, w, i):
        r = self.rds(w, i.args[1]) - self.rds(w, i.args[2]) - w.scc
        w.scc = 1 if r < 0 else 0
        self.wrs(w, i.args[0], r)

    def x_s_add_i32(self, w, i):
        r = self.rds(w, i.args[1]) + self.rds(w, i.args[2])
        self.wrs(w, i.args[0], r)
        w.scc = 0      # (overflow flag: never consumed by the generator)

    def x_s_sub_i32(self, w, i):
        self.wrs(w, i.args[0], self.rds(w, i.args[1]) - self.rds(w, i.args[2]))
        w.scc = 0

    def x_s_mul_i32(self, w, i):
        self.wrs(w, i.args[0], self.rds(w, i.args[1]) * self.rds(w, i.args[2]))

    def x_s_mul_hi_u32(self, w, i):
        self.wrs(w, i.args[0], (self.rds(w, i.args[1]) * self.rds(w, i.args[2])) >> 32)

    def x_s_lshl_b32(self, w, i):
        r = (self.rds(w, i.args[1]) << (self.rds(w, i.args[2]) & 31)) & 0xFFFFFFFF
        self.wrs(w, i.args[0], r)
        w.scc = 1 if r else 0

    def x_s_lshr_b32(self, w, i):
        r = self.rds(w, i.args[1]) >> (self.rds(w, i.args[2]) & 31)
        self.wrs(w, i.args[0], r)
        w.scc = 1 if r else 0

    def x_s_and_b32(self, w, i):
        r = self.rds(w, i.args[1]) & self.rds(w, i.args[2])
        self.wrs(w, i.args[0], r)
        w.scc = 1 if r else 0

    def x_s_or_b32(self, w, i):
        r = self.rds(w, i.args[1]) | self.rds(w, i.args[2])
        self.wrs(w, i.args[0], r)
        w.scc = 1 if r else 0

    def x_s_min_u32(self, w, i):
        a, b = self.rds(w, i.args[1]), self.rds(w, i.args[2])
        self.wrs(w, i.args[0], min(a, b))
        w.scc = 1 if a <= b else 0

    def _cmp(self, w, i, fn, signed=False):
        a, b = self.rds(w, i.args[0]), self.rds(w, i.args[1])
        if signed:
            a = a - (1 << 32) if a & 0x80000000 else a
            b = b - (1 << 32) if b & 0x80000000 else b
        w.scc = 1 if fn(a, b) else 0

    def x_s_cmp_eq_u32(self, w, i):
        self._cmp(w, i, lambda a, b: a == b)

    def x_s_cmp_lg_u32(self, w, i):
        self._cmp(w, i, lambda a, b: a != b)

    def x_s_cmp_lt_u32(self, w, i):
        self._cmp(w, i, lambda a, b: a < b)

    def x_s_cmp_le_u32(self, w, i):
        self._cmp(w, i, lambda a, b: a <= b)

    def x_s_cmp_gt_u32(self, w, i):
        self._cmp(w, i, lambda a, b: a > b)

    def x_s_cmp_ge_u32(self, w, i):
        self._cmp(w, i, lambda a, b: a >= b)

    def x_s_cmp_lt_i32(self, w, i):
        self._cmp(w, i, lambda a, b: a < b, True)

    def x_s_cmp_gt_i32(self, w, i):
        self._cmp(w, i, lambda a, b: a > b, True)

    def x_s_cmp_ge_i32(self, w, i):
        self._cmp(w, i, lambda a, b: a >= b, True)

    def x_s_min_i32(self, w, i):
        sg = lambda x: x - (1 << 32) if x & 0x80000000 else x
        self.wrs(w, i.args[0], min(sg(self.rds(w, i.args[1])), sg(self.rds(w, i.args[2]))))

    def x_s_max_i32(self, w, i):
        sg = lambda x: x - (1 << 32) if x & 0x80000000 else x
        self.wrs(w, i.args[0], max(sg(self.rds(w, i.args[1])), sg(self.rds(w, i.args[2]))))

    def x_s_cselect_b32(self, w, i):
        self.wrs(w, i.args[0], self.rds(w, i.args[1]) if w.scc else self.rds(w, i.args[2]))

    def x_s_branch(self, w, i):
        w.pc = self.labels[i.args[0]]

    def x_s_cbranch_scc0(self, w, i):
        if not w.scc:
            w.pc = self.labels[i.args[0]]

    def x_s_cbranch_scc1(self, w, i):
        if w.scc:
            w.pc = self.labels[i.args[0]]

    def x_s_nop(self, w, i):
        pass

    def x_s_setprio(self, w, i):
        pass

    def x_s_endpgm(self, w, i):
        if w.vm:
            raise RuntimeError("wave %d ends with %d vector-memory operations outstanding" % (w.wid, len(w.vm)))
        w.done = True

    def x_s_barrier(self, w, i):
        w.at_barrier = True

    def x_s_waitcnt(self, w, i):
        if "vmcnt" in i.mods:
            self.wait_vm(w, i.mods["vmcnt"])
        if "lgkmcnt" in i.mods:
            self.wait_lgkm(w, i.mods["lgkmcnt"])

    def _s_load(self, w, i, n):
        d, base, off = i.args
        addr = self.rds64(w, base) + (off if isinstance(off, int) else self.rds(w, off))
        data = self.mem.gather(np.array([addr], np.int64), 4 * n)[0].view(np.uint32).copy()
        regs = [d.idx + k for k in range(n)]
        for r in regs:
            w.spend.add(r)
            w.s[r] = POISON

        def done():
            for k, r in enumerate(regs):
                w.s[r] = data[k]
                w.spend.discard(r)
        w.lgkm.append(done)

    def x_s_memtime(self, w, i):
        d = i.args[0]
        regs = [d.idx, d.idx + 1]
        val = w.nissued * 4
        for r in regs:
            w.spend.add(r)

        def done():
            w.s[regs[0]] = val & 0xFFFFFFFF
            w.s[regs[1]] = val >> 32
            for r in regs:
                w.spend.discard(r)
        w.lgkm.append(done)

    def x_s_load_dword(self, w, i):
        self._s_load(w, i, 1)

    def x_s_load_dwordx2(self, w, i):
        self._s_load(w, i, 2)

    def x_s_load_dwordx4(self, w, i):
        self._s_load(w, i, 4)

    def x_s_load_dwordx8(self, w, i):
        self._s_load(w, i, 8)

    def x_s_load_dwordx16(self, w, i):
        self._s_load(w, i, 16)

    # ---- VALU integer
    def x_v_mov_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]).copy())

    def x_v_add_u32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]) + self.rd(w, i.args[2]))

    def x_v_sub_u32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]) - self.rd(w, i.args[2]))

    def x_v_mul_lo_u32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]) * self.rd(w, i.args[2]))

    def x_v_mul_u32_u24(self, w, i):
        self.wrv(w, i.args[0], (self.rd(w, i.args[1]) & 0xFFFFFF) * (self.rd(w, i.args[2]) & 0xFFFFFF))

    def x_v_lshlrev_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[2]) << (self.rd(w, i.args[1]) & 31))

    def x_v_lshrrev_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[2]) >> (self.rd(w, i.args[1]) & 31))

    def x_v_and_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]) & self.rd(w, i.args[2]))

    def x_v_or_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]) | self.rd(w, i.args[2]))

    def x_v_xor_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]) ^ self.rd(w, i.args[2]))

    def x_v_min_u32(self, w, i):
        self.wrv(w, i.args[0], np.minimum(self.rd(w, i.args[1]), self.rd(w, i.args[2])))

    def x_v_lshl_add_u32(self, w, i):
        self.wrv(w, i.args[0], (self.rd(w, i.args[1]) << (self.rd(w, i.args[2]) & 31)) + self.rd(w, i.args[3]))

    def x_v_add_lshl_u32(self, w, i):
        self.wrv(w, i.args[0], (self.rd(w, i.args[1]) + self.rd(w, i.args[2])) << (self.rd(w, i.args[3]) & 31))

    def x_v_and_or_b32(self, w, i):
        self.wrv(w, i.args[0], (self.rd(w, i.args[1]) & self.rd(w, i.args[2])) | self.rd(w, i.args[3]))

    def x_v_bfe_u32(self, w, i):
        off, width = self.rd(w, i.args[2]) & 31, self.rd(w, i.args[3]) & 31
        self.wrv(w, i.args[0], (self.rd(w, i.args[1]) >> off) & ((np.uint32(1) << width) - np.uint32(1)))

    def x_v_mad_u32_u24(self, w, i):
        self.wrv(w, i.args[0], (self.rd(w, i.args[1]) & 0xFFFFFF) * (self.rd(w, i.args[2]) & 0xFFFFFF) + self.rd(w, i.args[3]))

    def x_v_mbcnt_lo_u32_b32(self, w, i):
        assert i.args[1] == -1
        self.wrv(w, i.args[0], np.minimum(np.arange(64), 32).astype(np.uint32) + self.rd(w, i.args[2]))

    def x_v_mbcnt_hi_u32_b32(self, w, i):
        assert i.args[1] == -1
        self.wrv(w, i.args[0], np.maximum(np.arange(64) - 32, 0).astype(np.uint32) + self.rd(w, i.args[2]))

    def x_v_readfirstlane_b32(self, w, i):
        self.wrs(w, i.args[0], int(self.rd(w, i.args[1])[0]))

    # ---- VALU float
    def _fma(self, a, b, c):
        with np.errstate(all="ignore"):
            return (self.f(a).astype(np.float64) * self.f(b).astype(np.float64) + self.f(c).astype(np.float64)).astype(np.float32)

    def x_v_fma_f32(self, w, i):
        self.wrv(w, i.args[0], self.u(self._fma(self.rd(w, i.args[1]), self.rd(w, i.args[2]), self.rd(w, i.args[3]))))

    def x_v_fmaak_f32(self, w, i):      # d = s0 * s1 + K
        self.wrv(w, i.args[0], self.u(self._fma(self.rd(w, i.args[1]), self.rd(w, i.args[2]), self.rd(w, i.args[3]))))

    def x_v_fmamk_f32(self, w, i):      # d = s0 * K + s1
        self.wrv(w, i.args[0], self.u(self._fma(self.rd(w, i.args[1]), self.rd(w, i.args[2]), self.rd(w, i.args[3]))))

    def x_v_fmac_f32(self, w, i):       # d = s0 * s1 + d
        self.wrv(w, i.args[0], self.u(self._fma(self.rd(w, i.args[1]), self.rd(w, i.args[2]), self.rd(w, i.args[0]))))

    def x_v_mul_f32(self, w, i):
        with np.errstate(all="ignore"):
            self.wrv(w, i.args[0], self.u(self.f(self.rd(w, i.args[1])) * self.f(self.rd(w, i.args[2]))))

    def x_v_add_f32(self, w, i):
        with np.errstate(all="ignore"):
            self.wrv(w, i.args[0], self.u(self.f(self.rd(w, i.args[1])) + self.f(self.rd(w, i.args[2]))))

    # transcendentals: 1-ulp approximations in hardware, correctly rounded here (results are compared with a tolerance)
    def x_v_exp_f32(self, w, i):
        with np.errstate(all="ignore"):
            self.wrv(w, i.args[0], self.u(np.exp2(self.f(self.rd(w, i.args[1])).astype(np.float64)).astype(np.float32)))

    def x_v_rcp_f32(self, w, i):
        with np.errstate(all="ignore"):
            self.wrv(w, i.args[0], self.u((1.0 / self.f(self.rd(w, i.args[1])).astype(np.float64)).astype(np.float32)))

    def x_v_sub_f32(self, w, i):
        with np.errstate(all="ignore"):
            self.wrv(w, i.args[0], self.u(self.f(self.rd(w, i.args[1])) - self.f(self.rd(w, i.args[2]))))

    # ---- packed fp32 (two lanes' worth of fp32 per instruction: 64-bit register pairs, element k of every operand)
    def _rd2(self, w, o):
        if isinstance(o, Reg):
            assert o.n == 2 and o.idx % 2 == 0, o
            return self.rd(w, o[0]), self.rd(w, o[1])
        raise TypeError(o)

    def x_v_pk_mul_f32(self, w, i):
        a, b = self._rd2(w, i.args[1]), self._rd2(w, i.args[2])
        with np.errstate(all="ignore"):
            for k in range(2):
                self.wrv(w, i.args[0][k], self.u(self.f(a[k]) * self.f(b[k])))

    def x_v_pk_add_f32(self, w, i):
        a, b = self._rd2(w, i.args[1]), self._rd2(w, i.args[2])
        with np.errstate(all="ignore"):
            for k in range(2):
                self.wrv(w, i.args[0][k], self.u(self.f(a[k]) + self.f(b[k])))

    def x_v_pk_fma_f32(self, w, i):
        a, b, c = self._rd2(w, i.args[1]), self._rd2(w, i.args[2]), self._rd2(w, i.args[3])
        res = [self.u(self._fma(a[k], b[k], c[k])) for k in range(2)]
        for k in range(2):
            self.wrv(w, i.args[0][k], res[k])

    def x_v_add_f32_dpp(self, w, i):
        """dst = dpp(src0) + src1; quad_perm / row_half_mirror with full row and bank masks (every lane enabled)"""
        assert w.exec.all()
        a, b = self.rd(w, i.args[1]), self.rd(w, i.args[2])
        lane = np.arange(64)
        if "quad_perm" in i.mods:
            perm = [int(x) for x in i.mods["quad_perm"].strip("[]").split(",")]
            src = (lane & ~3) | np.array(perm)[lane & 3]
        elif i.mods.get("row_half_mirror"):
            src = (lane & ~7) | (7 - (lane & 7))
        else:
            raise NotImplementedError(i.mods)
        with np.errstate(all="ignore"):
            self.wrv(w, i.args[0], self.u(self.f(a[src]) + self.f(b)))

    def _dot2c(self, w, i, dec):
        d, a, b = i.args
        av, bv, dv = self.rd(w, a), self.rd(w, b), self.rd(w, d)
        with np.errstate(all="ignore"):
            r = (self.f(dv).astype(np.float64) + dec(av & 0xFFFF).astype(np.float64) * dec(bv & 0xFFFF).astype(np.float64)
                 + dec(av >> 16).astype(np.float64) * dec(bv >> 16).astype(np.float64)).astype(np.float32)
        self.wrv(w, d, self.u(r))

    def x_v_dot2c_f32_bf16(self, w, i):
        self._dot2c(w, i, lambda u: _bf16_to_f32(u.astype(np.uint16)))

    def x_v_dot2c_f32_f16(self, w, i):
        self._dot2c(w, i, lambda u: u.astype(np.uint16).view(np.float16).astype(np.float32))

    def x_v_med3_f32(self, w, i):
        a, b, c = (self.f(self.rd(w, x)) for x in i.args[1:4])
        with np.errstate(all="ignore"):
            r = np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))
        self.wrv(w, i.args[0], self.u(r))

    def x_v_cvt_pk_bf16_f32(self, w, i):
        lo = _f32_to_bf16_rne(self.f(self.rd(w, i.args[1])))
        hi = _f32_to_bf16_rne(self.f(self.rd(w, i.args[2])))
        self.wrv(w, i.args[0], lo | (hi << 16))

    def x_v_cvt_pk_f16_f32(self, w, i):
        with np.errstate(all="ignore"):
            lo = self.f(self.rd(w, i.args[1])).astype(np.float16).view(np.uint16).astype(np.uint32)
            hi = self.f(self.rd(w, i.args[2])).astype(np.float16).view(np.uint16).astype(np.uint32)
        self.wrv(w, i.args[0], lo | (hi << 16))

    def x_v_cvt_pkrtz_f16_f32(self, w, i):
        """round toward zero: never overflows to infinity (a finite input beyond the f16 range gives +-65504)"""
        out = []
        for k in (1, 2):
            x = self.f(self.rd(w, i.args[k]))
            with np.errstate(all="ignore"):
                h = x.astype(np.float16)
                hb = h.view(np.uint16).astype(np.uint32)
                over = np.abs(h.astype(np.float32)) > np.abs(x)           # rounded away from zero (or to infinity): one step back
                hb = np.where(over & np.isfinite(x), hb - 1, hb)
            out.append(hb & 0xFFFF)
        self.wrv(w, i.args[0], out[0] | (out[1] << 16))

    # ---- packed f16 (VOP3P): lane 0 of source k = its half op_sel[k] (default low), lane 1 = its half op_sel_hi[k] (default high)
    @staticmethod
    def _sel(mods, key, default, n):
        v = mods.get(key)
        return [default] * n if v is None else [int(x) for x in str(v).strip("[]").split(",")]

    def _pk16_src(self, w, i, nsrc):
        lo_sel, hi_sel = self._sel(i.mods, "op_sel", 0, nsrc), self._sel(i.mods, "op_sel_hi", 1, nsrc)
        lanes = ([], [])
        for k in range(nsrc):
            o = i.args[1 + k]
            if isinstance(o, H):
                assert hi_sel[k] == 0 and lo_sel[k] == 0, "an inline f16 constant sits in the LOW half only: op_sel_hi must be 0 for it"
            v = self.rd(w, o)
            for lane, sel in ((0, lo_sel[k]), (1, hi_sel[k])):
                half = ((v >> 16) if sel else v) & 0xFFFF
                lanes[lane].append(half.astype(np.uint16).view(np.float16).astype(np.float64))
        return lanes

    def _pk16_wr(self, w, i, res):
        out = []
        for r in res:
            with np.errstate(all="ignore"):
                h = r.astype(np.float16)
            if i.mods.get("clamp"):
                h = np.where(np.isnan(h), np.float16(0), np.clip(h, np.float16(0), np.float16(1))).astype(np.float16)
            out.append(h.view(np.uint16).astype(np.uint32))
        self.wrv(w, i.args[0], out[0] | (out[1] << 16))

    def x_v_pk_mul_f16(self, w, i):
        lanes = self._pk16_src(w, i, 2)
        with np.errstate(all="ignore"):
            self._pk16_wr(w, i, [l[0] * l[1] for l in lanes])

    def x_v_pk_fma_f16(self, w, i):
        lanes = self._pk16_src(w, i, 3)
        with np.errstate(all="ignore"):
            self._pk16_wr(w, i, [l[0] * l[1] + l[2] for l in lanes])       # (exact in fp64 up to astronomically rare ties, then ONE rounding)

    def x_v_fma_mix_f32(self, w, i):
        """fp32 fma whose source k is fp32 (op_sel_hi[k] = 0) or the f16 half op_sel[k] of its register (op_sel_hi[k] = 1)"""
        sel, is16 = self._sel(i.mods, "op_sel", 0, 3), self._sel(i.mods, "op_sel_hi", 0, 3)
        src = []
        for k in range(3):
            v = self.rd(w, i.args[1 + k])
            if is16[k]:
                src.append((((v >> 16) if sel[k] else v) & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float64))
            else:
                src.append(self.f(v).astype(np.float64))
        with np.errstate(all="ignore"):
            r = (src[0] * src[1] + src[2]).astype(np.float32)
        if i.mods.get("clamp"):
            r = np.where(np.isnan(r), np.float32(0), np.clip(r, 0, 1)).astype(np.float32)
        self.wrv(w, i.args[0], self.u(r))

    def x_v_cvt_f32_f16(self, w, i):
        self.wrv(w, i.args[0], self.u((self.rd(w, i.args[1]) & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)))

    def x_v_accvgpr_read_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]).copy())

    def x_v_accvgpr_write_b32(self, w, i):
        self.wrv(w, i.args[0], self.rd(w, i.args[1]).copy())

    def x_v_permlane32_swap_b32(self, w, i):
        d, s = i.args
        a, b = self.rd(w, d).copy(), self.rd(w, s).copy()
        a2, b2 = a.copy(), b.copy()
        a2[32:] = b[:32]
        b2[:32] = a[32:]
        self.wrv(w, d, a2)
        self.wrv(w, s, b2)

    # ---- MFMA
    def _mfma32(self, w, i, dec):
        d, sa, sb, sc = i.args
        assert d.n == 16 and sa.n == 4 and sb.n == 4
        lane = np.arange(64)

        def frag(r):      # -> [32 rows, 16 k]
            bank = w.v if r.kind == "v" else w.a
            regs = bank[r.idx:r.idx + 4]                                   # [4, 64]
            lo, hi = (regs & 0xFFFF).astype(np.uint16), (regs >> 16).astype(np.uint16)
            el = np.empty((8, 64), np.float32)
            el[0::2] = dec(lo)
            el[1::2] = dec(hi)
            m = np.zeros((32, 16), np.float32)
            for e in range(8):
                m[lane & 31, 8 * (lane >> 5) + e] = el[e]
            return m
        Am, Bm = frag(sa), frag(sb)                                        # A[i][k], B[j][k]
        with np.errstate(all="ignore"):
            P = Am.astype(np.float64) @ Bm.astype(np.float64).T            # [i][j]
        dbank = w.a if d.kind == "a" else w.v
        if isinstance(sc, Reg):
            cbank = w.a if sc.kind == "a" else w.v
            C = cbank[sc.idx:sc.idx + 16].view(np.float32).astype(np.float64)
        else:
            assert sc == 0
            C = np.zeros((16, 64))
        out = np.empty((16, 64), np.float32)
        with np.errstate(all="ignore"):
            for r in range(16):
                row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                out[r] = (C[r] + P[row, lane & 31]).astype(np.float32)
        dbank[d.idx:d.idx + 16] = out.view(np.uint32)

    def x_v_mfma_f32_32x32x16_bf16(self, w, i):
        self._mfma32(w, i, _bf16_to_f32)

    def x_v_mfma_f32_32x32x16_f16(self, w, i):
        self._mfma32(w, i, lambda u: u.view(np.float16).astype(np.float32))

    # ---- LDS
    def _ds_read(self, w, i, nbytes):
        d, a = i.args
        addr = self.rd(w, a).astype(np.int64) + i.mods.get("offset", 0)
        if addr.max() + nbytes > len(self.lds):
            raise RuntimeError("LDS read out of range: %d" % addr.max())
        data = self.lds[addr[:, None] + np.arange(nbytes)[None, :]].copy().view(np.uint32)      # [64, n]
        n = nbytes // 4
        bank = w.v if d.kind == "v" else w.a
        for k in range(n):
            bank[d.idx + k] = POISON

        def done():
            for k in range(n):
                bank[d.idx + k] = data[:, k]
        w.lgkm.append(done)

    def x_ds_read_b128(self, w, i):
        self._ds_read(w, i, 16)

    def x_ds_read_b64(self, w, i):
        self._ds_read(w, i, 8)

    def x_ds_read_b32(self, w, i):
        self._ds_read(w, i, 4)

    def _ds_write(self, w, i, nbytes):
        a, d = i.args
        addr = self.rd(w, a).astype(np.int64) + i.mods.get("offset", 0)
        if addr.max() + nbytes > len(self.lds):
            raise RuntimeError("LDS write out of range: %d" % addr.max())
        n = nbytes // 4
        bank = w.v if d.kind == "v" else w.a
        data = np.stack([bank[d.idx + k] for k in range(n)], axis=1).copy()
        if np.any(data == POISON):
            raise RuntimeError("ds_write of a register that still holds poison (pc %d)" % w.pc)
        self.lds[addr[:, None] + np.arange(nbytes)[None, :]] = data.view(np.uint8).reshape(64, nbytes)
        w.lgkm.append(lambda: None)

    def x_ds_write_b32(self, w, i):
        self._ds_write(w, i, 4)

    def x_ds_write_b64(self, w, i):
        self._ds_write(w, i, 8)

    def x_ds_write_b128(self, w, i):
        self._ds_write(w, i, 16)

    # ---- global
    def _gaddr(self, w, i, voff, sbase):
        off = i.mods.get("offset", 0)
        if sbase == "off":
            assert voff.n == 2
            base = w.v[voff.idx].astype(np.int64) | (w.v[voff.idx + 1].astype(np.int64) << 32)
            return base + off
        return self.rds64(w, sbase) + self.rd(w, voff).astype(np.int64) + off

    def x_global_load_lds_dwordx4(self, w, i):
        voff, sbase = i.args
        # the immediate offset moves BOTH sides (measured on gfx950, tools/ubench/lds_dma_offset.hip: offset N with m0 = b lands the
        # bytes of global address + N at LDS byte b + N, negative N included)
        addr = self._gaddr(w, i, voff, sbase)
        dst = w.m0 + i.mods.get("offset", 0) + 16 * np.arange(64)
        if dst.min() < 0 or dst.max() + 16 > len(self.lds):
            raise RuntimeError("LDS-DMA destination out of range: m0 = %d offset %d" % (w.m0, i.mods.get("offset", 0)))

        def land():
            self.lds[dst[:, None] + np.arange(16)[None, :]] = self.mem.gather(addr, 16)
        if self.dma_mode == "early":
            land()
            w.vm.append(("dma", None))
        else:
            w.vm.append(("dma", land))

    def _gload(self, w, i, n):
        d, voff, sbase = i.args
        addr = self._gaddr(w, i, voff, sbase)
        ex = w.exec.copy()
        data = np.zeros((64, n), np.uint32)
        if ex.any():
            data[ex] = self.mem.gather(addr[ex], 4 * n).copy().view(np.uint32)
        bank = w.v if d.kind == "v" else w.a
        for k in range(n):
            bank[d.idx + k][ex] = POISON

        def done():
            for k in range(n):
                bank[d.idx + k][ex] = data[ex, k]
        w.vm.append(("load", done))

    def x_global_load_dword(self, w, i):
        self._gload(w, i, 1)

    def x_global_load_dwordx2(self, w, i):
        self._gload(w, i, 2)

    def x_global_load_dwordx4(self, w, i):
        self._gload(w, i, 4)

    def _gstore(self, w, i, n):
        voff, d, sbase = i.args
        addr = self._gaddr(w, i, voff, sbase)
        bank = w.v if d.kind == "v" else w.a
        data = np.stack([bank[d.idx + k] for k in range(n)], axis=1).copy()           # [64, n]
        if np.any(data[w.exec] == POISON):
            raise RuntimeError("store of a register that still holds poison (pc %d)" % w.pc)
        self.mem.scatter(addr[w.exec], data.view(np.uint8).reshape(64, 4 * n)[w.exec])
        w.vm.append(("store", None))

    def x_global_store_dwordx4(self, w, i):
        self._gstore(w, i, 4)

    def x_global_store_dwordx2(self, w, i):
        self._gstore(w, i, 2)

    def x_global_store_dword(self, w, i):
        self._gstore(w, i, 1)


# ------------------------------------------------------------------ hazard lint
def _regs_of(o):
    if isinstance(o, (Neg, Abs)):
        o = o.r
    if isinstance(o, Reg):
        return {(o.kind, o.idx + k) for k in range(o.n)}
    if o == "m0":
        return {("m0", 0)}
    return set()


_NO_DST = ("ds_write_b32", "ds_write_b64", "ds_write_b128", "s_waitcnt", "s_barrier", "s_nop", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_endpgm", "s_setprio", "label",
           "global_load_lds_dwordx4", "global_store_dwordx4", "global_store_dwordx2", "global_store_dword",
           "s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_lt_u32", "s_cmp_le_u32", "s_cmp_gt_u32", "s_cmp_ge_u32", "s_cmp_lt_i32", "s_cmp_gt_i32")


def defs_uses(i):
    if i.op in _NO_DST:
        d, u = set(), set()
        for a in i.args:
            u |= _regs_of(a)
        if i.op == "global_load_lds_dwordx4":
            u.add(("m0", 0))
        return d, u
    d = _regs_of(i.args[0]) if i.args else set()
    u = set()
    for a in i.args[1:]:
        u |= _regs_of(a)
    if i.op in ("v_fmac_f32", "v_dot2c_f32_bf16", "v_dot2c_f32_f16"):
        u |= d
    if i.op == "v_permlane32_swap_b32":
        d |= _regs_of(i.args[1])
        u |= d
    return d, u


TRANS_OPS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")


def lint(asm, mfma_gap=16, verbose=False):
    """Distances are counted in issued instructions along the LINEAR listing (s_nop N = N + 1), which is what the
    generator's straight-line bodies need; loop back-edges are covered by the padding the generator puts at loop heads."""
    problems = []
    last_def = {}          # reg -> (position in wait states, kind of producer)
    pos = 0
    for k, i in enumerate(asm.ins):
        if i.op == "label":
            continue
        for o in i.args:
            if isinstance(o, (Neg, Abs)):
                o = o.r
            if isinstance(o, Reg) and o.n >= 2 and o.kind in ("v", "a", "s") and o.idx % 2:
                problems.append((k, "register tuple %s is not 64-bit aligned" % o.text()))
        if i.op.startswith("ds_") and not 0 <= i.mods.get("offset", 0) < 65536:
            problems.append((k, "DS offset %d does not fit 16 bits" % i.mods["offset"]))
        if i.op.startswith("global_") and not -4096 <= i.mods.get("offset", 0) < 4096:
            problems.append((k, "global offset %d does not fit 13 signed bits" % i.mods["offset"]))
        d, u = defs_uses(i)
        is_mfma = i.op.startswith("v_mfma")
        is_valu = i.op.startswith("v_") and not is_mfma
        for r in u:
            if r not in last_def:
                continue
            p, kind = last_def[r]
            dist = pos - p - 1         # wait states in between
            if kind == "mfma" and not is_mfma and dist < mfma_gap:
                problems.append((k, "%s reads %s%d %d states after the MFMA that writes it (< %d)" % (i.op, r[0], r[1], dist, mfma_gap)))
            if kind == "mfma" and is_mfma and r[0] != "a" and dist < mfma_gap and not (
                    isinstance(i.args[3], Reg) and i.args[3].kind == i.args[0].kind and i.args[3].idx == i.args[0].idx
                    and r in _regs_of(i.args[3]) and r not in _regs_of(i.args[1]) and r not in _regs_of(i.args[2])):
                problems.append((k, "MFMA operand written by an MFMA %d states before" % dist))
            if r == ("m0", 0) and i.op == "global_load_lds_dwordx4" and dist < 1:
                problems.append((k, "LDS-DMA straight after the m0 write"))
            if i.op.endswith("_dpp") and kind in ("valu", "dot") and dist < 2 and r in _regs_of(i.args[1]):
                problems.append((k, "DPP source %s%d written by a VALU instruction %d states before (< 2)" % (r[0], r[1], dist)))
            if kind == "dot" and is_valu and not i.op.startswith("v_dot2c") and dist < 3:
                problems.append((k, "%s reads the dot-product result %s%d %d states after it was written (< 3)" % (i.op, r[0], r[1], dist)))
            if i.op == "v_permlane32_swap_b32" and kind == "valu" and dist < 2:
                problems.append((k, "v_permlane32_swap %d states after a VALU write of %s%d (< 2)" % (dist, r[0], r[1])))
            if kind == "valu_sgpr" and (i.op.startswith("global_") or i.op.startswith("s_load")) and dist < 5:
                problems.append((k, "VMEM/SMEM reads s%d %d states after a VALU wrote it (< 5)" % (r[1], dist)))
            if kind == "trans" and is_valu and i.op not in TRANS_OPS and dist < 1:
                # gfx940+ trans forwarding hazard: a non-transcendental VALU instruction reads a transcendental's result >= 1 state later
                problems.append((k, "%s reads the transcendental result %s%d straight after it was written" % (i.op, r[0], r[1])))
            if kind == "pk16" and is_valu and dist < 1:
                # gfx940+ destination-forwarding hazard of the 16-bit packed (VOP3P) results: hipcc pads a dependent v_pk_fma_f16 with s_nop 0
                problems.append((k, "%s reads the packed-f16 result %s%d straight after it was written" % (i.op, r[0], r[1])))
            if kind in ("valu", "trans", "pk16") and is_mfma and dist < 2:
                problems.append((k, "MFMA reads %s%d %d states after a VALU write (< 2)" % (r[0], r[1], dist)))
        for r in d:
            if r in last_def:
                p, kind = last_def[r]
                dist = pos - p - 1
                if kind == "mfma" and not is_mfma and dist < mfma_gap:
                    problems.append((k, "%s overwrites %s%d %d states after the MFMA that writes it" % (i.op, r[0], r[1], dist)))
                if kind == "store4" and dist < 2:
                    problems.append((k, "%s overwrites store data %s%d %d states after the wide store (< 2)" % (i.op, r[0], r[1], dist)))
        for r in d:
            if is_mfma:
                last_def[r] = (pos, "mfma")
            elif i.op == "v_readfirstlane_b32":
                last_def[r] = (pos, "valu_sgpr")
            elif i.op.startswith("v_dot2c"):
                last_def[r] = (pos, "dot")
            elif i.op in TRANS_OPS:
                last_def[r] = (pos, "trans")
            elif i.op in ("v_pk_fma_f16", "v_pk_mul_f16", "v_pk_add_f16", "v_pk_max_f16", "v_pk_min_f16", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32"):
                last_def[r] = (pos, "pk16")
            elif is_valu:
                last_def[r] = (pos, "valu")
            else:
                last_def[r] = (pos, "other")
        if i.op == "global_store_dwordx4":
            for r in _regs_of(i.args[1]):
                last_def[r] = (pos, "store4")
        pos += (i.args[0] + 1) if i.op == "s_nop" else 1
    if verbose:
        for k, msg in problems[:50]:
            print("lint @%d: %s   [%s]" % (k, msg, asm.ins[k].text()))
    return problems
