"""Generator of the "q4" NT GEMM: one wave per SIMD, the previous tile's epilogue issued as fillers behind the MFMAs.

    C[m, n] = epi( sum_k A[m, k] * B[n, k] )      A: M x K, B: N x K, K-contiguous, 16-bit; C 16-bit row-major

Why this shape: on gfx950 the VALU work of one wave does not overlap the MFMAs of ANOTHER wave of
the same SIMD (tools/ubench/issue_rate.hip), but in ONE wave's own stream five plain VALU / LDS / SALU instructions issue for
free behind every v_mfma_f32_32x32x16 (profiles/r02_mfma_valu_mix.txt).  The persistent 8-wave tile (gemm_nt_p8_kernel) ran its
epilogue -- bias, folded LayerNorm, GELU, convert, store: as long as the 12 K-slabs of Mixer-B fc1 -- with the matrix pipe idle.
Here a workgroup is 4 waves (one per SIMD, 512 registers each) on a 256 x 128 tile; a wave owns 128 x 64 = 8 blocks of 32 x 32
in ONE of two accumulator sets (a[0:127] / a[128:255]) and, while it multiplies tile T into one set, drains tile T - 1 from the
other: v_accvgpr_read, the element math in scalar fp32, v_cvt_pk, v_permlane32_swap, 16-byte stores -- all placed by this
program between the MFMAs, at a fixed number per MFMA.  Nothing is left to a compiler: the kernel body is one asm block with
every register named here, the same instruction list runs on the numpy emulator of isa.py (tests/test_q4_emulated.py).

Pipeline of one wave (slab = 64 k = one 128-byte row per tile row; 3 LDS stages of 48 KiB: A 256 rows, B 128 rows):
    iteration t:   MFMA k-steps  (slab t-1, step 3), (slab t, steps 0 1 2)      8 MFMAs each, fragment buffer alternating
                   ds_read       fragments of slab t steps 0..3, one step ahead of their MFMAs
                   LDS-DMA       the wave's 12 pieces (8 A + 4 B, 1 KiB each) of slab t+2 into stage (t+2) % 3
                   s_waitcnt vmcnt(12) lgkmcnt(0) ; s_barrier        -> slab t+1 landed for everyone, slab t fully read
A tile's K loop is `nkf` unrolled iterations carrying the fillers followed by a rolled loop of plain iterations; the stream of
slabs runs on across tiles (the DMA is always two slabs ahead), so the pipeline never drains inside a launch.
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa import A, F, S, V, Asm, Neg, Reg  # noqa: E402

GELU_SCALE = 0.314269681
GELU_COEFS = [0.00260713836, -0.00718860654, 0.00979797821, -0.0172248576, 0.0355015062, -0.0601866171, 0.090279378, -0.127707109,
              0.174028099, -0.245624334, 0.499268919]
SQRT2 = 1.41421356237

STAGE_B = 49152          # one LDS stage: A 256 x 128 B, then B 128 x 128 B
B_OFF = 32768
LDS_BYTES = 3 * STAGE_B

# kernarg layout (bytes) -- mirrored by struct Q4Args in mlpk_gemm_q4.hip
KA = dict(A=0, B=8, C=16, R=24, bias=32, ln_mean=40, ln_rstd=48, ln_csum=56,
          lda=64, ldb=68, ldc=72, ldr=76, nk=80, cg=84, cg_magic=88, U=92, Q=96, log2X=100, m_base=104, grid=108, prof=112)


class Alloc:
    def __init__(self, kind, start, limit):
        self.kind, self.next, self.limit = kind, start, limit
        self.names = {}

    def __call__(self, name, n=1, align=1):
        self.next = (self.next + align - 1) // align * align
        r = Reg(self.kind, self.next, n)
        self.next += n
        assert self.next <= self.limit, "out of %s registers at %s" % (self.kind, name)
        self.names[name] = r
        return r


class Q4:
    def __init__(self, dtype="bf16", gelu=False, ln=False, res=False, nkf=4, dbg=0, name=None):
        assert nkf >= 2
        self.dtype, self.gelu, self.ln, self.res, self.nkf = dtype, gelu, ln, res, nkf
        # tuning ablations (results are wrong by construction): 1 no LDS-DMA, 2 no stores, 4 no epilogue fillers, 8 no fragment reads,
        # 16 minimal iteration tail (no stage rotation)
        self.dbg = dbg
        self.fillers_on, self.dma_on, self.stores_on, self.reads_on, self.tail_on = not (dbg & 4), not (dbg & 1), not (dbg & 2), not (dbg & 8), not (dbg & 16)
        self.name = name or "q4_%s%s%s%s_f%d" % (dtype, "_gelu" if gelu else "", "_ln" if ln else "", "_res" if res else "", nkf)
        self.a = Asm()
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.build()

    # ------------------------------------------------------------------ registers
    def regs(self):
        s = Alloc("s", 33, 96)          # (s32 is the ABI stack pointer: hipcc warns when an asm block clobbers it)
        v = Alloc("v", 1, 248)
        self.s_karg, self.s_bid, self.v_tid = S(0, 2), S(2), V(0)
        self.p = {k: S(4 + 2 * i, 2) for i, k in enumerate(["A", "B", "C", "R", "bias", "ln_mean", "ln_rstd", "ln_csum"])}
        self.k = {k: S(20 + i) for i, k in enumerate(["lda", "ldb", "ldc", "ldr", "nk", "cg", "cg_magic", "U", "Q", "log2X", "m_base", "grid"])}
        # work list
        self.s_wave, self.s_u0, self.s_lend, self.s_ncol0, self.s_lstep = s("wave"), s("u0"), s("lend"), s("ncol0"), s("lstep")
        self.s_lnext = s("lnext")              # l of the tile whose coordinates were computed last
        self.s_left = s("left")                # blocks still to run (tiles after the first + the draining one)
        self.s_roll = s("roll")                # rolled iterations of the current block
        self.s_cnt = s("cnt")
        # tile coordinates: p = the tile being drained, c = the tile being multiplied, n = the next one (DMA target after c)
        self.s_pm0, self.s_pn0, self.s_cm0, self.s_cn0, self.s_nm0, self.s_nn0 = (s(x) for x in ["pm0", "pn0", "cm0", "cn0", "nm0", "nn0"])
        # DMA stream
        self.s_dA, self.s_dB = s("dA", 2, 2), s("dB", 2, 2)
        self.s_dAn, self.s_dBn = s("dAn", 2, 2), s("dBn", 2, 2)
        self.s_dcnt = s("dcnt")
        self.s_rd, self.s_wr = s("rd"), s("wr")            # LDS byte offsets of the stage read / written in this iteration
        self.s_wrA, self.s_wrB = s("wrA"), s("wrB")        # + this wave's share
        self.s_wvA, self.s_wvB = s("wvA"), s("wvB")        # wave * 8192, 32768 + wave * 4096
        # epilogue bases of tile p
        self.s_eC, self.s_eR = s("eC", 2, 2), s("eR", 2, 2)
        self.s_eBias, self.s_eCsum, self.s_eMu, self.s_eRstd = s("eBias", 2, 2), s("eCsum", 2, 2), s("eMu", 2, 2), s("eRstd", 2, 2)
        self.s_r2 = s("r2")
        self.s_prof0, self.s_prof1, self.s_profp, self.s_ntiles = s("prof0", 2, 2), s("prof1", 2, 2), s("profp", 2, 2), s("ntiles")
        self.s_t = [s("t%d" % i) for i in range(6)]
        self.s_t64 = s("t64", 2, 2)
        # vector registers
        self.FA = [[v("FA%d_%d" % (b, i), 4, 4) for i in range(4)] for b in range(2)]
        self.FB = [[v("FB%d_%d" % (b, j), 4, 4) for j in range(2)] for b in range(2)]
        self.v_curA = [v("curA%d" % k) for k in range(4)]
        self.v_curB = [v("curB%d" % k) for k in range(4)]
        self.v_rdA0 = [v("rdA0_%d" % k) for k in range(4)]
        self.v_rdB0 = [v("rdB0_%d" % k) for k in range(4)]
        self.voffA = [v("voffA%d" % k) for k in range(8)]
        self.voffB = [v("voffB%d" % k) for k in range(4)]
        self.voffC = [v("voffC%d" % i) for i in range(4)]
        self.voffR = [v("voffR%d" % i) for i in range(4)]
        self.voffCol, self.voffRow = v("voffCol"), v("voffRow")
        self.v_bias = [[v("bias%d_%d" % (j, g), 4, 4) for g in range(4)] for j in range(2)]
        self.v_csum = [[v("csum%d_%d" % (j, g), 4, 4) for g in range(4)] for j in range(2)] if self.ln else None
        self.v_mu = [v("mu%d" % i) for i in range(4)] if self.ln else None
        self.v_rstd = [v("rstd%d" % i) for i in range(4)] if self.ln else None
        self.v_c0 = v("c0")
        self.v_x = [[v("x%d_%d" % (b, r)) for r in range(4)] for b in range(2)]
        self.v_t = [[v("t%d_%d" % (b, r)) for r in range(4)] for b in range(2)]
        self.v_u = [[v("u%d_%d" % (b, r)) for r in range(4)] for b in range(2)]
        self.v_q = [[v("q%d_%d" % (b, r)) for r in range(4)] for b in range(2)]
        self.v_quad = [v("quad%d" % k, 4, 4) for k in range(2)]
        self.v_res = [[[v("res%d_%d_%d" % (i, j, h), 4, 4) for h in range(2)] for j in range(2)] for i in range(4)] if self.res else None
        self.v_tmp = [v("tmp%d" % i) for i in range(8)]
        self.v_pair = v("pair", 2, 2)
        self.nv, self.ns = v.next, s.next

    def acc(self, set_, b):
        return A(128 * set_ + 16 * b, 16)

    # ------------------------------------------------------------------ helpers
    def add64(self, dst, src, lo, hi=None):
        """dst(64) = src(64) + (hi:lo); hi None = 0"""
        a = self.a
        a("s_add_u32", dst[0], src[0], lo)
        a("s_addc_u32", dst[1], src[1], hi if hi is not None else 0)

    def coords(self, l, m0, n0):
        """(m0, n0) of this workgroup's tile l (clamped to its last tile: the draining block re-reads valid memory)"""
        a, t = self.a, self.s_t
        a("s_sub_u32", t[0], self.s_lend, 1)
        a("s_min_u32", t[0], l, t[0])
        a("s_add_u32", t[0], t[0], self.s_u0)                  # u
        a("s_lshl_b32", t[1], t[0], 1)
        a("s_mul_hi_u32", t[1], t[1], self.k["cg_magic"])      # panel = u / cg = (2 u * ceil(2^31 / cg)) >> 32
        a("s_mul_i32", t[2], t[1], self.k["cg"])
        a("s_sub_u32", t[0], t[0], t[2])
        a("s_add_u32", t[0], t[0], self.s_ncol0)               # column tile
        a("s_lshl_b32", t[1], t[1], 8)
        a("s_add_u32", m0, t[1], self.k["m_base"])
        a("s_lshl_b32", n0, t[0], 7)

    def dma_base(self, dA, dB, m0, n0):
        """dA = A + m0 * lda * 2, dB = B + n0 * ldb * 2 (64-bit)"""
        a, t = self.a, self.s_t
        for d, ptr, r0, ld in ((dA, self.p["A"], m0, self.k["lda"]), (dB, self.p["B"], n0, self.k["ldb"])):
            a("s_mul_i32", t[0], r0, ld)
            a("s_mul_hi_u32", t[1], r0, ld)
            a("s_lshl_b32", t[1], t[1], 1)
            a("s_lshr_b32", t[2], t[0], 31)
            a("s_or_b32", t[1], t[1], t[2])
            a("s_lshl_b32", t[0], t[0], 1)
            self.add64(d, ptr, t[0], t[1])

    def epi_bases_ops(self):
        """SALU ops (a list of closures, one instruction each) that set the epilogue bases from (pm0, pn0)"""
        a, t = self.a, self.s_t
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))
        for dst, ptr, ld, on in ((self.s_eC, self.p["C"], self.k["ldc"], True), (self.s_eR, self.p["R"], self.k["ldr"], self.res)):
            if not on:
                continue
            # (pm0 * ld + pn0) * 2
            E("s_mul_i32", t[0], self.s_pm0, ld)
            E("s_mul_hi_u32", t[1], self.s_pm0, ld)
            E("s_add_u32", t[0], t[0], self.s_pn0)
            E("s_addc_u32", t[1], t[1], 0)
            E("s_lshl_b32", t[1], t[1], 1)
            E("s_lshr_b32", t[2], t[0], 31)
            E("s_or_b32", t[1], t[1], t[2])
            E("s_lshl_b32", t[0], t[0], 1)
            E("s_add_u32", dst[0], ptr[0], t[0])
            E("s_addc_u32", dst[1], ptr[1], t[1])
        E("s_lshl_b32", t[0], self.s_pn0, 2)
        E("s_add_u32", self.s_eBias[0], self.p["bias"][0], t[0])
        E("s_addc_u32", self.s_eBias[1], self.p["bias"][1], 0)
        if self.ln:
            E("s_add_u32", self.s_eCsum[0], self.p["ln_csum"][0], t[0])
            E("s_addc_u32", self.s_eCsum[1], self.p["ln_csum"][1], 0)
            E("s_lshl_b32", t[0], self.s_pm0, 2)
            E("s_add_u32", self.s_eMu[0], self.p["ln_mean"][0], t[0])
            E("s_addc_u32", self.s_eMu[1], self.p["ln_mean"][1], 0)
            E("s_add_u32", self.s_eRstd[0], self.p["ln_rstd"][0], t[0])
            E("s_addc_u32", self.s_eRstd[1], self.p["ln_rstd"][1], 0)
        return ops

    def block_start_ops(self):
        """shift the tile coordinates, compute the next tile's and its DMA bases, the epilogue bases of the drained tile"""
        a = self.a
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))
        E("s_mov_b32", self.s_pm0, self.s_cm0)
        E("s_mov_b32", self.s_pn0, self.s_cn0)
        E("s_mov_b32", self.s_cm0, self.s_nm0)
        E("s_mov_b32", self.s_cn0, self.s_nn0)
        ops += self.epi_bases_ops()
        if self.fillers_on:
            ops += self.param_load_ops()
        E("s_add_u32", self.s_lnext, self.s_lnext, self.s_lstep)
        n0 = len(a.ins)
        self.coords(self.s_lnext, self.s_nm0, self.s_nn0)
        self.dma_base(self.s_dAn, self.s_dBn, self.s_nm0, self.s_nn0)
        captured = a.ins[n0:]
        del a.ins[n0:]
        for ins in captured:
            ops.append(lambda ins=ins: a.ins.append(ins))
        # rolled iterations of this block: nk - nkf, or 0 for the draining block (no block left after it)
        t = self.s_t
        E("s_sub_u32", t[0], self.k["nk"], self.nkf)
        E("s_cmp_lg_u32", self.s_left, 0)
        E("s_cselect_b32", self.s_roll, t[0], 0)
        return ops

    def param_load_ops(self):
        """the column / row parameters (and the residual tile) of the drained tile, as loads"""
        a = self.a
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))
        for j in range(2):
            for g in range(4):
                E("global_load_dwordx4", self.v_bias[j][g], self.voffCol, self.s_eBias, offset=(j * 32 + g * 8) * 4)
                if self.ln:
                    E("global_load_dwordx4", self.v_csum[j][g], self.voffCol, self.s_eCsum, offset=(j * 32 + g * 8) * 4)
        if self.ln:
            for i in range(4):
                E("global_load_dword", self.v_mu[i], self.voffRow, self.s_eMu, offset=i * 128)
                E("global_load_dword", self.v_rstd[i], self.voffRow, self.s_eRstd, offset=i * 128)
        if self.res:
            for i in range(4):
                for j in range(2):
                    for h in range(2):
                        E("global_load_dwordx4", self.v_res[i][j][h], self.voffR[i], self.s_eR, offset=(j * 32 + h * 16) * 2)
        return ops

    def n_param_loads(self):
        return 8 * (2 if self.ln else 1) + (8 if self.ln else 0) + (16 if self.res else 0)

    def gelu_ops(self, E, x, t, u, q):
        """x[r] <- gelu(x[r]) for the 4 chains abreast (the operation sequence of gelu16_f in mlpk_common.h)"""
        c = GELU_COEFS
        for r in range(4):
            E("v_mul_f32", t[r], F(GELU_SCALE), x[r])
        for r in range(4):
            E("v_med3_f32", t[r], t[r], Neg(self.s_r2), self.s_r2)
        for r in range(4):
            E("v_fma_f32", u[r], t[r], t[r], F(-1.0))
        for r in range(4):
            E("v_fmaak_f32", q[r], u[r], self.v_c0, F(c[1]))
        for k in range(2, 11):
            for r in range(4):
                E("v_fmaak_f32", q[r], q[r], u[r], F(c[k]))
        for r in range(4):
            E("v_fma_f32", t[r], t[r], q[r], F(0.5))
        for r in range(4):
            E("v_mul_f32", x[r], x[r], t[r])

    def epilogue_ops(self, set_):
        """the drain of accumulator set `set_` as a flat list of one-instruction closures"""
        a = self.a
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))
        bank = 0
        for b in range(8):
            i, j = b >> 1, b & 1
            for g in range(4):
                x, t, u, q = self.v_x[bank], self.v_t[bank], self.v_u[bank], self.v_q[bank]
                bank ^= 1
                quad = self.v_quad[(b * 2 + (g >> 1)) & 1]
                for r in range(4):
                    E("v_accvgpr_read_b32", x[r], A(128 * set_ + 16 * b + 4 * g + r))
                if self.ln:
                    for r in range(4):
                        E("v_fma_f32", x[r], Neg(self.v_mu[i]), self.v_csum[j][g][r], x[r])
                    for r in range(4):
                        E("v_fma_f32", x[r], x[r], self.v_rstd[i], self.v_bias[j][g][r])
                else:
                    for r in range(4):
                        E("v_add_f32", x[r], x[r], self.v_bias[j][g][r])
                if self.gelu:
                    self.gelu_ops(E, x, t, u, q)
                E(self.cvt, quad[2 * (g & 1)], x[0], x[1])
                E(self.cvt, quad[2 * (g & 1) + 1], x[2], x[3])
                if g & 1:
                    # groups g-1, g packed in quad: exchange halves so that every lane holds 8 consecutive columns
                    E("s_nop", 1)          # 2 wait states between the v_cvt_pk that wrote quad[3] and the swaps
                    E("v_permlane32_swap_b32", quad[0], quad[2])
                    E("v_permlane32_swap_b32", quad[1], quad[3])
                    if self.res:
                        rr = self.v_res[i][j][g >> 1]
                        tm = self.v_tmp
                        for k in range(4):
                            if self.dtype == "bf16":
                                E("v_lshlrev_b32", tm[0], 16, quad[k])
                                E("v_and_b32", tm[1], 0xFFFF0000, quad[k])
                                E("v_lshlrev_b32", tm[2], 16, rr[k])
                                E("v_and_b32", tm[3], 0xFFFF0000, rr[k])
                            else:
                                E("v_lshrrev_b32", tm[1], 16, quad[k])
                                E("v_lshrrev_b32", tm[3], 16, rr[k])
                                E("v_cvt_f32_f16", tm[0], quad[k])
                                E("v_cvt_f32_f16", tm[2], rr[k])
                                E("v_cvt_f32_f16", tm[1], tm[1])
                                E("v_cvt_f32_f16", tm[3], tm[3])
                            E("v_add_f32", tm[0], tm[0], tm[2])
                            E("v_add_f32", tm[1], tm[1], tm[3])
                            E(self.cvt, quad[k], tm[0], tm[1])
                    if self.stores_on:
                        E("global_store_dwordx4", self.voffC[i], quad, self.s_eC, offset=(j * 32 + (g >> 1) * 16) * 2)
        return ops

    # ------------------------------------------------------------------ one iteration of the K loop
    def iteration(self, set_first, set_main, zero_c, fill, extra_loads=0, head_ops=None):
        """Four k-steps of 8 MFMAs with the loads of this iteration between them.
        set_first / set_main: accumulator set of step 0 / steps 1..3; zero_c: step 1 starts the accumulators (C = 0);
        fill(n): emit up to n filler instructions; extra_loads: VGPR loads among the fillers of this iteration (they count in
        the vmcnt in front of the barrier)."""
        a = self.a
        nread = 0           # ds_reads issued so far in this iteration; 6 from the previous iteration are outstanding at entry
        ready = {}          # fragment -> index of its read
        prev = {("B", 0): -6, ("B", 1): -5, ("A", 0): -4, ("A", 1): -3, ("A", 2): -2, ("A", 3): -1}

        def need(frag, idx_of):
            # wait until read number idx is complete: allow (issued - idx - 1) younger reads outstanding
            if not self.reads_on:
                return
            n = nread - idx_of[frag] - 1
            a("s_waitcnt", lgkmcnt=n)
        dma_slots = {(0, 1): ("A", 0), (0, 3): ("A", 1), (0, 6): ("A", 2), (0, 7): ("A", 3),
                     (1, 1): ("A", 4), (1, 3): ("A", 5), (1, 6): ("A", 6), (1, 7): ("A", 7),
                     (2, 6): ("B", 0), (2, 7): ("B", 1), (3, 6): ("B", 2), (3, 7): ("B", 3)}
        head = list(head_ops or [])
        i_start = len(a.ins)
        for s in range(4):
            mbuf, rbuf = (s + 1) & 1, s & 1
            cur = prev if s == 0 else ready
            acc_set = set_first if s == 0 else set_main
            read_order = [("B", 0), ("B", 1), ("A", 0), ("A", 1), ("A", 2), ("A", 3)]
            new_ready = {}
            for qm in range(8):
                i, j = qm >> 1, qm & 1
                if j == 0:
                    # first use of A_i (and of both B fragments when i == 0)
                    if i == 0:
                        need(("A", 0), cur)       # B0, B1, A0 were read before A0
                    else:
                        need(("A", i), cur)
                d = self.acc(acc_set, 2 * i + j)
                slot = dma_slots.get((s, qm))
                if slot and self.dma_on:
                    kind, pc = slot
                    a("s_add_u32", "m0", self.s_wrA if kind == "A" else self.s_wrB, pc * 1024)
                csrc = 0 if (zero_c and s == 1) else d
                a(self.mfma, d, self.FB[mbuf][j], self.FA[mbuf][i], csrc)
                if slot and self.dma_on:
                    kind, pc = slot
                    if kind == "A":
                        a("global_load_lds_dwordx4", self.voffA[pc], self.s_dA)
                    else:
                        a("global_load_lds_dwordx4", self.voffB[pc], self.s_dB)
                if qm < 6 and self.reads_on:
                    kind, idx = read_order[qm]
                    if kind == "A":
                        a("ds_read_b128", self.FA[rbuf][idx], self.v_curA[s], offset=idx * 4096)
                    else:
                        a("ds_read_b128", self.FB[rbuf][idx], self.v_curB[s], offset=idx * 4096)
                    new_ready[(kind, idx)] = nread
                    nread += 1
                # fillers
                if head:
                    for _ in range(4):
                        if head:
                            head.pop(0)()
                else:
                    fill(s, qm)
            ready = new_ready
        while head:
            head.pop(0)()
        # vmcnt in front of the barrier: the pieces of the PREVIOUS iteration must have landed (<= 12 loads of this one may be
        # in flight: loads return in order, stores may not be counted on), and so must every VGPR load issued in this one
        kinds = [("load" if x.op.startswith("global_load_dword") else "dma") for x in a.ins[i_start:]
                 if x.op.startswith("global_load")]
        allow = len(kinds)
        if "load" in kinds:
            allow = len(kinds) - 1 - max(k for k, x in enumerate(kinds) if x == "load")
        return min(allow, 12 if self.dma_on else 0)

    def iter_tail(self, vm_allow):
        """advance the DMA stream and the LDS stages, then wait + barrier"""
        a, t = self.a, self.s_t
        if not self.tail_on:
            a("s_waitcnt", vmcnt=vm_allow, lgkmcnt=0)
            a("s_barrier")
            return
        if self.dma_on:
            self.add64(self.s_dA, self.s_dA, 128)
            self.add64(self.s_dB, self.s_dB, 128)
            a("s_sub_u32", self.s_dcnt, self.s_dcnt, 1)
            a("s_cmp_eq_u32", self.s_dcnt, 0)
            a("s_cselect_b32", self.s_dA[0], self.s_dAn[0], self.s_dA[0])
            a("s_cselect_b32", self.s_dA[1], self.s_dAn[1], self.s_dA[1])
            a("s_cselect_b32", self.s_dB[0], self.s_dBn[0], self.s_dB[0])
            a("s_cselect_b32", self.s_dB[1], self.s_dBn[1], self.s_dB[1])
            a("s_cselect_b32", self.s_dcnt, self.k["nk"], self.s_dcnt)
        # stages: rd <- rd + 1, wr <- wr + 1 (mod 3)
        for r in (self.s_rd, self.s_wr):
            a("s_add_u32", r, r, STAGE_B)
            a("s_cmp_ge_u32", r, 3 * STAGE_B)
            a("s_cselect_b32", t[0], 3 * STAGE_B, 0)
            a("s_sub_u32", r, r, t[0])
        a("s_add_u32", self.s_wrA, self.s_wr, self.s_wvA)
        a("s_add_u32", self.s_wrB, self.s_wr, self.s_wvB)
        for k in range(4):
            a("v_add_u32", self.v_curA[k], self.s_rd, self.v_rdA0[k])
            a("v_add_u32", self.v_curB[k], self.s_rd, self.v_rdB0[k])
        a("s_waitcnt", vmcnt=vm_allow, lgkmcnt=0)
        a("s_barrier")

    # ------------------------------------------------------------------ the kernel
    def build(self):
        a = self.a
        self.regs()
        t = self.s_t
        k, p = self.k, self.p
        L_end = a.newlabel("END")
        # ---- arguments
        a("s_load_dwordx16", S(4, 16), self.s_karg, 0)
        a("s_load_dwordx8", S(20, 8), self.s_karg, 64)
        a("s_load_dwordx4", S(28, 4), self.s_karg, 96)
        a("s_load_dwordx2", self.s_profp, self.s_karg, KA["prof"])
        # ---- lane constants (independent of the arguments)
        vt = self.v_tmp
        lane, l31, h, l3, l7, x = vt[0], vt[1], vt[2], vt[3], vt[4], vt[5]
        a("v_and_b32", lane, 63, self.v_tid)
        a("v_lshrrev_b32", vt[6], 6, self.v_tid)
        a("s_nop", 0)
        a("v_readfirstlane_b32", self.s_wave, vt[6])
        a("v_and_b32", l31, 31, lane)
        a("v_lshrrev_b32", h, 5, lane)
        a("v_lshrrev_b32", l3, 3, lane)
        a("v_and_b32", l7, 7, lane)
        a("v_bfe_u32", x, lane, 4, 1)
        a("v_xor_b32", x, x, l7)                   # read swizzle (lane & 7) ^ ((lane >> 4) & 1)
        a("v_xor_b32", x, x, h)                    # ^ k-half of the lane
        a("s_mov_b32", self.s_r2, F(SQRT2))
        a("v_mov_b32", self.v_c0, F(GELU_COEFS[0]))
        a("s_waitcnt", lgkmcnt=0)
        a("s_memtime", self.s_prof0)
        # wave position: wm = wave >> 1, wn = wave & 1
        wm, wn = t[3], t[4]
        a("s_lshr_b32", wm, self.s_wave, 1)
        a("s_and_b32", wn, self.s_wave, 1)
        a("s_lshl_b32", self.s_wvA, self.s_wave, 13)
        a("s_lshl_b32", self.s_wvB, self.s_wave, 12)
        a("s_add_u32", self.s_wvB, self.s_wvB, B_OFF)
        # fragment read addresses in stage 0: row * 128 + ((2 ks) ^ x) * 16
        a("s_lshl_b32", t[0], wm, 7)               # wm * 128 rows
        a("v_add_u32", vt[6], t[0], l31)
        a("v_lshlrev_b32", vt[6], 7, vt[6])        # A row byte offset
        a("s_lshl_b32", t[0], wn, 6)
        a("v_add_u32", vt[7], t[0], l31)
        a("v_lshlrev_b32", vt[7], 7, vt[7])
        a("v_add_u32", vt[7], B_OFF, vt[7])        # B row byte offset
        for ks in range(4):
            a("v_xor_b32", lane, 2 * ks, x)        # (lane is free from here on)
            a("v_lshlrev_b32", lane, 4, lane)
            a("v_add_u32", self.v_rdA0[ks], vt[6], lane)
            a("v_add_u32", self.v_rdB0[ks], vt[7], lane)
        # DMA source offsets: ((rows0 + 8 p + l3) * ld + ((l7 ^ l3 ^ ((p >> 1) & 1)) * 8)) * 2
        a("v_xor_b32", vt[6], l7, l3)
        for arr, npc, rows_shift, ld in ((self.voffA, 8, 6, k["lda"]), (self.voffB, 4, 5, k["ldb"])):
            a("s_lshl_b32", t[0], self.s_wave, rows_shift)          # first row of this wave's share
            for pc in range(npc):
                a("v_add_u32", vt[7], t[0], l3)
                a("v_add_u32", vt[7], 8 * pc, vt[7])
                a("v_mul_lo_u32", vt[7], vt[7], ld)
                a("v_xor_b32", lane, (pc >> 1) & 1, vt[6])
                a("v_lshl_add_u32", vt[7], lane, 3, vt[7])
                a("v_lshlrev_b32", arr[pc], 1, vt[7])
        # epilogue offsets
        a("s_lshl_b32", t[0], wm, 7)
        a("v_add_u32", vt[6], t[0], l31)                            # row inside the tile (block row 0)
        a("v_lshlrev_b32", self.voffRow, 2, vt[6])
        a("s_lshl_b32", t[0], wn, 6)
        a("v_lshl_add_u32", vt[7], h, 3, t[0])                      # wn * 64 + h * 8
        for arr, ld, on in ((self.voffC, k["ldc"], True), (self.voffR, k["ldr"], self.res)):
            if not on:
                continue
            for i in range(4):
                a("v_add_u32", lane, 32 * i, vt[6])
                a("v_mul_lo_u32", lane, lane, ld)
                a("v_add_u32", lane, lane, vt[7])
                a("v_lshlrev_b32", arr[i], 1, lane)
        a("v_lshl_add_u32", vt[7], h, 2, t[0])                      # wn * 64 + h * 4
        a("v_lshlrev_b32", self.voffCol, 2, vt[7])
        # ---- work list (gemm_nt_p8_kernel's): XCD = bid & 7 -> column group, member; tiles u0 + l, l = bid >> 3, += grid >> 3
        xcd, cgrp, xj = t[0], t[1], t[2]
        a("s_and_b32", xcd, self.s_bid, 7)
        a("s_lshr_b32", cgrp, xcd, k["log2X"])
        a("s_lshl_b32", t[3], cgrp, k["log2X"])
        a("s_sub_u32", xj, xcd, t[3])
        a("s_mul_i32", self.s_u0, xj, k["Q"])
        a("s_mul_i32", self.s_ncol0, cgrp, k["cg"])
        a("s_lshr_b32", self.s_lstep, k["grid"], 3)
        a("s_sub_u32", t[3], k["U"], self.s_u0)                      # may wrap when u0 > U: compared as signed below
        a("s_cmp_lt_i32", t[3], k["Q"])
        a("s_cselect_b32", self.s_lend, t[3], k["Q"])
        a("s_mov_b32", self.s_ntiles, 0)
        a("s_lshr_b32", self.s_lnext, self.s_bid, 3)                 # l of the first tile
        a("s_cmp_lt_i32", self.s_lnext, self.s_lend)
        a("s_cbranch_scc0", L_end)
        # number of tiles of this workgroup: count l, l + lstep, ... < lend
        a("s_mov_b32", self.s_left, 0)
        a("s_mov_b32", t[3], self.s_lnext)
        L_cnt = a.newlabel("CNT")
        a.label(L_cnt)
        a("s_add_u32", self.s_left, self.s_left, 1)
        a("s_add_u32", t[3], t[3], self.s_lstep)
        a("s_cmp_lt_u32", t[3], self.s_lend)
        a("s_cbranch_scc1", L_cnt)
        a("s_mov_b32", self.s_ntiles, self.s_left)
        # ---- first tile: coordinates, DMA bases; the next tile's
        self.coords(self.s_lnext, self.s_cm0, self.s_cn0)
        self.dma_base(self.s_dA, self.s_dB, self.s_cm0, self.s_cn0)
        a("s_add_u32", self.s_lnext, self.s_lnext, self.s_lstep)
        self.coords(self.s_lnext, self.s_nm0, self.s_nn0)
        self.dma_base(self.s_dAn, self.s_dBn, self.s_nm0, self.s_nn0)
        a("s_mov_b32", self.s_dcnt, k["nk"])
        # ---- accumulators of the first tile = 0, fragment buffer 1 = 0 (step 0 of the first iteration multiplies it)
        for r in range(128):
            a("v_accvgpr_write_b32", A(r), 0)
        for i in range(4):
            for r in range(4):
                a("v_mov_b32", self.FA[1][i][r], 0)
        for j in range(2):
            for r in range(4):
                a("v_mov_b32", self.FB[1][j][r], 0)
        # ---- prologue DMA: slabs 0 and 1 into stages 0 and 1
        a("s_mov_b32", self.s_wr, 0)
        for slab in range(2):
            a("s_add_u32", self.s_wrA, self.s_wr, self.s_wvA)
            a("s_add_u32", self.s_wrB, self.s_wr, self.s_wvB)
            for pc in range(8):
                a("s_add_u32", "m0", self.s_wrA, pc * 1024)
                a("s_nop", 0)
                a("global_load_lds_dwordx4", self.voffA[pc], self.s_dA)
            for pc in range(4):
                a("s_add_u32", "m0", self.s_wrB, pc * 1024)
                a("s_nop", 0)
                a("global_load_lds_dwordx4", self.voffB[pc], self.s_dB)
            self.add64(self.s_dA, self.s_dA, 128)
            self.add64(self.s_dB, self.s_dB, 128)
            a("s_sub_u32", self.s_dcnt, self.s_dcnt, 1)
            a("s_add_u32", self.s_wr, self.s_wr, STAGE_B)
        # (nk >= 3: the stream cannot switch tiles inside the prologue)
        a("s_mov_b32", self.s_rd, 0)
        a("s_add_u32", self.s_wrA, self.s_wr, self.s_wvA)
        a("s_add_u32", self.s_wrB, self.s_wr, self.s_wvB)
        for kk in range(4):
            a("v_mov_b32", self.v_curA[kk], self.v_rdA0[kk])
            a("v_mov_b32", self.v_curB[kk], self.v_rdB0[kk])
        a("s_mov_b32", self.s_roll, k["nk"])
        a("s_waitcnt", vmcnt=12)
        a("s_barrier")
        # no ds_read is outstanding here, but the iteration's first waits are counted as if six were: harmless (they wait for less)
        L_roll = [a.newlabel("ROLL0"), a.newlabel("ROLL1")]
        L_block = [a.newlabel("BLK0"), a.newlabel("BLK1")]
        L_rtest = [a.newlabel("RT0"), a.newlabel("RT1")]
        a("s_branch", L_rtest[0])

        def block(P):
            """tile multiplied into set P, set 1 - P drained"""
            a.label(L_block[P])
            # --- unrolled iterations with the fillers
            ops = (self.epilogue_ops(1 - P) if self.fillers_on else [])
            head = self.block_start_ops()
            ngaps = (self.nkf - 1) * 32
            state = {"done": 0, "gap": 0}

            def fill(s, qm):
                # spread the ops evenly over the gaps of iterations 1 .. nkf-1
                state["gap"] += 1
                target = (len(ops) * state["gap"] + ngaps - 1) // ngaps
                while state["done"] < min(target, len(ops)):
                    ops[state["done"]]()
                    state["done"] += 1
            for u in range(self.nkf):
                if u == 0:
                    self.iter_tail(self.iteration(1 - P, P, True, lambda s, qm: None, head_ops=head))
                else:
                    self.iter_tail(self.iteration(P, P, False, fill))
            while state["done"] < len(ops):
                ops[state["done"]]()
                state["done"] += 1
            # --- rolled plain iterations
            a.label(L_rtest[P])
            a("s_cmp_eq_u32", self.s_roll, 0)
            a("s_cbranch_scc1", L_rdone[P])
            a.label(L_roll[P])
            self.iter_tail(self.iteration(P, P, False, lambda s, qm: None))
            a("s_sub_u32", self.s_roll, self.s_roll, 1)
            a("s_cmp_lg_u32", self.s_roll, 0)
            a("s_cbranch_scc1", L_roll[P])
            a.label(L_rdone[P])
            # --- next block?
            a("s_cmp_eq_u32", self.s_left, 0)
            a("s_cbranch_scc1", L_end)
            a("s_sub_u32", self.s_left, self.s_left, 1)
            # fall through / jump to the other parity
        L_rdone = [a.newlabel("RD0"), a.newlabel("RD1")]
        block(0)
        a("s_branch", L_block[1])          # (block 1 follows in the listing; kept explicit)
        block(1)
        a("s_branch", L_block[0])
        a.label(L_end)
        a("s_waitcnt", vmcnt=0, lgkmcnt=0)
        # tuning: prof != 0 -> wave 0 of every workgroup stores (cycles of the whole kernel body, tiles)
        L_np = a.newlabel("NOPROF")
        a("s_memtime", self.s_prof1)
        a("s_cmp_eq_u32", self.s_profp[0], 0)
        a("s_cbranch_scc1", L_np)
        a("s_cmp_lg_u32", self.s_wave, 0)
        a("s_cbranch_scc1", L_np)
        a("s_waitcnt", lgkmcnt=0)
        a("s_sub_u32", self.s_prof1[0], self.s_prof1[0], self.s_prof0[0])
        a("s_lshl_b32", self.s_t[0], self.s_bid, 3)
        a("v_mov_b32", self.v_tmp[0], self.s_t[0])
        a("v_mov_b32", self.v_pair[0], self.s_prof1[0])
        a("v_mov_b32", self.v_pair[1], self.s_ntiles)
        a("global_store_dwordx2", self.v_tmp[0], self.v_pair, self.s_profp)
        a("s_waitcnt", vmcnt=0)
        a.label(L_np)


# ------------------------------------------------------------------ emission
# variant table: (class name, gelu, ln, res, unrolled iterations)
CLASSES = {"p": (False, False, False), "l": (False, True, False), "g": (True, False, False), "gl": (True, True, False), "r": (False, False, True)}
NKF = {"p": (3, 4), "l": (3, 4, 6), "g": (3, 4, 6, 12), "gl": (3, 4, 6, 12), "r": (3, 4, 6)}
DTYPES = ("bf16", "f16")


def variants():
    for dt in DTYPES:
        for cls, (gelu, ln, res) in CLASSES.items():
            for nkf in NKF[cls]:
                yield "q4_%s_%s_f%d" % (dt, cls, nkf), dict(dtype=dt, gelu=gelu, ln=ln, res=res, nkf=nkf)
    # tuning ablations (wrong results by construction; bits in Q4.__init__)
    for cls, nkf in (("gl", 12), ("r", 6)):
        gelu, ln, res = CLASSES[cls]
        for x in (1, 2, 4, 5, 13, 21, 29):
            yield "q4_bf16_%s_f%d_x%d" % (cls, nkf, x), dict(dtype="bf16", gelu=gelu, ln=ln, res=res, nkf=nkf, dbg=x)


def kernel_text(name, gen):
    """one __global__ function whose body is the generated asm block"""
    clob = ['"v%d"' % i for i in range(248)] + ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in range(96) if i != 32] + ['"vcc"', '"memory"']
    body = ['"s_mov_b64 s[0:1], %0\\n\\t"', '"s_mov_b32 s2, %1\\n\\t"', '"v_mov_b32 v0, %2\\n\\t"', gen.a.c_string()]
    return ("extern \"C\" __global__ void __launch_bounds__(256, 1) %s(const mlpk::Q4Args args) {\n"
            "    asm volatile(\n%s\n        :\n        : \"s\"(__builtin_amdgcn_kernarg_segment_ptr()), \"s\"(blockIdx.x), \"v\"(threadIdx.x)\n"
            "        : %s);\n}\n" % (name, "\n".join(body), ", ".join(clob)))


def emit(path):
    import isa
    out = ["// GENERATED by csrc/gen/q4gen.py -- do not edit.  One asm block per kernel: every register is named by the generator.\n"]
    table = []
    for name, kw in variants():
        g = Q4(**kw)
        pr = isa.lint(g.a)
        if pr:
            raise RuntimeError("%s: %d hazard lint findings, first: %s" % (name, len(pr), pr[0]))
        out.append(kernel_text(name, g))
        table.append((name, kw))
    out.append("namespace mlpk {\nstruct Q4Variant { const char* name; const void* fn; int dtype, gelu, ln, res, nkf, dbg; };\n"
               "static const Q4Variant kQ4Variants[] = {\n")
    for name, kw in table:
        dbg = kw.get("dbg", 0)
        out.append("    {\"%s\", reinterpret_cast<const void*>(&%s), %s, %d, %d, %d, %d, %d},\n" %
                   (name, name, "MLPK_BF16" if kw["dtype"] == "bf16" else "MLPK_F16", kw["gelu"], kw["ln"], kw["res"], kw["nkf"], dbg))
    out.append("};\n}  // namespace mlpk\n")
    text = "".join(out)
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return len(table)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print("wrote %d kernels to %s" % (emit(sys.argv[1]), sys.argv[1]))
    else:
        g = Q4(gelu=True, ln=True, nkf=12)
        print("instructions:", len(g.a.ins), "vgprs:", g.nv, "sgprs:", g.ns)
