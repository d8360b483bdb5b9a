"""Generator of the "q4" NT GEMM: one wave per SIMD, the previous tile's epilogue issued as fillers behind the MFMAs.

    C[m, n] = epi( sum_k A[m, k] * B[n, k] )      A: M x K, B: N x K, K-contiguous, 16-bit; C 16-bit row-major

Why this shape (docs/DESIGN_LOG_r1-r4.md 3.1f): on gfx950 the VALU work of one wave does not overlap the MFMAs of ANOTHER wave of
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
from isa import A, F, H, S, V, Abs, Asm, Neg, Reg, h2bits  # noqa: E402

# the 16-bit GELU polynomials of mlpk_common.h (MLPK_GELUP_*) by storage type.  f16: (scale, Horner coefficients) of the centred form
# t = clamp(x * scale, -sqrt2, sqrt2), u = t * t - 1; bf16: (clamp, coefficients) of the raw form t = clamp(x, -clamp, clamp), u = t * t
GELU = {"f16": (0.314269681, [0.00260713836, -0.00718860654, 0.00979797821, -0.0172248576, 0.0355015062, -0.0601866171, 0.090279378,
                              -0.127707109, 0.174028099, -0.245624334, 0.499268919]),
        "bf16": (4.0, [-1.58078628e-09, 1.21711111e-07, -4.10086659e-06, 8.06673925e-05, -0.00104820437, 0.00966487452, -0.0661753789,
                       0.39884752])}
GELU_RAW = {"f16": False, "bf16": True}
# round 4, bf16 grade: gelu(x) = x / (1 + 2^(x (k0 + k1 |x| + k2 x^2)))  (mlpk_common.h MLPK_GELUS_K*, tools/fit_gelu_sig.py): SEVEN
# instructions per element instead of eleven (two of them transcendental: ~2 extra cycles each beside an MFMA, tools/ubench/q4_slots.py).
# MLPK_GELU_BF16_POLY=1 at generation time (+ -DMLPK_GELU_BF16_POLY for the HIP sources) keeps the polynomial for A/B builds.
GELU_SIG = {"bf16": (-2.28684449, -0.0305621661, -0.0905431807)}
# round 5: the default bf16 form is "h2b" -- Phi in packed f16 (GELU_H2 below), the product in fp32 on the unrounded x (mlpk_common.h gelu_h2b_f);
# MLPK_GELU_BF16_SIG=1 / MLPK_GELU_BF16_POLY=1 at generation time (+ the -D of the same name for the HIP sources) keep the older forms for A/B builds
GELU_FORM = {"f16": "poly", "bf16": "poly" if os.environ.get("MLPK_GELU_BF16_POLY") == "1" else ("sig" if os.environ.get("MLPK_GELU_BF16_SIG") == "1" else "h2b")}


# round 5, the fused token-mixing kernel's bf16 grade ("h2", t4gen.py): Phi as a polynomial in PACKED f16 -- two elements per instruction and no
# transcendental: h = rtz_f16(x); t = h * SCALE; u = t * t - 1; Phi = clamp01(0.5 + t * Q(u)), Q = 7 coefficients by Horner; gelu = h * Phi, which IS
# the f16 operand of the second product (the hidden is kept in f16 there: 11 bits instead of bf16's 8).  11 instructions per PAIR against 15 (two of
# them transcendental) -- profiles/r05_issue_slots_packed_gelu.txt: 43.7 -> 38.6 cycles per MFMA in the kernel's loop.  Every value is exact in f16
# (tools/fit_gelu_h2.py: fitted on |x| <= 4 with the coefficients rounded one by one; the leading coefficient is positive, so beyond the interval the
# polynomial runs off in the direction the clamp wants and no operand clamp is needed; rtz never produces an infinity, so h * Phi is never inf * 0).
GELU_H2 = {"scale": 0.353515625,
           "coefs": [0.01392364501953125, -0.046875, 0.07305908203125, -0.100830078125, 0.155029296875, -0.2386474609375, 0.497802734375]}


def h2_gelu_ops(E, x, hdst, t, u, q, v_c0, s_scale, s_c):
    """hdst[k] <- packed f16 (gelu(x[2k]), gelu(x[2k+1])) for the pairs k abreast; t, u, q: one scratch register per pair; v_c0 / s_scale / s_c[0:5]:
    the packed constants (both halves) -- a VOP3P instruction reads at most one SGPR, so the first Horner step takes c0 from a VGPR."""
    n = len(hdst)
    for k in range(n):
        E("v_cvt_pkrtz_f16_f32", hdst[k], x[2 * k], x[2 * k + 1])
    for k in range(n):
        E("v_pk_mul_f16", t[k], hdst[k], s_scale)
    for k in range(n):
        E("v_pk_fma_f16", u[k], t[k], t[k], H(-1.0), op_sel_hi="[1,1,0]")
    for k in range(n):
        E("v_pk_fma_f16", q[k], v_c0, u[k], s_c[0])
    for j in range(1, 6):
        for k in range(n):
            E("v_pk_fma_f16", q[k], q[k], u[k], s_c[j])
    for k in range(n):
        E("v_pk_fma_f16", q[k], t[k], q[k], H(0.5), op_sel_hi="[1,1,0]", clamp=True)
    for k in range(n):
        E("v_pk_mul_f16", hdst[k], hdst[k], q[k])


def h2b_gelu_ops(E, x, hp, t, u, q, v_c, s_scale, s_c6, v_nz):
    """x[r] <- gelu(x[r]) in fp32 for four chains = two pairs abreast, the operation sequence of gelu_h2b_f (mlpk_common.h): Phi of the
    nearest-even f16 of x in packed f16, the product on the unrounded fp32 x.  hp, t, u, q: two scratch registers each; v_c: c0 .. c5 as
    packed constants in VGPRs, s_scale / s_c6 in SGPRs (a VOP3P instruction reads at most one SGPR); v_nz holds -0.0: the product is an fma
    and x * 0 = -0 for x < 0 must stay -0 (a +0 addend would turn it into +0; the HIP kernels multiply, and the tiles are held bit-equal)."""
    for k in range(2):
        E("v_cvt_pk_f16_f32", hp[k], x[2 * k], x[2 * k + 1])
    for k in range(2):
        E("v_pk_mul_f16", t[k], hp[k], s_scale)
    for k in range(2):
        E("v_pk_fma_f16", u[k], t[k], t[k], H(-1.0), op_sel_hi="[1,1,0]")
    for k in range(2):
        E("v_pk_fma_f16", q[k], v_c[0], u[k], v_c[1])
    for j in range(2, 6):
        for k in range(2):
            E("v_pk_fma_f16", q[k], q[k], u[k], v_c[j])
    for k in range(2):
        E("v_pk_fma_f16", q[k], q[k], u[k], s_c6)
    for k in range(2):
        E("v_pk_fma_f16", q[k], t[k], q[k], H(0.5), op_sel_hi="[1,1,0]", clamp=True)
    for r in range(4):
        E("v_fma_mix_f32", x[r], x[r], q[r >> 1], v_nz, op_sel="[0,%d,0]" % (r & 1), op_sel_hi="[0,1,0]")


def sig_gelu_ops(E, x, q, v_k2, s_k1, s_k0):
    """x[r] <- gelu(x[r]) for four chains abreast, the operation sequence of gelu16_f<bf16> (mlpk_common.h); q: one scratch register
    per chain.  (A transcendental's result is read >= 4 instructions after it was written: the gfx940 forwarding hazard needs 1.)"""
    for r in range(4):
        E("v_fma_f32", q[r], Abs(x[r]), v_k2, s_k1)
    for r in range(4):
        E("v_fma_f32", q[r], Abs(x[r]), q[r], s_k0)
    for r in range(4):
        E("v_mul_f32", q[r], x[r], q[r])
    for r in range(4):
        E("v_exp_f32", q[r], q[r])
    for r in range(4):
        E("v_add_f32", q[r], F(1.0), q[r])
    for r in range(4):
        E("v_rcp_f32", q[r], q[r])
    for r in range(4):
        E("v_mul_f32", x[r], x[r], q[r])
SQRT2 = 1.41421356237

STAGE_B = 49152          # one LDS stage: A 256 x 128 B, then B 128 x 128 B
# round 6: a workgroup's last block drains its last tile WITHOUT multiplying a dummy tile (Q4.build drain_only); MLPK_Q4_DRAIN_ONLY=0 at
# generation time rebuilds the round-5 kernels (A/B)
DRAIN_ONLY = os.environ.get("MLPK_Q4_DRAIN_ONLY", "1") != "0"
B_OFF = 32768
OUT_OFF = 3 * STAGE_B     # 4 x 4 KiB: one staging tile per wave for the stores
LDS_BYTES = OUT_OFF + 16384

# kernarg layout (bytes) -- mirrored by struct Q4Args in mlpk_gemm_q4.hip
KA = dict(A=0, B=8, C=16, R=24, bias=32, ln_mean=40, ln_rstd=48, ln_csum=56,
          lda=64, ldb=68, ldc=72, ldr=76, nk=80, cg=84, cg_magic=88, U=92, Q=96, log2X=100, m_base=104, grid=108, prof=112,
          row_part=120, row_part_ld=128)


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
    def __init__(self, dtype="bf16", gelu=False, ln=False, res=False, stats=False, nkf=4, dbg=0, name=None, static=False):
        assert nkf >= 2
        self.dtype, self.gelu, self.ln, self.res, self.stats, self.nkf = dtype, gelu, ln, res, stats, nkf
        # static (round 4): the kernel is built for ONE K = 64 * nkf with nkf % 3 == 0, so a tile always starts in LDS stage 0 and
        # every stage / slab / tile-switch decision of the K loop is made HERE instead of by SALU instructions at run time: the loop's
        # own overhead drops from 81 to 41 instructions per 32 MFMAs (no stage rotation, no DMA pointer arithmetic, one m0 write per
        # FOUR LDS-DMA pieces -- the instruction's immediate offset moves both the global and the LDS address, tools/ubench/lds_dma_offset.hip)
        self.static = static
        assert not static or (nkf % 3 == 0 and nkf <= 32)
        # tuning ablations (results are wrong by construction): 1 no LDS-DMA, 2 no stores, 4 no epilogue fillers, 8 no fragment reads,
        # 16 minimal iteration tail (no stage rotation), 64 every tile stored over tile (0, 0); variants under test: A/B switches: 256 the four
        # stores of a block row back to back (default: spread over the next block row), 512 ordinary instead of non-temporal stores
        self.dbg = dbg
        self.fillers_on, self.dma_on, self.stores_on, self.reads_on, self.tail_on = not (dbg & 4), not (dbg & 1), not (dbg & 2), not (dbg & 8), not (dbg & 16)
        self.name = name or "q4_%s%s%s%s_%s%d" % (dtype, "_gelu" if gelu else "", "_ln" if ln else "", "_res" if res else "", "s" if static else "f", nkf)
        self.a = Asm()
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.build()

    # ------------------------------------------------------------------ registers
    def regs(self):
        s = Alloc("s", 33, 96)          # (s32 is the ABI stack pointer: hipcc warns when an asm block clobbers it; s96..s101 stay free for the
                                        #  asm block's scalar inputs and the wrapper's own needs -- with fewer "inline assembly requires more
                                        #  registers than available"; s102.. are flat_scratch / xnack / vcc)
        v = Alloc("v", 1, 248)
        self.s_karg, self.s_bid, self.v_tid = S(0, 2), S(2), V(0)
        self.p = {k: S(4 + 2 * i, 2) for i, k in enumerate(["A", "B", "C", "R", "bias", "ln_mean", "ln_rstd", "ln_csum"])}
        self.k = {k: S(20 + i) for i, k in enumerate(["lda", "ldb", "ldc", "ldr", "nk", "cg", "cg_magic", "U", "Q", "log2X", "m_base", "grid"])}
        # work list
        self.s_wave, self.s_u0, self.s_lend, self.s_ncol0, self.s_lstep = s("wave"), s("u0"), s("lend"), s("ncol0"), s("lstep")
        self.s_lnext = s("lnext")              # l of the tile whose coordinates were computed last
        self.s_left = s("left")                # blocks still to run (tiles after the first + the draining one)
        self.s_roll = s("roll")                # rolled iterations of the current block
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
        self.s_eC = s("eC", 2, 2)
        self.s_eR = s("eR", 2, 2) if self.res else None
        self.s_rowC = s("rowC")                                # bytes of 32 rows of C / R
        self.s_rowR = s("rowR") if self.res else None
        self.s_eBias = s("eBias", 2, 2)
        self.s_eCsum, self.s_eMu, self.s_eRstd = (s("eCsum", 2, 2), s("eMu", 2, 2), s("eRstd", 2, 2)) if self.ln else (None, None, None)
        if self.stats:
            self.s_part, self.s_partld = s("part", 2, 2), s("partld")
            self.s_eP = s("eP", 2, 2)
        self.sig = self.gelu and GELU_FORM[self.dtype] == "sig"
        self.h2b = self.gelu and GELU_FORM[self.dtype] == "h2b"
        self.s_r2 = s("r2") if (self.gelu and not self.sig and not self.h2b and not GELU_RAW[self.dtype]) else None      # sqrt 2: the centred polynomial's clamp
        self.s_k0, self.s_k1 = (s("gk0"), s("gk1")) if (self.sig or self.h2b) else (None, None)      # (h2b: the packed scale and c6)
        # (prof1 = the end stamp, taken after the last block: it lives in the next-tile DMA base, which is dead by then)
        self.s_prof0, self.s_prof1, self.s_profp, self.s_ntiles = s("prof0", 2, 2), self.s_dAn, s("profp", 2, 2), s("ntiles")
        self.s_t = [s("t%d" % i) for i in range(6)]
        # vector registers
        self.FA = [[v("FA%d_%d" % (b, i), 4, 4) for i in range(4)] for b in range(2)]
        self.FB = [[v("FB%d_%d" % (b, j), 4, 4) for j in range(2)] for b in range(2)]
        # fragment read addresses: dynamic kernels keep "current stage" copies; static ones a second base for stage 2 (a DS
        # immediate offset has 16 bits: stages 0 and 1 are offsets of the first base)
        self.v_curA = [v(("rdA2_%d" if self.static else "curA%d") % k) for k in range(4)]
        self.v_curB = [v(("rdB2_%d" if self.static else "curB%d") % k) for k in range(4)]
        self.v_rdA0 = [v("rdA0_%d" % k) for k in range(4)]
        self.v_rdB0 = [v("rdB0_%d" % k) for k in range(4)]
        self.voffA = [v("voffA%d" % k) for k in range(8)]
        self.voffB = [v("voffB%d" % k) for k in range(4)]
        self.voffC = [v("voffC%d" % i) for i in range(4)]      # store offsets of the 4 row groups of 8 rows of a block row
        self.voffR = [v("voffR%d" % i) for i in range(4)] if self.res else None
        self.voffCol, self.voffRow = v("voffCol"), v("voffRow")
        self.v_bias = [[v("bias%d_%d" % (j, g), 4, 4) for g in range(4)] for j in range(2)]
        self.v_csum = [[v("csum%d_%d" % (j, g), 4, 4) for g in range(4)] for j in range(2)] if self.ln else None
        self.v_mu = [v("mu%d" % i) for i in range(4)] if self.ln else None
        self.v_rstd = [v("rstd%d" % i) for i in range(4)] if self.ln else None
        self.v_c0 = v("c0") if self.gelu else None
        self.v_x = [[v("x%d_%d" % (b, r)) for r in range(4)] for b in range(2)]
        ge = self.gelu
        ns_ = 2 if self.h2b else 4                     # scratch registers per chain set (h2b works on PAIRS: half as many, + the packed f16 of x)
        self.v_t = [[v("t%d_%d" % (b, r)) for r in range(ns_)] for b in range(2)] if ge else [None, None]
        self.v_u = [[v("u%d_%d" % (b, r)) for r in range(ns_)] for b in range(2)] if ge else [None, None]
        self.v_q = [[v("q%d_%d" % (b, r)) for r in range(ns_)] for b in range(2)] if ge else [None, None]
        self.v_hp = [[v("hp%d_%d" % (b, r)) for r in range(2)] for b in range(2)] if self.h2b else [None, None]
        self.v_hc = [v("hc%d" % j) for j in range(1, 6)] if self.h2b else None      # c1 .. c5 (c0 is v_c0)
        self.v_nz = v("negzero") if self.h2b else None
        self.v_pk = [v("pk%d" % k, 2, 2) for k in range(2)]
        self.v_stw = [v("stw%d" % c) for c in range(8)]        # LDS staging: write address of 16-byte chunk c of this lane's row
        self.v_strd = [v("strd%d" % k) for k in range(4)]      # ... and the read addresses (row (lane >> 3) + 8 k, chunk lane & 7)
        self.v_out = [v("out%d" % k, 4, 4) for k in range(4)]
        self.v_res = [[v("res%d_%d" % (i, k), 4, 4) for k in range(4)] for i in range(2)] if self.res else None    # block rows i, i + 2
        if self.stats:
            self.v_sp = v("sp", 2, 2)                          # (sum, sum of squares) of a lane's 8 stored values, then of its row
            self.v_ones = v("ones")
            self.voffP = [v("voffP%d" % k) for k in range(4)]
        self.v_tmp = [v("tmp%d" % i) for i in range(8)]
        self.v_pair = v("pair", 2, 2)
        self.nv, self.ns = v.next, s.next

    def acc(self, set_, b):
        return A(128 * set_ + 16 * b, 16)

    # ------------------------------------------------------------------ helpers
    SCC_READERS = ("s_addc_u32", "s_subb_u32", "s_cselect_b32", "s_cbranch_scc0", "s_cbranch_scc1")

    def capture(self, fn):
        """run fn (which emits plain instructions), take the instructions back out of the listing and return them as filler
        closures -- an instruction that READS SCC stays glued to its predecessors (fillers from different lists are interleaved
        gap by gap, and nearly every SALU instruction writes SCC)"""
        a = self.a
        n0 = len(a.ins)
        fn()
        ins = a.ins[n0:]
        del a.ins[n0:]
        groups = []
        for x in ins:
            if x.op in self.SCC_READERS and groups:
                groups[-1].append(x)
            else:
                groups.append([x])
        return [(lambda g=g: a.ins.extend(g)) for g in groups]

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
        """SALU fillers that set the epilogue bases from (pm0, pn0)"""
        a, t = self.a, self.s_t

        def emit():
            E = a
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
            if self.stats:
                # plane (pn0 / 32 + 2 wn) (+ 1 for the lanes of the wave's second 32 columns: voffP), row pm0:
                # ((pn0 >> 5) + 2 wn) * ld + pm0 pairs of 8 bytes
                E("s_lshr_b32", t[0], self.s_pn0, 5)
                E("s_and_b32", t[1], self.s_wave, 1)
                E("s_lshl_b32", t[1], t[1], 1)
                E("s_add_u32", t[0], t[0], t[1])
                E("s_mul_hi_u32", t[1], t[0], self.s_partld)
                E("s_mul_i32", t[0], t[0], self.s_partld)
                E("s_add_u32", t[0], t[0], self.s_pm0)
                E("s_addc_u32", t[1], t[1], 0)
                E("s_lshl_b32", t[1], t[1], 3)
                E("s_lshr_b32", t[2], t[0], 29)
                E("s_or_b32", t[1], t[1], t[2])
                E("s_lshl_b32", t[0], t[0], 3)
                E("s_add_u32", self.s_eP[0], self.s_part[0], t[0])
                E("s_addc_u32", self.s_eP[1], self.s_part[1], t[1])
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
        return self.capture(emit)

    def block_start_ops(self):
        """shift the tile coordinates, the epilogue bases and parameter loads of the drained tile, the next tile's coordinates and
        DMA bases, the number of rolled iterations"""
        a, t = self.a, self.s_t

        def shift():
            if self.static:
                # the DMA stream moves on to this tile's panels (the previous block's last two iterations already requested its
                # slabs 0 and 1 through dAn / dBn); these four lead the head so that they precede the iteration's first piece
                for d_, n_ in ((self.s_dA, self.s_dAn), (self.s_dB, self.s_dBn)):
                    a("s_mov_b32", d_[0], n_[0])
                    a("s_mov_b32", d_[1], n_[1])
            a("s_mov_b32", self.s_pm0, 0 if (self.dbg & 64) else self.s_cm0)       # (64: every tile is stored over tile (0, 0))
            a("s_mov_b32", self.s_pn0, 0 if (self.dbg & 64) else self.s_cn0)
            a("s_mov_b32", self.s_cm0, self.s_nm0)
            a("s_mov_b32", self.s_cn0, self.s_nn0)

        def nxt():
            a("s_add_u32", self.s_lnext, self.s_lnext, self.s_lstep)
            self.coords(self.s_lnext, self.s_nm0, self.s_nn0)
            self.dma_base(self.s_dAn, self.s_dBn, self.s_nm0, self.s_nn0)
            if not self.static:
                # rolled iterations of this block: nk - nkf, or 0 for the draining block (no block left after it)
                a("s_sub_u32", t[0], self.k["nk"], self.nkf)
                a("s_cmp_lg_u32", self.s_left, 0)
                a("s_cselect_b32", self.s_roll, t[0], 0)
        ops = self.capture(shift) + self.epi_bases_ops()
        if self.fillers_on:
            ops += self.param_load_ops()
        return ops + self.capture(nxt)

    def param_load_ops(self):
        """the column / row parameters (and the residual tile) of the drained tile, as loads"""
        a = self.a
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))
        for j in range(2):
            for g in range(4):
                ops.append(lambda j=j, g=g: self.vload("global_load_dwordx4", self.v_bias[j][g], self.voffCol, self.s_eBias, offset=(j * 32 + g * 8) * 4))
                if self.ln:
                    ops.append(lambda j=j, g=g: self.vload("global_load_dwordx4", self.v_csum[j][g], self.voffCol, self.s_eCsum, offset=(j * 32 + g * 8) * 4))
        if self.ln:
            for i in range(4):
                ops.append(lambda i=i: self.vload("global_load_dword", self.v_mu[i], self.voffRow, self.s_eMu, offset=i * 128))
                ops.append(lambda i=i: self.vload("global_load_dword", self.v_rstd[i], self.voffRow, self.s_eRstd, offset=i * 128))
        if self.res:
            for i in range(2):
                ops += self.res_load_ops(i)
        return ops

    def vload(self, op, *args, **kw):
        """emit a VGPR load and number it (vmcnt bookkeeping: loads and LDS-DMA complete in order)"""
        self.a(op, *args, **kw)
        self.vm_loads += 1
        return self.vm_loads - 1

    def wait_vload(self, idx):
        n = self.vm_loads - idx - 1
        self.a("s_waitcnt", vmcnt=min(n, 63))

    def res_load_ops(self, i):
        """residual tile of block row i, in the layout of the stores: lane = (row lane >> 3 of row group k, 16-byte chunk
        lane & 7); s_eR walks the block rows.  Registers are shared by block rows i and i + 2."""
        ops = []
        rec = [None] * 4
        self._res_loads[i] = rec
        for k in range(4):
            ops.append(lambda k=k: rec.__setitem__(k, self.vload("global_load_dwordx4", self.v_res[i & 1][k], self.voffR[k], self.s_eR)))
        ops += self.capture(lambda: self.add64(self.s_eR, self.s_eR, self.s_rowR))
        return ops

    def n_param_loads(self):
        return 8 * (2 if self.ln else 1) + (8 if self.ln else 0) + (16 if self.res else 0)

    def gelu_ops(self, E, x, t, u, q):
        """x[r] <- gelu(x[r]) for the 4 chains abreast (the operation sequence of gelu16_f in mlpk_common.h)"""
        if self.sig:
            return sig_gelu_ops(E, x, q, self.v_c0, self.s_k1, self.s_k0)
        if self.h2b:
            return h2b_gelu_ops(E, x, self._hp, t, u, q, [self.v_c0] + self.v_hc, self.s_k0, self.s_k1, self.v_nz)
        scale, c = GELU[self.dtype]
        if GELU_RAW[self.dtype]:
            for r in range(4):
                E("v_med3_f32", t[r], x[r], F(-scale), F(scale))
            for r in range(4):
                E("v_mul_f32", u[r], t[r], t[r])
        else:
            for r in range(4):
                E("v_mul_f32", t[r], F(scale), x[r])
            for r in range(4):
                E("v_med3_f32", t[r], t[r], Neg(self.s_r2), self.s_r2)
            for r in range(4):
                E("v_fma_f32", u[r], t[r], t[r], F(-1.0))
        for r in range(4):
            E("v_fmaak_f32", q[r], u[r], self.v_c0, F(c[1]))
        for k in range(2, len(c)):
            for r in range(4):
                E("v_fmaak_f32", q[r], q[r], u[r], F(c[k]))
        for r in range(4):
            E("v_fma_f32", t[r], t[r], q[r], F(0.5))
        for r in range(4):
            E("v_mul_f32", x[r], x[r], t[r])

    def ds(self, op, *args, **kw):
        """emit an LDS instruction and count it (lgkmcnt bookkeeping: LDS operations of a wave complete in order)"""
        self.a(op, *args, **kw)
        self.lgkm_issued += 1
        return self.lgkm_issued - 1

    def wait_lds(self, idx):
        """wait until LDS operation number idx has completed"""
        n = self.lgkm_issued - idx - 1
        self.a("s_waitcnt", lgkmcnt=min(n, 15))

    def epilogue_ops(self, set_):
        """The drain of accumulator set `set_` as a flat list of one-instruction closures.  Per block row (32 rows x 64 columns of the
        wave): the 8 groups of 4 columns go acc -> fp32 math -> packed 16-bit pair -> ds_write_b64 into the wave's 4 KiB staging tile
        ([row][128 B], 16-byte chunk c of row r at chunk c ^ (r & 7)); then the tile is read back as rows (lane = row lane >> 3 of a
        group of 8 rows, chunk lane & 7), so that every global store instruction writes 8 whole 128-byte lines.  (Stored straight from
        the accumulator layout -- every lane its own row, 32 rows x 32 bytes per instruction -- a store took ~590 cycles to issue:
        profiles/r03_q4_cycles_v1.txt.)  The stores of block row i are issued two groups into block row i + 1."""
        a = self.a
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))

        def store_part(i, k):
            """row group k of block row i: (residual add) + store; the first part waits for the staging reads"""
            rd = self._st_reads[i]
            if k == 0:
                ops.append(lambda: self.wait_lds(rd[3]))
                if self.res:
                    ops.append(lambda: self.wait_vload(self._res_loads[i][3]))
            o = self.v_out[k]
            if self.res:
                rr = self.v_res[i & 1][k]
                tm = self.v_tmp
                for c in range(4):
                    if self.dtype == "bf16":
                        E("v_lshlrev_b32", tm[0], 16, o[c])
                        E("v_and_b32", tm[1], 0xFFFF0000, o[c])
                        E("v_lshlrev_b32", tm[2], 16, rr[c])
                        E("v_and_b32", tm[3], 0xFFFF0000, rr[c])
                    else:
                        E("v_lshrrev_b32", tm[1], 16, o[c])
                        E("v_lshrrev_b32", tm[3], 16, rr[c])
                        E("v_cvt_f32_f16", tm[0], o[c])
                        E("v_cvt_f32_f16", tm[2], rr[c])
                        E("v_cvt_f32_f16", tm[1], tm[1])
                        E("v_cvt_f32_f16", tm[3], tm[3])
                    E("v_add_f32", tm[0], tm[0], tm[2])
                    E("v_add_f32", tm[1], tm[1], tm[3])
                    E(self.cvt, o[c], tm[0], tm[1])
            if self.stats:
                sp, dot = self.v_sp, ("v_dot2c_f32_bf16" if self.dtype == "bf16" else "v_dot2c_f32_f16")
                E("v_mov_b32", sp[0], 0)
                E("v_mov_b32", sp[1], 0)
                for c in range(4):
                    E(dot, sp[0], o[c], self.v_ones)
                    E(dot, sp[1], o[c], o[c])
                # planes of 32 columns, reduced in ONE order by every tile of the library (mlpk.h row_part): a lane's chunk of 8 by four
                # dot products, then (c0 + c1) + (c2 + c3) over the four lanes that hold 32 consecutive columns
                for n_, mod in enumerate((dict(quad_perm="[1,0,3,2]"), dict(quad_perm="[2,3,0,1]"))):
                    # wait states: a VALU result is read through DPP after >= 2, a dot-product result by another opcode after >= 3
                    E("s_nop", 1 if n_ == 0 else 0)
                    E("v_add_f32_dpp", sp[0], sp[0], sp[0], **mod, row_mask="0xf", bank_mask="0xf")
                    E("v_add_f32_dpp", sp[1], sp[1], sp[1], **mod, row_mask="0xf", bank_mask="0xf")
                # two pairs per row: the lanes with (lane & 3) == 0
                ops.append(lambda k=k, i=i: (a("s_mov_b32", "exec_lo", 0x11111111), a("s_mov_b32", "exec_hi", 0x11111111),
                                             a("global_store_dwordx2", self.voffP[k], sp, self.s_eP, offset=i * 256),
                                             a("s_mov_b64", "exec", -1)) and None)
            if self.stores_on:
                if not (self.dbg & 512):
                    E("global_store_dwordx4", self.voffC[k], o, self.s_eC, nt=True)
                else:
                    E("global_store_dwordx4", self.voffC[k], o, self.s_eC)
            if k == 3:
                ops.extend(self.capture(lambda: self.add64(self.s_eC, self.s_eC, self.s_rowC)))
                if self.res and i + 2 < 4:
                    ops.extend(self.res_load_ops(i + 2))

        def stores(i):
            for k in range(4):
                store_part(i, k)
        self._st_reads = {}
        bank = 0
        for i in range(4):
            for j in range(2):
                for g in range(4):
                    b = 2 * i + j
                    x, t, u, q = self.v_x[bank], self.v_t[bank], self.v_u[bank], self.v_q[bank]
                    self._hp = self.v_hp[bank]
                    pk = self.v_pk[bank]
                    bank ^= 1
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
                    E(self.cvt, pk[0], x[0], x[1])
                    E(self.cvt, pk[1], x[2], x[3])
                    ops.append(lambda c=4 * j + g, pk=pk: self.ds("ds_write_b64", self.v_stw[c], pk))
                    if i > 0 and not (self.dbg & 256):
                        if (4 * j + g) in (1, 3, 5, 7):
                            store_part(i - 1, (4 * j + g) >> 1)
                    elif i > 0 and j == 0 and g == 1:
                        stores(i - 1)
            rd = [None] * 4
            self._st_reads[i] = rd
            for k in range(4):
                ops.append(lambda k=k, rd=rd: rd.__setitem__(k, self.ds("ds_read_b128", self.v_out[k], self.v_strd[k])))
        stores(3)
        return ops

    # ------------------------------------------------------------------ one iteration of the K loop
    def tail_ops(self):
        """advance the DMA stream and the LDS stages (issued in the gaps of the last k-step: every piece of the iteration has been
        issued by then, the fragment reads of the step use the old addresses until its sixth MFMA)"""
        a, t = self.a, self.s_t
        if not self.tail_on or self.static:
            return [], []

        def salu():
            E = a
            if self.dma_on:
                self.add64(self.s_dA, self.s_dA, 128)
                self.add64(self.s_dB, self.s_dB, 128)
                E("s_sub_u32", self.s_dcnt, self.s_dcnt, 1)
                E("s_cmp_eq_u32", self.s_dcnt, 0)
                E("s_cselect_b32", self.s_dA[0], self.s_dAn[0], self.s_dA[0])
                E("s_cselect_b32", self.s_dA[1], self.s_dAn[1], self.s_dA[1])
                E("s_cselect_b32", self.s_dB[0], self.s_dBn[0], self.s_dB[0])
                E("s_cselect_b32", self.s_dB[1], self.s_dBn[1], self.s_dB[1])
                E("s_cselect_b32", self.s_dcnt, self.k["nk"], self.s_dcnt)
            # stages: rd <- rd + 1, wr <- wr + 1 (mod 3)   (t[5] is the tail's own scratch register)
            for r in (self.s_rd, self.s_wr):
                E("s_add_u32", r, r, STAGE_B)
                E("s_cmp_ge_u32", r, 3 * STAGE_B)
                E("s_cselect_b32", t[5], 3 * STAGE_B, 0)
                E("s_sub_u32", r, r, t[5])
            E("s_add_u32", self.s_wrA, self.s_wr, self.s_wvA)
            E("s_add_u32", self.s_wrB, self.s_wr, self.s_wvB)

        def valu():
            for k in range(4):
                a("v_add_u32", self.v_curA[k], self.s_rd, self.v_rdA0[k])
                a("v_add_u32", self.v_curB[k], self.s_rd, self.v_rdB0[k])
        return self.capture(salu), self.capture(valu)

    def dma_piece(self, kind, pc, base, slab, stage):
        """static kernels: LDS-DMA piece pc (A: 0..7, B: 0..3) of K slab `slab` (counted inside the tile whose panels `base` addresses)
        into LDS stage `stage`.  Four pieces share one m0 value: the instruction's immediate offset
            imm = (pc % 4) * 1024 + slab * 128 - 4096
        is added to the LDS destination AND to the global address (measured, tools/ubench/lds_dma_offset.hip), so
            m0   = wave share + stage * STAGE_B + (pc // 4) * 4096 + 4096 - slab * 128     (written once per group, by the caller)
            voff = piece offset + 4096 - (pc % 4) * 1024                                     (set up once, build())
        give LDS piece pc of the stage and global base + piece offset + slab * 128, with no pointer arithmetic per slab."""
        imm = (pc % 4) * 1024 + slab * 128 - 4096
        assert -4096 <= imm < 4096
        voff = self.voffA[pc] if kind == "A" else self.voffB[pc]
        return self.vload("global_load_lds_dwordx4", voff, base, offset=imm)

    def dma_m0(self, kind, grp, slab, stage):
        c = stage * STAGE_B + grp * 4096 + 4096 - slab * 128
        assert c >= 0
        self.a("s_add_u32", "m0", self.s_wvA if kind == "A" else self.s_wvB, c)

    def iteration(self, set_first, set_main, zero_c, fill, head_ops=None, u=None):
        """Four k-steps of 8 MFMAs with the loads of this iteration between them, then the wait and the barrier.
        set_first / set_main: accumulator set of step 0 / steps 1..3; zero_c: step 1 starts the accumulators (C = 0);
        fill(s, qm): emit the filler instructions of the gap behind MFMA qm of step s.
        u (static kernels): the iteration's index inside its tile -- it reads slab u from stage u % 3 and requests slab u + 2 (of
        this tile, or slab u + 2 - nk of the next one) into stage (u + 2) % 3."""
        a = self.a
        ready = {}          # fragment -> number of its read
        # 4 pieces behind MFMAs 1, 3, 6, 7 of steps 0, 1 (A) and 2 (B); step 3 carries the bookkeeping
        dma_slots = {}
        for s_, (kind, base) in enumerate((("A", 0), ("A", 4), ("B", 0))):
            for n_, qm_ in enumerate((1, 3, 6, 7)):
                dma_slots[(s_, qm_)] = (kind, base + n_)
        if self.static:
            assert u is not None
            nk = self.nkf
            d_slab, d_next = (u + 2, False) if u + 2 < nk else (u + 2 - nk, True)
            d_stage = (u + 2) % 3
            r_stage = u % 3
        head = list(head_ops or [])
        tail_s, tail_v = self.tail_ops()
        i_start = len(a.ins)
        for s in range(4):
            mbuf, rbuf = (s + 1) & 1, s & 1
            acc_set = set_first if s == 0 else set_main
            read_order = [("B", 0), ("B", 1), ("A", 0), ("A", 1), ("A", 2), ("A", 3)]
            new_ready = {}
            for qm in range(8):
                i, j = qm >> 1, qm & 1
                if j == 0 and s > 0 and self.reads_on and (not self.static or i in (0, 2)):
                    # first use of A_i (and of both B fragments when i == 0: they were read before A_0); the fragments of step 0
                    # were waited for in front of the previous barrier.  Static kernels wait twice per step instead of four times
                    # (for A_1 in front of MFMA 0, for A_3 in front of MFMA 4: the reads were issued 5 / 3 MFMAs earlier) -- every
                    # s_waitcnt is an issue slot of the one wave that feeds this SIMD
                    self.wait_lds(ready[("A", i if not self.static else i + 1)])
                d = self.acc(acc_set, 2 * i + j)
                slot = dma_slots.get((s, qm))
                if slot and self.dma_on:
                    kind, pc = slot
                    if not self.static:
                        a("s_add_u32", "m0", self.s_wrA if kind == "A" else self.s_wrB, pc * 1024)
                    elif pc % 4 == 0:
                        self.dma_m0(kind, pc // 4, d_slab, d_stage)
                csrc = 0 if (zero_c and s == 1) else d
                a(self.mfma, d, self.FB[mbuf][j], self.FA[mbuf][i], csrc)
                if slot and self.dma_on:
                    kind, pc = slot
                    if self.static:
                        base = ((self.s_dAn if d_next else self.s_dA) if kind == "A" else (self.s_dBn if d_next else self.s_dB))
                        self.dma_piece(kind, pc, base, d_slab, d_stage)
                    elif kind == "A":
                        self.vload("global_load_lds_dwordx4", self.voffA[pc], self.s_dA)
                    else:
                        self.vload("global_load_lds_dwordx4", self.voffB[pc], self.s_dB)
                if qm < 6 and self.reads_on:
                    kind, idx = read_order[qm]
                    if self.static:
                        # stages 0 / 1: immediate offsets of the first base; stage 2: the second base
                        regs = (self.v_rdA0, self.v_rdB0) if r_stage < 2 else (self.v_curA, self.v_curB)
                        addr = (regs[0] if kind == "A" else regs[1])[s]
                        off = idx * 4096 + (STAGE_B if r_stage == 1 else 0)
                    else:
                        addr, off = (self.v_curA[s] if kind == "A" else self.v_curB[s]), idx * 4096
                    new_ready[(kind, idx)] = self.ds("ds_read_b128", (self.FA if kind == "A" else self.FB)[rbuf][idx], addr, offset=off)
                if s == 3:
                    for _ in range(3):
                        if tail_s:
                            tail_s.pop(0)()
                    if qm >= 6:
                        for _ in range(4):
                            if tail_v:
                                tail_v.pop(0)()
                # fillers
                if head:
                    for _ in range(4):
                        if head:
                            head.pop(0)()
                else:
                    fill(s, qm)
            ready = new_ready
        for rest in (head, tail_s, tail_v):
            while rest:
                rest.pop(0)()
        # vmcnt in front of the barrier: the pieces of the PREVIOUS iteration must have landed (<= 12 loads of this one may be
        # in flight: loads return in order, stores may not be counted on), and so must every VGPR load issued in this one
        # Round 6 experiment, kept as an option (MLPK_Q4_COUNT_STORES=1 at generation time; the emulator then needs STORES_IN_ORDER): counting
        # this iteration's STORES in the wait as well.  A wave's vector-memory operations report completion in issue order on this family
        # (LLVM AMDGPUUsage, memory model GFX6-GFX9 / GFX90A / GFX942), so vmcnt(operations issued in this iteration) would be enough
        # for "everything older has completed"; counting the loads only also covers this iteration's first pieces, one per store behind
        # them.  Hypothesis: that is what makes gMLP's proj1 (K = 256: 16 stores per 4 slabs) run 144 us in the model against 83 alone.
        # Measured same-box, three alternations (profiles/r06_q4_count_stores_ab.txt): gMLP-S 9.932 / 9.946 / 9.928 against 9.939 / 9.940 /
        # 9.912 ms, Mixer-B/16 7.39 vs 7.40, ViP-S7 27.81 vs 27.62 (-0.6 %), ResMLP-24 5.07 vs 5.07: nothing -- the slab waits are not
        # what the stores hold up.  Default off.
        count_stores = os.environ.get("MLPK_Q4_COUNT_STORES", "0") == "1"
        kinds = [("load" if x.op.startswith("global_load_dword") else ("store" if x.op.startswith("global_store") else "dma"))
                 for x in a.ins[i_start:] if x.op.startswith("global_load") or (count_stores and x.op.startswith("global_store"))]
        allow = len(kinds)
        if "load" in kinds:
            allow = len(kinds) - 1 - max(k for k, x in enumerate(kinds) if x == "load")
        cap = 63 if count_stores else 12
        a("s_waitcnt", vmcnt=min(allow, cap if self.dma_on else 0), lgkmcnt=0)
        a("s_barrier")

    # ------------------------------------------------------------------ the kernel
    def build(self):
        a = self.a
        self.regs()
        self.lgkm_issued = 0
        self.vm_loads = 0
        self._res_loads = {}
        t = self.s_t
        k, p = self.k, self.p
        L_end = a.newlabel("END")
        # ---- arguments
        a("s_load_dwordx16", S(4, 16), self.s_karg, 0)
        a("s_load_dwordx8", S(20, 8), self.s_karg, 64)
        a("s_load_dwordx4", S(28, 4), self.s_karg, 96)
        a("s_load_dwordx2", self.s_profp, self.s_karg, KA["prof"])
        if self.stats:
            a("s_load_dwordx2", self.s_part, self.s_karg, KA["row_part"])
            a("s_load_dword", self.s_partld, self.s_karg, KA["row_part_ld"])
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
        if self.s_r2 is not None:
            a("s_mov_b32", self.s_r2, F(SQRT2))
        if self.sig:
            gk = GELU_SIG[self.dtype]
            a("v_mov_b32", self.v_c0, F(gk[2]))
            a("s_mov_b32", self.s_k1, F(gk[1]))
            a("s_mov_b32", self.s_k0, F(gk[0]))
        elif self.h2b:
            a("v_mov_b32", self.v_c0, h2bits(GELU_H2["coefs"][0]))
            for cj in range(5):
                a("v_mov_b32", self.v_hc[cj], h2bits(GELU_H2["coefs"][cj + 1]))
            a("s_mov_b32", self.s_k0, h2bits(GELU_H2["scale"]))
            a("s_mov_b32", self.s_k1, h2bits(GELU_H2["coefs"][6]))
            a("v_mov_b32", self.v_nz, 0x80000000)
        elif self.gelu:
            a("v_mov_b32", self.v_c0, F(GELU[self.dtype][1][0]))
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
            if self.static:
                a("v_add_u32", self.v_curA[ks], 2 * STAGE_B, self.v_rdA0[ks])
                a("v_add_u32", self.v_curB[ks], 2 * STAGE_B, self.v_rdB0[ks])
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
                if self.static:
                    a("v_add_u32", arr[pc], 4096 - (pc % 4) * 1024, arr[pc])       # see dma_piece
        # epilogue offsets
        a("s_lshl_b32", t[0], wm, 7)
        a("v_add_u32", vt[6], t[0], l31)                            # row inside the tile (block row 0), accumulator layout
        a("v_lshlrev_b32", self.voffRow, 2, vt[6])
        a("s_lshl_b32", t[1], wn, 6)
        a("v_lshl_add_u32", vt[7], h, 2, t[1])                      # wn * 64 + h * 4
        a("v_lshlrev_b32", self.voffCol, 2, vt[7])
        # stores / residual loads: row wm * 128 + (lane >> 3) + 8 k of block row 0, columns wn * 64 + (lane & 7) * 8
        a("v_add_u32", vt[6], t[0], l3)
        a("v_lshl_add_u32", vt[7], l7, 3, t[1])
        for arr, ld, on in ((self.voffC, k["ldc"], True), (self.voffR, k["ldr"], self.res)):
            if not on:
                continue
            for kk in range(4):
                a("v_add_u32", lane, 8 * kk, vt[6])
                a("v_mul_lo_u32", lane, lane, ld)
                a("v_add_u32", lane, lane, vt[7])
                a("v_lshlrev_b32", arr[kk], 1, lane)
        a("s_lshl_b32", self.s_rowC, k["ldc"], 6)
        if self.res:
            a("s_lshl_b32", self.s_rowR, k["ldr"], 6)
        if self.stats:
            # pair offsets: row wm * 128 + 8 k + (lane >> 3) of block row 0, 8 bytes per row; the lanes of the wave's second 32
            # columns (lane & 4) write the next plane
            a("s_lshl_b32", t[0], wm, 7)
            a("v_add_u32", vt[6], t[0], l3)
            a("v_bfe_u32", vt[7], l7, 2, 1)
            a("s_waitcnt", lgkmcnt=0)                                   # (s_partld: the s_load in front of the lane constants)
            a("v_mul_lo_u32", vt[7], vt[7], self.s_partld)
            for kk in range(4):
                a("v_add_u32", lane, 8 * kk, vt[6])
                a("v_add_u32", lane, lane, vt[7])
                a("v_lshlrev_b32", self.voffP[kk], 3, lane)
            a("v_mov_b32", self.v_ones, 0x3F803F80 if self.dtype == "bf16" else 0x3C003C00)
        # LDS staging tile of this wave: OUT_OFF + wave * 4096, [32 rows][128 B]; chunk c of row r sits at chunk c ^ (r & 7) ^ (2 (r >> 3)).
        # (Round 4: the second term.  Rows are 128 B = 32 banks apart, so a chunk position serves two bank groups (row parity) -- 16 slots
        # for the 32 rows of a ds_write_b64; with c ^ (r & 7) alone rows r, r + 8, r + 16, r + 24 fell on ONE slot, a 4-way conflict where 2 is
        # the natural rate: SQ_LDS_BANK_CONFLICT 2.4e6 per fc1 launch.  Padding the rows instead does not fit: LDS is full to the byte.)
        # write: row l31, half h; read k: row l3 + 8 k, chunk l7
        a("s_lshl_b32", t[0], self.s_wave, 12)
        a("s_add_u32", t[0], t[0], OUT_OFF)
        a("v_lshlrev_b32", vt[6], 7, l31)
        a("v_add_u32", vt[6], t[0], vt[6])
        a("v_lshl_add_u32", vt[6], h, 3, vt[6])
        a("v_bfe_u32", vt[7], l31, 3, 2)
        a("v_lshlrev_b32", vt[7], 1, vt[7])
        a("v_xor_b32", vt[7], vt[7], l7)                        # (l31 & 7) ^ (2 (l31 >> 3))
        for c in range(8):
            a("v_xor_b32", lane, c, vt[7])
            a("v_lshl_add_u32", self.v_stw[c], lane, 4, vt[6])
        a("v_lshlrev_b32", vt[6], 7, l3)
        a("v_add_u32", vt[6], t[0], vt[6])
        a("v_xor_b32", vt[7], l7, l3)
        for kk in range(4):
            a("v_xor_b32", lane, 2 * kk, vt[7])
            a("v_lshl_add_u32", self.v_strd[kk], lane, 4, vt[6])
            a("v_add_u32", self.v_strd[kk], kk * 1024, self.v_strd[kk])
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
        if not self.static:
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
        L_roll = [a.newlabel("ROLL0"), a.newlabel("ROLL1")]
        L_block = [a.newlabel("BLK0"), a.newlabel("BLK1")]
        L_rtest = [a.newlabel("RT0"), a.newlabel("RT1")]
        L_drain = [a.newlabel("DRN0"), a.newlabel("DRN1")]
        if self.static:
            for slab in range(2):
                for kind, npc in (("A", 8), ("B", 4)):
                    for pc in range(npc):
                        if pc % 4 == 0:
                            self.dma_m0(kind, pc // 4, slab, slab)
                            a("s_nop", 0)
                        a("global_load_lds_dwordx4", (self.voffA if kind == "A" else self.voffB)[pc], self.s_dA if kind == "A" else self.s_dB,
                          offset=(pc % 4) * 1024 + slab * 128 - 4096)
            a("s_waitcnt", vmcnt=12)
            a("s_barrier")
            # the first tile: nk plain iterations into set 0 (nothing to drain yet); its last two request the second tile's slabs
            for u in range(self.nkf):
                self.iteration(0, 0, False, lambda s, qm: None, u=u)
            a("s_cmp_eq_u32", self.s_left, 0)
            a("s_cbranch_scc1", L_end)
            a("s_sub_u32", self.s_left, self.s_left, 1)
            if DRAIN_ONLY:
                a("s_cmp_eq_u32", self.s_left, 0)
                a("s_cbranch_scc1", L_drain[1])
            a("s_branch", L_block[1])
        else:
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
                    self.iteration(1 - P, P, True, lambda s, qm: None, head_ops=head, u=u)
                else:
                    self.iteration(P, P, False, fill, u=u)
            while state["done"] < len(ops):
                ops[state["done"]]()
                state["done"] += 1
            if not self.static:
                # --- rolled plain iterations
                a.label(L_rtest[P])
                a("s_cmp_eq_u32", self.s_roll, 0)
                a("s_cbranch_scc1", L_rdone[P])
                a.label(L_roll[P])
                self.iteration(P, P, False, lambda s, qm: None)
                a("s_sub_u32", self.s_roll, self.s_roll, 1)
                a("s_cmp_lg_u32", self.s_roll, 0)
                a("s_cbranch_scc1", L_roll[P])
                a.label(L_rdone[P])
            # --- next block?
            a("s_cmp_eq_u32", self.s_left, 0)
            a("s_cbranch_scc1", L_end)
            a("s_sub_u32", self.s_left, self.s_left, 1)
            if DRAIN_ONLY:
                # no tile left to multiply: the draining block WITHOUT its dummy tile (round 6)
                a("s_cmp_eq_u32", self.s_left, 0)
                a("s_cbranch_scc1", L_drain[1 - P])
            # fall through / jump to the other parity

        def drain_only(Q):
            """Round 6: the LAST block of a workgroup.  Until round 5 it was an ordinary block (its tile: the clamped last tile once more, the
            result never stored) whose MFMAs only carried the fillers that drain the real last tile -- one tile's worth of matrix work and operand
            traffic per workgroup for nothing: +5.4 % at Mixer-B's fc1 (18.4 tiles per workgroup), +30 - 40 % at the K = N = 384 products of 50176
            rows (2.3 tiles per workgroup).  Here: the one k-step the K loop's skew leaves for the next block's step 0 (the finished tile's LAST
            k-step, into set 1 - Q, its fragments already in registers), the head (tile coordinates, epilogue bases, parameter loads), and the
            drain itself, straight through -- no operand stream, no barrier (the staging tile is the wave's own)."""
            a.label(L_drain[Q])
            head = self.block_start_ops()
            ops = self.epilogue_ops(1 - Q) if self.fillers_on else []
            for qm in range(8):
                i, j = qm >> 1, qm & 1
                d = self.acc(1 - Q, 2 * i + j)
                a(self.mfma, d, self.FB[1][j], self.FA[1][i], d)
                for _ in range(4):
                    if head:
                        head.pop(0)()
            while head:
                head.pop(0)()
            # (an ordinary block's first iteration ends with the wait that covers the head's parameter loads; here nothing else is in flight
            # that matters -- the operand pieces requested ahead belong to no tile)
            a("s_waitcnt", vmcnt=0, lgkmcnt=0)
            for op in ops:
                op()
            a("s_branch", L_end)
        L_rdone = [a.newlabel("RD0"), a.newlabel("RD1")]
        block(0)
        a("s_branch", L_block[1])          # (block 1 follows in the listing; kept explicit)
        block(1)
        a("s_branch", L_block[0])
        if DRAIN_ONLY:
            drain_only(0)
            drain_only(1)
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
# (gelu, ln, res, stats)
CLASSES = {"p": (False, False, False, False), "l": (False, True, False, False), "g": (True, False, False, False), "gl": (True, True, False, False),
           "r": (False, False, True, False), "rs": (False, False, True, True),
           # round 4: GELU + folded LayerNorm WITH the by-product statistics of what it stores -- the v half of gMLP's channel_proj1, whose
           # LayerNorm (g_mlp.py:19) is then the spatial product's operand loader (mlpk_token_gemm_ln) without a statistics pass
           "gls": (True, True, False, True)}
NKF = {c: (3, 4, 6, 12) for c in CLASSES}
# K = 192 / 384 / 576 / 768: the kernels built for one K (Q4.static).  Not 1152 (nk = 18): measured BEHIND the general f12 kernel on ViP's
# fc2 (0.0624-0.0633 vs 0.0584-0.0586 ms, profiles/r04_q4_static_probe_v1.txt): 36 unrolled iterations no longer sit in the instruction cache
NK_STATIC = (3, 6, 9, 12)
DTYPES = ("bf16", "f16")


def variants():
    for dt in DTYPES:
        for cls, (gelu, ln, res, stats) in CLASSES.items():
            for nkf in NKF[cls]:
                yield "q4_%s_%s_f%d" % (dt, cls, nkf), dict(dtype=dt, gelu=gelu, ln=ln, res=res, stats=stats, nkf=nkf)
            for nk in NK_STATIC:
                yield "q4_%s_%s_s%d" % (dt, cls, nk), dict(dtype=dt, gelu=gelu, ln=ln, res=res, stats=stats, nkf=nk, static=True)
    # tuning ablations (wrong results by construction; bits in Q4.__init__)
    for cls, nkf in (("gl", 12), ("r", 12)):
        gelu, ln, res, _ = CLASSES[cls]
        for x in (1, 2, 3, 4, 5, 13, 21, 29, 64, 65, 256, 512, 768):
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
    out.append("namespace mlpk {\nstruct Q4Variant { const char* name; const void* fn; int dtype, gelu, ln, res, stats, nkf, dbg, is_static; };\n"
               "static const Q4Variant kQ4Variants[] = {\n")
    for name, kw in table:
        dbg = kw.get("dbg", 0)
        out.append("    {\"%s\", reinterpret_cast<const void*>(&%s), %s, %d, %d, %d, %d, %d, %d, %d},\n" %
                   (name, name, "MLPK_BF16" if kw["dtype"] == "bf16" else "MLPK_F16", kw["gelu"], kw["ln"], kw["res"], kw.get("stats", False), kw["nkf"], dbg,
                    kw.get("static", False)))
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
