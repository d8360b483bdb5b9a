"""Generator of the "t4" fused token-mixing MLP (MLP-Mixer, mlp_mixer.py:16-27,34,37): one wave per SIMD, GELU as fillers.

    x[b, s, c] += sum_t W2[s, t] * gelu(sum_s' W1[t, s'] * xt[b*C + c, s'] + b1[t]) + b2[s]

Same operation and argument block as token_mlp_rr_kernel (mlpk_tokenmlp.hip); what changes is WHO overlaps with whom.  There, two
waves share a SIMD and the GELU of one was meant to run beside the MFMAs of the other -- on gfx950 those times ADD
(tools/ubench/issue_rate.hip).  Here a workgroup is 4 waves = one per SIMD with 512 registers; a wave owns 64 rows of xt (2 blocks of 32)
for all 196 tokens and runs, in ONE instruction stream, three stages of a software pipeline over the groups of 32 hidden units:

    iteration g:   MFMA  fc2(g-2)  28 x v_mfma_f32_32x32x16   D2[rb][tb] += h(g-2)[rb][kk] x W2frag(tb, kk)
                   MFMA  fc1(g)    26 x                        xg[rb] = b1(g) + sum_ks W1frag(ks) x X[rb][ks]      (13 k-steps: 196 tokens fit 208)
                   VALU  gelu(g-1) on the 32 fp32 values per lane that fc1(g-1) left, rounded into h(g-1): placed between the MFMAs

with the first product computed "transposed" (hidden x rows), so that a lane's 16 accumulators of a block are -- after GELU and
rounding -- exactly two A fragments of the second product once W2's columns are stored in that order inside every group of 32
(layout 2 of mlpk_token_mlp_layout: k slot 16 kk + 8 h + e  <-  hidden 16 kk + 8 (e >> 2) + 4 h + (e & 3)).  The hidden never leaves
the registers.  W1 / W2 groups stream through two 2-stage LDS rings by LDS-DMA one iteration ahead (one barrier per iteration);
LDS rows are PADDED by 16 bytes (528 / 80-byte pitch) instead of XOR-swizzled: the 16 lanes of a ds_read_b128 group then fall on 16
different bank quads and a k-step is an immediate offset of the one address register.
Pipeline fill and drain: shape 0 runs every iteration with all three stages -- groups outside [0, G) multiply by the zero W2 group
of the packed tensor, so they add nothing -- and works for any G; shapes 1 / 2 (G odd / even, what the host launches) emit the first
and last iterations of a tile with only the stages that have work, and request the next tile's X when the drain begins.
Epilogue per tile: accumulators (started from b2) staged as fp32 through LDS (two buffers), read back as (token, 8 channels) items,
residual added in fp32, one rounding, 16-byte stores of whole 128-byte lines; optional by-product (sum, sum of squares) per token
over the wave's 64 channels for the LayerNorm that follows (mlpk.h: stats).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa import A, F, S, V, Asm, Neg, Reg, h2bits  # noqa: E402
from q4gen import GELU, GELU_FORM, GELU_H2, GELU_RAW, GELU_SIG, SQRT2, Alloc, h2_gelu_ops, h2b_gelu_ops, sig_gelu_ops  # noqa: E402

KA = dict(xt=0, w1=8, w2=16, b1=24, b2=32, x=40, stats=48, prof=56,
          M=64, G=68, ldxt=72, ldx=76, ntiles=80, tpi=84, tpi_magic=88, grid=92, stat_ld=96, nit=100, lead=104, S=108,
          ln_mean=112, ln_rstd=120, gamma=128, beta=136)
ARG_BYTES = 144
LN_WAVE = 2 * 8 * 272                    # LayerNorm loader: two buffers of 8 channel planes ([8 channels-of-8][8 token pairs] dwords + 16 B) per wave

W1_PITCH, W1_PIECES = 528, 17            # 32 hidden rows x (512 B + 16): 16896 B
W2_PITCH, W2_PIECES = 80, 18             # 224 tokens x (64 B + 16): 17920 B
W1_STAGE, W2_STAGE = W1_PIECES * 1024, W2_PIECES * 1024
W1_OFF, W2_OFF = 0, 2 * W1_STAGE
B1_OFF = W2_OFF + 2 * W2_STAGE           # 1024 floats: entry j = bias of hidden j - 64 (two zero groups in front, zeros behind)
STG_PITCH = 272                          # fp32 staging row: 64 channels + 16 B
STG_WAVE = 32 * STG_PITCH
STG_OFF = B1_OFF + 4096
STG_BUF = 4 * STG_WAVE                   # two staging buffers (token blocks alternate): block tb + 1 is written while tb is read back
LDS_BYTES = STG_OFF + 2 * STG_BUF
W2_GROUP_BYTES = 224 * 64                # one hidden group of the packed W2: 224 token rows x 32 k slots
W1_MAGIC, W2_MAGIC = 1986, 13108         # (o * magic) >> 20 == o // pitch for the multiples of 16 below one stage
assert all((o * W1_MAGIC) >> 20 == o // W1_PITCH for o in range(0, W1_STAGE, 16))
assert all((o * W2_MAGIC) >> 20 == o // W2_PITCH for o in range(0, W2_STAGE, 16))


class T4:
    NKS, NTB = 13, 7                       # K: 196 tokens fit 13 steps of 16 (208; the 14th of xt's 224 columns is padding only);
                                           # 196 tokens = 7 blocks of 32 for the second product (the last one holds 4)
    NXA = 8                                # X fragments kept in the 32 AGPRs the accumulators leave over
    DEPTH = 3                              # W fragments read ahead of their MFMAs (ring of 4 register quads)

    def __init__(self, dtype="bf16", stats=False, dbg=0, name=None, shape=0, ln=False, h2=False):
        # ln: the token LayerNorm + transpose is this kernel's X loader -- x itself is read (token-major rows, statistics given),
        # normalised, transposed through LDS into the fragment registers; `xt` is not used (shaped kernels only)
        assert not ln or shape
        self.ln = ln
        # shape: 0 = every iteration runs all three stages (pipeline fill / drain on dummy groups; any G); 1 / 2 = G odd (>= 3) / even
        # (>= 2): the fill and drain iterations are emitted without the stages that have nothing to do, the next tile's X is
        # requested two iterations before the tile ends
        self.shape = shape
        # tuning ablations (wrong results by construction): 1 no LDS-DMA, 2 no GELU fillers, 4 no epilogue stores, 16 no residual loads,
        # 32 no X loads
        self.dtype, self.stats, self.dbg = dtype, stats, dbg
        # h2 (round 5, bf16 storage only; mlpk.h layout 3): the GELU in packed f16 and the hidden KEPT in f16 -- W2 is packed as f16 and the second
        # product runs on the f16 MFMA; x, W1, the first product and everything the kernel stores stay bf16
        assert not h2 or dtype == "bf16"
        self.h2 = h2
        self.name = name or "t4_%s%s%s%s" % (dtype, "_h2" if h2 else "", "_st" if stats else "", ("", "_odd", "_even")[shape] + ("_ln" if ln else ""))
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.mfma2 = "v_mfma_f32_32x32x16_f16" if h2 else self.mfma          # the second product
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.dot = "v_dot2c_f32_bf16" if dtype == "bf16" else "v_dot2c_f32_f16"
        self.raw = GELU_RAW[dtype]
        self.coefs = GELU[dtype][1]
        self.a = Asm()
        self.build()

    # ------------------------------------------------------------------ registers
    def regs(self):
        s = Alloc("s", 33, 100)
        v = Alloc("v", 1, 256)
        self.s_karg, self.s_bid, self.v_tid = S(0, 2), S(2), V(0)
        self.p = {k: S(4 + 2 * i, 2) for i, k in enumerate(["xt", "w1", "w2", "b1", "b2", "x", "stats", "prof"])}
        self.k = {k: S(20 + i) for i, k in enumerate(["M", "G", "ldxt", "ldx", "ntiles", "tpi", "tpi_magic", "grid", "stat_ld", "nit", "lead", "S"])}
        self.s_wave, self.s_tile, self.s_next, self.s_has = s("wave"), s("tile"), s("next"), s("has")
        self.s_g, self.s_cnt = s("g"), s("cnt")
        self.s_wv1k, self.s_w2x = s("wv1k"), s("w2x")
        self.s_w1src, self.s_w2src = s("w1src", 2, 2), s("w2src", 2, 2)
        self.s_xb = [s("xb%d" % rb, 2, 2) for rb in range(2)]
        self.s_ocur, self.s_lcur = s("ocur", 2, 2), s("lcur", 2, 2)      # store / residual-load cursors over the token rows of x
        self.s_scur = s("scur", 2, 2) if self.stats else None
        self.s_tok8 = s("tok8")
        self.s_r2 = s("r2")
        self.sig = GELU_FORM[self.dtype] == "sig" and not self.h2
        self.h2b = GELU_FORM[self.dtype] == "h2b" and not self.h2         # (layout 2 of a bf16 model: the library-wide bf16 GELU, the hidden bf16)
        self.s_k0, self.s_k1 = (s("gk0"), s("gk1")) if (self.sig or self.h2b) else (None, None)
        # packed-f16 constants: two in SGPRs (as many as the logistic form held -- the asm block may clobber no more scalar registers than
        # that: "inline assembly requires more registers than available"), c0 .. c5 in VGPRs
        self.s_hscale, self.s_hc6 = (s("hscale"), s("hc6")) if self.h2 else (None, None)
        self.s_mask = s("mask", 2, 2)
        self.s_t = [s("t%d" % i) for i in range(6)]
        self.s_t64 = s("t64", 2, 2)
        self.s_prof0 = s("prof0", 2, 2)
        self.s_prof1, self.s_epi = s("prof1", 2, 2), s("epi")          # tuning: cycles spent in the tile epilogues
        self.s_ph = [s("ph%d" % i) for i in range(3)]                  # ... in the fill iterations, the steady loop, the drain iterations
        if self.ln:
            self.s_lnp = s("lnp", 8, 4)                                # ln_mean, ln_rstd, gamma, beta
            self.s_lx, self.s_lmean, self.s_lrstd = s("lx", 2, 2), s("lmean", 2, 2), s("lrstd", 2, 2)
            self.s_l16 = s("l16")                                      # bytes of 16 token rows of x
        self.ns = s.next
        # vector registers.  AGPRs: D2[rb][tb] at 16 (7 rb + tb); a[224:255] = the last NXA X fragments of row block 1
        self.X = [[None] * self.NKS for _ in range(2)]
        na = 0
        for rb in range(2):
            for ks in range(self.NKS):
                if rb == 1 and ks >= self.NKS - self.NXA:
                    self.X[rb][ks] = A(224 + 4 * na, 4)
                    na += 1
                else:
                    self.X[rb][ks] = v("X%d_%d" % (rb, ks), 4, 2)
        self.Wf = [v("Wf%d" % b, 4, 2) for b in range(4)]
        self.b1 = v("b1", 16, 2)
        self.xg = [[v("xg%d_%d" % (par, rb), 16, 2) for rb in range(2)] for par in range(2)]
        self.h = [[[v("h%d_%d_%d" % (par, rb, kk), 4, 2) for kk in range(2)] for rb in range(2)] for par in range(2)]
        self.tmp = [[v("g%s%d" % (n, ch), 2, 2) for n in "tuq"] for ch in range(2)]     # GELU scratch: 4 chains x (t, u, q)
        flat = [r_[e] for grp_ in self.tmp for r_ in grp_ for e in range(2)]
        self.tmp_t, self.tmp_u, self.tmp_q = flat[0:4], flat[4:8], flat[8:12]
        self.v_c0 = v("c0")
        self.v_hc = [v("hc%d" % j) for j in range(1, 6)] if (self.h2 or self.h2b) else None
        self.v_hp = [v("hp%d" % j) for j in range(2)] if self.h2b else None
        self.v_nz = v("negzero") if self.h2b else None
        self.v_w1rd, self.v_w2rd, self.v_b1rd = v("w1rd"), v("w2rd"), v("b1rd")
        self.v_w1off = [v("w1off%d" % i) for i in range(5)]
        self.v_w2off = [v("w2off%d" % i) for i in range(5)]
        self.v_xoff = v("xoff")
        self.v_stw, self.v_strd, self.v_ooff = v("stw"), v("strd"), v("ooff")
        self.v_soff = v("soff") if self.stats else None
        self.v_b2 = [v("b2_%d" % tb) for tb in range(self.NTB)]        # b2 of token 32 tb + (lane & 31): what D2 starts from
        self.nv = v.next
        # the epilogue lives in registers that are dead by then
        xg, h = self.xg, self.h
        self.e_acc = [xg[0][0][4 * i:4 * i + 4] for i in range(4)]
        self.e_in = [xg[0][1][4 * i:4 * i + 4] for i in range(4)] + [xg[1][0][4 * i:4 * i + 4] for i in range(4)]
        hb = h[0][0][0].idx
        self.e_res = [[xg[1][1][4 * i:4 * i + 4] for i in range(4)], [Reg("v", hb + 4 * i, 4) for i in range(4)]]
        assert h[0][1][1].idx == hb + 12
        self.e_t = self.tmp[0][0]                      # pair: a residual dword as two fp32
        self.e_sp = self.tmp[0][1]                     # pair: (sum, sum of squares)
        self.e_ones = self.tmp[0][2][0]

    def D2(self, rb, tb):
        return A(16 * (self.NTB * rb + tb), 16)

    # ------------------------------------------------------------------ helpers
    def ds(self, op, *args, **kw):
        self.a(op, *args, **kw)
        self.lgkm_issued += 1
        return self.lgkm_issued - 1

    def wait_lds(self, idx):
        self.a("s_waitcnt", lgkmcnt=min(self.lgkm_issued - idx - 1, 15))

    def vload(self, op, *args, **kw):
        self.a(op, *args, **kw)
        self.vm_loads += 1
        return self.vm_loads - 1

    def wait_vload(self, idx):
        self.a("s_waitcnt", vmcnt=min(self.vm_loads - idx - 1, 63))

    def add64(self, dst, src, lo, hi=None):
        self.a("s_add_u32", dst[0], src[0], lo)
        self.a("s_addc_u32", dst[1], src[1], hi if hi is not None else 0)

    def mul64(self, dst, a_, b_, shift):
        """dst(64) = (a * b) << shift, 32-bit unsigned factors"""
        a, t = self.a, self.s_t
        a("s_mul_hi_u32", dst[1], a_, b_)
        a("s_mul_i32", dst[0], a_, b_)
        if shift:
            a("s_lshl_b32", dst[1], dst[1], shift)
            a("s_lshr_b32", t[5], dst[0], 32 - shift)
            a("s_or_b32", dst[1], dst[1], t[5])
            a("s_lshl_b32", dst[0], dst[0], shift)

    # ------------------------------------------------------------------ GELU of one group as fillers
    def gelu_ops(self, par_in, par_out):
        """xg[par_in] (fc1 + b1 of a group, the accumulator layout) -> h[par_out][rb][kk] (A fragments of the second product).
        Plain fp32 VALU operations, four independent chains abreast: v_pk_*_f32 would halve the count but does not overlap with
        the matrix pipe (one v_pk_fma_f32 between two MFMAs costs 17 cycles, tools/ubench/q4_slots.py, profiles/r03_t4_issue_slots.txt).
        The operation sequence of gelu16_f (mlpk_common.h): bf16 11 operations per element (raw form), f16 15 (centred form)."""
        a = self.a
        ops = []

        def E(*x, **kw):
            ops.append(lambda: a(*x, **kw))
        T, U, Q = self.tmp_t, self.tmp_u, self.tmp_q
        for rb in range(2):
            for grp in range(4):                      # accumulator registers 4 grp .. 4 grp + 3
                x = [self.xg[par_in][rb][4 * grp + r] for r in range(4)]
                scale, c = GELU[self.dtype]
                kk, e0 = grp >> 1, 4 * (grp & 1)
                hreg = self.h[par_out][rb][kk]
                if self.h2:
                    # accumulator registers (4 grp + 2 k, 4 grp + 2 k + 1) -> the packed pair (e0 >> 1) + k of A fragment kk
                    h2_gelu_ops(E, x, [hreg[(e0 >> 1) + k] for k in range(2)], T[0:2], U[0:2], Q[0:2], self.v_c0, self.s_hscale, self.v_hc + [self.s_hc6])
                    continue
                if self.sig:
                    sig_gelu_ops(E, x, Q, self.v_c0, self.s_k1, self.s_k0)
                elif self.h2b:
                    h2b_gelu_ops(E, x, self.v_hp, T[0:2], U[0:2], Q[0:2], [self.v_c0] + self.v_hc, self.s_k0, self.s_k1, self.v_nz)
                elif self.raw:
                    for r in range(4):
                        E("v_med3_f32", T[r], x[r], F(-scale), F(scale))
                    for r in range(4):
                        E("v_mul_f32", U[r], T[r], T[r])
                else:
                    for r in range(4):
                        E("v_mul_f32", T[r], F(scale), x[r])
                    for r in range(4):
                        E("v_med3_f32", T[r], T[r], Neg(self.s_r2), self.s_r2)
                    for r in range(4):
                        E("v_fma_f32", U[r], T[r], T[r], F(-1.0))
                if not self.sig and not self.h2b:
                    for r in range(4):
                        E("v_fmaak_f32", Q[r], U[r], self.v_c0, F(c[1]))
                    for kx in range(2, len(c)):
                        for r in range(4):
                            E("v_fmaak_f32", Q[r], Q[r], U[r], F(c[kx]))
                    for r in range(4):
                        E("v_fma_f32", T[r], T[r], Q[r], F(0.5))
                    for r in range(4):
                        E("v_mul_f32", x[r], x[r], T[r])
                # accumulator register 8 kk + e -> A fragment kk, packed pair e >> 1
                E(self.cvt, hreg[e0 >> 1], x[0], x[1])
                E(self.cvt, hreg[(e0 >> 1) + 1], x[2], x[3])
        return ops

    # ------------------------------------------------------------------ DMA sources of group index gn (in s_t[0]; may be out of range)
    def dma_sources(self):
        a, t, k, p = self.a, self.s_t, self.k, self.p
        a("s_add_i32", t[1], k["G"], -1)
        a("s_min_i32", t[3], t[0], t[1])
        a("s_max_i32", t[3], t[3], 0)                         # any valid W1 group will do outside [0, G): its W2 slab is zero
        a("s_lshl_b32", t[3], t[3], 14)                       # 32 rows x 512 B
        self.add64(self.s_w1src, p["w1"], t[3])
        a("s_add_i32", t[3], t[0], -2)                        # the second product runs two groups behind
        a("s_cmp_lt_i32", t[3], 0)
        a("s_cselect_b32", t[3], k["G"], t[3])                # group G of the packed W2 = zeros
        a("s_mul_i32", t[3], t[3], W2_GROUP_BYTES)
        self.add64(self.s_w2src, p["w2"], t[3])

    def dma_piece(self, kind, i, stage):
        """(m0 value as (sgpr, literal), offset register, source) of piece i of this wave"""
        if kind == "w1":
            if i < 4:
                return self.s_wv1k, W1_OFF + stage * W1_STAGE + i * 4096, self.v_w1off[i], self.s_w1src
            return None, W1_OFF + stage * W1_STAGE + 16384, self.v_w1off[4], self.s_w1src
        if i < 4:
            return self.s_wv1k, W2_OFF + stage * W2_STAGE + i * 4096, self.v_w2off[i], self.s_w2src
        return self.s_w2x, W2_OFF + stage * W2_STAGE + 16384, self.v_w2off[4], self.s_w2src

    def emit_m0(self, kind, i, stage):
        sg, lit, _, _ = self.dma_piece(kind, i, stage)
        if sg is None:
            self.a("s_mov_b32", "m0", lit)
        else:
            self.a("s_add_u32", "m0", sg, lit)

    def emit_dma(self, kind, i, stage):
        _, _, voff, src = self.dma_piece(kind, i, stage)
        self.a("global_load_lds_dwordx4", voff, src)

    # ------------------------------------------------------------------ one pipeline iteration (static parity)
    def iteration(self, par, fc2=True, fc1=True, gelu=True, xload=False, vm_allow=0):
        """fc2(g-2) from W2 stage par and h[par]; fc1(g) from W1 stage par into xg[par]; gelu(g-1): xg[1-par] -> h[1-par];
        LDS-DMA of W1(g+1) / W2(g-1) into the stages 1-par.  Stages can be left out (fill / drain of the pipeline); xload: the next
        tile's X fragments are requested behind this iteration's DMA pieces (the registers are dead: no fc1 from here to the tile's
        end); vm_allow: vector-memory operations that may stay in flight across the iteration's barrier (those X loads)."""
        a, t, k = self.a, self.s_t, self.k
        if xload:
            a("s_add_u32", self.s_next, self.s_tile, k["grid"])
            a("s_cmp_lt_u32", self.s_next, k["ntiles"])
            a("s_cselect_b32", self.s_has, 1, 0)
            a("s_cselect_b32", t[4], self.s_next, self.s_tile)      # no next tile: this one again (the wait counts stay static)
            self.tile_xbase(t[4])
        # group of the next iteration (wraps to the first iteration of the next tile)
        a("s_add_i32", t[0], self.s_g, 1)
        a("s_add_i32", t[1], k["G"], 2)
        a("s_sub_u32", t[2], 0, k["lead"])
        a("s_cmp_ge_i32", t[0], t[1])
        a("s_cselect_b32", t[0], t[2], t[0])
        self.dma_sources()
        a("s_waitcnt", vmcnt=vm_allow, lgkmcnt=0)
        a("s_barrier")
        self.lgkm_issued = 0
        fill = [] if ((self.dbg & 2) or not gelu) else self.gelu_ops(1 - par, 1 - par)
        dma = [] if (self.dbg & 1) else [(kind, i) for i in range(5) for kind in ("w1", "w2")]
        xl = []
        if xload and not (self.dbg & 32):
            xl = [(rb, ks) for rb in range(2) for ks in range(self.NKS)]
        state = {"done": 0, "gap": 0}
        frs = ([("w2", kk, tb) for kk in range(2) for tb in range(self.NTB)] if fc2 else []) + \
              ([("w1", ks, None) for ks in range(self.NKS)] if fc1 else [])
        ngaps, total = 2 * len(frs), len(fill)

        def emit_xload():
            rb, ks = xl.pop(0)
            self.a("global_load_dwordx4", self.X[rb][ks], self.v_xoff, self.s_xb[rb], offset=32 * ks)
        if not frs:
            # nothing to multiply (the lead iteration of an odd group count): the DMA pieces alone
            while dma:
                kind, i = dma.pop(0)
                self.emit_m0(kind, i, 1 - par)
                a("s_nop", 0)
                self.emit_dma(kind, i, 1 - par)
            assert not fill and not xl
            a("s_add_i32", self.s_g, self.s_g, 1)
            a("v_add_u32", self.v_b1rd, 128, self.v_b1rd)
            return

        def gap():
            state["gap"] += 1
            target = (total * state["gap"] + ngaps - 1) // ngaps
            while state["done"] < min(target, total):
                fill[state["done"]]()
                state["done"] += 1

        def mfma(*margs, op=None):
            if dma:
                self.emit_m0(dma[0][0], dma[0][1], 1 - par)
            a(op or self.mfma, *margs)
            if dma:
                kind, i = dma.pop(0)
                self.emit_dma(kind, i, 1 - par)
            elif xl:
                emit_xload()
            gap()

        def read(n):
            kind, x, tb = frs[n]
            if kind == "w2":
                return self.ds("ds_read_b128", self.Wf[n & 3], self.v_w2rd, offset=par * W2_STAGE + tb * 32 * W2_PITCH + x * 32)
            return self.ds("ds_read_b128", self.Wf[n & 3], self.v_w1rd, offset=W1_OFF + par * W1_STAGE + x * 32)

        def b1_reads():                          # b1(g): the accumulator initialiser of fc1(g)
            for qd in range(4):
                self.ds("ds_read_b128", self.b1[4 * qd:4 * qd + 4], self.v_b1rd, offset=32 * qd)
        if fc1 and not fc2:
            b1_reads()                           # (LDS operations complete in order: the wait for fragment 0 covers them)
        rd = {}
        for n in range(self.DEPTH):
            rd[n] = read(n)
        for n, (kind, x, tb) in enumerate(frs):
            if n + self.DEPTH < len(frs):
                rd[n + self.DEPTH] = read(n + self.DEPTH)
            if n == 6 and fc1 and fc2:
                b1_reads()
            self.wait_lds(rd[n])
            wf = self.Wf[n & 3]
            for rb in range(2):
                if kind == "w2":
                    mfma(self.D2(rb, tb), self.h[par][rb][x], wf, self.D2(rb, tb), op=self.mfma2)
                else:
                    mfma(self.xg[par][rb], wf, self.X[rb][x], self.b1 if x == 0 else self.xg[par][rb])
        while state["done"] < total:
            fill[state["done"]]()
            state["done"] += 1
        while xl:
            emit_xload()
        assert not dma
        a("s_add_i32", self.s_g, self.s_g, 1)
        a("v_add_u32", self.v_b1rd, 128, self.v_b1rd)

    # ------------------------------------------------------------------ tile geometry
    def tile_xbase(self, tile):
        """s_xb[rb] = xt + ((tile * 256 + 64 wave + 32 rb) * ldxt) * 2"""
        a, t, k = self.a, self.s_t, self.k
        a("s_lshl_b32", t[0], tile, 8)
        a("s_lshl_b32", t[1], self.s_wave, 6)
        a("s_add_u32", t[0], t[0], t[1])
        self.mul64(self.s_t64, t[0], k["ldxt"], 1)
        self.add64(self.s_xb[0], self.p["xt"], self.s_t64[0], self.s_t64[1])
        a("s_lshl_b32", t[0], k["ldxt"], 6)                   # 32 rows
        self.add64(self.s_xb[1], self.s_xb[0], t[0])

    def x_loads(self):
        if self.dbg & 32:
            return
        for rb in range(2):
            for ks in range(self.NKS):
                self.vload("global_load_dwordx4", self.X[rb][ks], self.v_xoff, self.s_xb[rb], offset=32 * ks)

    def ln_loader(self, tile):
        """X[rb][ks] of tile `tile` from x itself: LayerNorm_C(x[b, t, :]) of the wave's 64 channels, transposed (mlp_mixer.py:34 with
        :6-13; the statistics of the rows come in as ln_mean / ln_rstd).  Per k-step of 16 tokens: a lane = (token pair lane >> 3,
        8 channels lane & 7) loads two 16-byte pieces of full 128-byte lines, normalises 16 values, packs (token 2p, token 2p+1) pairs
        per channel and writes them as [channel][token pair] words into a per-wave LDS tile (in-order LDS: no wait between the writes
        and the two ds_read_b128 that pull the fragments of both row blocks out).  Runs with the pipeline idle: every pipeline register is
        scratch.  Tokens behind S (the last k-step holds 4) keep whatever the registers held: W1's columns there are zero."""
        a, t, k, p = self.a, self.s_t, self.k, self.p
        bf = self.dtype == "bf16"
        xg = self.xg
        ldr = [[xg[0][0][0:4], xg[0][0][4:8], xg[0][0][8:10], xg[0][0][10:12]],          # two load buffers: x of tokens 2p / 2p+1 (quads), mean pair, rstd pair
               [xg[0][1][0:4], xg[0][1][4:8], xg[0][1][8:10], xg[0][1][10:12]]]
        gam, bet = [xg[1][0][e] for e in range(8)], [xg[1][0][8 + e] for e in range(8)]
        fa, fb = [xg[1][1][e] for e in range(8)], [xg[1][1][8 + e] for e in range(8)]
        hb = self.h[0][0][0].idx
        pk = [Reg("v", hb + e) for e in range(8)]
        lane, pp, c8, va0, va1, vst, vgb, vw = [Reg("v", hb + 8 + i) for i in range(8)]
        vr = [Reg("v", hb + 16), Reg("v", hb + 17)]
        x1, x2, mneg = Reg("v", hb + 18), Reg("v", hb + 19), [Reg("v", hb + 20), Reg("v", hb + 21)]
        # ---- scalars of the tile
        a("s_lshl_b32", t[0], tile, 1)
        a("s_mul_hi_u32", t[1], t[0], k["tpi_magic"])         # img
        a("s_mul_i32", t[0], t[1], k["tpi"])
        a("s_sub_u32", t[0], tile, t[0])
        a("s_lshl_b32", t[0], t[0], 8)
        a("s_lshl_b32", t[2], self.s_wave, 6)
        a("s_add_u32", t[2], t[2], t[0])                      # c0: first channel of the wave
        a("s_mul_i32", t[3], t[1], k["S"])                    # first token row of the image
        self.mul64(self.s_t64, t[3], k["ldx"], 1)
        a("s_lshl_b32", t[0], t[2], 1)
        self.add64(self.s_t64, self.s_t64, t[0])
        self.add64(self.s_lx, p["x"], self.s_t64[0], self.s_t64[1])
        a("s_lshl_b32", t[0], t[3], 2)
        self.add64(self.s_lmean, self.s_lnp[0:2], t[0])
        self.add64(self.s_lrstd, self.s_lnp[2:4], t[0])
        a("s_lshl_b32", t[0], t[2], 2)
        self.add64(self.s_t64, self.s_lnp[4:6], t[0])         # gamma + c0
        a("s_lshl_b32", self.s_l16, k["ldx"], 5)
        # ---- lane constants
        a("v_and_b32", lane, 63, self.v_tid)
        a("v_lshrrev_b32", pp, 3, lane)
        a("v_and_b32", c8, 7, lane)
        a("v_lshlrev_b32", vgb, 5, c8)
        a("global_load_dwordx4", Reg("v", gam[0].idx, 4), vgb, self.s_t64)
        a("global_load_dwordx4", Reg("v", gam[4].idx, 4), vgb, self.s_t64, offset=16)
        self.add64(self.s_t64, self.s_lnp[6:8], t[0])         # beta + c0
        a("v_mul_lo_u32", x1, pp, k["ldx"])
        a("v_lshlrev_b32", x1, 2, x1)                         # 2 p rows, 2 bytes
        a("v_lshl_add_u32", va0, c8, 4, x1)
        a("s_lshl_b32", t[0], k["ldx"], 1)
        a("v_add_u32", va1, t[0], va0)
        a("v_lshlrev_b32", vst, 3, pp)
        a("global_load_dwordx4", Reg("v", bet[0].idx, 4), vgb, self.s_t64)
        a("global_load_dwordx4", Reg("v", bet[4].idx, 4), vgb, self.s_t64, offset=16)
        a("s_mul_i32", t[0], self.s_wave, LN_WAVE)
        a("s_add_u32", t[0], t[0], STG_OFF)
        a("v_lshl_add_u32", x1, c8, 3, pp)                    # c8 * 8 + p
        a("v_lshl_add_u32", vw, x1, 2, t[0])
        a("v_and_b32", x1, 31, lane)                          # j
        a("v_lshrrev_b32", x2, 5, lane)                       # h
        for rb in range(2):
            a("v_and_b32", vr[rb], 7, x1)                     # e = j & 7
            a("v_mul_u32_u24", vr[rb], 272, vr[rb])
            a("v_lshrrev_b32", mneg[0], 3, x1)                # j >> 3
            a("v_add_u32", mneg[0], 4 * rb, mneg[0])          # c8' = 4 rb + (j >> 3)
            a("v_lshl_add_u32", mneg[0], mneg[0], 1, x2)      # 2 c8' + h
            a("v_lshl_add_u32", vr[rb], mneg[0], 4, vr[rb])   # + (c8' * 8 + 4 h) * 4
            a("v_add_u32", vr[rb], t[0], vr[rb])
        nks = self.NKS

        def loads(ks):
            buf = ldr[ks & 1]
            last = ks == nks - 1
            if last:
                a("s_mov_b32", self.s_t64[0], 0xFFFF)         # tokens 192 .. 195: token pairs 0 and 1 = lanes 0 .. 15
                a("s_mov_b32", self.s_t64[1], 0)
                a("s_mov_b64", "exec", self.s_t64)
            a("global_load_dwordx4", buf[0], va0, self.s_lx)
            a("global_load_dwordx4", buf[1], va1, self.s_lx)
            a("global_load_dwordx2", buf[2], vst, self.s_lmean)
            a("global_load_dwordx2", buf[3], vst, self.s_lrstd)
            if last:
                a("s_mov_b64", "exec", -1)
            else:
                self.add64(self.s_lx, self.s_lx, self.s_l16)
                self.add64(self.s_lmean, self.s_lmean, 64)
                self.add64(self.s_lrstd, self.s_lrstd, 64)
        loads(0)
        for ks in range(nks):
            if ks + 1 < nks:
                loads(ks + 1)
            a("s_waitcnt", vmcnt=4 if ks + 1 < nks else 0)
            buf = ldr[ks & 1]
            for tok, f in ((0, fa), (1, fb)):
                q = buf[tok]
                for d in range(4):
                    if bf:
                        a("v_lshlrev_b32", f[2 * d], 16, q[d])
                        a("v_and_b32", f[2 * d + 1], 0xFFFF0000, q[d])
                    else:
                        a("v_lshrrev_b32", f[2 * d + 1], 16, q[d])
                        a("v_cvt_f32_f16", f[2 * d], q[d])
                        a("v_cvt_f32_f16", f[2 * d + 1], f[2 * d + 1])
                a("v_mul_f32", mneg[tok], Neg(buf[2][tok]), buf[3][tok])        # -mean * rstd
            for tok, f in ((0, fa), (1, fb)):
                for e in range(8):
                    a("v_fma_f32", f[e], f[e], buf[3][tok], mneg[tok])
            for tok, f in ((0, fa), (1, fb)):
                for e in range(8):
                    a("v_fma_f32", f[e], f[e], gam[e], bet[e])
            for e in range(8):
                a(self.cvt, pk[e], fa[e], fb[e])
            for e in range(8):
                a("ds_write_b32", vw, pk[e], offset=(ks & 1) * 8 * 272 + e * 272)
            for rb in range(2):
                a("ds_read_b128", self.X[rb][ks], vr[rb], offset=(ks & 1) * 8 * 272)
        a("s_waitcnt", lgkmcnt=0)

    def tile_scalars(self):
        """cursors of the tile's 64 channels of this wave: x + ((img * S) * ldx + c0) * 2 and the statistics plane"""
        a, t, k, p = self.a, self.s_t, self.k, self.p
        a("s_lshl_b32", t[0], self.s_tile, 1)
        a("s_mul_hi_u32", t[1], t[0], k["tpi_magic"])         # img = tile / tpi
        a("s_mul_i32", t[0], t[1], k["tpi"])
        a("s_sub_u32", t[0], self.s_tile, t[0])
        a("s_lshl_b32", t[0], t[0], 8)
        a("s_lshl_b32", t[2], self.s_wave, 6)
        a("s_add_u32", t[2], t[2], t[0])                      # c0 = first channel of the wave
        a("s_mul_i32", t[3], t[1], k["S"])                    # first token row of the image
        self.mul64(self.s_t64, t[3], k["ldx"], 1)
        a("s_lshl_b32", t[0], t[2], 1)
        self.add64(self.s_t64, self.s_t64, t[0])
        self.add64(self.s_ocur, p["x"], self.s_t64[0], self.s_t64[1])
        a("s_mov_b32", self.s_lcur[0], self.s_ocur[0])
        a("s_mov_b32", self.s_lcur[1], self.s_ocur[1])
        if self.stats:
            a("s_lshr_b32", t[0], t[2], 6)                    # plane of 64 channels
            self.mul64(self.s_t64, t[0], k["stat_ld"], 2)
            a("s_lshl_b32", t[0], t[3], 3)                    # (sum, sum of squares) per token row
            self.add64(self.s_t64, self.s_t64, t[0])
            self.add64(self.s_scur, p["stats"], self.s_t64[0], self.s_t64[1])

    # ------------------------------------------------------------------ tile epilogue
    def res_loads(self, tb):
        """residual of token block tb in the layout of the stores: round r = tokens 8 r + (lane >> 3), 8 channels lane & 7"""
        a = self.a
        rec = []
        rounds = 4 if tb < self.NTB - 1 else 1
        if rounds == 1:
            a("s_mov_b64", "exec", self.s_mask)               # tokens 192..195: lanes 0..31 of round 0
        for r in range(rounds):
            if self.dbg & 16:
                rec.append(self.vm_loads - 1)
                continue
            rec.append(self.vload("global_load_dwordx4", self.e_res[tb & 1][r], self.v_ooff, self.s_lcur))
            self.add64(self.s_lcur, self.s_lcur, self.s_tok8)
        if rounds == 1:
            a("s_mov_b64", "exec", -1)
        return rec

    def epilogue(self, res):
        """res: the residual loads of token blocks 0 and 1 (issued behind the next tile's X: requesting them first only moved the
        wait for X to block 2 and measured slower); block tb + 2 is requested when block tb has been stored"""
        a, t = self.a, self.s_t
        if self.stats:
            a("v_mov_b32", self.e_ones, 0x3F803F80 if self.dtype == "bf16" else 0x3C003C00)
        def stage(tb):
            """accumulators (started from b2) of token block tb -> fp32 staging [token][64 channels]; returns the last write"""
            last = None
            for rb in range(2):
                for q in range(4):
                    x = self.e_acc[q]
                    for r in range(4):
                        a("v_accvgpr_read_b32", x[r], A(16 * (self.NTB * rb + tb) + 4 * q + r))
                    last = self.ds("ds_write_b128", self.v_stw, x, offset=(tb & 1) * STG_BUF + 128 * rb + 32 * q)
            return last
        staged = {0: stage(0)}
        for tb in range(self.NTB):
            rounds = 4 if tb < self.NTB - 1 else 1
            if tb + 1 < self.NTB:
                staged[tb + 1] = stage(tb + 1)
            self.wait_lds(staged[tb])
            rds = []
            for r in range(rounds):
                self.ds("ds_read_b128", self.e_in[2 * r], self.v_strd, offset=(tb & 1) * STG_BUF + 8 * r * STG_PITCH)
                rds.append(self.ds("ds_read_b128", self.e_in[2 * r + 1], self.v_strd, offset=(tb & 1) * STG_BUF + 8 * r * STG_PITCH + 16))
            for r in range(rounds):
                self.wait_lds(rds[r])
                self.wait_vload(res[tb][r])
                rr = self.e_res[tb & 1][r]
                for c in range(4):
                    pin = self.e_in[2 * r + (c >> 1)][2 * (c & 1):2 * (c & 1) + 2]
                    if self.dtype == "bf16":
                        a("v_lshlrev_b32", self.e_t[0], 16, rr[c])
                        a("v_and_b32", self.e_t[1], 0xFFFF0000, rr[c])
                    else:
                        a("v_lshrrev_b32", self.e_t[1], 16, rr[c])
                        a("v_cvt_f32_f16", self.e_t[0], rr[c])
                        a("v_cvt_f32_f16", self.e_t[1], self.e_t[1])
                    a("v_pk_add_f32", pin, pin, self.e_t)
                    a(self.cvt, rr[c], pin[0], pin[1])
                if self.stats:
                    sp = self.e_sp
                    a("v_mov_b32", sp[0], 0)
                    a("v_mov_b32", sp[1], 0)
                    for c in range(4):
                        a(self.dot, sp[0], rr[c], self.e_ones)
                        a(self.dot, sp[1], rr[c], rr[c])
                    for n_, mod in enumerate((dict(quad_perm="[1,0,3,2]"), dict(quad_perm="[2,3,0,1]"), dict(row_half_mirror=True))):
                        a("s_nop", 1 if n_ == 0 else 0)
                        a("v_add_f32_dpp", sp[0], sp[0], sp[0], **mod, row_mask="0xf", bank_mask="0xf")
                        a("v_add_f32_dpp", sp[1], sp[1], sp[1], **mod, row_mask="0xf", bank_mask="0xf")
                    a("s_mov_b32", self.s_t64[0], 0x01010101)
                    a("s_mov_b32", self.s_t64[1], 0x01010101 if rounds == 4 else 0)
                    a("s_mov_b64", "exec", self.s_t64)
                    a("global_store_dwordx2", self.v_soff, sp, self.s_scur)
                    a("s_mov_b64", "exec", -1)
                    self.add64(self.s_scur, self.s_scur, 64)
                if not (self.dbg & 4):
                    if rounds == 1:
                        a("s_mov_b64", "exec", self.s_mask)
                    a("global_store_dwordx4", self.v_ooff, rr, self.s_ocur)
                    if rounds == 1:
                        a("s_mov_b64", "exec", -1)
                self.add64(self.s_ocur, self.s_ocur, self.s_tok8)
            if tb + 2 < self.NTB:
                res[tb + 2] = self.res_loads(tb + 2)

    # ------------------------------------------------------------------ the kernel
    def build(self):
        a = self.a
        self.regs()
        self.lgkm_issued = 0
        self.vm_loads = 0
        t, k, p = self.s_t, self.k, self.p
        L_end, L_tile, L_iter, L_nonext = a.newlabel("END"), a.newlabel("TILE"), a.newlabel("ITER"), a.newlabel("NONEXT")
        a("s_load_dwordx16", S(4, 16), self.s_karg, 0)
        a("s_load_dwordx8", S(20, 8), self.s_karg, 64)
        a("s_load_dwordx4", S(28, 4), self.s_karg, 96)
        if self.ln:
            a("s_load_dwordx4", self.s_lnp[0:4], self.s_karg, KA["ln_mean"])
            a("s_load_dwordx4", self.s_lnp[4:8], self.s_karg, KA["gamma"])
        # ---- lane constants (prologue temporaries: the GELU scratch and one staging quad)
        lane, j, h, l3, l7, x, y, z = (self.tmp[0][0][0], self.tmp[0][0][1], self.tmp[0][1][0], self.tmp[0][1][1], self.tmp[0][2][0],
                                       self.tmp[0][2][1], self.tmp[1][0][0], self.tmp[1][0][1])
        a("v_and_b32", lane, 63, self.v_tid)
        a("v_lshrrev_b32", x, 6, self.v_tid)
        a("s_nop", 0)
        a("v_readfirstlane_b32", self.s_wave, x)
        a("v_and_b32", j, 31, lane)
        a("v_lshrrev_b32", h, 5, lane)
        a("v_lshrrev_b32", l3, 3, lane)
        a("v_and_b32", l7, 7, lane)
        a("s_mov_b32", self.s_r2, F(SQRT2))
        if self.h2:
            a("v_mov_b32", self.v_c0, h2bits(GELU_H2["coefs"][0]))
            a("s_mov_b32", self.s_hscale, h2bits(GELU_H2["scale"]))
            for cj in range(5):
                a("v_mov_b32", self.v_hc[cj], h2bits(GELU_H2["coefs"][cj + 1]))
            a("s_mov_b32", self.s_hc6, h2bits(GELU_H2["coefs"][6]))
        elif self.h2b:
            a("v_mov_b32", self.v_c0, h2bits(GELU_H2["coefs"][0]))
            for cj in range(5):
                a("v_mov_b32", self.v_hc[cj], h2bits(GELU_H2["coefs"][cj + 1]))
            a("s_mov_b32", self.s_k0, h2bits(GELU_H2["scale"]))
            a("s_mov_b32", self.s_k1, h2bits(GELU_H2["coefs"][6]))
            a("v_mov_b32", self.v_nz, 0x80000000)
        elif self.sig:
            a("v_mov_b32", self.v_c0, F(GELU_SIG[self.dtype][2]))
            a("s_mov_b32", self.s_k1, F(GELU_SIG[self.dtype][1]))
            a("s_mov_b32", self.s_k0, F(GELU_SIG[self.dtype][0]))
        else:
            a("v_mov_b32", self.v_c0, F(self.coefs[0]))
        a("s_mov_b32", self.s_mask[0], -1)
        a("s_mov_b32", self.s_mask[1], 0)
        a("s_lshl_b32", self.s_wv1k, self.s_wave, 10)
        a("s_and_b32", self.s_w2x, self.s_wave, 1)
        a("s_lshl_b32", self.s_w2x, self.s_w2x, 10)
        a("v_mul_u32_u24", x, W1_PITCH, j)
        a("v_lshl_add_u32", self.v_w1rd, h, 4, x)
        a("v_mul_u32_u24", x, W2_PITCH, j)
        a("v_lshl_add_u32", self.v_w2rd, h, 4, x)
        a("v_add_u32", self.v_w2rd, W2_OFF, self.v_w2rd)          # (DS offsets are 16 bits: the region bases live in the registers)
        a("v_lshlrev_b32", self.v_b1rd, 4, h)
        a("v_add_u32", self.v_b1rd, B1_OFF, self.v_b1rd)             # + 128 (group + 2) at the start of a tile, walks back at its end
        # LDS-DMA source offsets: piece -> LDS offset o = piece * 1024 + lane * 16 -> (row, column) of the padded stage -> source byte
        for i in range(5):
            for kind, magic, pitch, colmax, rowmax, rshift, dst in (("w1", W1_MAGIC, W1_PITCH, 496, 31, 9, self.v_w1off[i]),
                                                                    ("w2", W2_MAGIC, W2_PITCH, 48, 223, 6, self.v_w2off[i])):
                sg, lit, _, _ = self.dma_piece(kind, i, 0)
                lit -= W1_OFF if kind == "w1" else W2_OFF
                if sg is None:
                    a("s_mov_b32", t[0], lit)
                else:
                    a("s_add_u32", t[0], sg, lit)
                a("v_lshl_add_u32", x, lane, 4, t[0])               # o
                a("v_mul_u32_u24", y, magic, x)
                a("v_lshrrev_b32", y, 20, y)                        # row = o / pitch
                a("v_mul_u32_u24", z, pitch, y)
                a("v_sub_u32", z, x, z)                             # column byte
                a("v_min_u32", z, colmax, z)                        # the 16 padding bytes re-read the last chunk
                a("v_min_u32", y, rowmax, y)
                a("v_lshl_add_u32", dst, y, rshift, z)
        a("s_waitcnt", lgkmcnt=0)
        a("s_memtime", self.s_prof0)
        a("s_mov_b32", self.s_epi, 0)
        for r_ in self.s_ph:
            a("s_mov_b32", r_, 0)
        # ---- argument-dependent lane constants
        a("v_mul_lo_u32", x, j, k["ldxt"])
        a("v_lshlrev_b32", x, 1, x)
        a("v_lshl_add_u32", self.v_xoff, h, 4, x)
        a("s_mul_i32", t[0], self.s_wave, STG_WAVE)
        a("s_add_u32", t[0], t[0], STG_OFF)
        a("v_mul_u32_u24", x, STG_PITCH, j)
        a("v_lshl_add_u32", x, h, 4, x)
        a("v_add_u32", self.v_stw, t[0], x)
        a("v_mul_u32_u24", x, STG_PITCH, l3)
        a("v_lshl_add_u32", x, l7, 5, x)
        a("v_add_u32", self.v_strd, t[0], x)
        a("v_mul_lo_u32", x, l3, k["ldx"])
        a("v_lshlrev_b32", x, 1, x)
        a("v_lshl_add_u32", self.v_ooff, l7, 4, x)
        if self.stats:
            a("v_lshlrev_b32", self.v_soff, 3, l3)
        a("s_lshl_b32", self.s_tok8, k["ldx"], 4)
        # ---- the bias table: 4 KiB, 16 bytes per thread
        a("v_lshlrev_b32", x, 4, self.v_tid)
        a("global_load_dwordx4", self.e_acc[0], x, p["b1"])
        a("s_mov_b32", self.s_tile, self.s_bid)
        a("s_cmp_ge_u32", self.s_tile, k["ntiles"])
        a("v_add_u32", x, B1_OFF, x)
        a("s_waitcnt", vmcnt=0)
        a("ds_write_b128", x, self.e_acc[0])
        a("s_cbranch_scc1", L_end)
        a("v_lshlrev_b32", y, 2, j)
        for tb in range(self.NTB):
            a("global_load_dword", self.v_b2[tb], y, p["b2"], offset=128 * tb)
        # ---- the first stages and the first tile's X
        a("s_sub_u32", t[0], 0, k["lead"])
        self.dma_sources()
        if not (self.dbg & 1):
            for i in range(5):
                for kind in ("w1", "w2"):
                    self.emit_m0(kind, i, 0)
                    a("s_nop", 0)
                    self.emit_dma(kind, i, 0)
        if self.ln:
            self.ln_loader(self.s_tile)
        else:
            self.tile_xbase(self.s_tile)
            self.x_loads()
        a("s_sub_u32", t[0], 2, k["lead"])
        a("s_lshl_b32", t[0], t[0], 7)
        a("v_add_u32", self.v_b1rd, t[0], self.v_b1rd)              # bias row of the first iteration's group
        a("s_waitcnt", vmcnt=0 if ((self.dbg & 33) or self.ln) else 10 + 2 * self.NKS)   # b2 (the loads before the 10 DMA pieces and the X quads)
        a.label(L_tile)
        self.tile_scalars()
        for rb in range(2):
            for tb in range(self.NTB):
                for r in range(16):
                    a("v_accvgpr_write_b32", A(16 * (self.NTB * rb + tb) + r), self.v_b2[tb])
        if self.shape == 0:
            # dummy groups run GELU on whatever these registers hold and multiply it by the zero W2 group: it has to be finite
            for par in range(2):
                for rb in range(2):
                    for r in range(16):
                        a("v_mov_b32", self.xg[par][rb][r], 0)
                    for kk in range(2):
                        for r in range(4):
                            a("v_mov_b32", self.h[par][rb][kk][r], 0)
        a("s_sub_u32", self.s_g, 0, k["lead"])
        nx = 0 if ((self.dbg & 32) or self.ln) else 2 * self.NKS
        xl_on = not self.ln
        if self.shape == 0:
            a("s_lshr_b32", self.s_cnt, k["nit"], 1)
            a.label(L_iter)
            self.iteration(0)
            self.iteration(1)
            a("s_sub_u32", self.s_cnt, self.s_cnt, 1)
            a("s_cmp_lg_u32", self.s_cnt, 0)
            a("s_cbranch_scc1", L_iter)
        else:
            # fill: group g's first product has nothing in front of it; drain: the last two groups' second products alone
            L_skip = a.newlabel("NOLOOP")

            def stamp(acc):
                """acc += cycles since the last stamp"""
                a("s_memtime", self.s_t64)
                a("s_waitcnt", lgkmcnt=0)
                a("s_sub_u32", t[0], self.s_t64[0], self.s_prof1[0])
                a("s_add_u32", acc, acc, t[0])
                a("s_mov_b32", self.s_prof1[0], self.s_t64[0])
            a("s_memtime", self.s_prof1)
            a("s_waitcnt", lgkmcnt=0)
            if self.shape == 1:                     # G odd: a lead iteration keeps the stage parity static (28 = 27 + 1 iterations for G = 25)
                self.iteration(0, fc2=False, fc1=False, gelu=False)          # g = -1: the DMA pieces only
                self.iteration(1, fc2=False, gelu=False)                     # g = 0
                self.iteration(0, fc2=False)                                 # g = 1
                first, second = 1, 0
                a("s_sub_u32", self.s_cnt, k["G"], 3)
            else:
                self.iteration(0, fc2=False, gelu=False)                     # g = 0
                self.iteration(1, fc2=False)                                 # g = 1
                first, second = 0, 1
                a("s_sub_u32", self.s_cnt, k["G"], 2)
            stamp(self.s_ph[0])
            a("s_lshr_b32", self.s_cnt, self.s_cnt, 1)
            a("s_cmp_eq_u32", self.s_cnt, 0)
            a("s_cbranch_scc1", L_skip)
            a.label(L_iter)
            self.iteration(first)
            self.iteration(second)
            a("s_sub_u32", self.s_cnt, self.s_cnt, 1)
            a("s_cmp_lg_u32", self.s_cnt, 0)
            a("s_cbranch_scc1", L_iter)
            a.label(L_skip)
            if self.shape == 1:
                self.iteration(1)                                            # g = G - 1
                stamp(self.s_ph[1])
                self.iteration(0, fc1=False, xload=xl_on)                    # g = G
                self.iteration(1, fc1=False, gelu=False, vm_allow=nx)        # g = G + 1
            else:
                stamp(self.s_ph[1])
                self.iteration(0, fc1=False, xload=xl_on)
                self.iteration(1, fc1=False, gelu=False, vm_allow=nx)
            stamp(self.s_ph[2])
        a("s_lshl_b32", t[0], k["nit"], 7)
        a("v_sub_u32", self.v_b1rd, self.v_b1rd, t[0])              # back to the first iteration's bias row
        # ---- next tile's X (the registers are dead from here on), then the epilogue
        if self.ln:
            a("s_add_u32", self.s_next, self.s_tile, k["grid"])
            a("s_cmp_lt_u32", self.s_next, k["ntiles"])
            a("s_cselect_b32", self.s_has, 1, 0)
        if self.shape == 0:
            a("s_add_u32", self.s_next, self.s_tile, k["grid"])
            a("s_cmp_lt_u32", self.s_next, k["ntiles"])
            a("s_cselect_b32", self.s_has, 1, 0)
            a("s_cbranch_scc0", L_nonext)
            self.tile_xbase(self.s_next)
            self.x_loads()
            a.label(L_nonext)
        self.lgkm_issued = 0
        self.vm_loads = 0
        a("s_memtime", self.s_prof1)
        a("s_nop", 7)                                                # the residual lands in registers the last MFMAs have just written
        a("s_nop", 7)
        res = {0: self.res_loads(0), 1: self.res_loads(1)}
        self.epilogue(res)
        a("s_memtime", self.s_t64)
        a("s_waitcnt", lgkmcnt=0)
        a("s_sub_u32", t[0], self.s_t64[0], self.s_prof1[0])
        a("s_add_u32", self.s_epi, self.s_epi, t[0])
        a("s_mov_b32", self.s_tile, self.s_next)
        a("s_cmp_lg_u32", self.s_has, 0)
        if self.ln:
            a("s_cbranch_scc0", L_end)
            self.ln_loader(self.s_tile)                              # the next tile's operands: LayerNorm + transpose of its rows of x
            a("s_branch", L_tile)
        else:
            a("s_cbranch_scc1", L_tile)
        a.label(L_end)
        # tuning: shader cycles of this workgroup -> prof[bid] (mlpk_token_mlp_debug), skipped when the pointer is null
        L_noprof = a.newlabel("NOPROF")
        a("s_memtime", self.s_t64)
        a("s_or_b32", t[0], p["prof"][0], p["prof"][1])
        a("s_cmp_eq_u32", t[0], 0)
        a("s_cbranch_scc1", L_noprof)
        a("s_waitcnt", lgkmcnt=0)
        a("s_sub_u32", self.s_t64[0], self.s_t64[0], self.s_prof0[0])
        a("s_lshl_b32", t[0], self.s_bid, 5)
        a("v_mov_b32", self.e_sp[0], t[0])
        a("v_mov_b32", self.e_t[0], self.s_t64[0])                  # (cycles of the whole kernel body, cycles of its tile epilogues)
        a("v_mov_b32", self.e_t[1], self.s_epi)
        a("global_store_dwordx2", self.e_sp[0], self.e_t, p["prof"])
        a("s_nop", 1)
        a("v_mov_b32", self.e_t[0], self.s_ph[0])                   # (fill iterations, steady loop)
        a("v_mov_b32", self.e_t[1], self.s_ph[1])
        a("global_store_dwordx2", self.e_sp[0], self.e_t, p["prof"], offset=8)
        a("s_nop", 1)
        a("v_mov_b32", self.e_t[0], self.s_ph[2])                   # (drain iterations)
        a("global_store_dword", self.e_sp[0], self.e_t[0], p["prof"], offset=16)
        a.label(L_noprof)
        a("s_waitcnt", vmcnt=0, lgkmcnt=0)
        a("s_endpgm")


def variants():
    out = []
    for dt in ("bf16", "f16"):
        for st in (False, True):
            for shape in (0, 1, 2):
                out.append(dict(dtype=dt, stats=st, shape=shape))
            for shape in (1, 2):
                out.append(dict(dtype=dt, stats=st, shape=shape, ln=True))
    for st in (False, True):          # round 5: the bf16 kernels with the packed-f16 GELU and the f16 hidden (mlpk.h layout 3)
        for shape in (0, 1, 2):
            out.append(dict(dtype="bf16", stats=st, shape=shape, h2=True))
        for shape in (1, 2):
            out.append(dict(dtype="bf16", stats=st, shape=shape, ln=True, h2=True))
    for dbg in (1, 2, 4, 3, 16, 32, 48, 52):
        out.append(dict(dtype="bf16", stats=True, dbg=dbg, name="t4_bf16_st_dbg%d" % dbg))
    return out


def kernel_text(gen):
    assert gen.nv <= 248, "the compiler needs a few VGPRs of its own (the thread id operand, WWM)"
    clob = ['"v%d"' % i for i in range(gen.nv)] + ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in range(gen.ns) if i != 32] + ['"vcc"', '"memory"']
    body = ['"s_mov_b64 s[0:1], %0\\n\\t"', '"s_mov_b32 s2, %1\\n\\t"', '"v_mov_b32 v0, %2\\n\\t"', gen.a.c_string()]
    return ("extern \"C\" __global__ void __launch_bounds__(256, 1) %s(const mlpk::T4Args args) {\n"
            "    asm volatile(\n%s\n        :\n        : \"s\"(__builtin_amdgcn_kernarg_segment_ptr()), \"s\"(blockIdx.x), \"v\"(threadIdx.x)\n"
            "        : %s);\n}\n" % (gen.name, "\n".join(body), ", ".join(clob)))


def emit(path):
    import isa
    out = ["// GENERATED by csrc/gen/t4gen.py -- do not edit.  One asm block per kernel: every register is named by the generator.\n"]
    table = []
    for kw in variants():
        g = T4(**kw)
        pr = isa.lint(g.a)
        if pr:
            raise RuntimeError("%s: %d hazard lint findings, first: %s" % (g.name, len(pr), pr[0]))
        out.append(kernel_text(g))
        table.append((g.name, kw))
    out.append("namespace mlpk {\nstruct T4Variant { const char* name; const void* fn; int dtype, stats, dbg, shape, ln, h2; };\n"
               "static const T4Variant kT4Variants[] = {\n")
    for name, kw in table:
        out.append("    {\"%s\", reinterpret_cast<const void*>(&%s), %s, %d, %d, %d, %d, %d},\n" %
                   (name, name, "MLPK_BF16" if kw["dtype"] == "bf16" else "MLPK_F16", kw["stats"], kw.get("dbg", 0), kw.get("shape", 0), kw.get("ln", False),
                    kw.get("h2", False)))
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
        import isa
        g = T4(stats=True)
        print("instructions:", len(g.a.ins), "vgprs:", g.nv, "sgprs:", g.ns)
        isa.lint(g.a, verbose=True)
