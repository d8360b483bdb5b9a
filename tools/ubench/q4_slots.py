#!/usr/bin/env python3
"""Micro-benchmark generator: what does an instruction cost when it is issued between the v_mfma_f32_32x32x16_bf16 of ONE wave
per SIMD (the regime of the generated q4 GEMM)?  Emits a .hip file with one kernel per pattern (the body is one asm block, as in
csrc/gen/q4gen.py) and a main() that launches each on every CU and prints shader cycles per MFMA.

usage: python tools/ubench/q4_slots.py /tmp/q4_slots.hip && hipcc --offload-arch=gfx950 -O2 /tmp/q4_slots.hip -o q4_slots.bin && ./q4_slots.bin
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "jittor-mlp_amd", "csrc", "gen"))
from isa import A, F, S, V, Asm  # noqa: E402

ITERS = 64          # loop iterations of 32 MFMAs


def kernel(name, pattern, per_iter=None, setup=None, vdst=False):
    """pattern(a, q): instructions after MFMA q (0..31) of an iteration; per_iter(a): at the end of an iteration"""
    a = Asm()
    # s[0:1] = kernarg, s2 = block id, v0 = thread id.  kernarg: out (8), src (8)
    a("s_load_dwordx4", S(4, 4), S(0, 2), 0)
    a("s_load_dwordx2", S(8, 2), S(0, 2), 16)
    a("v_and_b32", V(1), 63, V(0))
    a("v_lshlrev_b32", V(2), 4, V(1))                 # lane * 16: LDS read address / DMA source offset
    a("v_lshlrev_b32", V(67), 3, V(1))                # lane * 8
    a("v_mul_u32_u24", V(90), 0x3039, V(0))           # LCG seed per thread
    a("v_lshrrev_b32", V(68), 3, V(1))                # row lane >> 3 at a pitch of 6144 bytes, chunk lane & 7
    a("v_mul_u32_u24", V(68), 6144, V(68))
    a("v_and_b32", V(69), 7, V(1))
    a("v_lshl_add_u32", V(68), V(69), 4, V(68))
    a("v_lshrrev_b32", V(3), 6, V(0))
    a("s_nop", 0)
    a("v_readfirstlane_b32", S(10), V(3))             # wave
    a("s_waitcnt", lgkmcnt=0)
    a("s_lshl_b32", S(11), S(10), 14)                 # this wave's 16 KiB of LDS
    a("v_lshlrev_b32", V(66), 4, V(1))                # lane * 16
    a("s_lshl_b32", S(19), S(2), 2)
    a("s_add_u32", S(19), S(19), S(10))               # bid * 4 + wave
    a("s_lshl_b32", S(19), S(19), 18)                 # 256 KiB of fresh destination per wave
    a("s_add_u32", S(8), S(8), S(19))
    a("s_addc_u32", S(9), S(9), 0)
    a("v_add_u32", V(2), S(11), V(2))
    a("s_mov_b32", S(12), 0)
    a("s_mov_b32", S(26), 57344)
    for r in range(4, 64):
        a("v_mov_b32", V(r), 0)
    a("v_mov_b32", V(64), F(1.0))
    a("v_mov_b32", V(65), F(0.5))
    for r in range(128):
        a("v_accvgpr_write_b32", A(r), 0)
    if setup:
        setup(a)
    a("s_mov_b32", S(13), ITERS)
    a("s_barrier")
    a("s_memtime", S(14, 2))
    a("s_waitcnt", lgkmcnt=0)
    L = a.newlabel("LOOP")
    a.label(L)
    for q in range(32):
        if vdst:         # accumulators in VGPRs (two chains), B operand from AGPRs: the first product of the fused token MLP
            a("v_mfma_f32_32x32x16_bf16", V(96 + 16 * (q & 1), 16), V(4, 4), A(128 + 4 * (q & 3), 4), V(96 + 16 * (q & 1), 16))
        else:
            a("v_mfma_f32_32x32x16_bf16", A(16 * (q & 7), 16), V(4, 4), V(8, 4), A(16 * (q & 7), 16))
        pattern(a, q)
    if per_iter:
        per_iter(a)
    a("s_sub_u32", S(13), S(13), 1)
    a("s_cmp_lg_u32", S(13), 0)
    a("s_cbranch_scc1", L)
    a("s_memtime", S(16, 2))
    a("s_waitcnt", vmcnt=0, lgkmcnt=0)
    a("s_sub_u32", S(14), S(16), S(14))
    a("s_subb_u32", S(15), S(17), S(15))
    # out[(bid * 4 + wave)] = cycles (low 32 bits)
    a("s_lshl_b32", S(18), S(2), 2)
    a("s_add_u32", S(18), S(18), S(10))
    a("s_lshl_b32", S(18), S(18), 2)
    a("v_mov_b32", V(70), S(18))
    a("v_mov_b32", V(71), S(14))
    a("global_store_dword", V(70), V(71), S(4, 2))
    a("s_waitcnt", vmcnt=0)
    clob = ['"v%d"' % i for i in range(128)] + ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in range(40) if i != 32] + ['"vcc"', '"memory"']
    body = ['"s_mov_b64 s[0:1], %0\\n\\t"', '"s_mov_b32 s2, %1\\n\\t"', '"v_mov_b32 v0, %2\\n\\t"', a.c_string()]
    return ("extern \"C\" __global__ void __launch_bounds__(256, 1) %s(void* out, const void* src) {\n    asm volatile(\n%s\n        :\n"
            "        : \"s\"(__builtin_amdgcn_kernarg_segment_ptr()), \"s\"(blockIdx.x), \"v\"(threadIdx.x)\n        : %s);\n}\n"
            % (name, "\n".join(body), ", ".join(clob))).replace("(void* out, const void* src)", "(void* out, const void* src, void* dst)")


def fma(a, n, base=0):
    for k in range(n):
        r = 72 + ((base + k) % 16)
        a("v_fma_f32", V(r), V(r), V(64), V(65))


def fmaak(a, n, base=0):
    for k in range(n):
        r = 72 + ((base + k) % 16)
        a("v_fmaak_f32", V(r), V(r), V(64), F(0.123))


def ds_read(a, k):
    a("ds_read_b128", V(12 + 4 * (k % 6), 4), V(2), offset=(k % 8) * 1024)


def dma(a, k, nop=True):
    a("s_add_u32", "m0", S(11), (k % 12) * 1024)
    if nop:
        a("s_nop", 0)
    a("global_load_lds_dwordx4", V(2), S(6, 2))


PAT = []


def P(name, pattern, per_iter=None):
    PAT.append((name, pattern, per_iter))


READS68 = lambda q: (q & 7) < 6            # 6 reads per 8 MFMAs, as the GEMM
P("mfma_only", lambda a, q: None)
for n in (2, 4, 5, 6, 8):
    P("fma%d" % n, lambda a, q, n=n: fma(a, n, q * n))
P("fmaak4", lambda a, q: fmaak(a, 4, q * 4))
P("fmaak6", lambda a, q: fmaak(a, 6, q * 6))
P("accread4", lambda a, q: [a("v_accvgpr_read_b32", V(72 + k), A(128 + (q * 4 + k) % 128)) for k in range(4)] and None)
P("salu4", lambda a, q: [a("s_add_u32", S(20 + k), S(20 + k), 1) for k in range(4)] and None)
P("salu8", lambda a, q: [a("s_add_u32", S(20 + k % 4), S(20 + k % 4), 1) for k in range(8)] and None)
P("dsread_every", lambda a, q: ds_read(a, q))
P("dsread_6of8", lambda a, q: ds_read(a, q) if READS68(q) else None)
P("dsread_6of8_wait", lambda a, q: (ds_read(a, q) if READS68(q) else None, a("s_waitcnt", lgkmcnt=3) if (q & 1) == 0 else None) and None)
P("dsread_6of8_fma3", lambda a, q: (ds_read(a, q) if READS68(q) else None, fma(a, 3, q * 3)) and None)
P("dsread_4of8", lambda a, q: ds_read(a, q) if (q & 1) == 0 else None)
P("dsread_b64_6of8", lambda a, q: a("ds_read_b64", V(12 + 4 * (q % 6), 2), V(2), offset=(q % 8) * 1024) if READS68(q) else None)
P("dsread2x_b64", lambda a, q: [a("ds_read_b64", V(12 + 2 * ((2 * q + k) % 12), 2), V(2), offset=((2 * q + k) % 16) * 512) for k in range(2)] and None if READS68(q) else None)
P("dma_12of32", lambda a, q: dma(a, q) if q % 8 in (1, 3, 6) else None, lambda a: a("s_waitcnt", vmcnt=12))
P("dma_12of32_nonop", lambda a, q: dma(a, q, False) if q % 8 in (1, 3, 6) else None, lambda a: a("s_waitcnt", vmcnt=12))
P("dma_6of32", lambda a, q: dma(a, q) if q % 16 in (1, 6, 11) else None, lambda a: a("s_waitcnt", vmcnt=12))
P("dma_12_reads", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None) and None, lambda a: a("s_waitcnt", vmcnt=12))
P("dma_12_reads_fma3", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, fma(a, 3, 3 * q)) and None,
  lambda a: a("s_waitcnt", vmcnt=12))
P("barrier_per_iter", lambda a, q: None, lambda a: a("s_barrier"))
P("reads_barrier", lambda a, q: ds_read(a, q) if READS68(q) else None, lambda a: (a("s_waitcnt", lgkmcnt=0), a("s_barrier")) and None)
P("reads_salu30_barrier", lambda a, q: ds_read(a, q) if READS68(q) else None,
  lambda a: ([a("s_add_u32", S(20 + k % 4), S(20 + k % 4), 1) for k in range(30)], a("s_waitcnt", lgkmcnt=0), a("s_barrier")) and None)
P("store_1of16", lambda a, q: a("global_store_dwordx4", V(2), V(72, 4), S(6, 2), offset=2048) if q % 16 == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
def st_fresh(a, n, width=4, only_wave0=False, k=0):
    """n stores of 16 (8) bytes per lane to lines never written before; s[8:9] advances"""
    L = None
    if only_wave0:
        L = a.newlabel("SK")
        a("s_cmp_lg_u32", S(10), 0)
        a("s_cbranch_scc1", L)
    for _ in range(n):
        if width == 4:
            a("global_store_dwordx4", V(66), V(72, 4), S(8, 2))
            a("s_add_u32", S(8), S(8), 1024)
        else:
            a("global_store_dwordx2", V(67), V(72, 2), S(8, 2))
            a("s_add_u32", S(8), S(8), 512)
        a("s_addc_u32", S(9), S(9), 0)
    if L:
        a.label(L)


P("st_fresh_2of32", lambda a, q: st_fresh(a, 1) if q % 16 == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_fresh_2of32_wave0", lambda a, q: st_fresh(a, 1, only_wave0=True) if q % 16 == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_fresh_4of32_x2", lambda a, q: st_fresh(a, 1, width=2) if q % 8 == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_fresh_1of32", lambda a, q: st_fresh(a, 1) if q == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_fresh_4of32", lambda a, q: st_fresh(a, 1) if q % 8 == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_fresh_burst4", lambda a, q: st_fresh(a, 4) if q == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_fresh_2of32_nowait", lambda a, q: st_fresh(a, 1) if q % 16 == 5 else None)
def st_rows(a, n):
    """n stores of 8 rows x 128 bytes at a row pitch of 6144 bytes (the GEMM's epilogue), fresh lines"""
    for _ in range(n):
        a("global_store_dwordx4", V(68), V(72, 4), S(8, 2))
        a("s_add_u32", S(8), S(8), 128)
        a("s_addc_u32", S(9), S(9), 0)


P("st_rows_2of32", lambda a, q: st_rows(a, 1) if q % 16 == 5 else None, lambda a: a("s_waitcnt", vmcnt=8))
P("st_rows_2of32_dma_reads", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, st_rows(a, 1) if q % 16 == 5 else None) and None,
  lambda a: a("s_waitcnt", vmcnt=12))
P("st_fresh_2of32_dma_reads", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, st_fresh(a, 1) if q % 16 == 5 else None) and None,
  lambda a: a("s_waitcnt", vmcnt=12))
P("st_rows_burst4_every2", lambda a, q: st_rows(a, 4) if q == 5 else None, lambda a: a("s_waitcnt", vmcnt=12))
def gather(a, q, n=1, valu=0):
    """n ds_read_b64 at pseudo-random 8-byte slots of an 8 KiB table (LDS offset 56 KiB) + valu dependent-free VALU"""
    for k in range(n):
        a("v_mul_u32_u24", V(90), 0x9E3B, V(90))                      # per-lane LCG
        a("v_add_u32", V(90), 0x7F4A7, V(90))
        a("v_bfe_u32", V(91), V(90), 8, 10)
        a("v_lshl_add_u32", V(91), V(91), 3, S(26))
        a("ds_read_b64", V(92 + 2 * ((q * n + k) % 4), 2), V(91))
    fma(a, valu, q * valu)


P("gather1", lambda a, q: gather(a, q, 1), lambda a: a("s_waitcnt", lgkmcnt=0))
P("gather1_fma2", lambda a, q: gather(a, q, 1, 2), lambda a: a("s_waitcnt", lgkmcnt=0))
P("gather1_reads_dma", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, gather(a, q, 1)) and None,
  lambda a: a("s_waitcnt", vmcnt=12, lgkmcnt=0))
P("gather1_reads_dma_fma1", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, gather(a, q, 1, 1)) and None,
  lambda a: a("s_waitcnt", vmcnt=12, lgkmcnt=0))
P("reads_dma_fma4", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, fma(a, 4, 4 * q)) and None,
  lambda a: a("s_waitcnt", vmcnt=12, lgkmcnt=0))
P("reads_dma_fma5", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, fma(a, 5, 5 * q)) and None,
  lambda a: a("s_waitcnt", vmcnt=12, lgkmcnt=0))
P("reads_dma_fma7", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, fma(a, 7, 7 * q)) and None,
  lambda a: a("s_waitcnt", vmcnt=12, lgkmcnt=0))
P("st_only_nomfma_ref", lambda a, q: None)
P("cvt_perm", lambda a, q: (a("v_cvt_pk_bf16_f32", V(80 + q % 4), V(72), V(73)), a("s_nop", 1), a("v_permlane32_swap_b32", V(80 + q % 4), V(84 + q % 4))) and None)
P("waitcnt4", lambda a, q: [a("s_waitcnt", lgkmcnt=15) for k in range(4)] and None)
P("nop4", lambda a, q: [a("s_nop", 0) for k in range(4)] and None)


def pk(a, n, base=0, sgpr=False, chains=8):
    for k in range(n):
        r = 72 + 2 * ((base + k) % chains)
        a("v_pk_fma_f32", V(r, 2), V(r, 2), V(64, 2), S(20, 2) if sgpr else V(64, 2))


def med3(a, n, base=0):
    for k in range(n):
        r = 72 + ((base + k) % 16)
        a("v_med3_f32", V(r), V(r), V(64), V(65))


for n in (1, 2, 3, 4, 5):
    P("pkfma%d" % n, lambda a, q, n=n: pk(a, n, q * n))
P("pkfma3_sgpr", lambda a, q: pk(a, 3, q * 3, sgpr=True))
P("pkfma3_2chains", lambda a, q: pk(a, 3, q * 3, chains=2))
P("pkfma4_2chains", lambda a, q: pk(a, 4, q * 4, chains=2))
P("fma4_2chains", lambda a, q: [a("v_fma_f32", V(72 + (4 * q + k) % 2), V(72 + (4 * q + k) % 2), V(64), V(65)) for k in range(4)] and None)
P("pkmul3", lambda a, q: [a("v_pk_mul_f32", V(72 + 2 * ((3 * q + k) % 8), 2), V(72 + 2 * ((3 * q + k) % 8), 2), V(64, 2)) for k in range(3)] and None)
P("t4gap", lambda a, q: (pk(a, 3, q * 3, sgpr=True, chains=2), med3(a, 1, q)) and None)
P("t4gap_reads_dma", lambda a, q: (dma(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if (q & 1) else None, pk(a, 3, q * 3, sgpr=True, chains=2), med3(a, 1, q)) and None,
  lambda a: a("s_waitcnt", vmcnt=12, lgkmcnt=0))
P("fma8_fullgelu", lambda a, q: fma(a, 7, q * 7))
def t4mix(a, q, n, with_dma=True):
    if with_dma and q % 8 in (1, 3, 6):
        dma(a, q, nop=False)
    if (q & 1) == 0:
        ds_read(a, q)
        a("s_waitcnt", lgkmcnt=3)
    fmaak(a, n, q * n)


for n in (3, 4, 5, 6, 7):
    P("t4mix%d" % n, lambda a, q, n=n: t4mix(a, q, n), lambda a: (a("s_waitcnt", vmcnt=0, lgkmcnt=0), a("s_barrier")) and None)
P("t4mix6_nodma", lambda a, q: t4mix(a, q, 6, False), lambda a: (a("s_waitcnt", vmcnt=0, lgkmcnt=0), a("s_barrier")) and None)
P("t4mix6_nobar", lambda a, q: t4mix(a, q, 6), lambda a: a("s_waitcnt", vmcnt=12))
def pkbatch(a, q, n, every, plain, mix=False):
    """n packed fp32 operations behind every `every`-th MFMA, `plain` single fp32 operations behind all of them (mix: plus the
    fragment reads / LDS-DMA pieces of the GEMM loop)"""
    if mix:
        if q % 8 in (1, 3, 6):
            dma(a, q, nop=False)
        if READS68(q):
            ds_read(a, q)
    if q % every == every - 1:
        pk(a, n, q * n)
    fmaak(a, plain, q * plain)


for n, every, plain in ((8, 8, 0), (16, 8, 0), (16, 16, 0), (32, 16, 0), (16, 8, 2), (16, 8, 4), (32, 16, 4), (16, 8, 5), (8, 4, 4), (12, 8, 4), (24, 16, 4), (48, 32, 4)):
    P("pkb%d_e%d_p%d" % (n, every, plain), lambda a, q, n=n, every=every, plain=plain: pkbatch(a, q, n, every, plain))
for n, every, plain in ((16, 8, 2), (16, 8, 4), (32, 16, 4), (14, 8, 2), (28, 16, 2), (56, 32, 2)):
    P("pkbmix%d_e%d_p%d" % (n, every, plain), lambda a, q, n=n, every=every, plain=plain: pkbatch(a, q, n, every, plain, True),
      lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
P("mix_p5", lambda a, q: pkbatch(a, q, 0, 8, 5, True), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
P("mix_p6", lambda a, q: pkbatch(a, q, 0, 8, 6, True), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
P("mix_p2", lambda a, q: pkbatch(a, q, 0, 8, 2, True), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
VDST = set()
for nm, pat in (("vdst_mfma_only", lambda a, q: None), ("vdst_fma4", lambda a, q: fma(a, 4, q * 4)), ("vdst_pkfma3", lambda a, q: pk(a, 3, q * 3)),
                ("vdst_t4gap", lambda a, q: (pk(a, 3, q * 3, sgpr=True, chains=2), med3(a, 1, q)) and None)):
    P(nm, pat)
    VDST.add(nm)

# ---- round 4: what do transcendental / 16-bit packed instructions cost in the shadow of an MFMA, and LDS-DMA pieces that share
# one m0 write through the instruction's immediate offset (tools/ubench/lds_dma_offset.hip shows where the offset goes)
def trans(a, op, n, base=0):
    for k in range(n):
        r = 72 + ((base + k) % 16)
        a(op, V(r), V(r))


def pk16(a, op, n, base=0):
    for k in range(n):
        r = 72 + ((base + k) % 16)
        if op == "v_pk_fma_f16":
            a(op, V(r), V(r), V(64), V(65))
        else:
            a(op, V(r), V(r), V(64))


def dma_off(a, q):
    """three pieces behind MFMAs 1, 3, 6 of every 8 as in dma(), but m0 written once per FOUR pieces"""
    k = (q // 8) * 3 + {1: 0, 3: 1, 6: 2}[q % 8]
    if k % 4 == 0:
        a("s_add_u32", "m0", S(11), (k // 4) * 4096)
    a("global_load_lds_dwordx4", V(2), S(6, 2), offset=(k % 4) * 1024)


for n in (1, 2, 4):
    P("exp%d" % n, lambda a, q, n=n: trans(a, "v_exp_f32", n, q * n))
    P("rcp%d" % n, lambda a, q, n=n: trans(a, "v_rcp_f32", n, q * n))
P("exp1_fma4", lambda a, q: (trans(a, "v_exp_f32", 1, q), fma(a, 4, q * 4 + 4)) and None)
P("exp1_fma5", lambda a, q: (trans(a, "v_exp_f32", 1, q), fma(a, 5, q * 5 + 4)) and None)
P("exp1rcp1_fma3", lambda a, q: (trans(a, "v_exp_f32", 1, q), fma(a, 3, q * 3 + 4), trans(a, "v_rcp_f32", 1, q + 8)) and None)
P("exp1rcp1_fma4", lambda a, q: (trans(a, "v_exp_f32", 1, q), fma(a, 4, q * 4 + 4), trans(a, "v_rcp_f32", 1, q + 8)) and None)
P("alt_exp_rcp_fma4", lambda a, q: (trans(a, "v_exp_f32" if q & 1 else "v_rcp_f32", 1, q), fma(a, 4, q * 4 + 4)) and None)
P("alt_exp_rcp_fma3", lambda a, q: (trans(a, "v_exp_f32" if q & 1 else "v_rcp_f32", 1, q), fma(a, 3, q * 3 + 4)) and None)
for n in (2, 4, 5, 6):
    P("pkfma16_%d" % n, lambda a, q, n=n: pk16(a, "v_pk_fma_f16", n, q * n))
P("pkmul16_4", lambda a, q: pk16(a, "v_pk_mul_f16", 4, q * 4))
P("dmaoff_12of32", lambda a, q: dma_off(a, q) if q % 8 in (1, 3, 6) else None, lambda a: a("s_waitcnt", vmcnt=12))
P("dmaoff_12_reads_fma3", lambda a, q: (dma_off(a, q) if q % 8 in (1, 3, 6) else None, ds_read(a, q) if READS68(q) else None, fma(a, 3, 3 * q)) and None,
  lambda a: a("s_waitcnt", vmcnt=12))


def target(a, q, nf, off=True, transmix=True):
    """the loop round 4 aims at: fragment reads, 12 DMA pieces (3 m0 writes), nf fillers per gap of which every fourth is transcendental"""
    if q % 8 in (1, 3, 6):
        (dma_off if off else (lambda a, q: dma(a, q, nop=False)))(a, q)
    if READS68(q):
        ds_read(a, q)
    for k in range(nf):
        n = q * nf + k
        r = 72 + n % 16
        if transmix and n % 4 == 3:
            a("v_exp_f32" if n & 4 else "v_rcp_f32", V(r), V(r))
        else:
            a("v_fmaak_f32", V(r), V(r), V(64), F(0.123))


for nf in (3, 4, 5):
    P("target%d" % nf, lambda a, q, nf=nf: target(a, q, nf), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
P("target4_m0each", lambda a, q: target(a, q, 4, off=False), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
P("target4_plain", lambda a, q: target(a, q, 4, transmix=False), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)
P("target5_plain", lambda a, q: target(a, q, 5, transmix=False), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)

# ---- round 5: the epilogue as an instruction STREAM at its real rate -- the logistic GELU in fp32 (7 per element, 2 of them transcendental)
# against the packed-f16 polynomial (v_cvt_pk_f16_f32, v_pk_mul / v_pk_fma_f16, v_fma_mix_f32 back to fp32): what does the exchange buy?
def stream_ops(kind):
    """the instruction kinds of one group of FOUR elements, in issue order"""
    R = lambda n: 72 + n % 16
    ops = []
    acc = [lambda a, k=k: a("v_accvgpr_read_b32", V(R(k)), A(128 + k)) for k in range(4)]
    ln = [lambda a, k=k: a("v_fma_f32", V(R(k)), V(R(k)), V(64), V(65)) for k in range(8)]
    cvtbf = [lambda a, k=k: a("v_cvt_pk_bf16_f32", V(88 + k), V(R(2 * k)), V(R(2 * k + 1))) for k in range(2)]
    dsw = [lambda a: a("ds_write_b64", V(2), V(88, 2), offset=8192)]
    f32 = lambda n: [lambda a, k=k: a("v_fma_f32", V(R(k)), V(R(k)), V(64), V(65)) for k in range(n)]
    tr = lambda op: [lambda a, k=k: a(op, V(R(k + 8)), V(R(k + 8))) for k in range(4)]
    pk = lambda n, op="v_pk_fma_f16": [lambda a, k=k: (a(op, V(R(k + 4)), V(R(k + 4)), V(64), V(65)) if op == "v_pk_fma_f16" else a(op, V(R(k + 4)), V(R(k + 4)), V(64))) for k in range(n)]
    cvth = [lambda a, k=k: a("v_cvt_pk_f16_f32", V(R(k + 4)), V(R(2 * k)), V(R(2 * k + 1))) for k in range(2)]
    mix = [lambda a, k=k: a("v_fma_mix_f32", V(R(k)), V(R(k)), V(R(4 + k // 2)), 0, op_sel="[0,%d,0]" % (k & 1), op_sel_hi="[0,1,0]") for k in range(4)]
    sig = f32(4) + f32(4) + f32(4) + tr("v_exp_f32") + f32(4) + tr("v_rcp_f32") + f32(4)
    h2 = lambda nh: cvth + pk(2, "v_pk_mul_f16") + pk(2) + pk(2 * nh) + pk(2)          # cvt, t, u, Horner, Phi
    if kind == "q4_sig":
        return acc + ln + sig + cvtbf + dsw
    if kind == "q4_h2":            # 7 coefficients
        return acc + ln + h2(6) + mix + cvtbf + dsw
    if kind == "q4_h2n5":
        return acc + ln + h2(4) + mix + cvtbf + dsw
    if kind == "q4_h2_f16out":     # the hidden stored as f16: the product in packed f16, no conversion back
        return acc + ln + h2(6) + pk(2, "v_pk_mul_f16") + dsw
    if kind == "q4_none":
        return acc + ln + cvtbf + dsw
    if kind == "t4_sig":
        return sig + cvtbf
    if kind == "t4_h2":            # f16 hidden: x16 * Phi16 is the second product's operand
        return h2(6) + pk(2, "v_pk_mul_f16")
    if kind == "t4_h2_bf16":
        return h2(6) + mix + cvtbf
    raise KeyError(kind)


class Stream:
    def __init__(self, kind, per_iter):
        self.ops, self.rate, self.pos = stream_ops(kind), per_iter / 32.0, 0

    def __call__(self, a, q):
        n = int((q + 1) * self.rate) - int(q * self.rate)
        for _ in range(n):
            self.ops[self.pos % len(self.ops)](a)
            self.pos += 1


def q4loop(kind):
    """the q4 K loop of fc1 (K = 768: 32 groups of four elements drained over 12 iterations of 32 MFMAs) with the epilogue stream `kind`"""
    st = Stream(kind, len(stream_ops(kind)) * 32 / 12.0)
    def pat(a, q):
        if q % 8 in (1, 3, 6):
            dma_off(a, q)
        if READS68(q):
            ds_read(a, q)
        st(a, q)
    return pat


for kind in ("q4_none", "q4_sig", "q4_h2", "q4_h2n5", "q4_h2_f16out"):
    P("r5_" + kind, q4loop(kind), lambda a: (a("s_waitcnt", vmcnt=12, lgkmcnt=0), a("s_barrier")) and None)


def t4loop(kind):
    """the token kernel's group iteration: 54 MFMAs, 32 elements per lane -> 8 groups of four; here per 32 MFMAs"""
    st = Stream(kind, len(stream_ops(kind)) * 8 * 32 / 54.0)
    def pat(a, q):
        if q % 8 in (1, 3, 6):
            dma(a, q, nop=False)
        if (q & 1) == 0:
            ds_read(a, q)
            a("s_waitcnt", lgkmcnt=3)
        st(a, q)
    return pat


for kind in ("t4_sig", "t4_h2", "t4_h2_bf16"):
    P("r5_" + kind, t4loop(kind), lambda a: (a("s_waitcnt", vmcnt=0, lgkmcnt=0), a("s_barrier")) and None)
    VDSTN = kind
P("mix4", lambda a, q: [a("v_fma_mix_f32", V(72 + (4 * q + k) % 16), V(72 + (4 * q + k) % 16), V(64), 0, op_sel="[0,%d,0]" % (k & 1), op_sel_hi="[0,1,0]") for k in range(4)] and None)
P("mix5", lambda a, q: [a("v_fma_mix_f32", V(72 + (5 * q + k) % 16), V(72 + (5 * q + k) % 16), V(64), 0, op_sel="[0,%d,0]" % (k & 1), op_sel_hi="[0,1,0]") for k in range(5)] and None)
P("cvth4", lambda a, q: [a("v_cvt_pk_f16_f32", V(72 + (4 * q + k) % 16), V(64), V(65)) for k in range(4)] and None)
P("cvtbf4", lambda a, q: [a("v_cvt_pk_bf16_f32", V(72 + (4 * q + k) % 16), V(64), V(65)) for k in range(4)] and None)
P("pkfma16_dep4", lambda a, q: [a("v_pk_fma_f16", V(72), V(72), V(64), V(65)) for k in range(4)] and None)       # ONE dependent chain
P("fma_dep4", lambda a, q: [a("v_fma_f32", V(72), V(72), V(64), V(65)) for k in range(4)] and None)
ONLY = os.environ.get("ONLY", "")
if ONLY:
    PAT[:] = [x for x in PAT if any(x[0].startswith(t) for t in ONLY.split(","))]


def main(path):
    out = ["#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <vector>\n"]
    for name, pat, per in PAT:
        out.append(kernel("ub_" + name, pat, per, vdst=name in VDST))
    out.append("struct K { const char* name; const void* fn; };\nstatic const K ks[] = {\n")
    for name, _, _ in PAT:
        out.append("    {\"%s\", (const void*)&ub_%s},\n" % (name, name))
    out.append("};\n")
    out.append("""int main() {
    setvbuf(stdout, 0, _IONBF, 0);
    unsigned* out; char* src; char* dst;
    hipMalloc(&dst, (size_t)1024 << 18);
    hipMalloc(&out, 256 * 4 * 4);
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    std::vector<unsigned> h(1024);
    for (const K& k : ks) {
        hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        void* args[] = {&out, &src, &dst};
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernel(k.fn, dim3(256), dim3(256), args, 65536, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%%s: launch failed\\n", k.name); return 1; }
        hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost);
        double s = 0; unsigned mx = 0, mn = ~0u;
        for (unsigned v : h) { s += v; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        const double per = %d.0 * 32;
        printf("%%-24s cycles per MFMA: mean %%7.2f  min %%7.2f  max %%7.2f\\n", k.name, s / 1024 / per, mn / per, mx / per);
    }
    return 0;
}
""" % ITERS)
    with open(path, "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
