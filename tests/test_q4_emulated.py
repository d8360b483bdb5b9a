"""The generated q4 GEMM kernels (jittor-mlp_amd/csrc/gen/q4gen.py) WITHOUT a GPU: the same instruction list that becomes the
asm block is executed by the numpy emulator of csrc/gen/isa.py -- one workgroup of four 64-lane waves, LDS, LDS-DMA, MFMA
32x32x16, the counted s_waitcnt / s_barrier protocol modelled adversarially (DMA data lands as late as the issuing wave's own
vmcnt allows, or at once; loaded registers hold poison until the covering wait; waves run one after the other between barriers)
-- and compared with an fp64 restatement of the operation (mlp_mixer.py:16-27,38 / vip.py:82-88: Linear + bias [+ folded LayerNorm]
[+ GELU] [+ residual]).  Also: the software-hazard lint over every emitted kernel, and mutations of the wait / barrier protocol
that the emulator must catch (a missing wait shows up here as a wrong result instead of as a rare wrong tile on the GPU)."""
import os
import sys

import pytest

GEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jittor-mlp_amd", "csrc", "gen")
sys.path.insert(0, GEN)
import isa  # noqa: E402
import q4emu  # noqa: E402
import q4gen  # noqa: E402

CASES = [
    # kernel class, problem (M, N, K), grid, column groups, DMA landing model, wave order
    (dict(nkf=3), (768, 256, 192), 8, 2, "late", None),                                        # plain, three tiles per workgroup + drain
    (dict(gelu=True, ln=True, nkf=3), (512, 256, 256), 8, 2, "early", [3, 2, 1, 0]),           # fc1 class, rolled iterations
    (dict(res=True, stats=True, nkf=4), (512, 128, 320), 8, 1, "late", [2, 0, 3, 1]),          # fc2 class with by-product statistics
    (dict(dtype="f16", gelu=True, nkf=4), (256, 256, 256), 16, 1, "late", None),
    (dict(dtype="f16", res=True, nkf=3), (512, 128, 192), 8, 1, "early", None),
    # round 4: kernels built for ONE K (static LDS stages, four LDS-DMA pieces per m0 write through the immediate offset, two fragment
    # waits per k-step), several tiles per workgroup so that both accumulator parities and the tile switch of the DMA stream run
    (dict(gelu=True, ln=True, nkf=3, static=True), (3072, 256, 192), 8, 2, "late", [3, 1, 2, 0]),
    (dict(res=True, stats=True, nkf=6, static=True), (1024, 256, 384), 8, 1, "early", None),
    (dict(dtype="f16", gelu=True, nkf=6, static=True), (1024, 256, 384), 8, 1, "late", None),
    (dict(gelu=True, ln=True, nkf=12, static=True), (768, 128, 768), 8, 1, "late", [1, 3, 0, 2]),            # the Mixer-B/16 fc1 kernel
    (dict(gelu=True, ln=True, stats=True, nkf=4), (1024, 256, 256), 8, 1, "late", [2, 0, 3, 1]),               # gMLP channel_proj1, v half
]


@pytest.mark.parametrize("kw,shape,grid,cg,mode,order", CASES)
def test_generated_kernel_matches_fp64_in_emulation(kw, shape, grid, cg, mode, order):
    g = q4gen.Q4(**kw)
    assert isa.lint(g.a) == []
    assert q4emu.run_case(g, *shape, grid=grid, cgroups=cg, dma_mode=mode, order=order)


def test_every_shipped_variant_passes_the_hazard_lint():
    n = 0
    for name, kw in q4gen.variants():
        g = q4gen.Q4(**kw)
        assert isa.lint(g.a) == [], name
        assert g.nv <= 248 and g.ns <= 96, name          # registers the asm block may name (the rest carry the block's inputs)
        n += 1
    assert n >= 48


def _mutated(mut):
    g = q4gen.Q4(gelu=True, ln=True, nkf=2)
    n = sum(mut(i) for i in g.a.ins)
    assert n > 0
    return g


def test_emulator_catches_protocol_faults():
    def loose_vmcnt(i):            # one more LDS-DMA piece allowed in flight than the ring tolerates
        if i.op == "s_waitcnt" and i.mods.get("vmcnt") == 12 and "lgkmcnt" in i.mods:
            i.mods["vmcnt"] = 13
            return 1
        return 0

    def loose_lgkm(i):             # a fragment consumed one LDS read too early
        if i.op == "s_waitcnt" and "vmcnt" not in i.mods and i.mods.get("lgkmcnt") == 3:
            i.mods["lgkmcnt"] = 4
            return 1
        return 0

    def no_barrier(i):
        if i.op == "s_barrier":
            i.op, i.args = "s_nop", (0,)
            return 1
        return 0
    for mut in (loose_vmcnt, loose_lgkm, no_barrier):
        g = _mutated(mut)
        caught = False
        for mode, order in (("late", None), ("early", [3, 2, 1, 0]), ("late", [3, 1, 2, 0])):
            try:
                ok = q4emu.run_case(g, 512, 128, 256, grid=8, dma_mode=mode, order=order)
            except RuntimeError:
                ok = False
            caught = caught or not ok
        assert caught, mut.__name__


def test_emulator_catches_protocol_faults_of_the_static_kernels():
    """the same screen for the kernels built for one K: a loosened DMA wait, a removed barrier, a fragment wait that comes one read too
    late, an LDS-DMA piece whose immediate offset points at the neighbouring K slab, an m0 base one stage off"""
    def build():
        return q4gen.Q4(gelu=True, ln=True, nkf=3, static=True)

    def loose_vmcnt(i):
        if i.op == "s_waitcnt" and i.mods.get("vmcnt") == 12 and "lgkmcnt" in i.mods:
            i.mods["vmcnt"] = 13
            return 1
        return 0

    def no_barrier(i):
        if i.op == "s_barrier":
            i.op, i.args = "s_nop", (0,)
            return 1
        return 0

    def late_fragment_wait(i):
        if i.op == "s_waitcnt" and "vmcnt" not in i.mods and i.mods.get("lgkmcnt") in (2, 3, 4):
            i.mods["lgkmcnt"] += 2
            return 1
        return 0
    state = {"n": 0}

    def wrong_slab(i):
        if i.op == "global_load_lds_dwordx4" and i.mods.get("offset", 0) % 1024 == 128 - 4096 % 1024:
            state["n"] += 1
            if state["n"] % 7 == 3:
                i.mods["offset"] += 128
                return 1
        return 0

    def wrong_stage(i):
        if i.op == "s_add_u32" and i.args[0] == "m0" and isinstance(i.args[2], int) and i.args[2] >= q4gen.STAGE_B:
            state["n"] += 1
            if state["n"] % 5 == 1:
                i.args = (i.args[0], i.args[1], i.args[2] - q4gen.STAGE_B)
                return 1
        return 0
    for mut in (loose_vmcnt, no_barrier, late_fragment_wait, wrong_slab, wrong_stage):
        g = build()
        state["n"] = 0
        assert sum(mut(i) for i in g.a.ins) > 0, mut.__name__
        caught = False
        for mode, order in (("late", None), ("early", [3, 2, 1, 0]), ("late", [3, 1, 2, 0])):
            try:
                ok = q4emu.run_case(g, 2048, 128, 192, grid=8, dma_mode=mode, order=order)
            except RuntimeError:
                ok = False
            caught = caught or not ok
        assert caught, mut.__name__


def test_hazard_lint_rules():
    from isa import A, Asm, V
    a = Asm()
    a("v_dot2c_f32_bf16", V(1), V(2), V(3))
    a("s_nop", 0)
    a("v_add_f32_dpp", V(1), V(1), V(1), quad_perm="[1,0,3,2]", row_mask="0xf", bank_mask="0xf")      # dot result read after 1 state
    assert isa.lint(a)
    a = Asm()
    a("v_mfma_f32_32x32x16_bf16", A(0, 16), V(0, 4), V(4, 4), A(0, 16))
    a("v_accvgpr_read_b32", V(9), A(3))                                                              # XDL result read at once
    assert isa.lint(a)
    a = Asm()
    a("s_add_u32", "m0", isa.S(4), 1024)
    a("global_load_lds_dwordx4", V(1), isa.S(6, 2))                                                  # no wait state after the m0 write
    assert isa.lint(a)
    a = Asm()
    a("global_store_dwordx2", V(1), V(3, 2), isa.S(6, 2))                                            # odd-aligned 64-bit VGPR tuple
    assert isa.lint(a)
    a = Asm()
    a("v_exp_f32", V(1), V(1))
    a("v_add_f32", V(1), isa.F(1.0), V(1))                                                           # transcendental result read at once (gfx940+)
    assert isa.lint(a)
    a = Asm()
    a("v_exp_f32", V(1), V(1))
    a("v_exp_f32", V(2), V(2))
    a("v_add_f32", V(1), isa.F(1.0), V(1))
    assert isa.lint(a) == []
