"""The generated fused token-mixing kernels (jittor-mlp_amd/csrc/gen/t4gen.py: both Conv1d(k=1) of the Mixer token FeedForward +
GELU + residual, mlp_mixer.py:16-27,34,37) WITHOUT a GPU: the instruction list that becomes the asm block runs on the numpy
emulator of csrc/gen/isa.py (four waves, LDS rings filled by LDS-DMA, MFMA 32x32x16, one counted wait + barrier per iteration,
modelled adversarially) and is compared with an fp64 restatement, by-product statistics included.  Also the hazard lint over
every shipped variant and mutations of the synchronisation the emulator must catch.  (The GELU polynomial is q4gen.GELU =
mlpk_common.h's: tests/test_host_cpu.py::test_division_free_gelu_coefficients.)"""
import os
import sys

import pytest

GEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jittor-mlp_amd", "csrc", "gen")
sys.path.insert(0, GEN)
import isa  # noqa: E402
import t4emu  # noqa: E402
import t4gen  # noqa: E402

CASES = [
    # kernel, images, channels per image, hidden size, grid, DMA landing model, wave order
    (dict(stats=True), 2, 256, 128, 2, "late", None),                       # even group count (no lead iteration), one tile per image
    (dict(stats=True), 2, 512, 100, 3, "early", [3, 2, 1, 0]),              # ragged hidden size, workgroups with one and two tiles
    (dict(stats=False), 1, 768, 33, 1, "late", [2, 0, 3, 1]),               # three tiles in one workgroup, odd group count (lead iteration)
    (dict(dtype="f16", stats=True), 1, 512, 80, 2, "late", None),
    # the kernels without dummy stages in the pipeline's fill / drain iterations (per parity of the group count)
    (dict(stats=True, shape=1), 3, 256, 160, 2, "late", None),              # G = 5
    (dict(stats=True, shape=1), 1, 512, 80, 1, "early", [3, 2, 1, 0]),      # G = 3: no steady-state loop at all
    (dict(stats=False, shape=2), 2, 256, 128, 2, "late", [2, 0, 3, 1]),     # G = 4
    (dict(dtype="f16", stats=True, shape=2), 1, 768, 64, 1, "early", None), # G = 2
    # the token LayerNorm + transpose as the kernel's operand loader (mlpk_token_mlp_ln: x itself is read, no xt)
    (dict(stats=True, shape=1, ln=True), 3, 256, 160, 2, "late", None),
    (dict(stats=False, shape=2, ln=True), 2, 512, 128, 3, "early", [2, 0, 3, 1]),
    (dict(dtype="f16", stats=True, shape=1, ln=True), 1, 768, 96, 1, "late", [3, 2, 1, 0]),
    # round 5: the bf16 kernels with the GELU in packed f16 and the hidden kept in f16 (mlpk.h layout 3) -- additionally held, operation
    # by operation, to the numpy restatement of that GELU (t4emu.h2_gelu_ref): within one ulp of the output everywhere
    (dict(stats=True, shape=1, ln=True, h2=True), 3, 256, 160, 2, "late", None),
    (dict(stats=False, shape=0, h2=True), 1, 768, 33, 1, "late", [2, 0, 3, 1]),
    (dict(stats=True, shape=2, h2=True), 2, 512, 128, 3, "early", [3, 2, 1, 0]),
    (dict(stats=False, shape=2, ln=True, h2=True), 2, 512, 128, 3, "early", [2, 0, 3, 1]),
]


@pytest.mark.parametrize("kw,nimg,t_rows,T,grid,mode,order", CASES)
def test_generated_kernel_matches_fp64_in_emulation(kw, nimg, t_rows, T, grid, mode, order):
    g = t4gen.T4(**kw)
    assert isa.lint(g.a) == []
    assert t4emu.run_case(g, nimg=nimg, t_rows=t_rows, T=T, grid=grid, dma_mode=mode, order=order)


def test_every_shipped_variant_passes_the_hazard_lint():
    n = 0
    for kw in t4gen.variants():
        g = t4gen.T4(**kw)
        assert isa.lint(g.a) == [], g.name
        assert g.nv <= 248 and g.ns <= 100, g.name
        n += 1
    assert n >= 30


def test_emulator_catches_protocol_faults():
    def no_barrier(i):
        if i.op == "s_barrier":
            i.op, i.args = "s_nop", (0,)
            return 1
        return 0

    def loose_vmcnt(i):            # the iteration barrier no longer waits for this wave's own LDS-DMA pieces
        if i.op == "s_waitcnt" and i.mods.get("vmcnt") == 0 and i.mods.get("lgkmcnt") == 0:
            i.mods["vmcnt"] = 10
            return 1
        return 0

    def loose_lgkm(i):             # a weight fragment consumed one LDS read too early
        if i.op == "s_waitcnt" and "vmcnt" not in i.mods and i.mods.get("lgkmcnt") == 3:
            i.mods["lgkmcnt"] = 4
            return 1
        return 0
    for mut in (no_barrier, loose_vmcnt, loose_lgkm):
        g = t4gen.T4(stats=False, shape=mut is not loose_vmcnt)
        assert sum(mut(i) for i in g.a.ins) > 0
        caught = False
        for mode, order in (("late", None), ("early", [3, 2, 1, 0]), ("late", [3, 1, 2, 0])):
            try:
                ok = t4emu.run_case(g, nimg=1, t_rows=512, T=100, grid=1, dma_mode=mode, order=order)
            except RuntimeError:
                ok = False
            caught = caught or not ok
        assert caught, mut.__name__


def test_host_constants_match_the_generators():
    """the host side launches with the LDS size and reads the argument block the generators laid out"""
    import re
    import q4gen
    csrc = os.path.join(os.path.dirname(GEN))
    t4 = open(os.path.join(csrc, "mlpk_tokenmlp_t4.hip")).read()
    assert int(re.search(r"#define T4_LDS_BYTES (\d+)", t4).group(1)) == t4gen.LDS_BYTES <= 160 * 1024
    assert int(re.search(r"static_assert\(sizeof\(T4Args\) == (\d+)", t4).group(1)) == t4gen.ARG_BYTES
    q4h = open(os.path.join(csrc, "mlpk_gemm_q4.h")).read()
    m = re.search(r"#define Q4_LDS_BYTES \((\d+) \* (\d+) \+ (\d+)\)", q4h)
    assert int(m.group(1)) * int(m.group(2)) + int(m.group(3)) == q4gen.LDS_BYTES <= 160 * 1024
