"""-m gpu: whole-model parity of the HIP path (through the C ABI) against
  (1) the committed golden logits the reference produced (tiny full-state fixtures and
      real-shape fixtures with portable weights), and
  (2) the CPU oracle on the same seeded inputs.
Tolerances (north star: 1e-5 fp32 / 1e-3 fp16; SURVEY Appendix C for bf16):
  fp32  abs 1e-5 (tiny and real depth);
  fp16  abs 1e-3 * max(1, max|ref|);
  bf16  abs 5e-3 * max(1, max|ref|) at real depth (the reference's own bf16 run is 2.7-4.7e-3 away
        from its fp32 logits, so 1e-3 is not attainable by any bf16 implementation), with the per-model exceptions of
        BF16_REAL_EXCEPTIONS; 1.5e-2 on the tiny configs (random norm statistics, logits of magnitude ~1).
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import load_pkg
from oracle.portable_init import portable_input, portable_state_dict
from test_oracle_golden import run_oracle  # noqa: F401  (re-exported for ad-hoc debugging sessions)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


# bf16 at real depth: the gate is 5e-3 * max(1, max|ref|) (Mixer-S/B/L 2.3-2.8e-3, ResMLP 3.9e-3, Swin-MLP 3.8e-3,
# ConvMixer 6e-4 measured; the reference's OWN bf16 forward is 2.7e-3 on Mixer-B and 4.7e-3 on gMLP-S).
# Deeper networks add 2 x depth bf16-rounded updates to a bf16 residual stream -- exactly like the reference run in bf16 -- and are 30-38
# blocks deep or carry a residual of magnitude 4-8 (half-ulp 2^-7 .. 2^-6 per add) where a 12-block Mixer does not: their gates are
# derived from the reference's own bf16 forward (REF_LOWP below); what is left here is a measured value + head-room.
BF16_REAL_EXCEPTIONS = {
    # the one deep three-branch model whose reference cannot be run in 16 bit here (its torchvision stand-in builds the sampling grid in
    # the input dtype): measured 6.2e-3 (the same three-branch + reweight structure as ViP, whose reference loses 9.3e-3 in bf16)
    "cyclemlp_b1": 8.0e-3,
}


# Round 4: where the reference ITSELF was run in bf16 on the fixture's inputs (tests/golden/make_golden.py --only lowp -> real_lowp.json:
# AS-MLP-T 9.7e-3, gMLP-S 4.5e-3, ViP-S7 9.3e-3, Sparse-MLP 8.2e-3, Hire-MLP 6.3e-3, MS-MLP 2.0e-2 away from its own fp32 logits), the
# gate is DERIVED -- 1.25 x the reference's own bf16 distance (never below the 5e-3 rule) -- instead of a measured value plus head-room
# (round-3 review: AS-MLP-T passed 7.1e-3 against a tuned 8.0e-3).  Measured here against those gates: 7.3e-3 / 4.2e-3 / 8.7e-3 /
# 6.8e-3 / 7.0e-3 / 1.7e-2 -- the kernels lose LESS than the reference's own bf16 forward on five of the six.
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_lowp.json")) as _f:
    REF_LOWP = json.load(_f)


# Regression guard (advisor, round 4: "accuracy drift cannot hide under a looser gate").  The bf16 distance of every real-depth fixture as
# MEASURED at the end of round 5 (profiles/r05_real_golden_lines.txt, the shipped packed-f16 GELU); a run may not exceed 1.3 x its entry,
# whatever the derived gate allows (the max over 2000-8000 logits moves by +-20 % between equally valid roundings).  History of the entries
# that moved: ViP-S7 6.2e-3 (rounds 1-3) -> 8.7e-3 (round 4) was the logistic bf16 GELU -- the SAME tree built with the polynomial
# (-DMLPK_GELU_BF16_POLY) gave 5.6e-3 on the same box, while AS-MLP-T / Sparse-MLP moved the other way (7.8e-3 vs 8.5e-3, 6.6e-3 vs 7.5e-3) and
# gMLP-S / Hire / MS-MLP by +-10-15 % (profiles/r05_gelu_accuracy_ab.txt): a property of which roundings the maximum lands on, not a trend;
# with round 5's form ViP-S7 is 7.0e-3, gMLP-S 4.3e-3, ResMLP-24 3.9e-3 (was 3.3e-3), the Mixers 2.3 - 2.6e-3.
BF16_RECORDED = {"mixer_s16": 2.34e-3, "mixer_b16": 2.51e-3, "gmlp_s": 4.27e-3, "resmlp_24": 3.91e-3, "vip_s7": 6.98e-3, "asmlp_t": 7.28e-3,
                 "convmixer_1536_20": 8.3e-4, "mixer_l16": 2.56e-3, "sparsemlp_t": 7.44e-3, "hiremlp_s": 6.78e-3, "msmlp_t": 1.53e-2,
                 "swinmlp_t": 3.76e-3, "cyclemlp_b1": 5.20e-3}
REGRESSION_HEADROOM = 1.3


# fp16 at real depth: north_star's 1e-3 (x max(1, |ref|)), or -- where the REFERENCE ITSELF, run in fp16 on the same input, is further than
# that from its own fp32 output (tests/golden/real_lowp.json, err_fp16; ViP-S7: 1.6e-3 on logits of magnitude 0.44 after 18 blocks x 6
# GEMMs with fp16 storage) -- 0.8 x the reference's own fp16 error: never less accurate than the reference in the same dtype, with
# margin.  (Round 6: this rule replaces a hand-set 1.2e-3 for ViP-S7, which measures 1.03e-3.  Same-box switches, round 2:
# SplitAttention's `a` from fp32 sums of the branch inputs -- 2.5e-4 from an fp64 evaluation of the same operands, tools/vip_a_check.py --
# 1.03e-3 | `a` summed from the fp16-ROUNDED branch outputs (4.5e-2 from fp64) 8.3e-4 | without the channel-branch LayerNorm fold
# 1.15e-3: the max over 8000 logits moves by +-20 % between equally valid roundings, and the more exact evaluation is kept.)
FP16_REF_FRACTION = 0.8


def tol_for(dtype, ref, real=False, name=None):
    m = max(1.0, float(ref.abs().max()))
    if dtype == torch.float32:
        return 1e-5
    if dtype == torch.float16:
        if real and name in REF_LOWP:
            return max(1e-3 * m, FP16_REF_FRACTION * REF_LOWP[name]["err_fp16"])
        return 1e-3 * m
    if real:
        if name in REF_LOWP:
            return max(5e-3 * m, 1.25 * REF_LOWP[name]["err_bf16"])
        return BF16_REAL_EXCEPTIONS.get(name, 5e-3) * m
    return 1.5e-2 * m


def flat(o):
    """a list of feature maps (CycleNet fork_feat) as one (B, sum C H W) matrix, the way its fixture holds the reference's output"""
    return torch.cat([t.reshape(t.shape[0], -1) for t in o], dim=1) if isinstance(o, (list, tuple)) else o


def ctor_for(pkg, name):
    table = {"mixer": "MLPMixerForImageClassification", "gmlp": "gMLPForImageClassification",
             "resmlp": "ResMLPForImageClassification", "vip": "ViP", "s2mlpv2": "S2MLPv2", "s2mlpv1": "S2MLPv1",
             "asmlp": "AS_MLP", "convmixer": "ConvMixer", "sparsemlp": "SparseMLP", "hiremlp": "HireMLP", "msmlp": "MS_MLP", "swinmlp": "SwinMLP",
             "cyclemlp_b1": "CycleMLP_B1", "cyclemlp": "CycleNet"}
    for k, v in table.items():
        if name.startswith(k):
            mod = pkg.models_pytorch
            if not hasattr(mod, v):
                pytest.skip("%s not built yet" % v)
            return getattr(mod, v)
    raise KeyError(name)


def build_from_tiny(pkg, name):
    z = np.load(os.path.join(GOLDEN, "tiny_%s.npz" % name))
    kw = json.loads(str(z["kwargs"]))
    for k in ("patch_size", "image_size"):
        if isinstance(kw.get(k), list) and not name.startswith("s2"):
            kw[k] = tuple(kw[k])
    model = ctor_for(pkg, name)(**kw).eval()
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    model.load_state_dict(sd, strict=True)                       # drop-in contract: reference keys load strictly
    if name.endswith("cleanshift"):
        model.set_shift_mode("shift")
    return model, torch.from_numpy(z["input"]), torch.from_numpy(z["logits"]), kw, sd


TINY_TOKEN = ["mixer", "mixer_nonsquare", "gmlp", "resmlp", "vip_weighted", "vip_unweighted", "vip_rect", "s2mlpv2",
              "s2mlpv2_cleanshift", "s2mlpv1", "asmlp", "convmixer", "convmixer_k4", "convmixer_k11", "sparsemlp", "sparsemlp_norm", "hiremlp", "hiremlp_rect", "msmlp", "msmlp_s3", "swinmlp", "swinmlp_small", "swinmlp_ape", "cyclemlp", "cyclemlp_rect", "cyclemlp_fork"]


@pytest.mark.parametrize("name", TINY_TOKEN)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_tiny_golden(name, dtype):
    pkg = load_pkg()
    model, x, ref, kw, sd = build_from_tiny(pkg, name)
    model = model.to(DEV)
    with torch.no_grad():
        out = flat(model(x.to(DEV).to(dtype)))
        out2 = flat(model(x.to(DEV).to(dtype)))
    torch.cuda.synchronize()
    assert out.dtype == dtype and out.shape == ref.shape
    assert torch.equal(out, out2), "forward is not deterministic"
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < tol_for(dtype, ref), (name, str(dtype), err)


@pytest.mark.parametrize("name", TINY_TOKEN)
def test_tiny_fp32_input_bf16_compute(name):
    """set_compute_dtype: fp32 images in, bf16 MFMA path, fp32 logits out."""
    pkg = load_pkg()
    model, x, ref, kw, sd = build_from_tiny(pkg, name)
    model = model.to(DEV).set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        out = flat(model(x.to(DEV)))
    assert out.dtype == torch.float32
    assert (out.cpu() - ref).abs().max().item() < tol_for(torch.bfloat16, ref)


REAL = [("mixer_s16", 8), ("mixer_b16", 4), ("gmlp_s", 2), ("resmlp_24", 2), ("vip_s7", 1), ("s2mlpv2", 2), ("asmlp_t", 2),
        ("convmixer_1536_20", 1), ("mixer_l16", 1), ("sparsemlp_t", 2), ("hiremlp_s", 2), ("msmlp_t", 2), ("swinmlp_t", 2), ("cyclemlp_b1", 2)]


@pytest.mark.parametrize("name,bs", REAL)
def test_real_golden_fp32_and_fp16(name, bs):
    """BASELINE configs at the fixture batch size, weights/inputs rebuilt from the portable generator,
    against the logits the reference produced with them."""
    pkg = load_pkg()
    z = np.load(os.path.join(GOLDEN, "real_%s.npz" % name))
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)["state_dicts"][name]
    model = ctor_for(pkg, name)(**man["kwargs"]).eval()
    shapes = {k: tuple(s) for k, s in man["keys"]}
    sd = {k: torch.from_numpy(v) for k, v in portable_state_dict(shapes, seed=0).items()}
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    x = torch.from_numpy(portable_input((bs, 3, 224, 224), seed=0))
    ref = torch.from_numpy(z["logits"])
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        with torch.no_grad():
            out = model(x.to(DEV).to(dtype))
        torch.cuda.synchronize()
        err = (out.float().cpu() - ref).abs().max().item()
        rms = (out.float().cpu() - ref).pow(2).mean().sqrt().item()
        print("real %-10s %-8s max|d| = %.3e  rms %.3e  (max|ref| %.3f)" % (name, str(dtype)[6:], err, rms, ref.abs().max()))
        tol = tol_for(dtype, ref, real=True, name=name)
        if name == "s2mlpv2":
            tol = s2_real_gate(dtype)
        elif dtype == torch.bfloat16 and os.environ.get("MLPK_LIB_PATH") is None:
            tol = min(tol, REGRESSION_HEADROOM * BF16_RECORDED[name])         # (an A/B build of another arithmetic keeps the plain gate)
        assert err < tol, (name, str(dtype), err)
    if name == "s2mlpv2":
        model.set_shift_mode("shift")
        ref2 = oracle.s2mlpv2_forward(sd, x, mode="shift")
        for dtype in (torch.float32, torch.float16, torch.bfloat16):
            with torch.no_grad():
                out = model(x.to(DEV).to(dtype))
            err = (out.float().cpu() - ref2).abs().max().item()
            print("real %-10s %-8s clean-shift mode vs oracle max|d| = %.3e (max|ref| %.3f)" % (name, str(dtype)[6:], err, ref2.abs().max()))
            # SplitAttention sums (not averages) 3*H*W pixels before its softmax, so this 18-block network amplifies
            # round-off ~100x more than the other families (fp32: 4e-5 here vs 5e-7 elsewhere).  There is no reference run
            # of this mode in 16 bit (the reference has no clean shift), so the 16-bit gates here are the same multiples of
            # the fp32 one that the reference's own runs show in its own mode (fp16 0.386 / fp32-vs-fp64 8.8e-4 = 440x,
            # bf16 1100x), times the factor 2 used everywhere else.
            gate = {torch.float32: 2e-4, torch.float16: 2e-4 * 440 * 2, torch.bfloat16: 2e-4 * 1100 * 2}[dtype]
            assert bool(torch.isfinite(out).all()) and err < gate, (name, str(dtype), err)


def s2_real_gate(dtype):
    """S2-MLPv2 at config depth (BASELINE configs[3]), reference_inplace mode.  The network is ill-conditioned (the in-place
    'smear' shift + SplitAttention's sum over all pixels amplify round-off ~1.6x per block): the reference's OWN fp32 logits
    differ from the fp64 oracle by 8.8e-4 (real_s2mlpv2.npz: oracle_fp64_maxdiff), and the reference's OWN 16-bit forwards
    are 0.386 (fp16) / 0.967 (bf16) away from its fp32 logits (real_s2mlpv2_lowp.npz, generated by running the reference
    in those dtypes, make_golden.py --only s2lowp).  Gates: fp32 = 5x the reference's fp32-vs-fp64 distance; 16 bit = 2x
    the reference's own 16-bit distance.  The conditioning-free check is test_s2mlpv2_teacher_forced_blocks below."""
    if dtype == torch.float32:
        return 5e-3
    z = np.load(os.path.join(GOLDEN, "real_s2mlpv2_lowp.npz"))
    return 2.0 * float(z["err_fp16" if dtype == torch.float16 else "err_bf16"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_s2mlpv2_teacher_forced_blocks(dtype):
    """Per-block parity at the real shapes (C = 192 @ 32x32, C = 384 @ 16x16) with no error amplification, for every one of the 18
    blocks of the BASELINE configuration: feed the reference's own block INPUT (fp32 golden, bs = 1) to the HIP block and compare
    with the reference's block OUTPUT.
    Gates, relative to max|out|: fp32 1e-5; 16 bit: measured against what the reference itself achieves on the same blocks in that
    dtype (stored next to the tensors: fp16 7e-4..1.1e-3, bf16 5.7e-3..1.1e-2) -- see the two bars below.
    s2_mlp_v2.py:53-92."""
    pkg = load_pkg()
    z = np.load(os.path.join(GOLDEN, "real_s2mlpv2_blocks.npz"))
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)["state_dicts"]["s2mlpv2"]
    model = ctor_for(pkg, "s2mlpv2")(**man["kwargs"]).eval()
    shapes = {k: tuple(s) for k, s in man["keys"]}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in portable_state_dict(shapes, seed=0).items()}, strict=True)
    model = model.to(DEV)
    gate = {torch.float32: 1e-5, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
    # round 6: ALL 18 blocks (4 at C = 192, 14 at C = 384).  A stage is an nn.Sequential, so the fixture stores the stage's first input and
    # every block's output: block i is fed the REFERENCE's output of block i - 1 (teacher forcing), never this library's
    nblocks = [int(n) for n in z["nblocks"]]
    assert nblocks == [4, 14]
    worst = 0.0
    ratios = []
    sfx = "bf16" if dtype == torch.bfloat16 else "fp16"
    ref_worst = max(float(z["s%d.b%02d/ref_relerr_%s" % (s, i, sfx)]) for s, nb in enumerate(nblocks) for i in range(nb)) if dtype != torch.float32 else 0.0
    for s, nb in enumerate(nblocks):
        xin = torch.from_numpy(z["s%d/in" % s])
        for i in range(nb):
            key = "s%d.b%02d" % (s, i)
            yout = torch.from_numpy(z[key + "/out"])
            with torch.no_grad():
                y = model.forward_block(s, i, xin.to(DEV).to(dtype))
            torch.cuda.synchronize()
            assert y.shape == yout.shape and y.dtype == dtype
            rel = ((y.float().cpu() - yout).abs().max() / yout.abs().max()).item()
            ref_rel = float(z[key + "/ref_relerr_" + ("bf16" if dtype == torch.bfloat16 else "fp16")]) if dtype != torch.float32 else 0.0
            print("block %-7s %-8s rel err %.3e (2^%.1f)   reference's own: %.3e" % (key, str(dtype)[6:], rel, np.log2(max(rel, 1e-30)), ref_rel))
            if dtype == torch.float32:
                assert rel < gate, (key, str(dtype), rel)
            else:
                # 16 bit: the bar is the reference's OWN error in this dtype.  The measure is ONE element (max-norm over 98 k .. 196 k
                # values), so block by block it scatters both ways -- over the 18 blocks this library's largest deviation is 0.6 .. 1.5 x
                # the reference's own on the same block (s1.b03 0.62, s1.b04 1.00 to the digit, s1.b05 1.46).  Two bars: no block may be
                # worse than 1.25 x the worst block of the reference itself, and on (geometric) average over the blocks this library
                # may not be less accurate than the reference (1.05: the box's rounding).
                assert rel < 1.25 * ref_worst, (key, str(dtype), rel, ref_worst)
                worst = max(worst, rel / ref_rel)
                ratios.append(rel / ref_rel)
            xin = yout
    if dtype != torch.float32:
        gmean = float(np.exp(np.mean(np.log(ratios))))
        print("block error relative to the reference's own on that block: worst %.2f, geometric mean over %d blocks %.2f" % (worst, len(ratios), gmean))
        assert len(ratios) == 18 and gmean < 1.05, (str(dtype), gmean)


BS256 = [
    # name, constructor kwargs (bench.py MODELS), images of the small batch, oracle family
    ("mixer_b16", "MLPMixerForImageClassification", dict(d_model=768, depth=12, patch_size=16, image_size=224), 4, "mixer"),
    ("gmlp_s", "gMLPForImageClassification", dict(image_size=224), 2, "gmlp"),
    ("resmlp_24", "ResMLPForImageClassification", dict(depth=24), 2, "resmlp"),
    ("vip_s7", "ViP", dict(image_size=224, patch_size=7, d_model=384, depth=18, segments=12, expansion_factor=3), 2, "vip"),
    ("s2mlpv2", "S2MLPv2", dict(), 2, "s2mlpv2"),
    ("s2mlpv2_cleanshift", "S2MLPv2", dict(), 2, "s2mlpv2"),
    ("asmlp_t", "AS_MLP", dict(), 2, "asmlp"),
    ("convmixer_1536_20", "ConvMixer", dict(dim=1536, depth=20), 2, "convmixer"),
    ("mixer_l16", "MLPMixerForImageClassification", dict(d_model=1024, depth=24, patch_size=16, image_size=224), 2, "mixer"),
    # round 5: the SURVEY 8(f) models have bs=256-only kernel choices too (mlpk_channel_mlp, mlpk_swin_spatial, persistent tiles)
    ("sparsemlp_t", "SparseMLP", dict(), 2, "sparsemlp"),
    ("hiremlp_s", "HireMLP", dict(), 2, "hiremlp"),
    ("msmlp_t", "MS_MLP", dict(), 2, "msmlp"),
    ("swinmlp_t", "SwinMLP", dict(), 2, "swinmlp"),
    ("cyclemlp_b1", "CycleMLP_B1", dict(), 2, "cyclemlp"),
]


# batch-independent to the bit.  Round 4: every tile of the library reduces its by-product LayerNorm statistics over planes of 32 columns in
# ONE order (include/mlpk.h row_part), so the models whose LayerNorms read those statistics -- the Mixers since their token LayerNorm moved
# into the token kernel -- are back in this set (round 3: sums over 32 / 64 / 128 columns "depending on the tile the batch size selects").
# (... and with them every other BASELINE configuration: gMLP-S, ViP-S7, S2-MLPv2 in both shift modes -- profiles/r04_bs256_parity.txt)
BIT_EQUAL = {"resmlp_24", "asmlp_t", "convmixer_1536_20", "mixer_b16", "mixer_l16", "gmlp_s", "vip_s7", "s2mlpv2", "s2mlpv2_cleanshift",
             "sparsemlp_t", "hiremlp_s", "msmlp_t", "swinmlp_t", "cyclemlp_b1"}


@pytest.mark.parametrize("name,ctor,kw,k,family", BS256)
def test_batch_256_rows_match_small_batch(name, ctor, kw, k, family):
    """Parity AT the benchmarked batch for every BASELINE configuration (configs 2-5: Mixer-B/16, gMLP-S, ResMLP-24, ViP-Small/7,
    S2-MLPv2 in both shift modes, AS-MLP-T, ConvMixer-1536/20, Mixer-L/16 at 256 images, bf16).  At bs = 256 the GEMM tiles are the
    persistent 256 x 256 tile and the generated q4 tile, the token kernels run full grids, the depthwise convolution its MFMA form --
    none of which the small golden batches reach.  Size-independent property: an image's logits do not depend on the batch it is in
    (eval mode), so rows [100, 100 + k) and the LAST k rows (the partial last round of the persistent kernels) of the
    bs = 256 forward must reproduce the same images run as a batch of k, and BOTH must meet the usual gate against the CPU oracle.
    Every tile computes a row's dot products in the same K order, so the rows are bit-equal (asserted for the models listed in
    BIT_EQUAL) unless something on the model's path still depends on the batch (a kernel choice with another summation order): one
    flipped rounding then propagates through the depth like any other 16-bit rounding, and the two evaluations end up one noise
    level apart -- asserted: within the gate of each other, reported: the actual difference.
    (Round 3: this test found mlpk_token_gemm multiplying the first token group of every tile after a workgroup's first by the
    wrong weights -- gMLP-S 5e-2 off at 256 images, correct at every golden batch size.)"""
    pkg = load_pkg()
    model = getattr(pkg.models_pytorch, ctor)(**kw).eval()
    mode = "shift" if name.endswith("cleanshift") else "reference_inplace"
    if name.endswith("cleanshift"):
        model.set_shift_mode("shift")
    # Round 5: weights from the portable generator (oracle/portable_init.py: LayerScale gamma 0.05-0.2, BatchNorm running statistics
    # and every norm's affine non-trivial), NOT torch's default init -- with that, ResMLP's gamma = 1e-5 put every token / channel
    # product 5 orders below bf16 resolution of the residual and ConvMixer's BatchNorm epilogue was the identity, so neither the oracle
    # gate nor bit-equality could see a wrong bs=256-only tile (res_mlp.py:38-43, conv_mixer.py:24-31).
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)["state_dicts"][name.replace("_cleanshift", "")]
    assert man["kwargs"] == kw
    sd = {kk: torch.from_numpy(v) for kk, v in portable_state_dict({kk: tuple(sh) for kk, sh in man["keys"]}, seed=0).items()}
    model.load_state_dict(sd, strict=True)
    sd = {kk: v.float() for kk, v in sd.items()}
    model = model.to(DEV)
    x = torch.from_numpy(portable_input((256, 3, 224, 224), seed=3)).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        big = model(x)
        small = model(x[100:100 + k].contiguous())
        last = model(x[256 - k:].contiguous())
    torch.cuda.synchronize()
    assert big.shape == (256, 1000) and bool(torch.isfinite(big.float()).all())
    ref = run_oracle(family, sd, x[100:100 + k].float().cpu(), kw, mode=mode) if family.startswith("s2") else run_oracle(family, sd, x[100:100 + k].float().cpu(), kw)
    gate = 2.0 * float(np.load(os.path.join(GOLDEN, "real_s2mlpv2_lowp.npz"))["err_bf16"]) if family == "s2mlpv2" else tol_for(torch.bfloat16, ref, real=True, name=name)
    d1 = (big[100:100 + k].float() - small.float()).abs().max().item()
    d2 = (big[256 - k:].float() - last.float()).abs().max().item()
    err = (small.float().cpu() - ref).abs().max().item()
    err_big = (big[100:100 + k].float().cpu() - ref).abs().max().item()
    print("%-20s bs=256 rows vs CPU oracle %.3e, batch of %d vs oracle %.3e (gate %.3e); bs=256 rows vs batch of %d: %.3e (mid) %.3e (last rows)%s"
          % (name, err_big, k, err, gate, k, d1, d2, "  [bit-equal]" if d1 == 0.0 and d2 == 0.0 else ""))
    assert err_big < gate and err < gate, (name, err_big, err, gate)
    assert d1 <= gate and d2 <= gate, (name, d1, d2, gate)
    if name in BIT_EQUAL:
        assert d1 == 0.0 and d2 == 0.0, (name, d1, d2)


def test_backbone_forward_tokens():
    pkg = load_pkg()
    from importlib import import_module
    mm = import_module("jittor-mlp_amd.models_pytorch.mlp_mixer")
    torch.manual_seed(1)
    bb = mm.MLPMixer(16, 32, 2).eval().to(DEV)
    t = torch.randn(3, 16, 32)
    sd = {k: v.detach().cpu() for k, v in bb.state_dict().items()}
    ref = t.clone()
    for i in range(2):
        ref = oracle.functional.mixer_block(sd, ref, "model.%d." % i)
    with torch.no_grad():
        out = bb(t.to(DEV))
    assert (out.cpu() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mixer_submodules_callable_like_the_reference(dtype):
    """mlp_mixer.py:6-27,33-38: `model.model[i]` (a block), `[i][0]` / `[i][1]` (token / channel PreNormResidual) and their `.fn`
    (FeedForward over Conv1d(k=1) / Linear) run on their own in the reference; here they do too, through the same HIP kernels, and
    chaining the blocks by hand reproduces the backbone's forward."""
    import torch.nn.functional as F
    pkg = load_pkg()
    mp = pkg.models_pytorch
    S, C, depth, ef = 49, 64, 2, 2
    torch.manual_seed(3)
    bb = mp.MLPMixer(num_patches=S, d_model=C, depth=depth, expansion_factor=ef).eval()
    for p in bb.parameters():
        p.data.add_(0.05 * torch.randn_like(p))            # non-trivial LayerNorm gamma / beta, biases
    x = torch.randn(3, S, C)
    xd = x.to(dtype).double()

    def ff_ref(ff, v, token):
        w1, b1, w2, b2 = (t.detach().double() for t in (ff.net[0].weight, ff.net[0].bias, ff.net[3].weight, ff.net[3].bias))
        if token:                                          # Conv1d(kernel_size=1) over dimension 1
            w1, w2 = w1.squeeze(-1), w2.squeeze(-1)
            h = oracle.gelu(torch.einsum("hs,bsc->bhc", w1, v) + b1.view(1, -1, 1))
            return torch.einsum("sh,bhc->bsc", w2, h) + b2.view(1, -1, 1)
        return F.linear(oracle.gelu(F.linear(v, w1, b1)), w2, b2)

    def pn_ref(pn, v, token):
        n = F.layer_norm(v, (C,), pn.norm.weight.detach().double(), pn.norm.bias.detach().double(), pn.norm.eps)
        return ff_ref(pn.fn, n, token) + v

    tol = 2e-5 if dtype == torch.float32 else 6e-2
    blk = bb.model[0]
    cases = [(blk[0].fn, ff_ref(blk[0].fn, xd, True)), (blk[1].fn, ff_ref(blk[1].fn, xd, False)),
             (blk[0], pn_ref(blk[0], xd, True)), (blk[1], pn_ref(blk[1], xd, False)),
             (blk, pn_ref(blk[1], pn_ref(blk[0], xd, True), False))]         # references from the CPU parameters, in float64
    bb = bb.to(DEV)
    xg = x.to(DEV).to(dtype)
    for mod, ref in cases:
        got = mod(xg)
        assert got.dtype == dtype and got.shape == ref.shape
        err = (got.double().cpu() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item()), (type(mod).__name__, str(dtype), err)
    # the blocks chained by hand == the backbone (same kernels up to the fused token kernel's rounding of the hidden)
    chained = bb.model(xg)
    whole = bb(xg)
    assert (chained.double() - whole.double()).abs().max().item() < tol * max(1.0, whole.abs().max().item())
    with pytest.raises(NotImplementedError):
        blk[0](x)                                          # CPU tensor: no fallback


def test_gmlp_resmlp_blocks_callable_like_the_reference():
    """g_mlp.py:33-39, res_mlp.py:50-57: `model.model[i](x)` runs one block on (B, S, C) tokens in the reference; here the block
    calls back into its backbone (same packed weights, same kernels) -- also inside the image-classification models -- and
    chaining the blocks by hand reproduces the whole backbone."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    torch.manual_seed(5)
    cases = [(mp.gMLP(d_model=32, d_ffn=64, seq_len=16, depth=2), oracle.functional.gmlp_block, (16, 32)),
             (mp.ResMLP(16, 32, 2, 2), oracle.functional.resmlp_block, (16, 32)),
             (mp.gMLPForImageClassification(image_size=32, patch_size=8, d_model=32, d_ffn=64, depth=2, num_classes=10),
              oracle.functional.gmlp_block, (16, 32))]
    for model, block_ref, (S, C) in cases:
        model = model.eval()
        for p in model.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        t = torch.randn(3, S, C)
        model = model.to(DEV)
        cur = t.to(DEV)
        ref = t.clone()
        for i in range(2):
            ref = block_ref(sd, ref, "model.%d." % i)
            cur = model.model[i](cur)
            assert (cur.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (type(model).__name__, i)
        if not type(model).__name__.endswith("Classification"):
            assert (model(t.to(DEV)).cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    lone = mp.g_mlp.gMLPBlock(32, 64, 16)
    with pytest.raises(NotImplementedError):
        lone(torch.randn(1, 16, 32).to(DEV))                      # a block outside a backbone stays a parameter container


def test_cycle_block_callable_like_the_reference():
    """cycle_mlp.py:194-197: `model.network[si][bi](x)` on channel-last (B, H, W, C)."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    Fo = oracle.functional
    torch.manual_seed(7)
    model = mp.CycleNet([1, 1], img_size=32, embed_dims=[16, 32], transitions=[True, True], mlp_ratios=[2, 2], num_classes=10,
                        mlp_fn=mp.cycle_mlp.CycleMLP).eval()
    for p in model.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    for (si, C, H, W) in ((0, 16, 8, 6), (2, 32, 4, 5)):
        t = torch.randn(2, H, W, C)
        pre = "network.%d.0." % si
        n = Fo.layer_norm(t, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
        ref = t + Fo.cyclemlp_attn(sd, n, pre + "attn.")
        n = Fo.layer_norm(ref, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
        ref = ref + Fo.linear(Fo.gelu(Fo.linear(n, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"],
                              sd[pre + "mlp.fc2.bias"])
        got = model.network[si][0](t.to(DEV))
        assert got.shape == ref.shape
        assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), si
        assert torch.equal(model.network[si](t.to(DEV)), got)                  # a stage is an nn.Sequential of such blocks
    # round 5: PatchEmbedOverlapping (cycle_mlp.py:213-215, NCHW out) and Downsample (:227-231, channel-last) on their own
    img = torch.randn(2, 3, 30, 26)
    ref = Fo.conv2d_im2col(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], 4, 2).permute(0, 3, 1, 2)
    got = model.patch_embed(img.to(DEV))
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    t = torch.randn(2, 7, 6, 16)
    ref = Fo.conv2d_im2col(t.permute(0, 3, 1, 2), sd["network.1.proj.weight"], sd["network.1.proj.bias"], 2, 1)
    got = model.network[1](t.to(DEV))
    assert got.shape == ref.shape == (2, 4, 3, 32) and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_asmlp_block_callable_like_the_reference(dtype):
    """as_mlp.py:149-162: `model.layers[l].blocks[b](x)` on (B, C, H, W); fp32 takes the unfused kernel sequence, bf16 the fused one."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    torch.manual_seed(11)
    model = mp.AS_MLP(img_size=32, patch_size=4, embed_dim=64, depths=[1, 2], shift_size=5, num_classes=10).eval()
    for p in model.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    tol = 2e-5 if dtype == torch.float32 else 6e-2
    for (li, bi, C, H, W) in ((0, 0, 64, 8, 8), (1, 1, 128, 4, 4)):
        t = torch.randn(2, C, H, W)
        ref = oracle.functional.asmlp_block(sd, t.to(dtype).float(), "layers.%d.blocks.%d." % (li, bi), 5)
        got = model.layers[li].blocks[bi](t.to(DEV).to(dtype))
        assert got.shape == ref.shape and got.dtype == dtype
        assert (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item()), (li, bi)
    # round 5: the stage-level modules run on their own too -- PatchEmbed (as_mlp.py:323-333), PatchMerging (:197-216), BasicLayer (:258-266)
    Fo = oracle.functional
    img = torch.randn(2, 3, 32, 32)
    ref = Fo.patch_embed(img.to(dtype).float(), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"]).permute(0, 3, 1, 2)
    ref = Fo.group_norm1(ref, sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"])
    got = model.patch_embed(img.to(DEV).to(dtype))
    assert got.shape == ref.shape == (2, 64, 8, 8)
    assert (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    t = torch.randn(2, 64, 8, 8)
    ref = Fo.asmlp_patch_merging(sd, t.to(dtype).float(), "layers.0.downsample.")
    got = model.layers[0].downsample(t.to(DEV).to(dtype))
    assert got.shape == ref.shape == (2, 128, 4, 4)
    assert (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    ref = Fo.asmlp_patch_merging(sd, Fo.asmlp_block(sd, t.to(dtype).float(), "layers.0.blocks.0.", 5), "layers.0.downsample.")
    got = model.layers[0](t.to(DEV).to(dtype))                                 # stage 0: one block, then the merging
    assert got.shape == ref.shape and (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    t = torch.randn(2, 128, 4, 4)
    ref = t.to(dtype).float()
    for bi in range(2):
        ref = Fo.asmlp_block(sd, ref, "layers.1.blocks.%d." % bi, 5)
    got = model.layers[1](t.to(DEV).to(dtype))                                 # the last stage has no downsample
    assert got.shape == ref.shape and (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    with pytest.raises(NotImplementedError):
        mp.as_mlp.PatchMerging((8, 8), 64, norm_layer=mp.as_mlp.MyNorm)(t.to(DEV))     # outside a model: a parameter container


def test_swin_and_msmlp_blocks_callable_like_the_reference():
    """swin_mlp.py:113-157 (`blocks[b](x)` on (B, H*W, C)), ms_mlp.py:48-78 (on (B, C, H, W))."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    Fo = oracle.functional
    torch.manual_seed(13)
    swin = mp.SwinMLP(img_size=32, patch_size=4, embed_dim=32, depths=[2, 2], num_heads=[2, 4], window_size=4, num_classes=10).eval()
    ms = mp.MS_MLP(img_size=32, patch_size=4, embed_dim=40, depths=[2, 1], shift_size=5, num_classes=10).eval()
    for model in (swin, ms):
        for p in model.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in swin.state_dict().items()}
    swin = swin.to(DEV)
    for (li, bi, C, hw, nh) in ((0, 1, 32, 8, 2), (1, 0, 64, 4, 4)):                   # a shifted block, and a stage whose map == one window
        blk = swin.layers[li].blocks[bi]
        t = torch.randn(2, hw * hw, C)
        ref = Fo.swinmlp_block(sd, t, "layers.%d.blocks.%d." % (li, bi), hw, hw, nh, 4, blk.shift_size if hw > 4 else 0)
        got = blk(t.to(DEV))
        assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), ("swin", li, bi)
    # round 5: PatchEmbed (swin_mlp.py:324-333), PatchMerging (:193-212) and a whole stage (BasicLayer, :258-266) run on their own too
    img = torch.randn(2, 3, 32, 32)
    ref = Fo.patch_embed(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"]).reshape(2, 64, 32)
    ref = Fo.layer_norm(ref, sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"])
    got = swin.patch_embed(img.to(DEV))
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())

    def merging(t, pre, hw):
        g = t.reshape(2, hw, hw, t.shape[-1])
        g = torch.cat([g[:, 0::2, 0::2, :], g[:, 1::2, 0::2, :], g[:, 0::2, 1::2, :], g[:, 1::2, 1::2, :]], dim=-1).reshape(2, hw * hw // 4, -1)
        return Fo.linear(Fo.layer_norm(g, sd[pre + "norm.weight"], sd[pre + "norm.bias"]), sd[pre + "reduction.weight"], None)

    t = torch.randn(2, 64, 32)
    ref = merging(t, "layers.0.downsample.", 8)
    got = swin.layers[0].downsample(t.to(DEV))
    assert got.shape == ref.shape == (2, 16, 64) and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    ref = t
    for bi in range(2):
        ref = Fo.swinmlp_block(sd, ref, "layers.0.blocks.%d." % bi, 8, 8, 2, 4, swin.layers[0].blocks[bi].shift_size)
    ref = merging(ref, "layers.0.downsample.", 8)
    got = swin.layers[0](t.to(DEV))                                            # stage 0: two blocks, then the merging
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    t = torch.randn(2, 16, 64)
    ref = t
    for bi in range(2):
        ref = Fo.swinmlp_block(sd, ref, "layers.1.blocks.%d." % bi, 4, 4, 4, 4, 0)
    got = swin.layers[1](t.to(DEV))                                            # the last stage has no downsample
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(NotImplementedError):
        mp.swin_mlp.PatchMerging((8, 8), 32)(t.to(DEV))                        # outside a model: a parameter container
    sd = {k: v.detach().clone() for k, v in ms.state_dict().items()}
    ms = ms.to(DEV)
    for (li, bi, C, hw) in ((0, 1, 40, 8), (1, 0, 80, 4)):
        blk = ms.layers[li].blocks[bi]
        t = torch.randn(2, C, hw, hw)
        ref = Fo.msmlp_block(sd, t, "layers.%d.blocks.%d." % (li, bi), tuple(blk.shift_dist), tuple(k for k, _ in blk.kernel_size))
        got = blk(t.to(DEV))
        assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), ("msmlp", li, bi)
    # round 5: PatchEmbed -- the model's and a stage's downsampling one (ms_mlp.py:255-262) -- and a whole stage (BasicLayer, :177-185)

    def embed(t, pre):
        e = Fo.layer_norm(Fo.patch_embed(t, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), sd[pre + "norm.weight"], sd[pre + "norm.bias"], eps=1e-6)
        return e.permute(0, 3, 1, 2)

    img = torch.randn(2, 3, 32, 32)
    ref, got = embed(img, "patch_embed."), ms.patch_embed(img.to(DEV))
    assert got.shape == ref.shape == (2, 40, 8, 8) and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    t = torch.randn(2, 40, 8, 8)
    ref, got = embed(t, "layers.0.downsample."), ms.layers[0].downsample(t.to(DEV))
    assert got.shape == ref.shape == (2, 80, 4, 4) and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    ref = t
    for bi in range(2):
        blk = ms.layers[0].blocks[bi]
        ref = Fo.msmlp_block(sd, ref, "layers.0.blocks.%d." % bi, tuple(blk.shift_dist), tuple(k for k, _ in blk.kernel_size))
    ref = embed(ref, "layers.0.downsample.")
    got = ms.layers[0](t.to(DEV))                                              # stage 0: two blocks, then the downsampling
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    t = torch.randn(2, 80, 4, 4)
    blk = ms.layers[1].blocks[0]
    ref = Fo.msmlp_block(sd, t, "layers.1.blocks.0.", tuple(blk.shift_dist), tuple(k for k, _ in blk.kernel_size))
    got = ms.layers[1](t.to(DEV))                                              # the last stage has no downsampling
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # the leaf modules: MS-MLP's own LayerNorm class in both data formats (ms_mlp.py:287-298), Swin-MLP's Mlp (swin_mlp.py:20-26)
    ln = ms.norm
    t = torch.randn(3, 5, 80)
    ref = Fo.layer_norm(t, sd["norm.weight"], sd["norm.bias"], eps=1e-6)
    assert (ln(t.to(DEV)).cpu() - ref).abs().max().item() < 2e-5
    lf = mp.ms_mlp.LayerNorm(24, data_format="channels_first").to(DEV)
    with torch.no_grad():
        lf.weight.uniform_(0.5, 1.5); lf.bias.normal_(0, 0.2)
    t = torch.randn(2, 24, 3, 5)
    ref = Fo.layer_norm(t.permute(0, 2, 3, 1), lf.weight.cpu(), lf.bias.cpu(), eps=1e-6).permute(0, 3, 1, 2)
    assert (lf(t.to(DEV)).cpu() - ref).abs().max().item() < 2e-5
    sd = {k: v.detach().clone().cpu() for k, v in swin.state_dict().items()}
    mlp = swin.layers[0].blocks[0].mlp
    t = torch.randn(2, 7, 32)
    ref = Fo.linear(Fo.gelu(Fo.linear(t, sd["layers.0.blocks.0.mlp.fc1.weight"], sd["layers.0.blocks.0.mlp.fc1.bias"])),
                    sd["layers.0.blocks.0.mlp.fc2.weight"], sd["layers.0.blocks.0.mlp.fc2.bias"])
    got = mlp(t.to(DEV))
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vip_and_s2_blocks_callable_like_the_reference(dtype):
    """vip.py:85-93 (`backbone.model[i](x)`), s2_mlp_v2.py:86-92 (`model.stages[s][1].model[i](x)`): blocks that are plain
    nn.Sequential in the reference, on channel-last (B, H, W, C).  fp32 = unfused kernel sequences, bf16 = the fused ones."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    Fo = oracle.functional
    torch.manual_seed(17)
    tol = 2e-5 if dtype == torch.float32 else 6e-2
    vip = mp.WeightedPermutator(4, 6, 32, 2, 8, expansion_factor=2).eval()
    s2 = mp.S2MLPv2(image_size=32, patch_size=[4, 2], d_model=[32, 64], depth=[2, 1], expansion_factor=[2, 2], num_classes=10).eval()
    for model in (vip, s2):
        for p in model.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in vip.state_dict().items()}
    vip = vip.to(DEV)
    t = torch.randn(2, 4, 6, 32).to(dtype)
    ref = Fo.vip_block(sd, t.float(), "model.1.", 8, True)
    got = vip.model[1](t.to(DEV))
    assert got.shape == ref.shape and got.dtype == dtype
    assert (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    sd = {k: v.detach().clone() for k, v in s2.state_dict().items()}
    s2 = s2.to(DEV)
    for (st, i, C, hw) in ((0, 1, 32, 8), (1, 0, 64, 4)):
        t = torch.randn(2, hw, hw, C).to(dtype)
        ref = Fo.s2v2_block(sd, t.float(), "stages.%d.1.model.%d." % (st, i), "reference_inplace")
        got = s2.stages[st][1].model[i](t.to(DEV))
        assert (got.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item()), (st, i)


def test_s2v1_and_convmixer_blocks_callable_like_the_reference():
    """s2_mlp_v1.py:47-52 (`model.stages[s][1].model[i](x)` on (B, H, W, C)), conv_mixer.py:23-32 (`model.blocks[i](x)` on (B, C, H, W))."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    Fo = oracle.functional
    torch.manual_seed(19)
    s2 = mp.S2MLPv1(image_size=32, patch_size=[4, 2], d_model=[32, 64], depth=[2, 1], expansion_factor=[2, 2], num_classes=10).eval()
    cm = mp.ConvMixer(32, 2, kernel_size=5, patch_size=4, n_classes=10).eval()
    for model in (s2, cm):
        for p in model.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
    for m in cm.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    sd = {k: v.detach().clone() for k, v in s2.state_dict().items()}
    s2 = s2.to(DEV)
    for (st, i, C, hw) in ((0, 1, 32, 8), (1, 0, 64, 4)):
        t = torch.randn(2, hw, hw, C)
        ref = Fo.s2v1_block(sd, t, "stages.%d.1.model.%d." % (st, i), "reference_inplace")
        got = s2.stages[st][1].model[i](t.to(DEV))
        assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (st, i)
    sd = {k: v.detach().clone() for k, v in cm.state_dict().items()}
    cm = cm.to(DEV)
    t = torch.randn(2, 32, 8, 8)
    pre = "blocks.1."
    d = Fo.depthwise_conv_same(t, sd[pre + "0.fn.0.weight"], sd[pre + "0.fn.0.bias"])
    ref = t + Fo.batch_norm_eval(Fo.gelu(d), sd, pre + "0.fn.2")
    ref = Fo.batch_norm_eval(Fo.gelu(Fo.conv1x1(ref, sd[pre + "1.weight"], sd[pre + "1.bias"])), sd, pre + "3")
    got = cm.blocks[1](t.to(DEV))
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_s2_stage_callable_on_its_own_like_the_reference(dtype):
    """s2_mlp_v2.py:71-92 / s2_mlp_v1.py:27-53: `S2Block(d_model, depth, expansion_factor)` is a module of its own in the reference -- (B, C, H, W)
    in, permute, the depth blocks, permute back (SURVEY 8b lists it as a secondary constructor).  Built standalone AND taken out of a model
    (`model.stages[s][1]`), in both shift semantics, against the oracle's blocks chained by hand."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    Fo = oracle.functional
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    for ver, mod, block_ref in ((2, mp.s2_mlp_v2, Fo.s2v2_block), (1, mp.s2_mlp_v1, Fo.s2v1_block)):
        torch.manual_seed(31 + ver)
        stage = mod.S2Block(32, 2, expansion_factor=3).eval()
        for p in stage.parameters():
            p.data.add_(0.05 * torch.randn_like(p))
        sd = {k: v.detach().clone() for k, v in stage.state_dict().items()}
        x = torch.randn(2, 32, 6, 5)
        stage = stage.to(DEV)
        for mode in ("reference_inplace", "shift"):
            stage.shift_mode = mode
            ref = x.to(dtype).float().permute(0, 2, 3, 1)
            for i in range(2):
                ref = block_ref(sd, ref, "model.%d." % i, mode)
            ref = ref.permute(0, 3, 1, 2)
            got = stage(x.to(DEV).to(dtype))
            assert got.shape == ref.shape and got.dtype == dtype
            err = (got.float().cpu() - ref).abs().max().item()
            assert err < tol * max(1.0, ref.abs().max().item()), (ver, mode, str(dtype), err)
        with pytest.raises(NotImplementedError):
            stage(x)                                               # CPU tensor: no fallback
    # a stage taken out of a model runs with the model's parameters and shift mode
    torch.manual_seed(37)
    model = mp.S2MLPv2(image_size=32, patch_size=[4, 2], d_model=[16, 32], depth=[1, 2], expansion_factor=[3, 3], num_classes=10).eval()
    model.set_shift_mode("shift")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    x = torch.randn(2, 32, 4, 4)
    ref = x.permute(0, 2, 3, 1)
    for i in range(2):
        ref = Fo.s2v2_block(sd, ref, "stages.1.1.model.%d." % i, "shift")
    got = model.stages[1][1](x.to(DEV))
    assert (got.cpu() - ref.permute(0, 3, 1, 2)).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_hire_block_callable_like_the_reference():
    """hire_mlp.py:176-187: `model.layers[l].model[b](x)` on channel-last (B, H, W, C); block 1 of a stage is the cross-region one."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    Fo = oracle.functional
    z = np.load(os.path.join(GOLDEN, "tiny_hiremlp.npz"))
    kw = json.loads(str(z["kwargs"]))
    torch.manual_seed(23)
    model = mp.HireMLP(**kw).eval()
    for p in model.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    interval = kw.get("cross_region_interval", 2)
    for li, stage in enumerate(model.layers):
        h, w, C = stage.geom[:3]
        for bi in range(len(stage.model)):
            t = torch.randn(2, 7, 9, C)                               # neither side a multiple of the region size
            pre = "layers.%d.model.%d." % (li, bi)
            n = Fo.layer_norm(t, sd[pre + "0.norm.weight"], sd[pre + "0.norm.bias"])
            ref = t + Fo.hiremlp_block(sd, n, pre + "0.fn.0.", h, w, stage.model[bi][0].fn[0].step or 1, (bi + 1) % interval == 0)
            n = Fo.layer_norm(ref, sd[pre + "1.norm.weight"], sd[pre + "1.norm.bias"])
            ref = ref + Fo.linear(Fo.gelu(Fo.linear(n, sd[pre + "1.fn.0.weight"], sd[pre + "1.fn.0.bias"])), sd[pre + "1.fn.3.weight"],
                                  sd[pre + "1.fn.3.bias"])
            got = stage.model[bi](t.to(DEV))
            assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (li, bi)
        # round 5: the whole stage (hire_mlp.py:182-186) and its PatchEmbedding (:17-31, NCHW) on their own
        t = torch.randn(2, 7, 9, C)
        ref = t
        for bi in range(len(stage.model)):
            pre = "layers.%d.model.%d." % (li, bi)
            n = Fo.layer_norm(ref, sd[pre + "0.norm.weight"], sd[pre + "0.norm.bias"])
            ref = ref + Fo.hiremlp_block(sd, n, pre + "0.fn.0.", h, w, stage.model[bi][0].fn[0].step or 1, (bi + 1) % interval == 0)
            n = Fo.layer_norm(ref, sd[pre + "1.norm.weight"], sd[pre + "1.norm.bias"])
            ref = ref + Fo.linear(Fo.gelu(Fo.linear(n, sd[pre + "1.fn.0.weight"], sd[pre + "1.fn.0.bias"])), sd[pre + "1.fn.3.weight"],
                                  sd[pre + "1.fn.3.bias"])
        pm = "layers.%d.patch_merge.1.reduction.0." % li
        if stage.pooling:
            ref = Fo.conv2d_im2col(ref.permute(0, 3, 1, 2), sd[pm + "weight"], sd[pm + "bias"], 2, 1)
            tn = torch.randn(2, C, 7, 9)
            rm = Fo.conv2d_im2col(tn, sd[pm + "weight"], sd[pm + "bias"], 2, 1).permute(0, 3, 1, 2)
            gm = stage.patch_merge[1](tn.to(DEV))
            assert gm.shape == rm.shape and (gm.cpu() - rm).abs().max().item() < 2e-5 * max(1.0, rm.abs().max().item()), li
        else:
            with pytest.raises(NotImplementedError):
                stage.patch_merge[1](torch.randn(2, C, 7, 9).to(DEV))
        got = stage(t.to(DEV))
        assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), li
    img = torch.randn(2, 3, 36, 28)
    ref = Fo.conv2d_im2col(img, sd["patcher.reduction.0.weight"], sd["patcher.reduction.0.bias"], kw.get("patch_size", 4), 3)
    if "patcher.reduction.1.1.weight" in sd:
        ref = Fo.layer_norm(ref, sd["patcher.reduction.1.1.weight"], sd["patcher.reduction.1.1.bias"])
    got = model.patcher(img.to(DEV))
    assert got.shape == ref.permute(0, 3, 1, 2).shape and (got.cpu() - ref.permute(0, 3, 1, 2)).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_sparsemlp_block_callable_like_the_reference():
    """sparse_mlp.py:84-104: `model.layers[l].model[b](x)` on (B, C, H, W) at the stage's resolution."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    z = np.load(os.path.join(GOLDEN, "tiny_sparsemlp.npz"))
    kw = json.loads(str(z["kwargs"]))
    torch.manual_seed(29)
    model = mp.SparseMLP(**kw).eval()
    for p in model.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    for li, stage in enumerate(model.layers):
        H, W, C = stage.geom[:3]
        bi = len(stage.model) - 1
        t = torch.randn(2, C, H, W)
        ref = oracle.functional.sparsemlp_block(sd, t, "layers.%d.model.%d." % (li, bi))
        got = stage.model[bi](t.to(DEV))
        assert got.shape == ref.shape
        assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (li, bi)
        # round 5: the whole stage (sparse_mlp.py:106-110) and its PatchMerging (:33-50, channel-last) on their own
        Fo = oracle.functional
        ref = t
        for b_i in range(len(stage.model)):
            ref = Fo.sparsemlp_block(sd, ref, "layers.%d.model.%d." % (li, b_i))
        if stage.pooling:
            ref = Fo.sparsemlp_patch_merging(sd, ref.permute(0, 2, 3, 1), "layers.%d.patch_merge.1." % li).permute(0, 3, 1, 2)
            tl = torch.randn(2, H, W, C)
            rm = Fo.sparsemlp_patch_merging(sd, tl, "layers.%d.patch_merge.1." % li)
            gm = stage.patch_merge[1](tl.to(DEV))
            assert gm.shape == rm.shape and (gm.cpu() - rm).abs().max().item() < 2e-5 * max(1.0, rm.abs().max().item()), li
        else:
            with pytest.raises(NotImplementedError):
                stage.patch_merge[1](torch.randn(2, H, W, C).to(DEV))           # built but unused by the reference too (:80-84, :108-109)
        got = stage(t.to(DEV))
        assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), li


def test_cpu_input_raises():
    pkg = load_pkg()
    model = pkg.MLPMixerForImageClassification(d_model=32, depth=1, patch_size=8, image_size=32, num_classes=10)
    with pytest.raises(NotImplementedError):
        model(torch.randn(1, 3, 32, 32))


def test_shift_module_dropin():
    """`Shift` (the reference's native op): same call surface, identity for k=1, NotImplementedError on CPU."""
    pkg = load_pkg()
    Shift = pkg.models_pytorch.Shift
    x = torch.randn(2, 10, 6, 5)
    for k in (3, 5):
        for dim in (2, 3):
            out = Shift(k, dim)(x.to(DEV))
            assert torch.equal(out.cpu(), oracle.axial_shift_nchw(x, k, dim))
    assert Shift(1, 2)(x) is x
    with pytest.raises(NotImplementedError):
        Shift(3, 2)(x)
    with pytest.raises(AssertionError):
        Shift(4, 2)
    with pytest.raises(AssertionError):
        Shift(3, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inner_modules_callable_like_the_reference(dtype):
    """The sub-block boundary BASELINE.json's north star names: `spatial_shift1/2` (s2_mlp_v2.py:15-29) and `torch_shift`
    (shift_cuda.py:195-205) as functions, `SpatialGatingUnit` (g_mlp.py:10-22), `SplitAttention` / `ParallelWeightedSum`
    (vip.py:24-57), `S2Attention` (s2_mlp_v2.py:53-69), `Spatial_Shift` (s2_mlp_v1.py:15-25) and `AxialShift` (as_mlp.py:27-95)
    called on their own -- through the HIP kernels -- against the CPU oracle's restatement of the same reference lines."""
    pkg = load_pkg()
    mp = pkg.models_pytorch
    from importlib import import_module
    s2 = import_module("jittor-mlp_amd.models_pytorch.s2_mlp_v2")
    s1 = import_module("jittor-mlp_amd.models_pytorch.s2_mlp_v1")
    vip = import_module("jittor-mlp_amd.models_pytorch.vip")
    gm = import_module("jittor-mlp_amd.models_pytorch.g_mlp")
    asm = import_module("jittor-mlp_amd.models_pytorch.as_mlp")
    ut = import_module("jittor-mlp_amd.models_pytorch.utils")
    F = oracle.functional
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    torch.manual_seed(11)

    def close(got, ref, what):
        err = (got.float().cpu().double() - ref.double()).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (what, str(dtype), err)

    # spatial shifts: bit-exact moves, in place, both semantics
    x = torch.randn(2, 6, 5, 16).to(dtype)
    for fn, ref in ((s2.spatial_shift1, F.spatial_shift1), (s2.spatial_shift2, F.spatial_shift2)):
        for mode in ("reference_inplace", "shift"):
            xd = x.clone().to(DEV)
            y = fn(xd, mode=mode)
            assert y is xd
            assert torch.equal(y.cpu(), ref(x.clone(), mode=mode)), (fn.__name__, mode)
    xd = x.clone().to(DEV)
    assert torch.equal(s1.Spatial_Shift()(xd).cpu(), F.spatial_shift1(x.clone()))          # s2_mlp_v1.py:19-25 == spatial_shift1
    xn = torch.randn(2, 10, 7, 9).to(dtype)
    for dim in (2, 3):
        assert torch.equal(ut.torch_shift(xn.to(DEV), 5, dim).cpu(), F.axial_shift_nchw(xn, 5, dim))
    # SplitAttention (ViP's and S2's are the same module)
    for mod in (vip.SplitAttention(32), s2.SplitAttention(32)):
        mod = mod.eval()
        xa = torch.randn(2, 3, 4, 5, 32).to(dtype)
        ref = F.split_attention(xa[:, 0].double(), xa[:, 1].double(), xa[:, 2].double(), mod.mlp1.weight.detach().double(), mod.mlp2.weight.detach().double())
        close(mod.to(DEV)(xa.to(DEV)), ref, "SplitAttention")
    # S2Attention, both shift modes
    att = s2.S2Attention(32).eval()
    xs = torch.randn(2, 6, 5, 32).to(dtype)
    for mode in ("reference_inplace", "shift"):
        att.shift_mode = mode
        t = torch.nn.functional.linear(xs.double(), att.mlp1.weight.detach().double(), att.mlp1.bias.detach().double())
        x1, x2, x3 = F.spatial_shift1(t[..., :32].clone(), mode=mode), F.spatial_shift2(t[..., 32:64].clone(), mode=mode), t[..., 64:]
        a = F.split_attention(x1, x2, x3, att.split_attention.mlp1.weight.detach().double(), att.split_attention.mlp2.weight.detach().double())
        ref = torch.nn.functional.linear(a, att.mlp2.weight.detach().double(), att.mlp2.bias.detach().double())
        close(att.to(DEV)(xs.to(DEV)), ref, "S2Attention " + mode)
        att = att.cpu()
    # SpatialGatingUnit
    sgu = gm.SpatialGatingUnit(24, 10).eval()
    xg = torch.randn(3, 10, 48).to(dtype)
    u, v = xg.double().chunk(2, dim=-1)
    v = torch.nn.functional.layer_norm(v, (24,), sgu.norm.weight.detach().double(), sgu.norm.bias.detach().double())
    v = torch.einsum("ts,bsf->btf", sgu.spatial_proj.weight.detach().double().squeeze(-1), v) + sgu.spatial_proj.bias.detach().double().view(1, -1, 1)
    close(sgu.to(DEV)(xg.to(DEV)), u * v, "SpatialGatingUnit")
    # ParallelWeightedSum of a ViP block: the three branch Linears with their rearranges, then the split attention
    vm = mp.ViP(image_size=32, patch_size=8, d_model=32, depth=1, segments=4, expansion_factor=2).eval()
    pws = [m for m in vm.modules() if isinstance(m, vip.ParallelWeightedSum)][0]
    xv = torch.randn(2, 4, 4, 32).to(dtype)
    sd = {k: p.detach().double() for k, p in pws.state_dict().items()}
    xh = F.vip_unpermute_h(torch.nn.functional.linear(F.vip_permute_h(xv.double(), 4), sd["fns.0.1.weight"], sd["fns.0.1.bias"]), 4)
    xw = F.vip_unpermute_w(torch.nn.functional.linear(F.vip_permute_w(xv.double(), 4), sd["fns.1.1.weight"], sd["fns.1.1.bias"]), 4)
    xc = torch.nn.functional.linear(xv.double(), sd["fns.2.weight"], sd["fns.2.bias"])
    ref = F.split_attention(xh, xw, xc, sd["split_attention.mlp1.weight"], sd["split_attention.mlp2.weight"])
    close(pws.to(DEV)(xv.to(DEV)), ref, "ParallelWeightedSum")
    # AxialShift
    ax = asm.AxialShift(32, 5).eval()
    xa = torch.randn(2, 32, 7, 6).to(dtype)
    sda = {"a." + k: p.detach().double() for k, p in ax.state_dict().items()}
    close(ax.to(DEV)(xa.to(DEV)), F.asmlp_axial_shift(sda, xa.double(), "a.", 5), "AxialShift")
    # round 4: the small modules that were parameter containers -- ResMLP's Aff and FeedForward (res_mlp.py:11-32), AS-MLP's Mlp (as_mlp.py:8-24)
    rm = import_module("jittor-mlp_amd.models_pytorch.res_mlp")
    aff = rm.Aff(32).eval()
    with torch.no_grad():
        aff.alpha.copy_(torch.randn(1, 1, 32) * 0.3 + 1.0)
        aff.beta.copy_(torch.randn(1, 1, 32) * 0.2)
    xr = torch.randn(3, 5, 32).to(dtype)
    close(aff.to(DEV)(xr.to(DEV)), xr.double() * aff.alpha.detach().double().cpu() + aff.beta.detach().double().cpu(), "Aff")
    ff = rm.FeedForward(32, 80).eval()
    sdf = {k: p.detach().double() for k, p in ff.state_dict().items()}
    want = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr.double(), sdf["net.0.weight"], sdf["net.0.bias"])),
                                      sdf["net.3.weight"], sdf["net.3.bias"])
    close(ff.to(DEV)(xr.to(DEV)), want, "ResMLP FeedForward")
    mlp = asm.Mlp(32, 64).eval()
    sdm = {k: p.detach().double() for k, p in mlp.state_dict().items()}
    want = torch.nn.functional.conv2d(torch.nn.functional.gelu(torch.nn.functional.conv2d(xa.double(), sdm["fc1.weight"], sdm["fc1.bias"])),
                                      sdm["fc2.weight"], sdm["fc2.bias"])
    got = mlp.to(DEV)(xa.to(DEV))
    assert got.shape == (2, 32, 7, 6)
    close(got, want, "AS-MLP Mlp")
    with pytest.raises(NotImplementedError):
        mlp.cpu()(xa)                                                            # no CPU path, as everywhere
    # round 4: PreNormResidual of ViP / S2-MLP v1 / v2 (vip.py:6-13, s2_mlp_v2.py:31-38) on its own -- around the channel MLP Sequential
    # (two widths: the GEMM pair, and the one-kernel form of a narrow stage) and around a module that is callable itself (S2Attention)
    nn = torch.nn
    for mod_, width, hid in ((vip.PreNormResidual, 256, 512), (s2.PreNormResidual, 96, 384), (s1.PreNormResidual, 64, 200)):
        pn = mod_(width, nn.Sequential(nn.Linear(width, hid), nn.GELU(), nn.Dropout(0.), nn.Linear(hid, width), nn.Dropout(0.))).eval()
        with torch.no_grad():
            pn.norm.weight.copy_(torch.randn(width) * 0.3 + 1.0)
            pn.norm.bias.copy_(torch.randn(width) * 0.2)
        xp = torch.randn(2, 5, 7, width).to(dtype)
        pd = copy.deepcopy(pn).double()
        want = (pd.fn(pd.norm(xp.double())) + xp.double()).detach()               # the held torch modules ARE the reference's lines
        close(pn.to(DEV)(xp.to(DEV)), want, "%s around the channel MLP, width %d" % (mod_.__module__, width))
    pa = s2.PreNormResidual(32, s2.S2Attention(32)).eval()
    att2 = pa.fn
    xs2 = torch.randn(2, 6, 5, 32).to(dtype)
    ln = torch.nn.functional.layer_norm(xs2.double(), (32,), pa.norm.weight.detach().double(), pa.norm.bias.detach().double())
    t = torch.nn.functional.linear(ln, att2.mlp1.weight.detach().double(), att2.mlp1.bias.detach().double())
    x1, x2, x3 = F.spatial_shift1(t[..., :32].clone(), mode=att2.shift_mode), F.spatial_shift2(t[..., 32:64].clone(), mode=att2.shift_mode), t[..., 64:]
    a = F.split_attention(x1, x2, x3, att2.split_attention.mlp1.weight.detach().double(), att2.split_attention.mlp2.weight.detach().double())
    want = torch.nn.functional.linear(a, att2.mlp2.weight.detach().double(), att2.mlp2.bias.detach().double()) + xs2.double()
    close(pa.to(DEV)(xs2.to(DEV)), want, "PreNormResidual around S2Attention")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", ["hiremlp", "sparsemlp", "convmixer", "vip_unweighted"])
def test_leaf_modules_callable_like_the_reference(tag, dtype):
    """Round 6 (VERDICT r5 missing 4): the inner modules of a block that only held parameters -- Hire-MLP's two `PreNormResidual`s, `HireMLPBlock`
    (hire_mlp.py:8-15,97-152) and `FeedForward` (:33-42), Sparse-MLP's three `PreNormResidual`s and `sMLPBlock` (sparse_mlp.py:8-15,61-74),
    ConvMixer's `Residual` (conv_mixer.py:5-11), ViP's `ParallelSum` (vip.py:16-22) -- run on their own inside a model, fed what the REFERENCE fed them in the tiny fixture's
    forward and compared with what the reference's module returned (tests/golden/leaf_modules.npz, make_golden.py --only leaf: forward hooks
    on the reference model)."""
    pkg = load_pkg()
    z = np.load(os.path.join(GOLDEN, "leaf_modules.npz"))
    t = np.load(os.path.join(GOLDEN, "tiny_%s.npz" % tag))
    model = ctor_for(pkg, tag)(**json.loads(str(t["kwargs"]))).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(t[k]) for k in t.files if k.startswith("sd/")}, strict=True)
    model = model.to(DEV)
    mods = dict(model.named_modules())
    worst = 0.0
    for pth in json.loads(str(z[tag + "/paths"])):
        xin = torch.from_numpy(z["%s/%s/in" % (tag, pth)])
        want = torch.from_numpy(z["%s/%s/out" % (tag, pth)])
        with torch.no_grad():
            got = mods[pth](xin.to(DEV).to(dtype))
        torch.cuda.synchronize()
        assert got.shape == want.shape and got.dtype == dtype, (tag, pth, tuple(got.shape), tuple(want.shape))
        err = (got.float().cpu() - want).abs().max().item()
        tol = (1e-5 if dtype == torch.float32 else 1.5e-2) * max(1.0, want.abs().max().item())
        assert err < tol, (tag, pth, str(dtype), err, tol)
        worst = max(worst, err / max(1.0, want.abs().max().item()))
    print("leaf modules %s %s: worst relative deviation %.3e" % (tag, str(dtype)[6:], worst))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d_model", [100, 50, 33])
def test_mixer_takes_any_width_like_the_reference(d_model, dtype):
    """Round 6 (VERDICT r5 missing 5; mlp_mixer.py:46-54 takes any d_model): widths that are not whole 16-byte chunks run with zero padding
    channels (multiples of 4, of 2, odd; the expanded width 4 * 33 = 132 is no multiple of 8 either) -- logits against the CPU oracle on
    the same random weights, and the token-level backbone too."""
    pkg = load_pkg()
    torch.manual_seed(7 + d_model)
    kw = dict(d_model=d_model, depth=2, patch_size=8, image_size=32, num_classes=10, expansion_factor=4)
    model = pkg.models_pytorch.MLPMixerForImageClassification(**kw).eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():                       # non-trivial norms and biases
            if n_.endswith("norm.weight") or n_ == "active.weight":
                p_.copy_(1.0 + 0.3 * torch.randn_like(p_))
            elif n_.endswith("bias"):
                p_.copy_(0.2 * torch.randn_like(p_))
    x = torch.randn(3, 3, 32, 32)
    want = oracle.mixer_forward({k: v.detach() for k, v in model.state_dict().items()}, x)
    model = model.to(DEV)
    with torch.no_grad():
        got = model(x.to(DEV).to(dtype))
    torch.cuda.synchronize()
    assert got.dtype == dtype and tuple(got.shape) == (3, 10)
    err = (got.float().cpu() - want).abs().max().item()
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.5e-2}[dtype] * max(1.0, want.abs().max().item())
    assert err < tol, (d_model, str(dtype), err, tol)
    # a second batch size on the same model, and the same rows must not depend on it
    with torch.no_grad():
        one = model(x[:1].to(DEV).to(dtype))
    assert torch.equal(one, got[:1])
