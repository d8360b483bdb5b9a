"""the reference's own harness protocol (compare.py) on its six model
configurations (compare.py:35-138): uniform[0,1) input (compare.py:23), eval mode, shared weights,
10 warm-up + timed forwards bracketed by device synchronisation (compare.py:149-158), the two print lines
(:159) and the acceptance metric  mean(|x+1 - (y+1)| / |y+1|) < 1e-3  (compare.py:179-186) -- with y = the
logits the PyTorch reference produced (committed golden) instead of its Jittor twin."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle.portable_init import portable_state_dict, portable_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BS = 4
CONFIGS = {
    "MLPMixer": ("MLPMixerForImageClassification", dict(image_size=(224, 224), patch_size=16, in_channels=3, num_classes=1000, d_model=256, depth=12)),
    "gMLP": ("gMLPForImageClassification", dict(image_size=(224, 224), patch_size=16, in_channels=3, num_classes=1000, d_model=256, d_ffn=1536, depth=30)),
    "ResMLP": ("ResMLPForImageClassification", dict(in_channels=3, image_size=(224, 224), patch_size=16, d_model=384, depth=12, num_classes=1000, expansion_factor=4)),
    "ViP": ("ViP", dict(image_size=(224, 224), patch_size=(16, 8), in_channels=3, num_classes=1000, d_model=256, depth=30, segments=16, weighted=True)),
    "ConvMixer": ("ConvMixer", dict(dim=1568, depth=20)),
    "S2MLPv2": ("S2MLPv2", dict(in_channels=3, image_size=(224, 224), patch_size=[(7, 7), (2, 2)], d_model=[192, 384], depth=[4, 14], num_classes=1000, expansion_factor=[3, 3])),
}


def compare_metric(x, y):
    """compare.py:181-184."""
    x = x + 1
    y = y + 1
    return float(np.mean(np.abs(x - y) / np.abs(y)))


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", sorted(CONFIGS))
def test_compare_py_protocol(model_name):
    pkg = load_pkg()
    cls, kw = CONFIGS[model_name]
    model = getattr(pkg.models_pytorch, cls)(**kw).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in portable_state_dict(shapes, seed=1).items()}, strict=True)
    model = model.cuda()
    img = torch.from_numpy(portable_tensor("compare.input", (BS, 3, 224, 224), 0.0, 1.0, seed=1)).cuda()
    golden = np.load(os.path.join(GOLDEN, "compare_logits.npz"))
    ref = golden[model_name]
    turns = 20
    for dtype, threshold in ((torch.float32, 1e-3), (torch.float16, 1e-3)):
        x = img.to(dtype)
        with torch.no_grad():
            for _ in range(10):
                result = model(x)
            torch.cuda.synchronize()
            sta = time.time()
            for _ in range(turns):
                result = model(x)
            torch.cuda.synchronize()
            end = time.time()
        tc_time = round((end - sta) / turns, 5)
        tc_fps = round(BS * turns / (end - sta), 0)
        print(f"- MI355X {model_name} ({str(dtype)[6:]}) forward average time cost: {tc_time}, Batch Size: {BS}, FPS: {tc_fps}")
        diff = compare_metric(result.float().cpu().numpy(), ref)
        # conditioning floor: how far the reference's OWN logits move under a one-ulp input change (fp32) or
        # fp16-rounded parameters (16-bit), recorded by make_compare_golden.py.  Where that alone exceeds
        # compare.py's 1e-3 (ViP depth 30 in fp16; S2-MLPv2's in-place shift in any precision, see
        # tests/test_gpu_models.py) the gate is 8 x the floor, and beyond 1e-2 the configuration is chaotic
        # at that precision and the figure is only reported.
        floor = float(golden[model_name + ("__sens32" if dtype == torch.float32 else "__sens16")])
        if floor > 1e-2:
            print(f"[*] {model_name} ({str(dtype)[6:]}) relative error {diff}; reference self-sensitivity {floor} (not gated)")
            continue
        threshold = max(threshold, 8.0 * floor)
        assert diff < threshold, f"[*] {model_name} forward fails..., Relative Error: {diff}"
        print(f"[*] {model_name} forword passes with Relative Error {diff}")


@pytest.mark.gpu
def test_compare_s2mlpv2_clean_shift_against_oracle():
    """The S2MLPv2 compare.py configuration is chaotic with synthetic weights (the reference's own logits move
    3e-2 for a one-ulp input change; SplitAttention soft-maxes a sum over 3*H*W pixels), so its hard gate is
    the race-free `shift` mode against the fp64 CPU oracle in the same mode on one image, in compare.py's
    metric.  The bound is 4 x the measured conditioning floor: the larger of (fp32 oracle vs fp64 oracle; 5e-4
    to 6e-3 depending on the host's BLAS) and (this engine's own logits under a one-ulp input change, about
    2e-3), capped at 2e-2 -- a wrong kernel shows up as O(1)."""
    import oracle

    pkg = load_pkg()
    cls, kw = CONFIGS["S2MLPv2"]
    model = getattr(pkg.models_pytorch, cls)(**kw).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in portable_state_dict(shapes, seed=1).items()}
    model.load_state_dict(sd, strict=True)
    model.set_shift_mode("shift")
    model = model.cuda()
    x = torch.from_numpy(portable_tensor("compare.input", (1, 3, 224, 224), 0.0, 1.0, seed=1))
    ref64 = oracle.s2mlpv2_forward({k: v.double() for k, v in sd.items()}, x.double(), mode="shift").numpy()
    ref32 = oracle.s2mlpv2_forward(sd, x, mode="shift").numpy()
    sign = torch.from_numpy(portable_tensor("compare.sign", tuple(x.shape), -1.0, 1.0, seed=2)).sign()
    with torch.no_grad():
        out = model(x.cuda()).float().cpu().numpy().astype(np.float64)
        out_ulp = model((x * (1.0 + sign * 2.0 ** -23)).cuda()).float().cpu().numpy().astype(np.float64)
    floor = max(compare_metric(ref32, ref64), compare_metric(out_ulp, out))
    diff = compare_metric(out, ref64)
    print(f"[*] S2MLPv2 clean-shift vs fp64 oracle relative error {diff}; conditioning floor {floor}")
    assert diff < min(2e-2, max(1e-3, 4.0 * floor)), (diff, floor)


def test_compare_metric_is_the_reference_formula():
    x = np.array([[0.5, -0.25]], dtype=np.float32)
    y = np.array([[0.25, -0.5]], dtype=np.float32)
    assert abs(compare_metric(x, y) - np.mean(np.abs((x + 1) - (y + 1)) / np.abs(y + 1))) < 1e-12
