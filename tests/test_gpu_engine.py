"""-m gpu: host-side contracts of the engine that only show on a device: any input resolution for the models the
reference does not pin to one (workspaces keyed by the input shape), launches on the INPUT's device rather than the
current one, train-mode warning."""
import warnings

import pytest
import torch

import oracle
from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_convmixer_and_hiremlp_at_two_resolutions():
    """conv_mixer.py:13-41 / hire_mlp.py:155-229 derive H and W from the input; a second forward at another resolution
    with the same batch size must work (and the first resolution must still be right afterwards)."""
    pkg = load_pkg()
    torch.manual_seed(0)
    cm = pkg.ConvMixer(dim=32, depth=2, kernel_size=5, patch_size=4, n_classes=10).eval()
    sd = {k: v.detach().clone() for k, v in cm.state_dict().items()}
    cm = cm.to(DEV)
    for hw in ((32, 32), (48, 40), (32, 32)):
        x = torch.randn(2, 3, *hw)
        with torch.no_grad():
            out = cm(x.to(DEV))
        ref = oracle.convmixer_forward(sd, x)
        assert (out.cpu() - ref).abs().max().item() < 1e-5, hw
    hm = pkg.HireMLP(patch_size=4, d_model=[16, 32], h=[4, 3], w=[4, 3], cross_region_step=[2, 1], cross_region_interval=2,
                     depth=[2, 2], expansion_factor=2, num_classes=10).eval()
    sd = {k: v.detach().clone() for k, v in hm.state_dict().items()}
    hm = hm.to(DEV)
    for hw in ((64, 64), (96, 64), (64, 64)):
        x = torch.randn(2, 3, *hw)
        with torch.no_grad():
            out = hm(x.to(DEV))
        ref = oracle.hiremlp_forward(sd, x, [4, 3], [4, 3], [2, 1], 2, 4)
        assert (out.cpu() - ref).abs().max().item() < 1e-5, hw


def test_forward_on_a_device_that_is_not_current():
    """The model and the input live on cuda:1 while cuda:0 is current: every launch must go to cuda:1's stream."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    pkg = load_pkg()
    torch.manual_seed(0)
    model = pkg.MLPMixerForImageClassification(d_model=64, depth=2, patch_size=8, image_size=64, num_classes=16).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(4, 3, 64, 64)
    ref = oracle.mixer_forward(sd, x)
    torch.cuda.set_device(0)
    model = model.to("cuda:1")
    with torch.no_grad():
        out = model(x.to("cuda:1"))
    torch.cuda.synchronize(1)
    assert torch.cuda.current_device() == 0 and out.device.index == 1
    assert (out.cpu() - ref).abs().max().item() < 1e-5
    sh = pkg.models_pytorch.Shift(3, 2)
    xs = torch.randn(2, 6, 5, 5)
    assert torch.equal(sh(xs.to("cuda:1")).cpu(), oracle.axial_shift_nchw(xs, 3, 2))


def test_train_mode_warns_once():
    """a module whose train mode is not built says so once (the mechanism: EngineModule._resolve); round 6: every model family implements train()
    -- forward and backward -- so none of them warns any more"""
    pkg = load_pkg()
    x = torch.randn(1, 3, 32, 32, device=DEV)
    model = pkg.models_pytorch.gMLPForImageClassification(image_size=32, patch_size=8, d_model=32, d_ffn=64, depth=1, num_classes=10).to(DEV)
    model.train()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            model(x)
    assert not any("inference-only" in str(m.message) for m in w)
    model.__dict__["_train_forward"] = False                # what a family without a train path looks like to the engine
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            model(x)
            model(x)
    assert sum("inference-only" in str(m.message) for m in w) == 1
    mixer = pkg.MLPMixerForImageClassification(d_model=32, depth=1, patch_size=8, image_size=32, num_classes=10).to(DEV).train()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            mixer(x)
    assert not any("inference-only" in str(m.message) for m in w)
