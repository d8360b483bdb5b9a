#!/usr/bin/env python3
"""Golden logits for the six model configurations of the reference's own harness (compare.py:35-138),
on a uniform[0,1) input like compare.py:23 (bs reduced from 32 to 4 to keep the CPU run short), with
portable weights.  Authoring container only (imports /root/reference through make_golden's shim).

Besides the logits the fixture records how far the REFERENCE's own logits move, in compare.py's metric,
under two perturbations that are below what any other implementation can avoid:
  <name>__sens32  input multiplied by (1 +- 2^-23) elementwise (one fp32 ulp),
  <name>__sens16  input and every parameter rounded to fp16 (arithmetic still fp32).
tests/test_compare_protocol.py uses them as the conditioning floor of each configuration."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from oracle.portable_init import portable_tensor  # noqa: E402

CONFIGS = {  # name -> (module, class, kwargs exactly as compare.py passes them)
    "MLPMixer": ("mlp_mixer", "MLPMixerForImageClassification", dict(image_size=(224, 224), patch_size=16, in_channels=3, num_classes=1000, d_model=256, depth=12)),
    "gMLP": ("g_mlp", "gMLPForImageClassification", dict(image_size=(224, 224), patch_size=16, in_channels=3, num_classes=1000, d_model=256, d_ffn=1536, depth=30)),
    "ResMLP": ("res_mlp", "ResMLPForImageClassification", dict(in_channels=3, image_size=(224, 224), patch_size=16, d_model=384, depth=12, num_classes=1000, expansion_factor=4)),
    "ViP": ("vip", "ViP", dict(image_size=(224, 224), patch_size=(16, 8), in_channels=3, num_classes=1000, d_model=256, depth=30, segments=16, weighted=True)),
    "ConvMixer": ("conv_mixer", "ConvMixer", dict(dim=1568, depth=20)),
    "S2MLPv2": ("s2_mlp_v2", "S2MLPv2", dict(in_channels=3, image_size=(224, 224), patch_size=[(7, 7), (2, 2)], d_model=[192, 384], depth=[4, 14], num_classes=1000, expansion_factor=[3, 3])),
}
BS = 4


def metric(x, y):
    """compare.py:181-184."""
    return float(np.mean(np.abs((x + 1) - (y + 1)) / np.abs(y + 1)))


def main():
    ref = mg.load_reference()
    out = {}
    for name, (mod, cls, kw) in CONFIGS.items():
        torch.manual_seed(0)
        model = getattr(ref[mod], cls)(**kw).eval()
        mg.load_portable(model, seed=1)
        x = torch.from_numpy(portable_tensor("compare.input", (BS, 3, 224, 224), 0.0, 1.0, seed=1))
        y = mg.run_ref(model, x, one_thread=(name == "S2MLPv2"))
        out[name] = y.numpy()
        sign = torch.from_numpy(portable_tensor("compare.sign", tuple(x.shape), -1.0, 1.0, seed=1)).sign()
        y32 = mg.run_ref(model, x * (1.0 + sign * 2.0 ** -23), one_thread=(name == "S2MLPv2"))
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                if t.is_floating_point():
                    t.copy_(t.half().float())
        y16 = mg.run_ref(model, x.half().float(), one_thread=(name == "S2MLPv2"))
        out[name + "__sens32"] = np.float64(metric(y32.numpy(), y.numpy()))
        out[name + "__sens16"] = np.float64(metric(y16.numpy(), y.numpy()))
        print("  compare/%-10s logits max|.| %.3f  sens32 %.3g  sens16 %.3g" % (name, float(y.abs().max()), out[name + "__sens32"], out[name + "__sens16"]), flush=True)
        del model
    np.savez_compressed(os.path.join(HERE, "compare_logits.npz"), **out)


if __name__ == "__main__":
    main()
