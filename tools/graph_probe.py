#!/usr/bin/env python3
"""Tuning aid: Mixer-B/16 bs=256 bf16 forward, eager launches vs a captured HIP graph replay (torch.cuda.CUDAGraph)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "mixer"
if name == "mixer":
    model = pkg.MLPMixerForImageClassification(d_model=768, depth=12, patch_size=16, image_size=224, num_classes=1000)
elif name == "gmlp":
    model = pkg.gMLPForImageClassification(image_size=224, patch_size=16, d_model=256, d_ffn=1536, depth=30)
else:                                     # any bench.py model name
    bench = importlib.import_module("bench")
    ctor, kw, _ = bench.MODELS[name]
    model = getattr(pkg.models_pytorch, ctor)(**kw)
model = model.eval().cuda()
x = torch.rand(256, 3, 224, 224, device="cuda").bfloat16()
with torch.no_grad():
    for _ in range(5):
        y = model(x)
    torch.cuda.synchronize()
    def timeit(fn, n=30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("eager  %.3f ms" % timeit(lambda: model(x)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        yg = model(x)
    print("graph  %.3f ms" % timeit(lambda: g.replay()))
    print("max |eager - graph| = %.3e" % (y.float() - yg.float()).abs().max().item())
