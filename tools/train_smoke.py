"""One training step (train() forward, loss.backward()) of every model family at its BENCHMARK configuration (bench.MODELS), bf16: every
parameter the forward touches gets a finite fp32 gradient; prints the step time.  A size check of the round-6 backward (index tables, grid
caps, LDS limits at real widths), not a tuned training benchmark.  python tools/train_smoke.py [batch] [model ...]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("jittor-mlp_amd")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
names = sys.argv[2:] or [n for n in bench.MODELS if n != "mixer_l16"]
for name in names:
    ctor, kw, _ = bench.MODELS[name]
    torch.manual_seed(0)
    model = getattr(pkg.models_pytorch, ctor)(**kw).cuda().train()
    x = torch.rand(bs, 3, 224, 224, device="cuda").bfloat16()
    ts = []
    for step in range(2):
        model.zero_grad()
        torch.cuda.synchronize()
        t0 = time.time()
        out = model(x)
        loss = (out.float() ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    n_par = n_grad = 0
    for k, p in model.named_parameters():
        n_par += 1
        if p.grad is not None:
            n_grad += 1
            assert p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), (name, k)
    assert out.requires_grad and n_grad >= n_par - 4, (name, n_par, n_grad)
    gn = sum(float(p.grad.norm()) ** 2 for p in model.parameters() if p.grad is not None) ** 0.5
    print("%-20s batch %d  loss %.4e  |grad| %.3e  %3d / %3d parameters with a gradient  step %.1f ms (first %.0f ms)  peak %.1f GB" % (
        name, bs, float(loss), gn, n_grad, n_par, ts[1] * 1e3, ts[0] * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    del model, out, loss
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
