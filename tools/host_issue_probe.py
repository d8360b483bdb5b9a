"""Host time to ISSUE one forward (queue kept short, no sync inside) beside the GPU time per step: which models are bound by the Python / ctypes
side once two steps are in flight.  python tools/host_issue_probe.py [model ...]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("jittor-mlp_amd")
names = sys.argv[1:] or [n for n in bench.MODELS]
for name in names:
    ctor, kw, _ = bench.MODELS[name]
    torch.manual_seed(0)
    model = getattr(pkg.models_pytorch, ctor)(**kw).eval().cuda()
    x = torch.rand(256, 3, 224, 224, device="cuda").bfloat16()
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        issue = []
        for _ in range(5):
            t0 = time.perf_counter()
            model(x)
            issue.append(time.perf_counter() - t0)      # the queue is empty at the start: nothing blocks the host
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model(x)
        torch.cuda.synchronize()
        gpu = (time.perf_counter() - t0) / 10
    print("%-20s host issue %.2f ms   step (one at a time) %.2f ms   host / step %.0f %%" % (name, min(issue) * 1e3, gpu * 1e3, 100 * min(issue) / gpu), flush=True)
    del model
    torch.cuda.empty_cache()
