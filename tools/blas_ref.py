#!/usr/bin/env python3
"""Reference point only (not used by the product): what the vendor library reaches on the channel-MLP shapes."""
import torch
dt = torch.bfloat16
for name, M, N, K in [("fc1", 50176, 3072, 768), ("fc2", 50176, 768, 3072), ("tok_fc1", 196608, 784, 224), ("vip", 262144, 384, 384)]:
    A = (torch.rand((M, K), device="cuda") * 2 - 1).to(dt)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(dt)
    b = torch.rand(N, device="cuda").to(dt)
    for label, fn in [("linear", lambda: torch.nn.functional.linear(A, W, b)),
                      ("linear+gelu", lambda: torch.nn.functional.gelu(torch.nn.functional.linear(A, W, b)))]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-8s %-12s %.3f ms  %.1f TF" % (name, label, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
