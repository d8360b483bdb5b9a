"""The four 1 x 1 convolutions of an AS-MLP block (as_mlp.py:55-95) as the model calls them at 256 images, on every GEMM tile that takes
the call: which tile should own which (shape, epilogue).  usage: python tools/asmlp_conv_probe.py  (on a GPU box)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = torch.device("cuda:0")
dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ALGOS = [(0, "auto"), (11, "s3 256x128"), (12, "s3 128x128"), (13, "s3 128x256"), (14, "p8"), (15, "q4")]
for (M, C, HW) in [(802816, 96, 3136), (200704, 192, 784), (50176, 384, 196), (12544, 768, 49)]:
    A = (torch.rand((M, C), device=dev) * 2 - 1).to(dt)
    A2 = (torch.rand((M, C), device=dev) * 2 - 1).to(dt)
    B = ((torch.rand((C, C), device=dev) * 2 - 1) / C ** 0.5).to(dt)
    out = torch.zeros((M, C), dtype=dt, device=dev)
    bias = torch.rand(C, device=dev)
    mean = torch.rand(M // HW, device=dev) * 0.1; rstd = torch.rand(M // HW, device=dev) + 0.5
    csum = B.float().sum(1).contiguous()
    modes = {
        "conv1  (norm fold + bias)": dict(bias=bias, ln=(mean, rstd, csum), ln_group=HW),
        "conv2_1 (bias + gelu)": dict(bias=bias, act=N.ACT_GELU),
        "conv2_2 (bias + gelu + add)": dict(bias=bias, act=N.ACT_GELU, R=out, res=N.RES_ADD),
        "conv3  (norm fold + bias + residual in place)": dict(bias=bias, ln=(mean, rstd, csum), ln_group=HW, R=A2, res=N.RES_ADD, inplace=True),
    }
    for name, kw in modes.items():
        kw = dict(kw)
        dst = A2 if kw.pop("inplace", False) else out
        line = "M=%7d C=%3d %-48s" % (M, C, name)
        for algo, an in ALGOS:
            try:
                t = timeit(lambda: E.gemm(A, B, dst, M, C, C, algo=algo, **kw))
                nm = ""
                line += " | %s %6.1f" % (an, t)
            except Exception as e:
                line += " | %s    --" % an
        print(line, flush=True)
