"""mlpk_channel_mlp alone on the narrow stages' shapes, as the models call it (in place, residual = operand, by-product statistics):
time per call, and the split between per-iteration and per-tile cost (the same rows with twice the hidden width).
usage: [MLPK_LIB_PATH=variant.so] python tools/chanmlp_probe.py  (on a GPU box)"""
import importlib, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = torch.device("cuda:0")
dt = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("lib:", os.environ.get("MLPK_LIB_PATH", "default"), flush=True)
for (name, M, C, hid, group) in [("asmlp l0", 802816, 96, 384, 3136), ("asmlp l0 2x hidden", 802816, 96, 768, 3136), ("asmlp l1", 200704, 192, 768, 784),
                                 ("hire/cycle l0", 802816, 64, 256, 1), ("cycle l1", 200704, 128, 512, 1), ("l2 160", 50176, 160, 640, 1)]:
    g = torch.Generator().manual_seed(1)
    w1 = torch.randn((hid, C), generator=g) / math.sqrt(C); b1 = torch.randn((hid,), generator=g) * 0.1
    w2 = torch.randn((C, hid), generator=g) / math.sqrt(hid) * 0.1; b2 = torch.randn((C,), generator=g) * 0.01
    gamma = torch.ones(C); beta = torch.zeros(C)
    x0 = torch.randn((M, C), generator=g).to(dt).to(dev)
    x = x0.clone()
    ns = (M + group - 1) // group
    mean = torch.zeros((ns,), dtype=torch.float32, device=dev); rstd = torch.ones((ns,), dtype=torch.float32, device=dev)
    pack = E.pack_channel_mlp_fused(w1, b1, w2, b2, dt, dev, gamma, beta)
    ws = E.Workspace(dev, dt)
    out = torch.empty_like(x)
    def inplace(): E.channel_mlp_fused(x, M, C, pack, x, R=x, ln=(mean, rstd), ln_group=group, part=(ws, "p"))
    def apart(): E.channel_mlp_fused(x0, M, C, pack, out, R=x0, ln=(mean, rstd), ln_group=group, part=(ws, "p"))
    def other_res(): E.channel_mlp_fused(x0, M, C, pack, out, R=out, ln=(mean, rstd), ln_group=group, part=(ws, "p"))
    if os.environ.get("CM_DEBUG"):
        for nm, fn in (("inplace", inplace), ("apart", apart), ("other_res", other_res)):
            print("  running", name, nm, flush=True)
            fn(); torch.cuda.synchronize()
    ti, ta, to = timeit(inplace), timeit(apart), timeit(other_res)
    fl = 4.0 * M * C * hid
    print("%-20s M=%7d C=%3d hid=%4d  in place %7.1f us (%6.1f TFLOP/s, %5.2f TB/s x+out)  out!=x %7.1f us  R!=x %7.1f us" %
          (name, M, C, hid, ti, fl / ti / 1e6, 4.0 * M * C / ti / 1e6, ta, to), flush=True)
