"""Fused channel MLP (mlpk_channel_mlp) against the two GEMMs it replaces, on the narrow stages of the hierarchical families.
usage: python tools/chanmlp_ab.py  (on a GPU box)"""
import importlib, math, os, sys, torch
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
for (name, M, C, hid, group) in [("asmlp l0", 802816, 96, 384, 3136), ("asmlp l1", 200704, 192, 768, 784), ("swin l0", 802816, 96, 384, 1), ("hire/cycle l0", 802816, 64, 256, 1),
                                 ("cycle l1", 200704, 128, 512, 1), ("l2 160", 50176, 160, 640, 1)]:
    g = torch.Generator().manual_seed(1)
    w1 = torch.randn((hid, C), generator=g) / math.sqrt(C); b1 = torch.randn((hid,), generator=g) * 0.1
    w2 = torch.randn((C, hid), generator=g) / math.sqrt(hid); b2 = torch.randn((C,), generator=g) * 0.1
    gamma = torch.ones(C); beta = torch.zeros(C)
    x = torch.randn((M, C), generator=g).to(dt).to(dev)
    ns = (M + group - 1) // group
    mean = torch.zeros((ns,), dtype=torch.float32, device=dev); rstd = torch.ones((ns,), dtype=torch.float32, device=dev)
    pack = E.pack_channel_mlp_fused(w1, b1, w2, b2, dt, dev, gamma, beta)
    out = torch.empty_like(x)
    wq, bq, csum = E.pack_ln_folded(w1, b1, gamma, beta, dt, dev)
    w2q = E.pack_matrix(w2, dt, dev); b2d = b2.to(dev)
    hb = torch.empty((M, hid), dtype=dt, device=dev)
    def fused(): E.channel_mlp_fused(x, M, C, pack, out, R=x, ln=(mean, rstd), ln_group=group)
    def two():
        E.gemm(x, wq, hb, M, hid, C, bias=bq, act=N.ACT_GELU, ln=(mean, rstd, csum), ln_group=group)
        E.gemm(hb, w2q, out, M, C, hid, bias=b2d, R=x, res=N.RES_ADD)
    tf, tt = timeit(fused), timeit(two)
    fl = 4.0 * M * C * hid
    print("%-14s M=%7d C=%3d hid=%4d   fused %7.1f us (%6.1f TFLOP/s, %5.2f TB/s of x+out)   two GEMMs %7.1f us   x%.2f" % (name, M, C, hid, tf, fl / tf / 1e6, 4.0 * M * C / tf / 1e6, tt, tt / tf))
