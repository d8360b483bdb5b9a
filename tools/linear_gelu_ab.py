"""mlpk_linear_gelu (rows resident, no epilogue) against the GEMM tiles with a GELU epilogue, on the short-K shapes of the models.
usage: python tools/linear_gelu_ab.py  (on a GPU box)"""
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
for (name, M, K, Nn) in [("gmlp proj1", 50176, 256, 1536), ("vip fc1", 262144, 384, 1152), ("resmlp fc1", 50176, 384, 1536), ("s2 fc1", 50176, 384, 1536),
                         ("asmlp l2 fc1", 50176, 384, 1536), ("swin l1 fc1", 200704, 192, 768), ("k512", 50176, 512, 2048)]:
    g = torch.Generator().manual_seed(1)
    w = torch.randn((Nn, K), generator=g) / math.sqrt(K); b = torch.randn((Nn,), generator=g) * 0.1
    gamma = torch.ones(K); beta = torch.zeros(K)
    x = torch.randn((M, K), generator=g).to(dt).to(dev)
    mean = torch.zeros((M,), dtype=torch.float32, device=dev); rstd = torch.ones((M,), dtype=torch.float32, device=dev)
    pack = E.pack_linear_gelu(w, b, dt, dev, gamma, beta)
    wq, bq, csum = E.pack_ln_folded(w, b, gamma, beta, dt, dev)
    out = torch.empty((M, Nn), dtype=dt, device=dev)
    ws = E.Workspace(dev, dt)
    t_rr = timeit(lambda: E.linear_gelu(x, M, K, pack, out, ln=(mean, rstd)))
    t_rrs = timeit(lambda: E.linear_gelu(x, M, K, pack, out, ln=(mean, rstd), part=(ws, "a")))
    t_g = timeit(lambda: E.gemm(x, wq, out, M, Nn, K, bias=bq, act=N.ACT_GELU, ln=(mean, rstd, csum)))
    t_gs = timeit(lambda: E.gemm(x, wq, out, M, Nn, K, bias=bq, act=N.ACT_GELU, ln=(mean, rstd, csum), part=(ws, "b")))
    fl = 2.0 * M * K * Nn
    print("%-14s M=%7d K=%3d N=%4d   rows-resident %7.1f us (%6.1f TFLOP/s)  + statistics %7.1f us |  GEMM tile %7.1f us  + statistics %7.1f us" % (name, M, K, Nn, t_rr, fl / t_rr / 1e6, t_rrs, t_g, t_gs))
