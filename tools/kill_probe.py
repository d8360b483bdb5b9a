"""Upper bounds for the two remaining fc1 / launch items of the Mixer-B/16 step (VERDICT r5 item 1 b, c), measured without building them:
  (c) the 23 mlpk_stats_finalize_planar launches per forward: the same forward with those launches SKIPPED (stale statistics: the
      logits are wrong, the timing of everything else is what it was) -- what folding them into their consumers could give at most;
  (b) the draining block of the generated fc1 tile (dummy MFMAs): fc1 at M and at 2 M rows -- the per-workgroup constant
      2 T(M) - T(2 M) is everything a launch pays once per workgroup (ramp, first loads, the drain), an upper bound for the drain.
usage: python tools/kill_probe.py   (on a GPU box)"""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = torch.device("cuda:0")
dt = torch.bfloat16


def step_ms(model, x, steps=60, warm=8):
    with torch.no_grad():
        for _ in range(warm): model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): model(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


torch.manual_seed(0)
model = pkg.models_pytorch.MLPMixerForImageClassification(d_model=768, depth=12, patch_size=16, image_size=224).eval().to(dev)
x = torch.rand((256, 3, 224, 224)).to(dev).to(dt)
real = E.stats_finalize_planar
calls = [0]
def counted(*a, **k):
    calls[0] += 1
    return real(*a, **k)
E.stats_finalize_planar = counted
with torch.no_grad(): model(x); calls[0] = 0; model(x)
print("finalize launches per forward:", calls[0])
for rnd in range(3):
    E.stats_finalize_planar = real
    a = step_ms(model, x)
    E.stats_finalize_planar = lambda *a_, **k_: None
    b = step_ms(model, x)
    print("round %d: step %.4f ms   without the finalize launches %.4f ms   (%.2f %%)" % (rnd, a, b, (a - b) / a * 100))
E.stats_finalize_planar = real


def t_gemm(M, Nn, K, algo, n=30, **kw):
    A = (torch.rand((M, K), device=dev) * 2 - 1).to(dt)
    B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    bias = torch.rand(Nn, device=dev)
    mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev); csum = torch.zeros(Nn, device=dev)
    def run(): E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU, ln=(mean, rstd, csum), algo=algo)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rnd in range(2):
    t1 = t_gemm(50176, 3072, 768, 15)
    t2 = t_gemm(100352, 3072, 768, 15)
    t3 = t_gemm(150528, 3072, 768, 15)
    print("fc1 (generated tile, LN fold + GELU): M=50176 %.1f us, 2M %.1f us, 3M %.1f us -> per-M marginal %.1f us, per-launch constant %.1f us (%.1f %% of the call)"
          % (t1, t2, t3, (t3 - t1) / 2, t1 - (t3 - t1) / 2, (t1 - (t3 - t1) / 2) / t1 * 100))
