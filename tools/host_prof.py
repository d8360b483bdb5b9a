import cProfile, pstats, importlib, sys, os, torch, io
sys.path.insert(0, "/root/repo")
import bench
pkg = importlib.import_module("jittor-mlp_amd")
name = sys.argv[1]
ctor, kw, _ = bench.MODELS[name]
model = getattr(pkg.models_pytorch, ctor)(**kw).eval().cuda()
x = torch.rand(256, 3, 224, 224, device="cuda").bfloat16()
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5): model(x)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:5000])
