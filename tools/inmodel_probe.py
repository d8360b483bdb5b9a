"""gMLP-S at 256 images: every engine call of one block timed (HIP events) in the model, then the SAME calls (same tensors) replayed alone
in a loop, and alone behind a cache-flushing kernel -- which of the block's kernels lose time to what surrounds them.
usage: python tools/gmlp_inmodel_probe.py [model]  (on a GPU box)"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("jittor-mlp_amd")
bench = importlib.import_module("bench")
E = pkg.engine
name = sys.argv[1] if len(sys.argv) > 1 else "gmlp_s"
ctor, kw, _ = bench.MODELS[name]
torch.manual_seed(0)
model = getattr(pkg.models_pytorch, ctor)(**kw).eval().cuda()
x = torch.rand(256, 3, 224, 224, device="cuda").bfloat16()
NAMES = ["gemm", "linear_gelu", "token_gemm_ln", "token_gemm", "token_mlp", "token_mlp_ln", "stats_finalize_planar", "row_stats", "norm_apply", "channel_mlp_fused",
         "vip_branch", "vip_split_apply", "vip_unpermute", "split_sum", "split_softmax", "split_apply", "s2_shift", "dwconv_nhwc", "dwconv_affine_nhwc", "im2col",
         "patchify", "pool_mean", "as_conv2", "norm_shift_nhwc", "smlp_mix", "smlp_mix_dw", "swin_spatial", "hire_gather_ln", "hire_combine_from", "mixshift_nhwc",
         "cycle_shift_ln", "layernorm_transpose", "add_periodic", "convert", "stem7", "patch_embed4", "conv_gemm_nhwc", "merge2x2_row_stats", "hire_combine_stats",
         "im2col", "hire_gather_ln"]
NAMES = [n for n in NAMES if hasattr(E, n)]
orig = {n: getattr(E, n) for n in NAMES}
log = []          # (name, args, kwargs, ev0, ev1)
recording = [False]


def wrap(n):
    f = orig[n]
    def g(*a, **k):
        if not recording[0]:
            return f(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        log.append((n, a, k, e0, e1))
        return r
    return g


for n in NAMES:
    setattr(E, n, wrap(n))
for mod in list(sys.modules.values()):          # models bind E.<fn> at call time through the module attribute: nothing else to patch
    pass
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    recording[0] = True
    model(x)
    torch.cuda.synchronize()
    recording[0] = False
calls = [(n, a, k, e0.elapsed_time(e1) * 1e3) for n, a, k, e0, e1 in log]
print("%d engine calls recorded in one forward, %.2f ms between their events" % (len(calls), sum(c[3] for c in calls) / 1e3))


def key(c):
    n, a, k, _ = c
    shp = tuple(int(v) for v in a if isinstance(v, int))[:4]
    return (n, k.get("tag"), shp)


groups = {}
for c in calls:
    groups.setdefault(key(c), []).append(c)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def alone(c, mode, n=20):
    nme, a, k, _ = c
    f = orig[nme]
    ts = []
    for _ in range(n):
        if mode == "flush": flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(*a, **k); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print("%-22s %-14s %-28s %5s %9s %9s %9s" % ("call", "tag", "ints", "n", "in-model", "alone", "flushed"))
with torch.no_grad():
    for kq, cs in sorted(groups.items(), key=lambda kv: -sum(c[3] for c in kv[1])):
        tin = sorted(c[3] for c in cs)[len(cs) // 2]
        if tin * len(cs) < 40: continue
        mid = cs[len(cs) // 2]
        print("%-22s %-14s %-28s %5d %9.1f %9.1f %9.1f" % (kq[0], str(kq[1]), str(kq[2]), len(cs), tin, alone(mid, "loop"), alone(mid, "flush")), flush=True)
