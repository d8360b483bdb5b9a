import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_pkg
pkg = load_pkg(); E = pkg.engine
def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)
for dtype in (torch.float16, torch.bfloat16):
  for (B, H, W, C, shift, ks) in ((2, 56, 56, 96, [-2, -1, 0, 1, 2], [1, 1, 3, 5, 7]),):
    kmax = max(ks)
    x = rnd((B, H, W, C), dtype, 1500).cuda()
    w_lr, w_td = rnd((kmax * kmax, C), torch.float32, 1510, 0.5).cuda(), rnd((kmax * kmax, C), torch.float32, 1520, 0.5).cuda()
    b_lr, b_td = rnd((C,), torch.float32, 1530).cuda(), rnd((C,), torch.float32, 1540).cuda()
    outs = []
    for mode in ("1", "0", "1", "0"):
        os.environ["MLPK_MIXSHIFT_TILE"] = mode
        out = torch.full((B, H, W, C), float("nan"), dtype=dtype, device="cuda")
        E.mixshift_nhwc(x, out, B, H, W, C, shift, ks, w_lr, b_lr, w_td, b_td)
        torch.cuda.synchronize()
        outs.append(out)
    print(dtype, "tile==tile", torch.equal(outs[0], outs[2]), "band==band", torch.equal(outs[1], outs[3]), "tile==band", torch.equal(outs[0], outs[1]))
    d = (outs[0].float() - outs[1].float()).abs()
    idx = d.nonzero()
    print("n diff", idx.shape[0])
    if idx.shape[0]:
        print("channels", sorted(set(idx[:, 3].tolist()))[:40])
        print("rows", sorted(set(idx[:, 1].tolist()))[:60]); print("cols", sorted(set(idx[:, 2].tolist()))[:60])
        i = idx[0].tolist(); print(i, outs[0][tuple(i)].item(), outs[1][tuple(i)].item())
