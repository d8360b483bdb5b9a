import importlib, os, sys, torch, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
pkg = importlib.import_module("jittor-mlp_amd")
from oracle import portable_init
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    model = pkg.models_pytorch.AS_MLP(img_size=56, patch_size=4, embed_dim=384, depths=[3], num_classes=10).cuda().eval()
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if "norm" in n and n.endswith("weight"): p_.copy_(1 + 0.2 * torch.randn_like(p_))
            elif "norm" in n and n.endswith("bias"): p_.copy_(0.1 * torch.randn_like(p_))
            elif n.endswith("bias"): p_.copy_(0.05 * torch.randn_like(p_))
    x = torch.randn(5, 3, 56, 56, device="cuda").to(dt)
    outs = {}
    for flag in ("0", "1"):
        os.environ["MLPK_AS_BLOCK"] = flag
        with torch.no_grad():
            outs[flag] = model(x).float()
        torch.cuda.synchronize()
    ref32 = None
    os.environ["MLPK_AS_BLOCK"] = "0"
    with torch.no_grad():
        ref32 = model(x.float()).float()
    d = (outs["0"] - outs["1"]).abs().max().item()
    print(dt, "max |separate - fused| %.3e  |separate - fp32| %.3e  |fused - fp32| %.3e  scale %.3f" % (
        d, (outs["0"] - ref32).abs().max().item(), (outs["1"] - ref32).abs().max().item(), ref32.abs().max().item()))
    # batch invariance: image 3 alone
    os.environ["MLPK_AS_BLOCK"] = "1"
    with torch.no_grad():
        one = model(x[3:4]).float()
    print("   batch invariance (fused):", torch.equal(one[0], outs["1"][3]))
