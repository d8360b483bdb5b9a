import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo")
from conftest import load_pkg
from oracle.portable_init import portable_input
from test_oracle_golden import run_oracle
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "gmlp"
torch.manual_seed(0)
if name == "gmlp":
    model = pkg.models_pytorch.gMLPForImageClassification(image_size=224).eval(); fam = "gmlp"; kw = dict(image_size=224)
else:
    kw = dict(image_size=224, patch_size=7, d_model=384, depth=18, segments=12, expansion_factor=3)
    model = pkg.models_pytorch.ViP(**kw).eval(); fam = "vip"
sd = {k: v.detach().float().clone() for k, v in model.state_dict().items()}
model = model.to("cuda:0")
x = torch.from_numpy(portable_input((256, 3, 224, 224), seed=3)).to("cuda:0").to(torch.bfloat16)
with torch.no_grad():
    big = model(x); small = model(x[100:102].contiguous()); b8 = model(x[96:104].contiguous()); b64 = model(x[64:128].contiguous())
ref = run_oracle(fam, sd, x[100:102].float().cpu(), kw)
f = lambda a: (a.float().cpu() - ref).abs().max().item()
print(name, "Q4=%s" % os.environ.get("MLPK_GEMM_Q4"), "vs oracle: bs256 %.3e  bs2 %.3e  bs8 %.3e  bs64 %.3e | max|ref| %.3f" % (f(big[100:102]), f(small), f(b8[4:6]), f(b64[36:38]), ref.abs().max().item()))
