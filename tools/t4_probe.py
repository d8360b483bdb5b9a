"""Time the fused token-mixing kernels on the Mixer-B/16 bs=256 shape: layout 1 (token_mlp_rr_kernel) vs layout 2 (generated t4)."""
import os
import sys

import torch

import importlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("jittor-mlp_amd")
E = pkg.engine
dt = torch.bfloat16
B_, C, S, T, sp = int(os.environ.get("BS", 256)), int(os.environ.get("C", 768)), 196, int(os.environ.get("T", 384)), 224
torch.manual_seed(0)
xt = torch.zeros(B_ * C, sp, dtype=dt, device="cuda")
xt[:, :S] = torch.randn(B_ * C, S, device="cuda").to(dt)
x = torch.randn(B_ * S, C, device="cuda").to(dt)
w1, b1, w2, b2 = torch.randn(T, S) / 14, torch.randn(T), torch.randn(S, T) / 20, torch.randn(S)
flops = 2.0 * B_ * C * S * T * 2
for lay in (1, 2, 3):
    for st in (False, True):
        for dbg in ([0] if lay != 2 else [int(v) for v in os.environ.get("DBGS", "0,1,2,4,3,8,16,32,48,52").split(",")]):
            os.environ["MLPK_T4_DBG"] = str(dbg)
            pk = E.pack_token_mlp(w1, b1, w2, b2, dt, "cuda", sp, layout=lay, t_rows=C)
            part = torch.empty(E.token_mlp_stat_planes(C, lay), B_ * S, 2, device="cuda") if st else None
            if dbg and not st:
                continue
            for _ in range(3):
                E.token_mlp(xt, sp, B_ * C, S, pk[0], pk[1], pk[2], pk[3], pk[4], x, C, C, stats=part, layout=lay)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            n = 20
            ev[0].record()
            for _ in range(n):
                E.token_mlp(xt, sp, B_ * C, S, pk[0], pk[1], pk[2], pk[3], pk[4], x, C, C, stats=part, layout=lay)
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / n
            extra = ""
            if lay >= 2:
                import ctypes
                prof = torch.zeros(1024, dtype=torch.int64, device="cuda")
                fn = ctypes.CDLL(pkg._native.LIB_PATH).mlpk_token_mlp_debug
                fn.argtypes = [ctypes.c_void_p]
                fn(prof.data_ptr())
                E.token_mlp(xt, sp, B_ * C, S, pk[0], pk[1], pk[2], pk[3], pk[4], x, C, C, stats=part, layout=lay)
                torch.cuda.synchronize()
                fn(None)
                rec = prof[:1024].view(torch.int32).view(256, 8).float()
                cyc, epi = rec[:, 0], rec[:, 1]
                extra = "  ticks/WG mean %.0f max %.0f (epilogues %.0f, fill %.0f, steady %.0f, drain %.0f) -> %.2f ticks/ns" % (
                    cyc.mean().item(), cyc.max().item(), epi.mean().item(), rec[:, 2].mean().item(), rec[:, 3].mean().item(), rec[:, 4].mean().item(),
                    cyc.max().item() / (ms * 1e6))
            print("layout %d stats %d dbg %d: %.1f us  %.0f TFLOP/s%s" % (lay, st, dbg, ms * 1e3, flops / ms / 1e9, extra), flush=True)
os.environ["MLPK_T4_DBG"] = "0"
