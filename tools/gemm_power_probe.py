#!/usr/bin/env python3
"""GPU tuning aid: is the channel-MLP GEMM limited by its clock (power) or by its cycle count?
Loops one GEMM configuration for ~1.5 s while a thread samples the GPU's shader clock and socket power from sysfs
(hwmon freq1_input / power1_average / power1_input; falls back to `rocm-smi` once per configuration), then prints
TFLOP/s next to the sampled clock and power.  Configurations: fc1 / fc2 on random and on zero operands, fc2 on a third of
the CUs (a partial round), a pure MFMA-free HBM copy for reference.  usage: python tools/gemm_power_probe.py"""
import glob
import importlib
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jittor-mlp_amd")
E, N = pkg.engine, pkg._native
dev = "cuda:0"
dt = torch.bfloat16


def find_sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "power1_average", "power1_input", "freq2_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        out.setdefault("pp_dpm_sclk", p)
    return out


SENS = find_sensors()
print("sensors:", SENS, flush=True)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.clk, self.pw = [], []

    def run(self):
        while not self.stop:
            try:
                if "freq1_input" in SENS:
                    self.clk.append(int(open(SENS["freq1_input"]).read()) / 1e6)
                elif "pp_dpm_sclk" in SENS:
                    for l in open(SENS["pp_dpm_sclk"]):
                        if "*" in l:
                            self.clk.append(float(l.split(":")[1].replace("Mhz", "").replace("*", "").strip()))
                for k in ("power1_average", "power1_input"):
                    if k in SENS:
                        self.pw.append(int(open(SENS[k]).read()) / 1e6)
                        break
            except Exception:
                pass
            time.sleep(0.02)


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20)
        return " | ".join(l.strip() for l in r.stdout.splitlines() if ("sclk" in l or "Power" in l))[:300]
    except Exception as ex:
        return "rocm-smi failed: %s" % ex


def loop(name, fn, flops, seconds=1.5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    extra = smi() if not s.clk else ""
    s.stop = True
    ms = e0.elapsed_time(e1) / n
    clk = s.clk[len(s.clk) // 3:]
    pw = s.pw[len(s.pw) // 3:]
    print("%-34s %8.4f ms %8.1f TFLOP/s | sclk MHz avg %7.1f min %7.1f max %7.1f (%d samples) | power W avg %6.1f | %s"
          % (name, ms, flops / ms / 1e9, sum(clk) / max(1, len(clk)), min(clk or [0]), max(clk or [0]), len(clk),
             sum(pw) / max(1, len(pw)), extra), flush=True)


def gemm_case(M, Nn, K, gelu, zero=False, dbg=0):
    if zero:
        A = torch.zeros((M, K), dtype=dt, device=dev)
        B = torch.zeros((Nn, K), dtype=dt, device=dev)
    else:
        A = (torch.rand((M, K), device=dev) * 2 - 1).to(dt)
        B = ((torch.rand((Nn, K), device=dev) * 2 - 1) / K ** 0.5).to(dt)
    bias = torch.rand(Nn, device=dev)
    C = torch.zeros((M, Nn), dtype=dt, device=dev)
    kw = dict(R=C, res=N.RES_ADD) if not gelu else {}
    return lambda: E.gemm(A, B, C, M, Nn, K, bias=bias, act=N.ACT_GELU if gelu else 0, algo=14, dbg=dbg, **kw)


print("idle:", smi(), flush=True)
loop("fc1 random", gemm_case(50176, 3072, 768, True), 2.0 * 50176 * 3072 * 768)
loop("fc1 zeros", gemm_case(50176, 3072, 768, True, zero=True), 2.0 * 50176 * 3072 * 768)
loop("fc1 random, no gelu", gemm_case(50176, 3072, 768, False), 2.0 * 50176 * 3072 * 768)
loop("fc2 random", gemm_case(50176, 768, 3072, False), 2.0 * 50176 * 768 * 3072)
loop("fc2 zeros", gemm_case(50176, 768, 3072, False, zero=True), 2.0 * 50176 * 768 * 3072)
loop("fc2 random 256-row tiles only", gemm_case(50176, 768, 3072, False, dbg=16), 2.0 * 50176 * 768 * 3072)
loop("fc2 random, 75 tiles (1/3 CUs)", gemm_case(6400, 768, 3072, False, dbg=16), 2.0 * 6400 * 768 * 3072)
loop("fc2 random, 255 tiles (1 round)", gemm_case(21760, 768, 3072, False, dbg=16), 2.0 * 21760 * 768 * 3072)
loop("square 8192 random", gemm_case(8192, 8192, 8192, False, dbg=16), 2.0 * 8192 ** 3)
loop("square 8192 zeros", gemm_case(8192, 8192, 8192, False, zero=True, dbg=16), 2.0 * 8192 ** 3)
src = torch.empty((256 << 20,), dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
loop("hbm copy 256 MiB", lambda: dst.copy_(src), 0.0)
