#!/usr/bin/env python3
"""Headline benchmark: forward images/s of Mixer-B/16, 224^2, 256 images per GPU, bf16 MFMA path.

  python bench.py --gpus N --steps K --warmup W          (N=1: plain python;  N>1: launched by
  python -m torch.distributed.run --nproc-per-node N ... one rank per GPU, RCCL over xGMI)

A "step" is one forward pass of the hot path over one resident synthetic batch (BASELINE.json
configs[1]): patch embed -> 12 x (token-mixing MLP, channel MLP) -> LN/mean/head, plus -- for
N > 1 -- the single all-gather of the (256, 1000) logits that data-parallel inference needs
(SURVEY.md 8e).  Inputs are already in HBM when the timed region starts.  Protocol of the
reference's compare.py:149-158: warm-up, device sync, K timed forwards, device sync.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- the dominant kernel (channel-MLP GEMM, 79 % of the flops): algorithmic
                  2*M*N*K per launch / HIP-event time of those launches inside the timed region,
                  against the gfx950 dense bf16 MFMA peak (2.5 PFLOP/s);
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference forward) timed on this host,
                  best of a thread sweep, on a bounded sample (the benched model, fp32, bs=8) plus
                  BASELINE configs[0] (Mixer-S/16, bs=8, fp32) under "config1"; N=1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = {
    # name: (ctor name, kwargs, GFLOP per image)
    "mixer_b16": ("MLPMixerForImageClassification", dict(d_model=768, depth=12, patch_size=16, image_size=224), 28.094),
    "mixer_s16": ("MLPMixerForImageClassification", dict(d_model=512, depth=8, patch_size=16, image_size=224), 9.249),
    "mixer_l16": ("MLPMixerForImageClassification", dict(d_model=1024, depth=24, patch_size=16, image_size=224), 94.336),
    "gmlp_s": ("gMLPForImageClassification", dict(image_size=224), 17.491),
    "resmlp_24": ("ResMLPForImageClassification", dict(depth=24), 11.923),
    "vip_s7": ("ViP", dict(image_size=224, patch_size=7, d_model=384, depth=18, segments=12, expansion_factor=3), 54.496),
    "s2mlpv2": ("S2MLPv2", dict(), 13.817),
    "asmlp_t": ("AS_MLP", dict(), 8.701),
    "convmixer_1536_20": ("ConvMixer", dict(dim=1536, depth=20), 102.198),
    "sparsemlp_t": ("SparseMLP", dict(), 16.231),      # SURVEY.md 8(f) rank 2; 2*MAC of its GEMMs/convs counted by hand
    "hiremlp_s": ("HireMLP", dict(), 9.742),           # SURVEY.md 8(f) rank 2; counted by hand (padded region rows included)
    "msmlp_t": ("MS_MLP", dict(), 5.990),              # SURVEY.md 8(f) rank 3; counted by hand
    "swinmlp_t": ("SwinMLP", dict(), 6.110),           # SURVEY.md 8(f) rank 3; counted by hand (useful flops of the per-head window mixes)
    "cyclemlp_b1": ("CycleMLP_B1", dict(), 4.2),       # SURVEY.md 8(f) rank 3; 2 x the 2.1 GMACs the CycleMLP paper quotes for B1
}
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16/f16 MFMA (MI355X_MICROARCH.md); f32 MFMA 157.3
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


FAMILY = {"MLPMixerForImageClassification": "mixer", "gMLPForImageClassification": "gmlp",
          "ResMLPForImageClassification": "resmlp", "ViP": "vip", "S2MLPv2": "s2mlpv2", "AS_MLP": "asmlp",
          "ConvMixer": "convmixer", "SparseMLP": "sparsemlp", "HireMLP": "hiremlp", "MS_MLP": "msmlp", "SwinMLP": "swinmlp",
          "CycleMLP": "cyclemlp", "CycleMLP_B1": "cyclemlp"}


def _time_oracle(pkg, ctor_name, kwargs, bs, budget_s):
    """Thread sweep of the CPU oracle (fp32) on one resident batch: returns (best images/s, threads, description).
    Oversubscribing a 2-socket EPYC makes MKL/OpenMP collapse (0.14 img/s at 256 threads), too few threads leave
    cores idle: time a few counts and report the best, as the reference's user would pick."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import run_oracle
    ncpu = os.cpu_count() or 2
    sweep = sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} | {max(1, min(64, ncpu // 2))})
    torch.manual_seed(0)
    model = getattr(pkg.models_pytorch, ctor_name)(**kwargs).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.rand(bs, 3, 224, 224)
    fam = FAMILY[ctor_name]
    best, tried = (0.0, 0), []
    per = budget_s / len(sweep)
    for th in sweep:
        torch.set_num_threads(th)
        run_oracle(fam, sd, x, kwargs)                                  # warm-up at this thread count
        t0 = time.perf_counter()
        n = 0
        while True:
            run_oracle(fam, sd, x, kwargs)
            n += 1
            if time.perf_counter() - t0 > per or n >= 8:
                break
        rate = bs * n / (time.perf_counter() - t0)
        tried.append("%d:%.1f" % (th, rate))
        if rate > best[0]:
            best = (rate, th)
    return best[0], best[1], "threads:images/s " + " ".join(tried)


def run_cpu_baseline(model_name, kwargs, ctor_name, pkg):
    """The oracle (a port of the reference forward, oracle/) timed on this host, rank 0, N = 1 only:
      * the benched workload at bs = 8 (bounded sample of the metric's configuration), and
      * BASELINE.json configs[0]: Mixer-S/16, 224^2, bs = 8, fp32 -- the reference's own CPU-runnable case
        (BASELINE.md section 4 measured the reference itself at 24.5 images/s on 8 cores for it)."""
    import oracle  # noqa: F401  (the CPU baseline IS the oracle package: fail here, loudly, if it is missing)
    rate, th, desc = _time_oracle(pkg, ctor_name, kwargs, 8, 12.0)
    out = {"value": round(rate, 2), "unit": "images/s", "cores": th, "kind": "port",
           "sample": "%s fp32 bs=8, oracle/ restatement, best of a thread sweep (%s), host has %d logical CPUs"
                     % (model_name, desc, os.cpu_count() or 0),
           # the hosts of the pool differ and are shared: the same command gave 21 .. 37 images/s box to box in rounds 4-5
           "spread_note": "host-dependent: 21-37 images/s were seen for this sample on different boxes of the pool (shared 256-thread hosts)"}
    c1 = MODELS["mixer_s16"]
    r1, t1, d1 = _time_oracle(pkg, c1[0], c1[1], 8, 8.0)
    out["config1"] = {"workload": "BASELINE configs[0]: Mixer-S/16, 224^2, bs=8, fp32 on CPU", "value": round(r1, 2),
                      "unit": "images/s", "cores": t1, "kind": "port", "sample": d1}
    return out


GEMM_SOURCES = ("jittor-mlp_amd/csrc/mlpk_gemm.hip", "jittor-mlp_amd/csrc/mlpk_gemm_q4.hip", "jittor-mlp_amd/csrc/gen/q4gen.py",
                "jittor-mlp_amd/csrc/gen/isa.py", "jittor-mlp_amd/csrc/mlpk_common.h")


def gemm_source_digest():
    """sha256 over the sources of the channel-MLP GEMM kernels (hand-written tiles + the q4 generator)"""
    import hashlib
    h = hashlib.sha256()
    for rel in GEMM_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


class PowerSampler:
    """Shader clock and socket power of the GPU under test, sampled from sysfs (hwmon freq1_input / power1_average|input) every 20 ms
    while the timed loop runs -- the evidence behind "the channel-MLP GEMMs run against the 1400 W cap" (DESIGN.md section 3.1).  A box
    exposes one hwmon node per GPU of the host, not only the one this process may use: every node is sampled and the one that drew
    the most power during the loop is reported (null when sysfs is not readable)."""

    def __init__(self):
        import glob
        import threading
        self.nodes = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = os.path.join(hw, "freq1_input")
            pw = [os.path.join(hw, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, n))]
            if os.path.exists(f) and pw:
                self.nodes.append((hw, f, pw[0]))
        self.samples = {hw: ([], []) for hw, _, _ in self.nodes}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            for hw, f, pw in self.nodes:
                try:
                    with open(f) as fh:
                        c = int(fh.read()) / 1e6
                    with open(pw) as fh:
                        w = int(fh.read()) / 1e6
                    self.samples[hw][0].append(c)
                    self.samples[hw][1].append(w)
                except (OSError, ValueError):
                    pass
            self._stop.wait(0.02)

    def start(self):
        if self.nodes:
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=1.0)
        best = None
        for hw, (clk, pw) in self.samples.items():
            if len(pw) >= 3:
                # the first third of the loop is the ramp from idle clocks
                c, w = clk[len(clk) // 3:], pw[len(pw) // 3:]
                mean_w = sum(w) / len(w)
                if best is None or mean_w > best["power_w"]:
                    best = {"sclk_mhz": round(sum(c) / len(c), 1), "sclk_mhz_min": round(min(c), 1), "power_w": round(mean_w, 1),
                            "power_w_max": round(max(w), 1), "samples": len(w), "hwmon_nodes_seen": len(self.nodes)}
        return best


def measured_traffic(args):
    """HBM-side bytes per launch of the dominant kernel from the PMC passes of tools/pmc_bench.sh on this same
    command.  The JSON is stamped with the sha256 of the GEMM source it was measured on: a stale file (kernel
    changed since) yields null instead of a silently wrong number."""
    import glob
    import hashlib
    if not (args.model == "mixer_b16" and args.batch == 256 and args.dtype == "bf16"):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    if t.get("gemm_source_sha256") != gemm_source_digest():
        return None, "stale: %s was measured on other GEMM sources" % os.path.basename(files[-1])
    return t.get("channel_mlp_gemm_bytes_per_launch"), "%s (git %s)" % (os.path.basename(files[-1]), t.get("git", "?"))


def cpu_stub(args, world, rank):
    """The N-rank protocol of this file without a GPU: gloo, every rank "computes" deterministic logits for its shard, the
    single all-gather, barrier-bracketed timing with the max over ranks, one JSON line from rank 0."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    parallel = importlib.import_module("jittor-mlp_amd.parallel")

    class Stub(torch.nn.Module):
        def forward(self, x):
            return x.flatten(1)[:, :1000].float() + float(rank)
    runner = parallel.DataParallelForward(Stub(), world)
    x = torch.zeros((args.batch, 3, 24, 24))
    for _ in range(args.warmup):
        out = runner(x)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = runner(x)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert out.shape == (args.batch * world, 1000)
    # shard r of the gathered logits carries + r: the gather kept the rank order
    assert all(abs(float(out[r * args.batch, 0] - out[0, 0]) - r) < 1e-6 for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "images/sec fwd (cpu stub)", "value": round(args.batch * world * args.steps / elapsed, 1), "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
                          "config": {"workload": "launcher self-test, no GPU", "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                                     "collective": "all_gather(logits) over gloo" if world > 1 else "none"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="mixer_b16", choices=sorted(MODELS))
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=sorted(DT))
    ap.add_argument("--streams", type=int, default=None,
                    help="(default 2; 1 with --share-device, where two processes already time-slice the one GPU and their all_gather runs over gloo "
                         "through the host: 262 ms per step with two in flight against 15.9 ms one at a time) HIP streams the consecutive steps alternate over (round 6): every step is a whole forward of the whole batch; with 2, "
                         "step i + 1 is enqueued on the other stream while step i runs, so the partly empty last rounds of one step's persistent "
                         "kernels are filled by the next step's (two batches in flight, like a server with two request slots).  1 = strictly one step after the other")
    ap.add_argument("--gemm-plan", default="auto", choices=["auto", "mixed", "whole"],
                    help="tile heights of the persistent GEMM: mixed = shortest single launch, whole = least total CU time (engine.set_gemm_plan); "
                         "auto = whole when several steps are in flight, mixed otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the fp32-input variant line (profiling runs count forwards)")
    ap.add_argument("--algo", default="", help="tag=algo[,tag=algo] GEMM tile overrides (tuning)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm; gloo only for "
                    "exercising the multi-process path on a single-GPU box together with --share-device)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--cpu-stub", action="store_true", help="testing only (tests/test_parallel_gloo.py): no GPU, gloo, a stub forward "
                    "-- exercises the launcher, the barrier / max-over-ranks timing and the logits gather")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` on its own: become the launcher (one rank per GPU, rendezvous on 127.0.0.1) -- the same
            # command line the driver would have started through torch.distributed.run
            # --standalone: torch.distributed.run picks and binds its own free rendezvous port (no bind-close-rebind race)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   os.path.abspath(__file__)] + sys.argv[1:]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            sys.stdout.flush()
            os.execv(sys.executable, cmd)
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.cpu_stub:
        return cpu_stub(args, world, rank)
    if args.streams is None:
        args.streams = 1 if args.share_device else 2
    if args.share_device:
        local_rank = 0
        if args.backend == "nccl" and world > 1:
            # RCCL refuses two ranks on one device ("Duplicate GPU detected"): the single-GPU exercise of the N > 1 path
            # (HIP forward per rank + the logits gather) runs over gloo; the JSON line says so in config.collective
            args.backend = "gloo"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import __graft_entry__ as ge
    if world > 1:
        # one rank compiles (a no-op when the in-tree library is current), the others wait: eight ranks racing on the same
        # object files would corrupt a cold build
        if rank == 0:
            ge.build()
        dist.barrier()
        if rank != 0:
            ge.build()
    else:
        ge.build()
    pkg = importlib.import_module("jittor-mlp_amd")
    E = pkg.engine
    for item in filter(None, args.algo.split(",")):
        tag, algo = item.split("=")
        E.GEMM_ALGO[tag] = int(algo)

    ctor_name, kwargs, gflop_img = MODELS[args.model]
    cd = DT[args.dtype]
    torch.manual_seed(0)                                             # same random-init weights on every rank (replicated)
    model = getattr(pkg.models_pytorch, ctor_name)(**kwargs).eval().to(dev)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.rand((args.batch, 3, 224, 224), generator=g).to(dev).to(cd)   # uniform[0,1) like compare.py:23; resident in HBM
    parallel = importlib.import_module("jittor-mlp_amd.parallel")
    runner = parallel.DataParallelForward(model, world)

    # Consecutive steps are independent forwards of the same resident batch: they alternate over `--streams` HIP streams (each stream has its
    # own workspace inside the model), so that step i + 1 starts filling the CUs that the tail of step i's kernels leaves idle.  Same kernels,
    # same K whole steps inside the timed region, same bits; `single_stream` below is the strictly serial figure from the same process.
    nstreams = max(1, args.streams)
    side = parallel.InFlight(runner, nstreams, device=dev, throughput_plan=args.gemm_plan != "mixed") if nstreams > 1 else None     # (jittor-mlp_amd/parallel.py)
    if args.gemm_plan == "whole":
        E.set_gemm_plan(True)

    def step(i):
        return runner(x) if side is None else side(x)[0]

    def kernel_timing_pass():
        """Per-kernel durations for `roofline`: HIP events around the channel-MLP GEMM launches (on the stream they are launched on) in a
        SEPARATE short pass of whole steps, one step at a time on one stream -- same process, same resident batch -- so that the headline
        loop carries no event records at all (round-4 review: 24 pairs per step sat inside it)"""
        if args.no_kernel_timing:
            return None, 0
        timer = E.KernelTimer()
        n = max(2, min(args.steps, 6))
        E.TIMER = timer
        for _ in range(n):
            runner(x)
        E.TIMER = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return timer, n

    with torch.no_grad():
        # (1) two steps in flight only: the strictly serial figure first (one step after the other on ONE stream, = --streams 1), and the
        # kernel-timing pass in THAT regime -- a kernel's own duration at the clock the serial steady state runs at (with two steps in
        # flight the chip sits 4 % lower at the power cap and a kernel shares the CUs with the other step: not a statement about the kernel)
        serial = None
        timer, timing_steps, timing_where = None, 0, ""
        if side is not None:
            plan_in_flight, side_in_flight = E.GEMM_PLAN_WHOLE, E.SIDE_STREAMS     # the serial passes run the serial regime's own GEMM plan and side streams
            if args.gemm_plan == "auto":
                E.set_gemm_plan(False)
                E.set_side_streams(True)
            ns = max(5, min(args.steps, 30))
            for _ in range(3):
                out1 = runner(x)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ts0 = time.perf_counter()
            for _ in range(ns):
                out1 = runner(x)
            torch.cuda.synchronize()
            tser = (time.perf_counter() - ts0) / ns
            serial = {"value": round(args.batch * world / tser, 1), "unit": "images/s", "ms_per_step": round(tser * 1e3, 4), "steps": ns,
                      "what": "the same steps one after the other on ONE stream (--streams 1), measured in front of the timed region"}
            timer, timing_steps = kernel_timing_pass()
            timing_where = "%d serial steps straight after the %d steps of `single_stream`, in front of the timed region" % (timing_steps, ns)
            E.set_gemm_plan(plan_in_flight)
            E.set_side_streams(side_in_flight)
        # (2) the contract: W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs
        for it in range(args.warmup):
            out = step(it)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        E.TIMER = None
        sampler = PowerSampler().start() if rank == 0 else None
        t0 = time.perf_counter()
        for it in range(args.steps):
            out = step(it)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        sensors = sampler.stop() if sampler is not None else None
        timer_after = None
        if side is None:
            timer, timing_steps = kernel_timing_pass()
            timing_where = "%d extra steps straight after the %d timed ones" % (timing_steps, args.steps)
        else:
            serial["bits_equal_to_timed_steps"] = bool(torch.equal(out1, out))
            del out1
            if args.gemm_plan == "auto":
                E.set_gemm_plan(False)
                E.set_side_streams(True)
            timer_after, _ = kernel_timing_pass()               # the same pass behind the two-in-flight region (hotter chip): reported beside
            E.set_gemm_plan(plan_in_flight)
            E.set_side_streams(side_in_flight)
        # The drop-in contract hands over fp32 images (the reference's models take float tensors) and runs the 16-bit path through
        # set_compute_dtype: the image is converted while the patches are gathered and the logits come back in fp32.  The headline above
        # keeps the batch resident in the compute dtype; this variant line times the contract itself on the same model (N = 1 only).
        fp32_variant = None
        if world == 1 and cd != torch.float32 and hasattr(model, "set_compute_dtype") and not args.no_variants:
            x32 = x.float()
            model.set_compute_dtype(cd)
            for _ in range(3):
                out32 = runner(x32)
            torch.cuda.synchronize()
            n32 = max(5, min(args.steps, 30))
            tv0 = time.perf_counter()
            for _ in range(n32):
                out32 = runner(x32)
            torch.cuda.synchronize()
            tv = (time.perf_counter() - tv0) / n32
            model.set_compute_dtype(None)
            assert out32.dtype == torch.float32 and bool(torch.isfinite(out32).all())
            fp32_variant = {"value": round(args.batch / tv, 1), "unit": "images/s", "ms_per_step": round(tv * 1e3, 4), "steps": n32,
                            "what": "fp32 images resident in HBM, model.set_compute_dtype(%s): conversion inside the patch gather, fp32 logits" % args.dtype}
            del x32, out32
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert out.shape == (args.batch * world, 1000) and bool(torch.isfinite(out.float()).all())

    if rank == 0:
        global_batch = args.batch * world
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": ("images/sec fwd, 224^2 bs=%d/GPU, Mixer-B/16" % args.batch) if args.model == "mixer_b16" else "images/sec fwd " + args.model,
            "value": round(global_batch * args.steps / elapsed, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s forward, 224x224, %d images/GPU x %d GPU, random-init weights, uniform[0,1) input resident in HBM"
                                   % (args.model, args.batch, world),
                       "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "streams": nstreams, "steps_in_flight": nstreams,
                       "gemm_plan": "whole 256-row tiles where they fill a round (least total CU time: engine.set_gemm_plan)" if E.GEMM_PLAN_WHOLE else "mixed tile heights (shortest single launch)",
                       "collective": ("all_gather(logits) over %s%s" % (args.backend, ", all ranks on cuda:0" if args.share_device else "")) if world > 1 else "none"},
            "model_tflops": round(gflop_img * global_batch * args.steps / elapsed / 1e3, 1),
            # sysfs sensors of the GPU sampled every 20 ms DURING the timed loop (null if the box does not expose them)
            "sensors_timed_loop": sensors,
        }
        if serial is not None:
            line["single_stream"] = serial
        if fp32_variant is not None:
            line["fp32_input_variant"] = fp32_variant
        if timer is not None and timer.events:
            summ = timer.summary()
            dom = [t for t in ("channel_fc1", "channel_fc2") if t in summ]
            if dom:
                flops = sum(summ[t]["flops_per_launch"] * summ[t]["launches"] for t in dom)
                secs = sum(summ[t]["avg_ms"] * summ[t]["launches"] for t in dom) * 1e-3
                n_launch = sum(summ[t]["launches"] for t in dom)
                peak = PEAK_BF16_TFLOPS if args.dtype != "fp32" else 157.3
                ach = flops / secs / 1e12
                traffic, traffic_src = measured_traffic(args)
                names = "; ".join("%s: %s" % (t, ", ".join(summ[t].get("kernels") or ["?"])) for t in dom)      # from the dispatch, not a constant
                line["roofline"] = {"bound": "mfma", "kernel": names, "achieved": round(ach, 1),
                                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                                    "flops_per_launch": flops / n_launch, "avg_launch_ms": round(secs / n_launch * 1e3, 4),
                                    "launches_timed": n_launch, "timed_in": timing_where + "; one step at a time on one stream: a kernel's own duration (no event records inside the headline loop)",
                                    "traffic_source": traffic_src,
                                    # "traffic" is NOT measured in this run: it is the PMC result of tools/pmc_bench.sh on this command,
                                    # read from profiles/ and dropped (null) when the GEMM sources changed since it was taken
                                    "traffic_measured_in_run": False,
                                    # a "launch" here is one mlpk_gemm_nt CALL (236.8 GFLOP at Mixer-B/16): the kernel names above are what
                                    # the library's dispatch answered for the timed calls (mlpk_gemm_kernel_name); the persistent tile
                                    # lists the tile heights of its plan ("rows 256+192": both in one launch of the pair kernel)
                                    "launch_means": "one mlpk_gemm_nt call; kernel names as answered by the dispatch for the timed calls"}
            line["kernels"] = {t: {"avg_ms": round(v["avg_ms"], 4), "tflops": round(v["flops_per_launch"] / v["avg_ms"] / 1e9, 1),
                                   "launches": v["launches"]} for t, v in summ.items()}
            if timer_after is not None and timer_after.events and "roofline" in line:
                sa = timer_after.summary()
                da = [t for t in ("channel_fc1", "channel_fc2") if t in sa]
                if da:
                    fa = sum(sa[t]["flops_per_launch"] * sa[t]["launches"] for t in da) / (sum(sa[t]["avg_ms"] * sa[t]["launches"] for t in da) * 1e-3) / 1e12
                    line["roofline"]["frac_behind_timed_region"] = round(fa / line["roofline"]["peak"], 4)
                    line["roofline"]["frac_behind_timed_region_note"] = ("the same serial pass repeated straight after the timed region: with two steps in flight "
                                                                         "the chip sits lower at its power cap and the pass inherits that state")
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = run_cpu_baseline(args.model, kwargs, ctor_name, pkg)
            except Exception as e:      # a baseline failure must not discard the measured GPU line
                line["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
