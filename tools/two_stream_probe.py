#!/usr/bin/env python3
"""Round-6 probe: does running the batch as TWO half-batches on two HIP streams fill the tails of the persistent kernels?
Every big kernel of the path is a persistent grid of one workgroup per CU with a static tile list, so a launch ends with a tail
in which part of the chip idles (fc2: 588 tiles on 256 CUs; fc1: 18 or 19 tiles per workgroup + a draining block) and launches
are separated by a boundary.  Images are independent in eval mode and every kernel is batch-invariant, so the two halves give the
same bits; the second stream's workgroups take the CUs the first stream's launch has already left.
usage: python tools/two_stream_probe.py [model] [batch] [steps]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("jittor-mlp_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "mixer_b16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctor, kw, _ = bench.MODELS[name]
torch.manual_seed(0)
dev = "cuda:0"
model = getattr(pkg.models_pytorch, ctor)(**kw).eval().to(dev)
x = torch.rand(B, 3, 224, 224, device=dev).to(torch.bfloat16)


def run_one():
    with torch.no_grad():
        return model(x)


def make_split(nparts):
    streams = [torch.cuda.Stream() for _ in range(nparts)]
    parts = [p.contiguous() for p in x.chunk(nparts)]
    outs = [None] * nparts

    def f():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        for i, (s, p) in enumerate(zip(streams, parts)):
            s.wait_event(ev)
            with torch.cuda.stream(s), torch.no_grad():
                outs[i] = model(p)
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)
    return f


def timeit(f, n):
    for _ in range(6):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.cuda.synchronize()
for _ in range(3):
    run_one()
torch.cuda.synchronize()
t0 = time.perf_counter()
run_one()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time to ISSUE one forward (queue empty, no sync): %.2f ms" % ((t1 - t0) * 1e3), flush=True)
ref = run_one().float()
variants = [("one stream, %d images" % B, run_one)] + [("%d streams x %d images" % (n, B // n), make_split(n)) for n in (2, 4)]
for nm, f in variants[1:]:
    o = f().float()
    torch.cuda.synchronize()
    print("%-28s max |delta| vs one stream: %.3g" % (nm, (o - ref).abs().max().item()), flush=True)
for rep in range(3):
    for nm, f in variants:
        ms = timeit(f, steps)
        print("%-28s %8.3f ms per %d images  %9.1f images/s" % (nm, ms, B, B / ms * 1e3), flush=True)

# ---- round 6, second question: WHOLE batches on alternating streams (step i on stream i % n): consecutive steps are independent, so the
# tail of one step's kernels can be filled by the next step's -- throughput of back-to-back steps, not latency of one
if os.environ.get("ALTERNATE", "1") == "1":
    def make_alternate(nstreams):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        state = {"i": 0}

        def f():
            s = streams[state["i"] % nstreams]
            state["i"] += 1
            with torch.cuda.stream(s), torch.no_grad():
                return model(x)
        return f, streams

    for n in (2, 3):
        f, streams = make_alternate(n)
        for _ in range(6):
            f()
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e3
            print("whole batches alternating over %d streams   %.3f ms per %d images   %.1f images/s" % (n, dt, B, B / dt * 1e3), flush=True)
