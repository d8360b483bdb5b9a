"""The N>1 path on CPU: 2 processes, gloo backend.  Checks the sharding + all-gather contract:
the gathered logits of a sharded run equal a single run on the concatenated batch, row for row
(SURVEY.md section 4 tier 6).  The per-rank forward is the CPU oracle here (the HIP forward needs a
GPU); the collective code path (jittor-mlp_amd/parallel.py) is exactly the one bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parallel = importlib.import_module("jittor-mlp_amd.parallel")
        z = np.load(os.path.join(GOLDEN, "tiny_mixer.npz"))
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
        g = torch.Generator().manual_seed(7)
        x_global = torch.randn(4 * world, 3, 32, 32, generator=g)              # identical on every rank
        runner = parallel.DataParallelForward(lambda xs: oracle.mixer_forward(sd, xs), world)
        out = runner(parallel.shard_batch(x_global, rank, world))
        out2 = runner(parallel.shard_batch(x_global, rank, world))             # buffer re-use path
        single = oracle.mixer_forward(sd, x_global)
        ok = bool(torch.equal(out, out2)) and out.shape == single.shape and float((out - single).abs().max()) < 1e-6
        q.put((rank, ok, float((out - single).abs().max())))
    finally:
        dist.destroy_process_group()


def test_all_gather_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results), results


def test_shard_batch_contract():
    sys.path.insert(0, ROOT)
    import importlib
    parallel = importlib.import_module("jittor-mlp_amd.parallel")
    x = torch.arange(24).reshape(8, 3)
    parts = [parallel.shard_batch(x, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts), x) and all(p.shape[0] == 2 for p in parts)
    with pytest.raises(ValueError):
        parallel.shard_batch(x, 0, 3)
    one = parallel.DataParallelForward(lambda t: t * 2, world=1)
    assert torch.equal(one(x), x * 2)


def test_bench_launches_itself_for_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it must become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
    and print ONE JSON line with n_gpus = 2: driven here without a GPU through the stub forward (--cpu-stub: gloo, the same barrier /
    max-over-ranks timing and logits gather as the real path)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--cpu-stub"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    # and the one-process form still runs on its own
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--cpu-stub"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
