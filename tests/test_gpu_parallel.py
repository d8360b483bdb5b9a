"""-m gpu: the N > 1 path on hardware -- HIP forward + the logits all-gather in the same processes.

A gpurun box has ONE MI355X, so both ranks share cuda:0 (RCCL refuses two ranks on one device, hence gloo for the
two-rank case: the collective then stages through the host, but everything around it -- sharding, the HIP forward
on each rank's stream, the stream-ordered hand-over into torch.distributed, rank order of the gathered rows -- is
the code bench.py runs on the 8-GPU node).  The RCCL side is covered by a world-size-1 process group on the GPU:
ncclAllGather on the RCCL stream, ordered against the compute stream by the events parallel.py documents.
SURVEY.md 8e."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model_and_input(pkg, world, per_rank):
    torch.manual_seed(0)                                  # identical replicated weights on every rank
    # the benchmarked architecture itself (BASELINE configs[1], Mixer-B/16): fused token kernel with the LayerNorm loader, q4 / persistent
    # GEMM tiles with by-product statistics -- the sharded forward must equal the single forward on the concatenated batch BIT FOR BIT
    model = pkg.MLPMixerForImageClassification(d_model=768, depth=12, patch_size=16, image_size=224, num_classes=1000).eval().to("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.rand((world * per_rank, 3, 224, 224), generator=g).to("cuda:0").to(torch.bfloat16)
    return model, x


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("jittor-mlp_amd")
        parallel = importlib.import_module("jittor-mlp_amd.parallel")
        model, x = _model_and_input(pkg, world, 32)
        runner = parallel.DataParallelForward(model, world)
        with torch.no_grad():
            out = runner(parallel.shard_batch(x, rank, world))
            out_b = runner(parallel.shard_batch(x, rank, world))          # a second forward must not disturb the first result
            single = model(x)                                             # one rank on the concatenated batch
        torch.cuda.synchronize()
        ok = (out.shape == single.shape and bool(torch.equal(out, single)) and bool(torch.equal(out, out_b))
              and out.data_ptr() != out_b.data_ptr())
        q.put((rank, ok, float((out.float() - single.float()).abs().max())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_hip_forward_plus_all_gather_bit_equal():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results), results


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        pkg = importlib.import_module("jittor-mlp_amd")
        parallel = importlib.import_module("jittor-mlp_amd.parallel")
        model, x = _model_and_input(pkg, 1, 32)
        runner = parallel.DataParallelForward(model, 1, force_collective=True)
        with torch.no_grad():
            plain = model(x)
            outs = [runner(x) for _ in range(3)]           # back to back: gather i is ordered between forward i and i+1
        torch.cuda.synchronize()
        q.put(all(bool(torch.equal(o, plain)) for o in outs))
    finally:
        dist.destroy_process_group()


def test_rccl_all_gather_is_ordered_on_the_compute_stream():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    ok = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and ok


def test_two_forwards_in_flight_give_the_serial_bits():
    """parallel.InFlight (round 6): successive forwards alternate over two HIP streams -- each stream its own workspace inside the model -- and
    return exactly what one call after the other returns; a call on inputs that differ per request keeps the requests apart"""
    sys.path.insert(0, ROOT)
    DEV = "cuda:0"
    pkg = importlib.import_module("jittor-mlp_amd")
    parallel = importlib.import_module("jittor-mlp_amd.parallel")
    torch.manual_seed(0)
    model = pkg.models_pytorch.MLPMixerForImageClassification(d_model=128, depth=2, patch_size=16, image_size=64, num_classes=10).to(DEV).eval()
    xs = [torch.rand(8, 3, 64, 64, device=DEV).bfloat16() for _ in range(5)]
    with torch.no_grad():
        serial = [model(x).clone() for x in xs]
        slots = parallel.InFlight(lambda t: (hire if t.shape[0] == 4 else model)(t), 2, device=DEV)
        pending = [slots(x) for x in xs]                         # all five enqueued before any is waited for
        slots.synchronize()
    for (out, _), want in zip(pending, serial):
        assert torch.equal(out, want)
    assert len({id(s) for _, s in pending}) == 2
    # the in-flight regime switched the persistent GEMM to its whole-tile plan (engine.set_gemm_plan): same bits, and it can be undone
    assert pkg.engine.GEMM_PLAN_WHOLE and not pkg.engine.SIDE_STREAMS
    # ... and a model with a side chain of its own (Hire-MLP's second branch) issues it in line while forwards are in flight: same bits
    hire = pkg.models_pytorch.HireMLP(d_model=[32, 64], h=[4, 3], w=[4, 3], cross_region_step=[2, 1], depth=[2, 2], num_classes=10).to(DEV).eval()
    hx = [torch.rand(4, 3, 64, 64, device=DEV).bfloat16() for _ in range(3)]
    with torch.no_grad():
        got = [slots(x) for x in hx]
        slots.synchronize()
        slots.restore_plan()
        assert not pkg.engine.GEMM_PLAN_WHOLE and pkg.engine.SIDE_STREAMS
        for (out, _), x in zip(got, hx):
            assert torch.equal(out, hire(x))                     # (one call after the other, on its side stream again)
