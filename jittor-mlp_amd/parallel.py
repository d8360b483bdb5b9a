"""Data-parallel inference across the GPUs of one node: one process per GPU, weights replicated,
the batch sharded on dim 0 in equal contiguous slices, and ONE collective per forward -- an
all-gather of the (B_local, num_classes) logits (RCCL over xGMI).  Images are
independent in eval mode, so there is no exchange inside the model.  The reference has no
distributed code at all; this is new.

The message is tiny (256 x 1000 logits = 0.5-1 MB per rank), i.e. latency-bound on the xGMI mesh:
it is issued through torch.distributed (backend "nccl" == RCCL on ROCm) right behind the head GEMM,
stream-ordered (no host synchronisation), so successive forwards pipeline.

Ordering relied upon (ProcessGroupNCCL, synchronous-op form): the kernels of the forward are enqueued on
torch's current stream S of the rank's device (engine.stream()); all_gather_into_tensor records an event
on S and makes the process group's RCCL stream wait for it, enqueues ncclAllGather there, records the
collective's end event and makes S wait for that event before returning.  Everything later on S
(the next forward, a copy of the result) is therefore ordered after the gather, and the gather after the
head GEMM, with no host-side synchronisation anywhere; the result tensor is allocated on S and is only
ever touched on S, so no record_stream bookkeeping is needed.  tests/test_gpu_parallel.py runs exactly this
path with two ranks (HIP forward + collective) and checks the rows bit for bit.
"""
import torch


def shard_batch(x, rank, world):
    """Equal contiguous slice of a global batch for `rank` (dim 0 must divide evenly)."""
    n = x.shape[0]
    if n % world:
        raise ValueError("global batch %d is not divisible by %d ranks" % (n, world))
    per = n // world
    return x[rank * per:(rank + 1) * per]


class DataParallelForward:
    """Callable wrapper: local forward + all-gather of the logits in rank order.

    forward_fn : callable mapping the local shard (B_local, ...) to (B_local, num_classes)
    world      : number of ranks (1 = no collective, no process group needed)
    group      : optional torch.distributed process group (default group when None)
    force_collective : issue the all-gather for world == 1 too (a one-rank RCCL group on a single-GPU box: exercises
                 the collective and its stream ordering where no second GPU exists)
    """

    def __init__(self, forward_fn, world=1, group=None, force_collective=False):
        self.forward_fn = forward_fn
        self.world = world
        self.group = group
        self.force_collective = force_collective

    def __call__(self, x_local):
        logits = self.forward_fn(x_local)
        if self.world == 1 and not self.force_collective:
            return logits
        import torch.distributed as dist
        logits = logits.contiguous()
        shape = (self.world * logits.shape[0],) + tuple(logits.shape[1:])
        # a fresh result per call (torch's caching allocator makes this free): the caller may keep step i's logits
        # while step i+1 is already running
        out = torch.empty(shape, dtype=logits.dtype, device=logits.device)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, logits, group=self.group)       # one ncclAllGather straight into `out`
        else:
            # gloo (CPU tests, or two ranks sharing one GPU): the list form, gathering into the row blocks of `out`
            dist.all_gather(list(out.chunk(self.world, dim=0)), logits, group=self.group)
        return out


class InFlight:
    """Keep `n` forwards in flight on one GPU (round 6): successive calls alternate over `n` HIP streams, so the next request's workgroups
    fill the CUs that the tail of the current request's persistent kernels leaves idle (every big kernel of the path is a persistent grid
    with a static tile list: a launch ends with a partly empty last round).  Each stream has its own workspace inside the model
    (EngineModule._get_space is keyed by the stream), the kernels and hence the results are the same bits as one call after the other.
    Same-box: Mixer-B/16 at 256 images 34.65 -> 36.55 k images/s with n = 2; n = 3 adds nothing (profiles/r06_two_steps_in_flight_probe.txt).

        slots = InFlight(model, 2)
        out, stream = slots(x)          # enqueued, not waited for; x must stay alive and unmodified until the call has run
        ...
        torch.cuda.current_stream().wait_stream(stream)   # (or stream.synchronize()) before `out` is consumed elsewhere
    """

    def __init__(self, forward_fn, n=2, device=None, throughput_plan=True):
        self.forward_fn = forward_fn
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, int(n)))]
        self._i = 0
        if throughput_plan and len(self.streams) > 1:
            # with several forwards sharing the chip the persistent GEMM's plan with the least TOTAL work wins over the one with the shortest
            # single launch (engine.set_gemm_plan): whole 256-row tiles.  Process-wide, same bits; InFlight.restore_plan() undoes it.
            from . import engine as E
            self._old_plan = E.GEMM_PLAN_WHOLE
            E.set_gemm_plan(True)
            # ... and a model's own side stream (Hire-MLP's second branch chain) only couples the in-flight streams: issue in line
            self._old_side = E.SIDE_STREAMS
            E.set_side_streams(False)

    def restore_plan(self):
        if hasattr(self, "_old_plan"):
            from . import engine as E
            E.set_gemm_plan(self._old_plan)
            E.set_side_streams(self._old_side)
            del self._old_plan

    def __call__(self, x):
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(torch.cuda.current_stream(x.device))         # x may have been produced on the caller's stream
        with torch.cuda.stream(s):
            out = self.forward_fn(x)
        return out, s

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
