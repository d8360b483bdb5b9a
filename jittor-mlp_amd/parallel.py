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
