"""Data-parallel inference across the GPUs of one node: one process per GPU, weights replicated,
the batch sharded on dim 0 in equal contiguous slices, and ONE collective per forward -- an
all-gather of the (B_local, num_classes) logits (RCCL over xGMI).  Images are
independent in eval mode, so there is no exchange inside the model.  The reference has no
distributed code at all; this is new.

The message is tiny (256 x 1000 logits = 0.5-1 MB per rank), i.e. latency-bound on the xGMI mesh:
it is issued through torch.distributed (backend "nccl" == RCCL on ROCm) right behind the head GEMM,
stream-ordered (no host synchronisation), so successive forwards pipeline.
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
    """

    def __init__(self, forward_fn, world=1, group=None):
        self.forward_fn = forward_fn
        self.world = world
        self.group = group
        self._out = None

    def __call__(self, x_local):
        logits = self.forward_fn(x_local)
        if self.world == 1:
            return logits
        import torch.distributed as dist
        logits = logits.contiguous()
        shape = (self.world * logits.shape[0],) + tuple(logits.shape[1:])
        if self._out is None or self._out.shape != shape or self._out.dtype != logits.dtype or self._out.device != logits.device:
            self._out = torch.empty(shape, dtype=logits.dtype, device=logits.device)
        dist.all_gather_into_tensor(self._out, logits, group=self.group)
        return self._out
