"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) and xGMI.

Inference shards frames / tiles with no collective.  Training exchanges gradients once per optimiser step: the
generator's 4 920 545 fp32 gradients (19.7 MB) and the discriminator's 12 373, averaged over ranks in a few large
buckets (xGMI rings are per-link bound: few big messages, not many small ones).  Batch-coupled losses (contrastive
GAN, InfoNCE selection, pseudo label) use per-rank-local semantics: each rank equals the reference at batch N/world."""
import torch
import torch.distributed as td


def shard_range(n, rank, world):
    """Contiguous [lo, hi) block of n units for this rank (units = frames or tiles)."""
    per, rem = divmod(n, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def allreduce_gradients(params, bucket_bytes=32 << 20, group=None):
    """Average .grad over ranks in place, flattening into buckets of about `bucket_bytes`."""
    if not td.is_available() or not td.is_initialized() or td.get_world_size(group) == 1:
        return
    world = td.get_world_size(group)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        td.all_reduce(flat, op=td.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0

    for p in params:
        if p.grad is None:
            continue
        bucket.append(p.grad)
        size += p.grad.numel() * p.grad.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


class DistributedOptimizer:
    """Wraps an optimizer so that step() first averages the gradients across ranks (what DistributedDataParallel's
    reducer does for nn.DataParallel-free training); everything else is forwarded."""

    def __init__(self, optimizer, bucket_bytes=32 << 20):
        self.optimizer = optimizer
        self.bucket_bytes = bucket_bytes

    def step(self, *a, **k):
        allreduce_gradients([p for g in self.optimizer.param_groups for p in g["params"]], self.bucket_bytes)
        return self.optimizer.step(*a, **k)

    def __getattr__(self, name):
        return getattr(self.optimizer, name)
