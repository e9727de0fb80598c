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


class GradReducer:
    """In-place, overlapped gradient all-reduce for the HIP generator (uncltmo_amd/autograd.py:_GradSet.finish).

    The generator's backward pass produces its parameter gradients as views of three flat fp32 buffers.  With a reducer attached
    to the module, those buffers are all-reduced IN PLACE (no concatenation, no copy back): the decoder's weights on a side
    stream from the moment their event fires -- while the graph block and the encoder are still running their backward kernels
    on the caller's stream -- the encoder's weights and the small tensors right after the pass.  `finish` (called by
    DistributedOptimizer.step) waits, divides by the world size and points every .grad at its reduced view.  xGMI rings are
    per-link bound: three large messages per step, not 57 small ones."""

    def __init__(self, group=None):
        self.group = group
        self._streams = {}
        self._pending = []
        self._grads = None
        self._calls = 0

    def active(self):
        import os
        if not td.is_available() or not td.is_initialized():
            return False
        return td.get_world_size(self.group) > 1 or os.environ.get("UNCL_FORCE_DIST") == "1"

    def stream(self, dev):
        if dev not in self._streams:
            self._streams[dev] = torch.cuda.Stream(device=dev)
        return self._streams[dev]

    def launch(self, tensor, owner):
        self._pending.append((td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.group, async_op=True), tensor, owner))

    def keep(self, owner, grads):
        self._grads = grads
        self._calls += 1

    def reset(self):
        for work, _, _ in self._pending:
            work.wait()
        self._pending, self._grads, self._calls = [], None, 0

    def finish(self, named_params):
        """True if the gradients of this step were reduced here (exactly one backward pass since the last step: a second pass
        accumulates into .grad outside these buffers, and the generic bucketed path takes over)."""
        if self._calls != 1 or self._grads is None:
            self.reset()
            return False
        world = td.get_world_size(self.group)
        for work, t, _ in self._pending:
            work.wait()                      # the current stream waits for the collective; the host does not block
            if world > 1:
                t.div_(world)
        for k, p in named_params:
            g = self._grads.get(k)
            if g is not None and p.requires_grad:
                p.grad = g
        self._pending, self._grads, self._calls = [], None, 0
        return True


class DistributedOptimizer:
    """Wraps an optimizer so that step() first averages the gradients across ranks (what DistributedDataParallel's
    reducer does for nn.DataParallel-free training); everything else is forwarded.  `module` = the HIP generator whose
    parameters the optimizer updates: its backward pass then reduces its own flat gradient buffers in place, overlapped with
    the backward tail (GradReducer); any other case (the discriminator's 12 373 gradients, a two-pass backward) goes through
    allreduce_gradients."""

    def __init__(self, optimizer, bucket_bytes=32 << 20, module=None):
        self.optimizer = optimizer
        self.bucket_bytes = bucket_bytes
        self.module = module
        if module is not None and hasattr(module, "_packed_weights"):
            module._grad_reducer = GradReducer()

    def step(self, *a, **k):
        red = getattr(self.module, "_grad_reducer", None) if self.module is not None else None
        if red is None or not red.finish(list(self.module.named_parameters())):
            allreduce_gradients([p for g in self.optimizer.param_groups for p in g["params"]], self.bucket_bytes)
        return self.optimizer.step(*a, **k)

    def zero_grad(self, *a, **k):
        red = getattr(self.module, "_grad_reducer", None) if self.module is not None else None
        if red is not None:
            red.reset()
        return self.optimizer.zero_grad(*a, **k)

    def __getattr__(self, name):
        return getattr(self.optimizer, name)
