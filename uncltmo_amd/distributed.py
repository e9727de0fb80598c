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

    The generator's backward pass produces its parameter gradients as views of flat fp32 buffers that belong to the pass
    (`_GradSet`).  With a reducer attached to the module those buffers are all-reduced IN PLACE (no concatenation, no copy
    back): the decoder's weights on a side stream from the moment their event fires -- while the graph block and the encoder
    are still running their backward kernels on the caller's stream -- the encoder's weights and the small tensors right after
    the pass.  xGMI rings are per-link bound: a few large messages per step, not 57 small ones.

    Ownership rule (what keeps the collective away from autograd): while a reducer is active the backward pass returns NO
    parameter gradients to autograd (None for every parameter), so nothing clones or accumulates a buffer that RCCL is still
    writing.  The buffers of every pass since the last step stay with the reducer; `finish` (DistributedOptimizer.step /
    .synchronize) waits for their collectives, divides by the world size and only then ADDS them into .grad -- one pass, the
    reference's two passes with retain_graph (GanTrainerImg.py:338,460), or gradient accumulation over several batches all
    give the mean over ranks of what a single process would have accumulated (the all-reduce is linear).  `.grad` of the
    generator's parameters is therefore defined after step() / synchronize(), not between backward() and step().
    `reset` (module.zero_grad / optimizer.zero_grad) drops the passes of a step that was never taken."""

    def __init__(self, group=None):
        import os
        self.group = group
        # in_stream: the all-reduces are issued on the CALLER's stream at the end of the backward pass (synchronous collectives,
        # no side stream).  A captured optimisation step takes this form always -- a replayed hipGraph pays ~0.2 ms per
        # cross-stream edge on this runtime, more than the 19.7 MB exchange costs over xGMI -- and an eager step takes it when
        # UNCL_DP_INSTREAM=1 (A/B against the overlapped form).
        self.in_stream = os.environ.get("UNCL_DP_INSTREAM") == "1"
        self._streams = {}
        self._passes = []        # finished backward passes since the last step: {"work": [(handle, tensor)], "grads": {...}}
        self._open = None        # the pass whose collectives are being launched

    def active(self):
        import os
        if not td.is_available() or not td.is_initialized():
            return False
        return td.get_world_size(self.group) > 1 or os.environ.get("UNCL_FORCE_DIST") == "1"

    def stream(self, dev):
        if dev not in self._streams:
            # high priority = another hardware-queue pool than the compute stream's: a normal-priority stream shares the
            # compute stream's queue whenever the runtime's round-robin says so, and the exchange then cannot overlap it
            self._streams[dev] = torch.cuda.Stream(device=dev, priority=-1)
        return self._streams[dev]

    def launch(self, tensor, owner):
        if self._open is None or self._open["owner"] is not owner:
            self._open = {"owner": owner, "work": [], "grads": None}
        self._open["work"].append((td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.group, async_op=True), tensor))

    def launch_in_stream(self, tensor, owner):
        """the same collective on the current stream, in order behind the kernels that produced `tensor` (capturable)"""
        if self._open is None or self._open["owner"] is not owner:
            self._open = {"owner": owner, "work": [], "grads": None}
        td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.group)
        self._open["work"].append((None, tensor))

    def keep(self, owner, grads, post=None):
        """end of a pass: `grads` (state_dict name -> view of the buffers just launched) and the owner (which keeps the
        buffers alive) stay here until finish().  `post`: a function applied to `grads` in finish(), AFTER the collectives and the
        averaging -- the re-layout of a variant generator's parameters (slices / tap sums of the published-layout buffers:
        generator._variant_grads), which must not copy anything while the reductions are in flight"""
        ps = self._open if self._open is not None and self._open["owner"] is owner else {"owner": owner, "work": [], "grads": None}
        ps["grads"] = grads
        ps["post"] = post
        self._passes.append(ps)
        self._open = None

    def pending(self):
        return len(self._passes)

    def reset(self):
        for ps in self._passes + ([self._open] if self._open is not None else []):
            for work, _ in ps["work"]:
                if work is not None:
                    work.wait()
        self._passes, self._open = [], None

    def finish(self, named_params):
        """Wait for the collectives of every pass since the last step, average, and add the result into .grad.
        True if there was anything to do (False: no generator backward ran through this reducer since the last step)."""
        if not self._passes:
            return False
        world = td.get_world_size(self.group)
        for ps in self._passes:
            for work, t in ps["work"]:
                if work is not None:
                    work.wait()              # the current stream waits for the collective; the host does not block
                if world > 1:
                    t.div_(world)
            if ps.get("post") is not None:
                ps["grads"], ps["post"] = ps["post"](ps["grads"]), None
        for k, p in named_params:
            if not p.requires_grad:
                continue
            total = None
            for ps in self._passes:
                g = ps["grads"].get(k)
                if g is not None:
                    total = g if total is None else total + g
            if total is not None:
                total = total.view_as(p) if total.shape != p.shape else total
                p.grad = total if p.grad is None else p.grad + total
        self._passes, self._open = [], None
        return True


class DistributedOptimizer:
    """Wraps an optimizer so that step() first averages the gradients across ranks (what DistributedDataParallel's
    reducer does for nn.DataParallel-free training); everything else is forwarded.  `module` = the HIP generator whose
    parameters the optimizer updates: its backward pass then reduces its own flat gradient buffers in place, overlapped with
    the backward tail (GradReducer), and its .grad tensors appear in step() / synchronize(); any other parameter (the
    discriminator's 12 373 gradients) goes through allreduce_gradients."""

    def __init__(self, optimizer, bucket_bytes=32 << 20, module=None):
        self.optimizer = optimizer
        self.bucket_bytes = bucket_bytes
        self.module = module
        if module is not None and hasattr(module, "_packed_weights"):
            # `module` must be the network THIS optimizer updates: a second wrapper (say, the discriminator's optimizer given
            # module=netG) would replace the generator's reducer, its zero_grad() would drop the generator's pending passes and
            # its step() would take the generator's gradients for its own -- ranks then diverge without an error
            own = {id(p) for g in optimizer.param_groups for p in g["params"]}
            if not any(id(p) in own for p in module.parameters()):
                raise ValueError("DistributedOptimizer(module=...): the optimizer updates none of the module's parameters; "
                                 "pass module= only to the optimizer of that module (the generator's)")
            prev = module.__dict__.get("_grad_reducer")
            if prev is not None and getattr(prev, "owner_optimizer", None) is not optimizer and prev.pending():
                raise RuntimeError("the module already has a GradReducer with pending passes from another optimizer")
            module._grad_reducer = GradReducer()
            module._grad_reducer.owner_optimizer = optimizer

    def synchronize(self):
        """Make .grad of every parameter the mean over ranks (call before reading or clipping gradients; step() does).
        An event pair brackets it on the compute stream: the time between them is what the step waited for the exchange --
        collectives that the backward pass launched and that had not finished, plus the averaging (exposed_allreduce_ms)."""
        ev = None
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        capturing = bool(params) and params[0].is_cuda and torch.cuda.is_current_stream_capturing()
        if params and params[0].is_cuda and td.is_available() and td.is_initialized() and not capturing:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        red = getattr(self.module, "_grad_reducer", None) if self.module is not None else None
        if red is not None and red.finish(list(self.module.named_parameters())):
            # the module's gradients came reduced out of its backward pass; parameters of this optimizer that do NOT belong to
            # the module still take the bucketed path
            mine = {id(p) for p in self.module.parameters()}
            params = [p for p in params if id(p) not in mine]
        allreduce_gradients(params, self.bucket_bytes)
        if ev is not None:
            ev[1].record()
            self.__dict__.setdefault("_sync_events", []).append(ev)
            del self._sync_events[:-256]

    def step(self, *a, **k):
        self.synchronize()
        return self.optimizer.step(*a, **k)

    def zero_grad(self, *a, **k):
        red = getattr(self.module, "_grad_reducer", None) if self.module is not None else None
        if red is not None:
            red.reset()
        return self.optimizer.zero_grad(*a, **k)

    def __getattr__(self, name):
        return getattr(self.optimizer, name)


def exposed_allreduce_ms(optimizers):
    """Per step: milliseconds the compute stream spent inside DistributedOptimizer.synchronize() (summed over the given
    optimizers, oldest step first).  Call after a device synchronisation; clears the record."""
    per_opt = []
    for o in optimizers:
        evs = o.__dict__.get("_sync_events") or []
        per_opt.append([a.elapsed_time(b) for a, b in evs])
        if evs:
            del evs[:]
    n = min((len(v) for v in per_opt), default=0)
    return [sum(v[len(v) - n + i] for v in per_opt) for i in range(n)]
