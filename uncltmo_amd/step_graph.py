"""One optimisation step (train_D + train_G) as ONE hipGraph replay.

The image step is ~360 kernel launches issued from Python through autograd and ctypes: ~4 ms of host work per 8.4 ms step on
an idle host -- GPU-bound there, but a host that is 2x slower (a loaded or cold box; the round-2 driver measured 17.5 ms)
makes the step host-bound.  Captured once and replayed, the step costs the host one hipGraphLaunch.

What makes the step capturable: no host synchronisation on the step path (losses stay on the device), every workspace comes
from the caching allocator (the graph's private pool during capture), the Adam step reads its learning rate and step count
from device memory (optim.Adam -> uncl_adam_step_dev), DropPath masks come from the device generator, the library itself never
allocates or copies from host memory on this path.  Inputs are STATIC tensors: `load(hdr, gray, pos, neg)` copies a new batch
into them.  Data-parallel runs (RCCL inside the step) stay eager.
"""
import torch


def capture_error_mode():
    """Stream-capture error mode for torch.cuda.graph().  'global' (PyTorch's default) makes a potentially unsafe runtime call by ANY
    thread of the process an error while the capture runs -- and torch.distributed's RCCL watchdog thread polls the events of
    earlier collectives (hipEventQuery) whenever it likes: inside a capture window that query fails, the watchdog rethrows and the
    process dies with SIGABRT and no message (one run in seven of the world-1 RCCL step-graph test; backtrace in
    DESIGN.md section 5).  With a process group alive the capture is 'thread_local': only the capturing thread is policed; work the
    autograd engine's thread puts on the capturing stream is captured either way."""
    try:
        import torch.distributed as td
        if td.is_available() and td.is_initialized():
            return "thread_local"
    except Exception:
        pass
    return "global"


class _Snapshot:
    """Parameters, module buffers, optimiser state and the device's random-number state of a trainer before StepGraph's warm-up steps: the warm-up is there to bring the library,
    the allocator and the weight packs into their steady state, not to train -- restore() puts every value back IN PLACE (the
    addresses the capture records stay the ones the warm-up used)."""

    def __init__(self, trainer):
        self.tr = trainer
        self.opts = [o for o in (trainer.optimizerD, trainer.optimizerG)]
        self.params, self.state = [], []
        with torch.no_grad():
            for opt in self.opts:
                inner = getattr(opt, "optimizer", opt)               # DistributedOptimizer wraps one
                for group in inner.param_groups:
                    for p in group["params"]:
                        self.params.append((p, p.detach().clone()))
                        st = inner.state.get(p)
                        self.state.append((inner, group, p, None if not st else
                                           (int(st["step"]), st["exp_avg"].clone(), st["exp_avg_sq"].clone())))
        self.lists = [(lst, len(lst)) for lst in (trainer.D_losses, trainer.G_loss_d, trainer.G_loss_struct)]
        # module buffers too: a batch_norm generator's train-mode forward updates running_mean / running_var / num_batches_tracked
        # in place (several forwards per warm-up step), and the DropPath draws consume the device's random-number stream
        self.buffers = []
        with torch.no_grad():
            for net in (getattr(trainer, "netG", None), getattr(trainer, "netD", None)):
                if net is not None:
                    self.buffers += [(b, b.detach().clone()) for _, b in net.named_buffers()]
        # the random-number streams a warm-up step can draw from: the HOST generator (host-side augmentation, a keep mask built on the
        # host) and the generator of the device the trainer's parameters live on (DropPath).  Covered: netG / netD buffers only -- a
        # stateful loss module or a caller's own buffers are not snapshotted.
        self.dev = next((p.device for p, _ in self.params if p.is_cuda), None)
        self.rng_host = torch.get_rng_state()
        self.rng = torch.cuda.get_rng_state(self.dev) if self.dev is not None else None

    def restore(self):
        with torch.no_grad():
            for b, saved in self.buffers:
                b.copy_(saved)
            torch.set_rng_state(self.rng_host)
            if self.rng is not None:
                torch.cuda.set_rng_state(self.rng, self.dev)
            for p, saved in self.params:
                p.copy_(saved)                                       # bumps ._version: the weight packs are rebuilt
                p.grad = None
            for inner, group, p, saved in self.state:
                st = inner.state.get(p)
                if not st:
                    continue
                if saved is None:
                    st["step"] = 0
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
                else:
                    st["step"] = saved[0]
                    st["exp_avg"].copy_(saved[1])
                    st["exp_avg_sq"].copy_(saved[2])
                host = group.get("_uncl_hyper_host")
                if host is not None:                                 # device-side step count of optim.Adam
                    host[1] = st["step"]
                    group["_uncl_hyper"][1:2].fill_(float(st["step"]))
        for lst, n in self.lists:
            del lst[n:]


class StepGraph:
    def __init__(self, trainer, hdr_input, hdr_original_gray_norm, real_ldr_pos, real_ldr_neg, epoch, warmup=3,
                 keep_warmup_updates=False):
        """`warmup` eager steps run first (lazy initialisation inside the library, the allocator's pools, the weight packs);
        unless `keep_warmup_updates`, parameters and optimiser state are put back afterwards, so that building a StepGraph
        trains nothing: N replays == N eager steps."""
        self.tr, self.epoch = trainer, epoch
        self.inputs = [t.clone() for t in (hdr_input, hdr_original_gray_norm, real_ldr_pos, real_ldr_neg)]
        self.replays = 0
        snap = None if keep_warmup_updates else _Snapshot(trainer)
        # Warm-up AND capture on ONE side stream.  torch.cuda.graph() without `stream=` captures on a stream of its own: the
        # autograd nodes (AccumulateGrad) and allocator blocks the warm-up created on `side` then belong to another stream than
        # the capture -- autograd warns ("AccumulateGrad node's stream does not match"), inserts cross-stream edges into the
        # graph, and a block freed on one stream and reused on the other without record_stream is a use-after-free candidate.
        side = self.stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
            if snap is not None:
                snap.restore()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode=capture_error_mode()):
            self._eager()
        # the capture pass ran the Python of one step (optimizer step counts went up) but none of its kernels
        for opt in (trainer.optimizerD, trainer.optimizerG):
            adv = getattr(opt, "advance_host_steps", None)
            if adv is not None:
                adv(-1)
        for lst in (trainer.D_losses, trainer.G_loss_d, trainer.G_loss_struct):
            if lst:
                lst.pop()
        # The capture pass stored pack keys for weight packs whose kernels were only RECORDED (their buffers live in the graph's
        # pool and hold nothing until the first replay): an eager forward before that replay must re-pack, not trust the key.
        for net in (getattr(trainer, "netG", None), getattr(trainer, "netD", None)):
            if net is not None and hasattr(net, "_pack_key"):
                net._pack_key = None
                net._packed = None
        trainer._step_graph = self

    def _eager(self):
        hdr, gray, pos, neg = self.inputs
        self.tr.train_D(hdr, pos, neg, self.epoch)
        self.tr.train_G(hdr, gray, pos, neg, self.epoch)

    def load(self, hdr_input, hdr_original_gray_norm, real_ldr_pos, real_ldr_neg):
        for dst, src in zip(self.inputs, (hdr_input, hdr_original_gray_norm, real_ldr_pos, real_ldr_neg)):
            dst.copy_(src, non_blocking=True)

    def replay(self):
        """one optimisation step on the batch last given to load(); trainer.errD / errG_d / errG_struct are updated in place"""
        # a learning rate changed since the last step (lr_scheduler.step() edits param_groups): Adam reads it from device memory,
        # and the only code that writes it there is the Python step() a replay never runs -- push it eagerly, in stream order
        for opt in (self.tr.optimizerD, self.tr.optimizerG):
            sync = getattr(opt, "sync_device_hyper", None)
            if sync is not None:
                sync()
        self.graph.replay()
        self.replays += 1
        for opt in (self.tr.optimizerD, self.tr.optimizerG):
            adv = getattr(opt, "advance_host_steps", None)
            if adv is not None:
                adv(1)
        tr = self.tr
        tr.D_losses.append(tr.errD.detach())
        tr.G_loss_d.append(tr.errG_d.detach())
        if tr.struct_loss_factor:
            tr.G_loss_struct.append(tr.errG_struct.detach())
