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


class StepGraph:
    def __init__(self, trainer, hdr_input, hdr_original_gray_norm, real_ldr_pos, real_ldr_neg, epoch, warmup=3):
        self.tr, self.epoch = trainer, epoch
        self.inputs = [t.clone() for t in (hdr_input, hdr_original_gray_norm, real_ldr_pos, real_ldr_neg)]
        self.replays = 0
        # eager steps on a side stream first (the documented capture recipe): lazy initialisation inside the library, the
        # allocator's pools and the weight packs reach their steady state before anything is recorded
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._eager()
        # the capture pass ran the Python of one step (optimizer step counts went up) but none of its kernels
        for opt in (trainer.optimizerD, trainer.optimizerG):
            adv = getattr(opt, "advance_host_steps", None)
            if adv is not None:
                adv(-1)
        for lst in (trainer.D_losses, trainer.G_loss_d, trainer.G_loss_struct):
            if lst:
                lst.pop()
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
