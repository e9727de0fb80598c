"""torch.optim.Adam semantics (no weight decay / amsgrad) with the update done by one fused HIP launch per step
(uncl_adam_step_dev; uncl_adam_step when tensors sit at different step counts).  Drop-in for the `optim.Adam(net.parameters(), lr=..., betas=(0.5, 0.999))` the reference builds in
main_train_image.py:29-32; works with torch.optim.lr_scheduler.StepLR (it only edits param_groups[...]['lr'])."""
import ctypes as C

import torch

from . import _hip


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _hip.lib()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda:
                    raise _hip.HipError("uncltmo_amd.optim.Adam needs CUDA(HIP) parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                st["step"] += 1
            arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            b1, b2 = group["betas"]
            # the bias corrections depend on each tensor's OWN step count (torch.optim.Adam): tensors that got their first
            # gradient later, or a loaded optimizer state with uneven steps, go in a launch of their own
            by_step = {}
            for p in ps:
                by_step.setdefault(int(self.state[p]["step"]), []).append(p)
            if len(by_step) == 1:
                # the usual case, every tensor at the same step: learning rate and step count live in DEVICE memory and the step
                # count is advanced by a device-side add, so no kernel argument changes from step to step and a captured
                # optimisation step (uncltmo_amd.step_graph) replays with the right bias corrections
                (step, sub), = by_step.items()
                hyper = group.get("_uncl_hyper")
                if hyper is None or hyper.device != sub[0].device:
                    hyper = group["_uncl_hyper"] = torch.tensor([float(group["lr"]), float(step - 1)], dtype=torch.float32,
                                                                device=sub[0].device)
                    group["_uncl_hyper_host"] = [float(group["lr"]), step - 1]
                host = group["_uncl_hyper_host"]
                if host[1] != step - 1:                          # a loaded state dict moved the step count
                    hyper[1:2].fill_(float(step - 1))
                if host[0] != float(group["lr"]):                # lr_scheduler.step()
                    hyper[0:1].fill_(float(group["lr"]))
                hyper[1:2].add_(1.0)
                host[0], host[1] = float(group["lr"]), step
                grads = [p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous() for p in sub]
                n = (C.c_int * len(sub))(*[p.numel() for p in sub])
                _hip.check(lib.uncl_adam_step_dev(arr(sub), arr(grads), arr([self.state[p]["exp_avg"] for p in sub]),
                                                  arr([self.state[p]["exp_avg_sq"] for p in sub]), n, len(sub), hyper.data_ptr(),
                                                  float(b1), float(b2), float(group["eps"]), _hip.stream_ptr()), "uncl_adam_step_dev")
            else:
                for step, sub in sorted(by_step.items()):
                    grads = [p.grad.float().contiguous() for p in sub]
                    n = (C.c_int * len(sub))(*[p.numel() for p in sub])
                    _hip.check(lib.uncl_adam_step(arr(sub), arr(grads), arr([self.state[p]["exp_avg"] for p in sub]),
                                                  arr([self.state[p]["exp_avg_sq"] for p in sub]), n, len(sub), float(group["lr"]),
                                                  float(b1), float(b2), float(group["eps"]), int(step), _hip.stream_ptr()),
                               "uncl_adam_step")
            # parameters changed behind autograd's back (no ._version bump): mark THESE tensors so that the module that owns
            # them re-packs its weights -- and only that module (the discriminator's step leaves the generator's packs valid)
            for p in ps:
                p._uncl_epoch = getattr(p, "_uncl_epoch", 0) + 1
            _hip.PARAM_EPOCH[0] += 1
        return loss


    def sync_device_hyper(self):
        """Push a learning rate that changed on the host (lr_scheduler.step() edits param_groups[...]['lr']) into the device-side
        hyper-parameter buffer, in stream order.  step() does this itself; a REPLAYED hipGraph of step() never runs that Python,
        so step_graph.StepGraph.replay() calls this before every launch (one comparison per group when nothing changed)."""
        for group in self.param_groups:
            host = group.get("_uncl_hyper_host")
            if host is not None and host[0] != float(group["lr"]):
                group["_uncl_hyper"][0:1].fill_(float(group["lr"]))
                host[0] = float(group["lr"])

    def advance_host_steps(self, n=1):
        """A replayed hipGraph of step() advanced the device-side step count n times without running this Python: bring the
        host bookkeeping (state[p]['step'], what state_dict() saves) in line."""
        for group in self.param_groups:
            moved = False
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] += n
                    moved = True
                    p._uncl_epoch = getattr(p, "_uncl_epoch", 0) + n     # the module's weight packs are stale (see step())
            if moved and "_uncl_hyper_host" in group:
                group["_uncl_hyper_host"][1] += n
        _hip.PARAM_EPOCH[0] += n

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:            # device-side mirrors are rebuilt on demand, not saved
            g.pop("_uncl_hyper", None)
            g.pop("_uncl_hyper_host", None)
        return sd
