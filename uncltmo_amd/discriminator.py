"""SimpleDiscriminator, MI355X-native: same constructor / forward signature and state_dict keys as the reference's
`models/Discriminator.py:SimpleDiscriminator` (:87-126); forward and backward are HIP kernels (csrc/discriminator.hip).
The nn.Conv2d / nn.Linear members only hold the parameters (reference layout); they are never called."""
import torch
import torch.nn as nn

from . import _hip


class _SimpleDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2, w4, b4, wl):
        lib = _hip.lib()
        n = x.shape[0]
        xf = x.detach().reshape(n, 256, 256).float().contiguous()
        ps = [t.detach().float().contiguous() for t in (w0, b0, w2, b2, w4, b4, wl)]
        dev = xf.device
        ws = torch.empty(lib.uncl_simple_d_workspace_bytes(n), dtype=torch.uint8, device=dev)
        gs = torch.empty(lib.uncl_gauss_stats_workspace_bytes(n, 62, 1), dtype=torch.uint8, device=dev)
        out = torch.empty(n, 1, dtype=torch.float32, device=dev)
        fea = torch.empty(n, 2, dtype=torch.float32, device=dev)
        _hip.check(lib.uncl_simple_d_forward(_hip.ptr(xf), *[_hip.ptr(p) for p in ps], out.data_ptr(), fea.data_ptr(), n,
                                             ws.data_ptr(), gs.data_ptr(), _hip.stream_ptr()), "uncl_simple_d_forward")
        ctx.saved = (xf, ps, ws, n)
        return out, fea.reshape(n, 2, 1, 1)

    @staticmethod
    def backward(ctx, g_out, g_fea):
        lib = _hip.lib()
        xf, ps, ws, n = ctx.saved
        w0, b0, w2, b2, w4, b4, wl = ps
        dev = xf.device
        st = _hip.stream_ptr()
        g_out = g_out.reshape(n).float().contiguous()
        g_fea = g_fea.reshape(n, 2).float().contiguous()
        g_f1 = g_fea[:, 0].contiguous()
        g_f2 = g_fea[:, 1].contiguous()
        # gradient reaching `fea` through the Gaussian local-variance feature; fea lives in the forward workspace
        h1_sz, h2_sz = n * 127 * 127 * 16, n * 62 * 62 * 32
        fea_map = ws[(h1_sz + h2_sz) * 4:(h1_sz + h2_sz + n * 62 * 62) * 4].view(torch.float32).reshape(n, 62, 62)
        g_var = torch.empty(n, 62, 62, dtype=torch.float32, device=dev)
        _hip.check(lib.uncl_gauss_var_backward(fea_map.data_ptr(), g_f2.data_ptr(), g_var.data_ptr(), n, 62, 62, 0, st),
                   "uncl_gauss_var_backward")
        need_p = any(ctx.needs_input_grad[1:])
        need_x = ctx.needs_input_grad[0]
        gp = [torch.empty_like(p) for p in ps] if need_p else [None] * 7
        gx = torch.empty(n, 256, 256, dtype=torch.float32, device=dev) if need_x else None
        P = lambda t: t.data_ptr() if t is not None else None
        _hip.check(lib.uncl_simple_d_backward(P(xf), P(w0), P(w2), P(w4), P(wl), P(g_out), P(g_f1), P(g_var), P(gp[0]), P(gp[1]),
                                              P(gp[2]), P(gp[3]), P(gp[4]), P(gp[5]), P(gp[6]), P(gx), 0, n, ws.data_ptr(), st),
                   "uncl_simple_d_backward")
        return (gx.reshape(n, 1, 256, 256) if need_x else None,) + tuple(gp)


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size()[0], -1)


class SimpleDiscriminator(nn.Module):
    def __init__(self, input_size, input_dim, dim, norm, last_activation, simpleD_maxpool, padding):
        super().__init__()
        if input_size != 256 or input_dim != 1 or dim != 16 or simpleD_maxpool or padding or last_activation != "none":
            raise NotImplementedError("the HIP SimpleDiscriminator covers the published configuration "
                                      "(input 256, 1 channel, dim 16, no padding / maxpool / sigmoid)")
        last_dim = ((input_size // 2 - 1) // 2 - 1) ** 2
        self.model = nn.Sequential(nn.Conv2d(input_dim, dim, 4, 2, padding=padding, bias=True), nn.LeakyReLU(0.2, inplace=True),
                                   nn.Conv2d(dim, dim * 2, 4, 2, padding=padding, bias=True), nn.LeakyReLU(0.2, inplace=True),
                                   nn.Conv2d(dim * 2, 1, kernel_size=1, stride=1, padding=0, bias=True))
        self.tail = nn.Sequential(Flatten(), nn.Linear(last_dim, 1, bias=False))

    def forward(self, x):
        if not x.is_cuda:
            raise _hip.HipError("SimpleDiscriminator needs CUDA(HIP) tensors; there is no CPU path")
        if x.shape[-1] != 256 or x.shape[-2] != 256 or x.shape[1] != 1:
            raise ValueError("SimpleDiscriminator expects (N,1,256,256) inputs")
        m, t = self.model, self.tail
        return _SimpleDFn.apply(x, m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias, t[1].weight)
