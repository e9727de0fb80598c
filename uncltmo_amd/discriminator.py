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
        g_ft = g_fea.reshape(n, 2).float().t().contiguous()      # (2, n): one copy, both feature gradients as contiguous rows
        g_f1, g_f2 = g_ft[0], g_ft[1]
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


class Conv2dBlock(nn.Module):
    """Parameter holder with the reference's member names (models/Blocks.py:6-36): `conv` without bias, then instance
    norm (no affine parameters) and LeakyReLU(0.2) -- applied by the HIP kernels, never by these modules."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none", activation="none"):
        super().__init__()
        if norm != "instance_norm" or activation != "leakyReLU":
            raise NotImplementedError("the HIP PatchGAN covers Conv2dBlock(norm='instance_norm', activation='leakyReLU')")
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, padding, bias=False)


class _PatchDFn(torch.autograd.Function):
    """PatchGAN forward keeping what its backward needs (uncl_patch_d_forward_train / uncl_patch_d_backward, fp32)."""

    @staticmethod
    def forward(ctx, x, ndf, n_layers, b_first, b_last, *weights):
        import ctypes as C
        lib = _hip.lib()
        n, h = x.shape[0], x.shape[2]
        xf = x.detach().reshape(n, h, h).float().contiguous()
        ws_ = [w.detach().float().contiguous() for w in weights]
        b0, bl = b_first.detach().float().contiguous(), b_last.detach().float().contiguous()
        wp = (C.c_void_p * len(ws_))(*[w.data_ptr() for w in ws_])
        ho = lib.uncl_patch_d_out_size(h, n_layers)
        out = torch.empty(n, 1, ho, ho, dtype=torch.float32, device=x.device)
        arena = torch.empty(lib.uncl_patch_d_train_bytes(n, h, ndf, n_layers), dtype=torch.uint8, device=x.device)
        _hip.check(lib.uncl_patch_d_forward_train(xf.data_ptr(), wp, b0.data_ptr(), bl.data_ptr(), out.data_ptr(), n, h, ndf,
                                                  n_layers, arena.data_ptr(), _hip.stream_ptr()), "uncl_patch_d_forward_train")
        ctx.save_for_backward(xf, arena, *ws_)
        ctx.geom = (n, h, ndf, n_layers)
        return out

    @staticmethod
    def backward(ctx, g_out):
        import ctypes as C
        lib = _hip.lib()
        xf, arena, *ws_ = ctx.saved_tensors
        n, h, ndf, n_layers = ctx.geom
        g = g_out.detach().float().contiguous()
        gws = [torch.empty_like(w) for w in ws_]
        gb0 = torch.empty(ndf, dtype=torch.float32, device=g.device)
        gbl = torch.empty(1, dtype=torch.float32, device=g.device)
        gx = torch.empty(n, h, h, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        wp = (C.c_void_p * len(ws_))(*[w.data_ptr() for w in ws_])
        gp = (C.c_void_p * len(gws))(*[w.data_ptr() for w in gws])
        _hip.check(lib.uncl_patch_d_backward(xf.data_ptr(), wp, g.data_ptr(), gp, gb0.data_ptr(), gbl.data_ptr(),
                                             gx.data_ptr() if gx is not None else None, n, h, ndf, n_layers, arena.data_ptr(),
                                             _hip.stream_ptr()), "uncl_patch_d_backward")
        return (gx.reshape(n, 1, h, h) if gx is not None else None, None, None, gb0, gbl) + tuple(gws)


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator (reference: models/Discriminator.py:129-167; same constructor, state_dict keys and output
    shape (N,1,30,30) for 256x256 inputs).  Under torch.no_grad() the ping-pong forward runs; with gradients enabled the
    training form keeps the activations and `backward()` fills every parameter's .grad and the input gradient."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer="batch_norm", last_activation="none"):
        super().__init__()
        if input_nc != 1 or norm_layer != "instance_norm" or last_activation != "none" or ndf % 8 != 0:
            raise NotImplementedError("the HIP PatchGAN covers input_nc=1, norm_layer='instance_norm', no last activation")
        self.ndf, self.n_layers = ndf, n_layers
        seq = [nn.Conv2d(input_nc, ndf, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq.append(Conv2dBlock(ndf * prev, ndf * mult, 4, 2, 1, norm=norm_layer, activation="leakyReLU"))
        prev, mult = mult, min(2 ** n_layers, 8)
        seq.append(Conv2dBlock(ndf * prev, ndf * mult, 4, 1, 1, norm=norm_layer, activation="leakyReLU"))
        seq.append(nn.Conv2d(ndf * mult, 1, kernel_size=4, stride=1, padding=1))
        self.model = nn.Sequential(*seq)

    def forward(self, input):
        import ctypes as C
        x = input
        if not x.is_cuda:
            raise _hip.HipError("NLayerDiscriminator needs CUDA(HIP) tensors; there is no CPU path")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != x.shape[3]:
            raise ValueError("NLayerDiscriminator expects square one-channel frames (N,1,H,H)")
        convs = [self.model[0]] + [m.conv for m in self.model if isinstance(m, Conv2dBlock)] + [self.model[-1]]
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _PatchDFn.apply(x, self.ndf, self.n_layers, convs[0].bias, convs[-1].bias, *[c.weight for c in convs])
        lib = _hip.lib()
        n, h = x.shape[0], x.shape[2]
        xf = x.detach().reshape(n, h, h).float().contiguous()
        ws_ = [c.weight.detach().float().contiguous() for c in convs]
        b0, bl = convs[0].bias.detach().float().contiguous(), convs[-1].bias.detach().float().contiguous()
        wp = (C.c_void_p * len(ws_))(*[w.data_ptr() for w in ws_])
        ho = lib.uncl_patch_d_out_size(h, self.n_layers)
        out = torch.empty(n, 1, ho, ho, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.uncl_patch_d_workspace_bytes(n, h, self.ndf, self.n_layers), dtype=torch.uint8, device=x.device)
        _hip.check(lib.uncl_patch_d_forward(xf.data_ptr(), wp, b0.data_ptr(), bl.data_ptr(), out.data_ptr(), n, h, self.ndf,
                                            self.n_layers, ws.data_ptr(), _hip.stream_ptr()), "uncl_patch_d_forward")
        return out
