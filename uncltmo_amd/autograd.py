"""Autograd bridge of the image and video generators: forward = uncl_gen_forward with activations kept, backward = ONE call of
uncl_gen_backward (hand-written HIP dgrad / wgrad / element-wise kernels), then the packed weight gradients are
re-laid-out into the reference parameter layout.  compute_dtype 'bf16' is the training path (bf16 activations and activation
gradients, fp32 accumulation, fp32 parameter gradients); 'fp32' is the PARITY mode: everything in fp32, weight gradients by
deterministic fixed-order kernels (csrc/bwd_f32.hip) -- slower, used to check a trainer step against the CPU oracle at fp32
tolerances (tests/test_gpu_trainer.py, tests/test_gpu_backward.py)."""
import ctypes as C

import torch

from . import _hip
from .state_spec import generator_spec


class _GradSet:
    """Packed fp32 gradient buffers of one generator and the calls that fill them (one per frame batch)."""

    def __init__(self, module, dev):
        lib = _hip.lib()
        self.module, self.dev = module, dev
        self.spec = {k: (shape, kind) for k, shape, kind in generator_spec()}
        self.names = [lib.uncl_gen_layer_name(i).decode() for i in range(_hip.G_NUM_WEIGHTS)]
        self.sizes = [int(torch.tensor(self.spec[nm + ".weight"][0]).prod()) for nm in self.names]
        self.gw_flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)   # packed, accumulated with atomics
        self.gb = [torch.empty(self.spec[nm + ".bias"][0], dtype=torch.float32, device=dev) for nm in self.names]
        self.g_inc_w = torch.empty(32, 1, 3, 3, dtype=torch.float32, device=dev)
        self.g_inc_b = torch.empty(32, dtype=torch.float32, device=dev)
        self.g_oc_w = torch.empty(32, dtype=torch.float32, device=dev)
        self.g_oc_b = torch.empty(1, dtype=torch.float32, device=dev)
        self.g_pe = torch.empty(144, 256, dtype=torch.float32, device=dev)
        self.gws = None

    def run(self, xf, out, up, ws, ds, g_out, gup, accumulate=False, prev_ws=None, carry_in=None, carry_out=None):
        """One uncl_gen_backward call over the n samples of (xf, out, up, ws)."""
        lib = _hip.lib()
        module = self.module
        gwts, _keep = module._packed_weights()
        n = xf.shape[0]
        gbytes = lib.uncl_gen_backward_workspace_bytes_dt(n, module._dtype_code())
        if self.gws is None or self.gws.numel() < gbytes:
            self.gws = torch.empty(gbytes, dtype=torch.uint8, device=self.dev)
        b = _hip.GenBwd()
        b.N = n
        b.x, b.x_out, b.g_out, b.up_x = xf.data_ptr(), out.data_ptr(), g_out.data_ptr(), up.data_ptr()
        b.g_upx = gup.data_ptr() if gup is not None else None
        b.drop_scale = ds.data_ptr() if ds is not None else None
        b.workspace, b.grad_workspace, b.grad_workspace_bytes = ws.data_ptr(), self.gws.data_ptr(), gbytes
        off = 0
        for i in range(_hip.G_NUM_WEIGHTS):
            b.wd[i] = module._wd[i].data_ptr()
            b.gw[i] = self.gw_flat.data_ptr() + off * 4
            b.gb[i] = self.gb[i].data_ptr()
            off += self.sizes[i]
        b.g_inc0_w, b.g_inc0_b = self.g_inc_w.data_ptr(), self.g_inc_b.data_ptr()
        b.g_outc_w, b.g_outc_b, b.g_pos_embed = self.g_oc_w.data_ptr(), self.g_oc_b.data_ptr(), self.g_pe.data_ptr()
        b.accumulate = int(accumulate)
        b.prev_workspace = prev_ws.data_ptr() if prev_ws is not None else None
        b.carry_in = carry_in.data_ptr() if carry_in is not None else None
        b.carry_out = carry_out.data_ptr() if carry_out is not None else None
        _hip.check(lib.uncl_gen_backward(C.byref(gwts), C.byref(b), _hip.stream_ptr()), "uncl_gen_backward")

    def unpack(self):
        """packed [tap][Cout][Cin] -> reference parameter layout, keyed by state_dict name"""
        lib = _hip.lib()
        st = _hip.stream_ptr()
        grads = {"inc.conv.conv.weight": self.g_inc_w, "inc.conv.conv.bias": self.g_inc_b,
                 "outc.conv.weight": self.g_oc_w.reshape(1, 32, 1, 1), "outc.conv.bias": self.g_oc_b,
                 "gcn.pos_embed": self.g_pe.t().reshape(1, 256, 12, 12).contiguous()}
        off = 0
        items = (_hip.UnpackItem * len(self.names))()
        flat = torch.empty(sum(self.sizes), dtype=torch.float32, device=self.dev)      # one allocation, one launch
        for i, nm in enumerate(self.names):
            shape, kind = self.spec[nm + ".weight"]
            transposed = kind == "convT"
            k = shape[2]
            cout, cin = (shape[1], shape[0]) if transposed else (shape[0], shape[1])
            dst = flat[off:off + self.sizes[i]].view(shape)
            it = items[i]
            it.packed, it.dst = self.gw_flat.data_ptr() + off * 4, dst.data_ptr()
            it.Cout, it.Cin, it.k, it.transposed, it.flip, it.accumulate = cout, cin, k, int(transposed), \
                (1 if (transposed and k == 3) else 0), 0
            grads[nm + ".weight"] = dst
            grads[nm + ".bias"] = self.gb[i]
            off += self.sizes[i]
        _hip.check(lib.uncl_unpack_conv_wgrads(items, len(self.names), st), "uncl_unpack_conv_wgrads")
        return grads


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        n = x.shape[0]
        xf = x.detach().reshape(n, 256, 256).float().contiguous()
        # a private workspace per call: it must survive until backward (two generator passes are alive in a step)
        module._ws.pop(("train", xf.device), None)
        out, up, _, ws, ds = module._run(xf, need_feat=True, keep_act=True, slot="train", save_preact=True, return_drop=True)
        module._ws.pop(("train", xf.device), None)
        ctx.module = module
        ctx.saved = (xf, out, up, ws, ds)
        ctx.mark_non_differentiable()
        return out, up.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_out, g_upx):
        module = ctx.module
        xf, out, up, ws, ds = ctx.saved
        n = xf.shape[0]
        g_out = torch.zeros_like(out) if g_out is None else g_out.reshape(n, 1, 256, 256).float().contiguous()
        gup = None
        if g_upx is not None:
            gup = g_upx.permute(0, 2, 3, 1).to(_hip.torch_dtype(module._dtype_code())).contiguous()
        gs = _GradSet(module, xf.device)
        gs.run(xf, out, up, ws, ds, g_out, gup)
        grads = gs.unpack()
        return (None, None) + tuple(grads.get(k) for k in ctx.pnames)


class _VideoGeneratorFn(torch.autograd.Function):
    """Clip forward (frames sequential, Unet.py:213-289) and backward through time: frames are visited last to first, the
    head-channel gradients of the eight recurrent hand-offs travel between consecutive frames in a small carry arena and
    the parameter gradients of all frames accumulate in one packed set."""

    @staticmethod
    def forward(ctx, module, x, *params):
        from .generator import gauss_stats
        B, T = x.shape[0], x.shape[1]
        dev = x.device
        frames, outs, feats = [], [], []
        prev_ws = None
        for t in range(T):
            xf = x[:, t].detach().reshape(B, 256, 256).float().contiguous()
            module._ws.pop((("clip", t), dev), None)
            out, up, _, ws, ds = module._run(xf, need_feat=True, prev_ws=prev_ws, keep_act=True, slot=("clip", t),
                                             save_preact=True, return_drop=True)
            module._ws.pop((("clip", t), dev), None)
            feats.append(gauss_stats(up, B, 256, 256, 32).reshape(B, 1, 64, 1, 1))
            outs.append(out.reshape(B, 1, 1, 256, 256))
            frames.append((xf, out, up, ws, ds))
            prev_ws = ws
        ctx.module, ctx.frames = module, frames
        return torch.cat(outs, 1), torch.cat(feats, 1)

    @staticmethod
    def backward(ctx, g_out, g_feats):
        module, frames = ctx.module, ctx.frames
        lib = _hip.lib()
        T = len(frames)
        B = frames[0][0].shape[0]
        dev = frames[0][0].device
        gs = _GradSet(module, dev)
        code = module._dtype_code()
        cbytes = lib.uncl_gen_carry_bytes_dt(B, code)
        carries = [torch.empty(cbytes, dtype=torch.uint8, device=dev) for _ in range(2)] if T > 1 else []
        for t in range(T - 1, -1, -1):
            xf, out, up, ws, ds = frames[t]
            go = (torch.zeros_like(out) if g_out is None else g_out[:, t].reshape(B, 1, 256, 256).float().contiguous())
            gup = None
            if g_feats is not None:
                gst = g_feats[:, t].reshape(B, 2, 32).float().contiguous()
                gup = torch.empty_like(up)
                _hip.check(lib.uncl_gauss_stats_backward(up.data_ptr(), code, gst.data_ptr(), gup.data_ptr(), B, 256, 256, 32, 0,
                                                         _hip.stream_ptr()), "uncl_gauss_stats_backward")
            carry_in = carries[(t + 1) % 2] if t < T - 1 else None      # written by frame t+1
            carry_out = carries[t % 2] if t > 0 else None               # read by frame t-1
            gs.run(xf, out, up, ws, ds, go, gup, accumulate=(t != T - 1), prev_ws=frames[t - 1][3] if t > 0 else None,
                   carry_in=carry_in, carry_out=carry_out)
        grads = gs.unpack()
        # the saved activations stay with the graph node: the reference calls backward twice on it
        # (errG_d.backward(retain_graph=True), then the structural loss; GanTrainer.py:338,461)
        return (None, None) + tuple(grads.get(k) for k in ctx.pnames)


def _warn_bf16_norm(module):
    if module._dtype_code() == _hip.BF16 and getattr(module, "unet_norm", "none") == "instance_norm":
        import warnings
        warnings.warn("uncltmo_amd: training with unet_norm='instance_norm' in bf16 loses the gradient below the per-channel "
                      "mean that the norm's backward subtracts (activation gradients are stored in bf16); use "
                      "compute_dtype='fp32' for this configuration", UserWarning, stacklevel=3)


def generator_image_apply(module, x):
    _warn_bf16_norm(module)
    if module._dtype_code() not in (_hip.BF16, _hip.F32):
        raise NotImplementedError("uncltmo_amd: the HIP backward path is built for compute_dtype 'bf16' (training) and 'fp32' "
                                  "(parity mode); 'fp16' is an inference dtype")
    named = [(k, p) for k, p in module.named_parameters() if p.requires_grad]
    _GeneratorFn_pnames = [k for k, _ in named]

    class _Fn(_GeneratorFn):
        @staticmethod
        def forward(ctx, module_, x_, *params):
            ctx.pnames = _GeneratorFn_pnames
            return _GeneratorFn.forward(ctx, module_, x_, *params)

    return _Fn.apply(module, x, *[p for _, p in named])


def generator_video_apply(module, x):
    _warn_bf16_norm(module)
    if module._dtype_code() not in (_hip.BF16, _hip.F32):
        raise NotImplementedError("uncltmo_amd: the HIP backward path is built for compute_dtype 'bf16' (training) and 'fp32' "
                                  "(parity mode); 'fp16' is an inference dtype")
    named = [(k, p) for k, p in module.named_parameters() if p.requires_grad]
    pnames = [k for k, _ in named]

    class _Fn(_VideoGeneratorFn):
        @staticmethod
        def forward(ctx, module_, x_, *params):
            ctx.pnames = pnames
            return _VideoGeneratorFn.forward(ctx, module_, x_, *params)

    return _Fn.apply(module, x, *[p for _, p in named])
