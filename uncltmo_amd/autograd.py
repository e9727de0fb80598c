"""Autograd bridge of the image generator: forward = uncl_gen_forward with activations kept, backward = ONE call of
uncl_gen_backward (hand-written HIP dgrad / wgrad / element-wise kernels), then the packed weight gradients are
re-laid-out into the reference parameter layout.  bf16 compute, fp32 accumulation and fp32 parameter gradients."""
import ctypes as C

import torch

from . import _hip
from .state_spec import generator_spec


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
        lib = _hip.lib()
        gwts, _keep = module._packed_weights()
        n = xf.shape[0]
        dev = xf.device
        st = _hip.stream_ptr()
        g_out = torch.zeros_like(out) if g_out is None else g_out.reshape(n, 1, 256, 256).float().contiguous()
        gup = None
        if g_upx is not None:
            gup = g_upx.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
        spec = {k: (shape, kind) for k, shape, kind in generator_spec()}
        names = [lib.uncl_gen_layer_name(i).decode() for i in range(_hip.G_NUM_WEIGHTS)]
        sizes = [int(torch.tensor(spec[nm + ".weight"][0]).prod()) for nm in names]
        gw_flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)        # packed, accumulated with atomics
        gb = [torch.empty(spec[nm + ".bias"][0], dtype=torch.float32, device=dev) for nm in names]
        g_inc_w = torch.empty(32, 1, 3, 3, dtype=torch.float32, device=dev)
        g_inc_b = torch.empty(32, dtype=torch.float32, device=dev)
        g_oc_w = torch.empty(32, dtype=torch.float32, device=dev)
        g_oc_b = torch.empty(1, dtype=torch.float32, device=dev)
        g_pe = torch.empty(144, 256, dtype=torch.float32, device=dev)
        gbytes = lib.uncl_gen_backward_workspace_bytes(n)
        gws = torch.empty(gbytes, dtype=torch.uint8, device=dev)
        b = _hip.GenBwd()
        b.N = n
        b.x, b.x_out, b.g_out, b.up_x = xf.data_ptr(), out.data_ptr(), g_out.data_ptr(), up.data_ptr()
        b.g_upx = gup.data_ptr() if gup is not None else None
        b.drop_scale = ds.data_ptr() if ds is not None else None
        b.workspace, b.grad_workspace, b.grad_workspace_bytes = ws.data_ptr(), gws.data_ptr(), gbytes
        off = 0
        for i in range(_hip.G_NUM_WEIGHTS):
            b.wd[i] = module._wd[i].data_ptr()
            b.gw[i] = gw_flat.data_ptr() + off * 4
            b.gb[i] = gb[i].data_ptr()
            off += sizes[i]
        b.g_inc0_w, b.g_inc0_b = g_inc_w.data_ptr(), g_inc_b.data_ptr()
        b.g_outc_w, b.g_outc_b, b.g_pos_embed = g_oc_w.data_ptr(), g_oc_b.data_ptr(), g_pe.data_ptr()
        _hip.check(lib.uncl_gen_backward(C.byref(gwts), C.byref(b), st), "uncl_gen_backward")
        # packed [tap][Cout][Cin] -> reference layout
        grads = {"inc.conv.conv.weight": g_inc_w, "inc.conv.conv.bias": g_inc_b,
                 "outc.conv.weight": g_oc_w.reshape(1, 32, 1, 1), "outc.conv.bias": g_oc_b,
                 "gcn.pos_embed": g_pe.t().reshape(1, 256, 12, 12).contiguous()}
        off = 0
        for i, nm in enumerate(names):
            shape, kind = spec[nm + ".weight"]
            transposed = kind == "convT"
            k = shape[2]
            cout, cin = (shape[1], shape[0]) if transposed else (shape[0], shape[1])
            dst = torch.empty(shape, dtype=torch.float32, device=dev)
            _hip.check(lib.uncl_unpack_conv_wgrad(gw_flat.data_ptr() + off * 4, dst.data_ptr(), cout, cin, k, int(transposed),
                                                  1 if (transposed and k == 3) else 0, 0, st), "uncl_unpack_conv_wgrad")
            grads[nm + ".weight"] = dst
            grads[nm + ".bias"] = gb[i]
            off += sizes[i]
        pnames = ctx.pnames
        return (None, None) + tuple(grads.get(k) for k in pnames)


def generator_image_apply(module, x):
    if module._dtype_code() != _hip.BF16:
        raise NotImplementedError("uncltmo_amd: the HIP backward path is built for compute_dtype='bf16' (fp32 is the "
                                  "inference parity mode)")
    named = [(k, p) for k, p in module.named_parameters() if p.requires_grad]
    _GeneratorFn_pnames = [k for k, _ in named]

    class _Fn(_GeneratorFn):
        @staticmethod
        def forward(ctx, module_, x_, *params):
            ctx.pnames = _GeneratorFn_pnames
            return _GeneratorFn.forward(ctx, module_, x_, *params)

    return _Fn.apply(module, x, *[p for _, p in named])
