"""Generators of UnCLTMO, MI355X-native: same constructor / forward() signatures and state_dict keys as the
reference's `models/unet_multi_filters/Unet_singleFrame.py:UNet` (image) and `Unet.py:UNet` (video), with
the arithmetic done by hand-written gfx950 kernels behind the C ABI of include/uncltmo_hip.h.

The module only owns parameters (fp32 masters, reference layout); `forward` packs them once per parameter
version into the kernels' layout and enqueues the whole network with one C call (uncl_gen_forward).
There is no eager / CPU fallback: on a host tensor or without the built library, forward raises.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _hip, params
from .state_spec import generator_spec


def sincos_relative_pos(embed_dim=256, grid=12):
    """Fixed buffer `relative_pos` = -(2 P P^T / D) for the 2-D sin/cos table P (columns first).
    Reference: gcn_lib/pos_embed.py:21-83, gcn_lib/torch_vertex.py:203-209 (the bicubic resize to (n, n) is
    the identity at reduce ratio 1)."""
    quarter = embed_dim // 4
    omega = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / quarter)
    gy, gx = np.meshgrid(np.arange(grid, dtype=np.float64), np.arange(grid, dtype=np.float64), indexing="ij")

    def enc(pos):
        ang = pos.reshape(-1, 1) * omega[None, :]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    table = np.concatenate([enc(gx), enc(gy)], axis=1)
    rel = 2.0 * (table @ table.T) / table.shape[1]
    return -torch.from_numpy(rel.astype(np.float32)).unsqueeze(0)


def _attach(root, dotted, tensor, as_param):
    """Create plain container modules along a dotted state_dict key and register the leaf tensor."""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    if as_param is None:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=as_param))


# skip-concat layers whose data gradient carries the skip operator's backward in its epilogue (uncl_conv3x3_dgrad_ssr)
SSR_FUSED_LAYERS = ("up_path.2.conv.conv", "up_path.3.conv.conv")
_ACT = {"relu": _hip.ACT_RELU, "leakyrelu": _hip.ACT_LRELU}
_LAST = {"sigmoid": _hip.ACT_SIGMOID, "tanh": _hip.ACT_TANH, "msig": _hip.ACT_MSIG, "none": _hip.ACT_NONE}


class _GeneratorBase(nn.Module):
    """Parameter container + packing cache shared by the image and video generators."""

    def __init__(self, n_channels, output_dim, last_layer, depth, layer_factor, con_operator, filters, bilinear,
                 network, dilation, to_crop, unet_norm, stretch_g, activation, doubleConvTranspose,
                 padding_mode, convtranspose_kernel, up_mode=True, recurrent_ch_ratio=1 / 32,
                 compute_dtype="fp32", chunk=0):
        super().__init__()
        # The 12x12 learned pos_embed hard-wires the topology (Unet_singleFrame.py:66,94): only the published
        # configuration can produce a 12x12 bottleneck from a 256x256 input.
        unsupported = []
        if network != params.unet_network:
            assert 0, "Unsupported network request: {}".format(network)
        if activation not in _ACT:
            assert 0, "Unsupported activation: {%s}" % (activation)
        # skip operators that are SUB-SETS of the published four-member concatenation [x2, x1, x2^2, sqrt(x2 + eps)] run on its
        # kernels with zero weights for the members they leave out (unet_parts.py:311-332; exact: the left-out products are 0);
        # 'gamma' (a 0.02 power) and the manual-d form have members of their own and stay refused
        if con_operator not in (params.square_and_square_root, params.original_unet, params.square, params.square_root):
            unsupported.append("con_operator=%s" % con_operator)
        elif layer_factor != params.get_layer_factor(con_operator):
            unsupported.append("layer_factor=%s does not match con_operator=%s (the reference's first decoder convolution would "
                               "refuse the concatenation)" % (layer_factor, con_operator))
        # ... exact only while every product with a zero weight is finite: the loaders still evaluate sqrt(x2 + 1e-8) and x2^2 for the
        # members an operator leaves out, and with a leaky ReLU x2 < 0 makes the square root a NaN (NaN * 0 = NaN in the matrix
        # cores, forward and backward) where the reference has no such member and stays finite
        elif con_operator != params.square_and_square_root and activation == "leakyrelu":
            unsupported.append("con_operator=%s with activation=leakyrelu (the sub-set operators run on the four-member kernels, "
                               "whose square-root member is NaN for negative skip values)" % con_operator)
        if depth != 4 or filters != 32:
            unsupported.append("depth/filters=%s/%s" % (depth, filters))
        # bilinear=1: nn.Upsample(scale_factor=2) [nearest] + Conv2d 1x1 (unet_parts.py:256-259) IS a stride-2 2x2 transposed
        # convolution whose four taps all hold the 1x1 weight: same kernels, the weight is replicated when it is packed
        # up_mode=1: the parameter-free zero-insertion upsampling (unet_parts.py:284-288) is the same transposed convolution with an
        # identity on tap (0, 0), zeros on the other three and no bias
        if not doubleConvTranspose or convtranspose_kernel != 2:
            unsupported.append("decoder must be doubleConvTranspose=1, convtranspose_kernel=2")
        if n_channels != 1 or output_dim != 1:
            unsupported.append("n_channels/output_dim must be 1")
        if unet_norm not in ("none", None, "instance_norm", "batch_norm"):
            unsupported.append("unet_norm=%s (HIP path covers 'none', 'instance_norm' and 'batch_norm')" % unet_norm)
        if last_layer not in _LAST:
            unsupported.append("last_layer=%s" % last_layer)
        # stretch_g: the reference builds Blocks.BatchMaxNormalization / MinMaxNormalization into `self.stretch` (parameter-free,
        # Unet_singleFrame.py:169-175, Unet.py:203-209) and its forward never calls it: every accepted value computes what 'none'
        # computes, with the same state_dict.  An unknown name fails where the reference's dictionary lookup does.
        if stretch_g not in ("none", None, "batchMax", "instanceMinMax"):
            raise KeyError(stretch_g)
        self.stretch_g = "none" if stretch_g is None else stretch_g
        if unsupported:
            raise NotImplementedError("generator configuration outside the published topology: " + "; ".join(unsupported))
        self.to_crop = to_crop
        self.unet_norm = unet_norm if unet_norm not in (None,) else "none"
        self.con_operator = con_operator
        self.layer_factor = int(layer_factor)
        self.bilinear = int(bool(bilinear))
        self.up_mode = int(bool(up_mode))
        self.filters = filters
        self.network = network
        self.depth = depth
        self.activation = activation
        self.last_layer = last_layer
        self.recurrent_ch_ratio = recurrent_ch_ratio
        self.compute_dtype = compute_dtype
        self.chunk = chunk
        self.drop_path_prob = 0.05          # dpr = linspace(0.05, 0.1, 1)[0] (Unet_singleFrame.py:62)
        self.forced_drop_keep = None        # optional (2, N) 0/1 keep flags for deterministic train-mode runs
        for key, shape, kind in self._own_spec():
            if kind == "buffer":
                _attach(self, key, sincos_relative_pos(), False)
            elif kind == "embed":
                _attach(self, key, torch.zeros(shape), True)
            elif kind in ("bn_mean", "bn_var"):           # nn.BatchNorm2d buffers (unet_parts.py:20-21)
                _attach(self, key, torch.zeros(shape) if kind == "bn_mean" else torch.ones(shape), None)
            elif kind == "bn_count":
                _attach(self, key, torch.zeros((), dtype=torch.long), None)
            elif kind in ("bn_weight", "bn_bias"):
                _attach(self, key, torch.ones(shape) if kind == "bn_weight" else torch.zeros(shape), True)
            else:
                _attach(self, key, torch.empty(shape), True)
        self.reset_parameters()
        self._pack_key = None
        self._packed = None
        self._ws = {}

    def _warn_detached(self, x):
        """The reference back-propagates through an eval-mode BatchNorm; the folded inference path here cannot.  A caller with grad
        enabled and something that requires grad gets outputs detached from the graph: say so (once per module) instead of handing
        back zero / missing gradients silently."""
        if torch.is_grad_enabled() and not getattr(self, "_warned_detached", False) and \
                (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            import warnings
            warnings.warn("uncltmo_amd: the batch_norm generator in eval() mode runs the folded inference path, which has no backward "
                          "pass: its outputs are detached from the autograd graph (call it under torch.no_grad(), or use train() mode "
                          "to back-propagate)", RuntimeWarning, stacklevel=3)
            self._warned_detached = True

    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad + the data-parallel reducer's pending passes: gradients of a backward pass whose optimiser step
        was never taken must not be added to the next step's (distributed.GradReducer keeps them until step())."""
        red = self.__dict__.get("_grad_reducer")
        if red is not None:
            red.reset()
        return super().zero_grad(set_to_none=set_to_none)

    # --- initialisation: what `create_G_net*` + `set_parallel_net(use_xaviar=True)` leave behind
    def _own_spec(self):
        """state_dict layout of THIS configuration (skip operator's member count, bilinear up path)"""
        return generator_spec(self.filters, self.layer_factor, self.unet_norm, self.bilinear, self.up_mode)

    def _is_variant(self):
        return self.layer_factor != 4 or self.bilinear != 0 or self.up_mode != 0

    def _published_state(self, sd):
        """name -> tensor in the PUBLISHED layout the kernels are packed from: the skip-concat convolutions padded to four members
        with zero weights where this configuration's operator has none, the 1x1 weight of a bilinear `up` replicated over the four
        taps of the 2x2 transposed convolution it equals.  Identity for the published configuration."""
        if not self._is_variant():
            return sd
        out = dict(sd)
        for i in range(4):
            p = "up_path.%d" % i
            if self.layer_factor != 4:
                w = sd[p + ".conv.conv.weight"].detach().float()            # (m C, Cout, 3, 3), members [x2, x1, (x2^2 | sqrt)]
                c = w.shape[0] // self.layer_factor
                zero = torch.zeros_like(w[:c])
                third = w[2 * c:3 * c] if self.layer_factor == 3 else zero
                sq, rt = (third, zero) if self.con_operator == params.square else (zero, third)
                out[p + ".conv.conv.weight"] = torch.cat([w[:2 * c], sq, rt], 0)
            if self.up_mode:
                c = sd[p + ".conv.conv.weight"].shape[0] // self.layer_factor       # channels entering `up`
                w = torch.zeros(c, c, 2, 2, dtype=torch.float32, device=sd[p + ".conv.conv.weight"].device)
                w[:, :, 0, 0] = torch.eye(c, device=w.device)
                out[p + ".up.weight"], out[p + ".up.bias"] = w, torch.zeros(c, dtype=torch.float32, device=w.device)
            elif self.bilinear:
                v = sd[p + ".up.1.weight"].detach().float()                 # Conv2d (Cout, Cin, 1, 1)
                co, ci = v.shape[0], v.shape[1]
                out[p + ".up.weight"] = v.reshape(co, ci).t().reshape(ci, co, 1, 1).expand(ci, co, 2, 2).contiguous()
                out[p + ".up.bias"] = sd[p + ".up.1.bias"]
        return out

    def _variant_grads(self, g):
        """published-layout gradients (autograd._GradSet.grads) -> this configuration's parameters: the members the operator has,
        the sum over the four taps for a bilinear `up`"""
        if not self._is_variant():
            return g
        g = dict(g)
        for i in range(4):
            p = "up_path.%d" % i
            if self.layer_factor != 4:
                w = g[p + ".conv.conv.weight"]
                c = w.shape[0] // 4
                if self.layer_factor == 2:
                    g[p + ".conv.conv.weight"] = w[:2 * c]
                elif self.con_operator == params.square:
                    g[p + ".conv.conv.weight"] = w[:3 * c]
                else:
                    g[p + ".conv.conv.weight"] = torch.cat([w[:2 * c], w[3 * c:]], 0)
            if self.up_mode:
                g.pop(p + ".up.weight", None)
                g.pop(p + ".up.bias", None)
            elif self.bilinear:
                w = g.pop(p + ".up.weight")                                 # (Cin, Cout, 2, 2)
                g[p + ".up.1.weight"] = w.sum(dim=(2, 3)).t().reshape(w.shape[1], w.shape[0], 1, 1)
                g[p + ".up.1.bias"] = g.pop(p + ".up.bias")
        return g

    def reset_parameters(self):
        """Conv2d: xavier_normal(gain sqrt 2), zero bias (model_save_util.py:41-47); the class-name test there does
        not match ConvTranspose2d, which keeps PyTorch's default init; the graph block re-inits its convs with
        kaiming_normal / zero bias first (Unet_singleFrame.py:83-90) and xavier then overrides them."""
        own = self._own_spec()
        for key, shape, kind in own:
            t = dict(self.named_parameters()).get(key)
            if t is None:
                continue
            with torch.no_grad():
                if kind == "conv":
                    fan_in = shape[1] * shape[2] * shape[3]
                    fan_out = shape[0] * shape[2] * shape[3]
                    t.normal_(0.0, math.sqrt(2.0) * math.sqrt(2.0 / (fan_in + fan_out)))
                elif kind == "convT":
                    fan_in = shape[1] * shape[2] * shape[3]     # torch computes fan_in from dim 1 for any weight
                    bound = 1.0 / math.sqrt(fan_in)             # kaiming_uniform(a=sqrt 5) == U(-1/sqrt(fan_in), ..)
                    t.uniform_(-bound, bound)
                elif kind == "bias":
                    wkey = key[:-4] + "weight"
                    wshape = dict((k, s) for k, s, _ in own)[wkey]
                    wkind = dict((k, kd) for k, _, kd in own)[wkey]
                    if wkind == "conv":
                        t.zero_()
                    else:
                        bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
                        t.uniform_(-bound, bound)

    # --- packing
    def _dtype_code(self):
        return _hip.dtype_code(self.compute_dtype)

    def _packed_weights(self):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        code = self._dtype_code()
        # the fused skip backward (uncl_conv3x3_dgrad_ssr) exists in the producer / consumer kernel only and wants its weights in the
        # interleaved cout order: packed that way only while that structure is on (uncl_conv3x3_set_pc / UNCL_PC, UNCL_SSR_FUSED), and
        # both switches are part of the key -- toggling them re-packs instead of leaving a pack the other path cannot read
        ssr_on = code == _hip.BF16 and os.environ.get("UNCL_SSR_FUSED", "1") != "0" and _hip.lib().uncl_conv3x3_set_pc(-1) != 0
        key = (code, self.training and self.unet_norm == "batch_norm", ssr_on) + \
            tuple((k, v.data_ptr(), v._version, getattr(v, "_uncl_epoch", 0)) for k, v in sd.items())
        if key == self._pack_key:
            return self._packed
        sd = self._published_state(sd)
        lib = _hip.lib()
        dev = sd["outc.conv.weight"].device
        if dev.type != "cuda":
            raise _hip.HipError("generator parameters live on %s; move the module to the MI355X (.cuda()) first" % dev)
        tdt = _hip.torch_dtype(code)
        spec = {k: (shape, kind) for k, shape, kind in generator_spec()}
        keep = []          # tensors that must stay alive as long as the struct
        gw = _hip.GenWeights()
        gw.dtype = code
        st = _hip.stream_ptr()

        def f32(name):
            t = sd[name].detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()

        # unet_norm='batch_norm', eval mode: y = gamma (conv(x) + b - mean) / sqrt(var + eps) + beta is a convolution again -- the
        # running statistics are folded into weight and bias here, once per parameter version, and the forward runs the fused
        # conv + activation kernels of the norm-free topology (unet_parts.py:20-21, 34-35, 57-75; nn.BatchNorm2d eps 1e-5)
        fold = {}
        if self.unet_norm == "batch_norm" and not self.training:
            from .state_spec import batch_norm_layers
            for cname, nname in batch_norm_layers():
                scale = sd[nname + ".weight"].detach().double() / torch.sqrt(sd[nname + ".running_var"].detach().double() + 1e-5)
                fold[cname] = (scale, sd[nname + ".running_mean"].detach().double(), sd[nname + ".bias"].detach().double())

        def folded(name):
            """(weight, bias) of convolution `name` as fp32 tensors, BatchNorm folded in when there is one behind it"""
            w, b = sd[name + ".weight"].detach().float(), sd[name + ".bias"].detach().float()
            if name not in fold:
                return w, b
            scale, mean, beta = fold[name]
            transposed = dict((k, kd) for k, _, kd in generator_spec())[name + ".weight"] == "convT"
            shape = (1, -1, 1, 1) if transposed else (-1, 1, 1, 1)
            return (w.double() * scale.reshape(shape)).float(), ((b.double() - mean) * scale + beta).float()

        w0, b0 = folded("inc.conv.conv")
        w0, b0 = w0.contiguous(), b0.contiguous()
        keep += [w0, b0]
        gw.inc0_w, gw.inc0_b = w0.data_ptr(), b0.data_ptr()
        # every re-layout of the step goes into one flat buffer through one batched launch (uncl_pack_conv_weights):
        # jobs = (source fp32 tensor, offset in elements, Cout, Cin, k, transposed, flip)
        jobs, fwd_off, wd_off = [], [], []
        total = 0
        ssr_fused = 0

        def job(src, numel, co, ci, kk, tr, fl, order=0):
            nonlocal total
            src = src.contiguous()
            keep.append(src)
            jobs.append((src, total, co, ci, kk, tr, fl, order))
            off = total
            total += (numel + 127) & ~127          # 256-byte aligned slices
            return off

        for i in range(_hip.G_NUM_WEIGHTS):
            name = lib.uncl_gen_layer_name(i).decode()
            shape, kind = spec[name + ".weight"]
            transposed = kind == "convT"
            k = shape[2]
            cout, cin = (shape[1], shape[0]) if transposed else (shape[0], shape[1])
            src, bsrc = folded(name)
            fwd_off.append(job(src, src.numel(), cout, cin, k, int(transposed), 1 if (transposed and k == 3) else 0))
            bsrc = bsrc.contiguous()
            keep.append(bsrc)
            gw.b[i] = bsrc.data_ptr()
        # weights re-packed for the data-gradient convolutions (the two training dtypes: bf16, and fp32 = parity mode)
        if code in (_hip.BF16, _hip.F32):
            for i in range(_hip.G_NUM_WEIGHTS):
                name = lib.uncl_gen_layer_name(i).decode()
                shape, kind = spec[name + ".weight"]
                src = sd[name + ".weight"].detach().float().contiguous()
                k = shape[2]
                if kind == "convT" and k == 3:      # dgrad = valid conv with the weight read as a Conv2d weight
                    # the skip-concat layers of the last two decoder stages: cout order interleaved for the data-gradient launch
                    # whose epilogue is the skip operator's backward (uncl_conv3x3_dgrad_ssr; the up-sampled map has the skip's
                    # extent there -- up_path.1 has the 56 -> 57 replicate pad, up_path.0 runs on flat tiles)
                    order = 1 if (ssr_on and name in SSR_FUSED_LAYERS) else 0
                    if order:
                        ssr_fused |= 1 << int(name.split(".")[1])
                    wd_off.append(job(src, src.numel(), shape[0], shape[1], 3, 0, 0, order))
                elif kind == "convT":                # 2x2 stride 2: [4][Cin][Cout]
                    wd_off.append(job(src, src.numel(), shape[0], shape[1], 2, 0, 0))
                elif name.endswith("graph_conv.gconv.nn.0"):   # grouped 1x1: transpose every 128x128 block
                    g, blk = 4, shape[1]
                    offs = [job(src[j * blk:(j + 1) * blk], blk * blk, blk, blk, 1, 1, 0) for j in range(g)]
                    assert all(offs[j] == offs[0] + j * blk * blk for j in range(g))    # blk*blk is a multiple of 128
                    wd_off.append(offs[0])
                else:                                # Conv2d (Cout,Cin,k,k): read as a ConvTranspose2d weight, flipped for 3x3
                    wd_off.append(job(src, src.numel(), shape[1], shape[0], k, 1, 1 if k == 3 else 0))
        # the learned position table (1,256,12,12) as node-major (144,256) rows in the compute dtype: one more item of the batched
        # re-layout (a 1x1 "transposed" weight is exactly a 2-D transpose) instead of a transpose copy and a dtype copy per step
        pe_off = job(sd["gcn.pos_embed"].detach().reshape(256, 144), 256 * 144, 144, 256, 1, 1, 0)
        flat = torch.empty(total, dtype=tdt, device=dev)
        keep.append(flat)
        esz = flat.element_size()
        items = (_hip.PackItem * len(jobs))()
        for it, (src, off, co, ci, kk, tr, fl, order) in zip(items, jobs):
            it.src, it.dst = src.data_ptr(), flat.data_ptr() + off * esz
            it.Cout, it.Cin, it.k, it.transposed, it.flip, it.cout_order = co, ci, kk, tr, fl, order
        _hip.check(lib.uncl_pack_conv_weights(items, len(jobs), code, st), "uncl_pack_conv_weights")
        for i in range(_hip.G_NUM_WEIGHTS):
            gw.w[i] = flat.data_ptr() + fwd_off[i] * esz
        wd = [flat[o:] for o in wd_off]            # views: .data_ptr() is the packed data-gradient weight
        self._wd = wd
        self._ssr_fused = ssr_fused                # decoder stages whose skip-concat data-gradient weights are interleaved
        gw.pos_embed = flat.data_ptr() + pe_off * esz                                    # (144,256) NHWC
        gw.relative_pos = f32("gcn.module.0.0.relative_pos")
        gw.outc_w, gw.outc_b = f32("outc.conv.weight"), f32("outc.conv.bias")
        gw.act = _ACT[self.activation]
        # 1: InstanceNorm kernels; 2: BatchNorm in TRAINING mode (batch statistics, uncl_gen_set_bn announces the layers' tensors);
        # BatchNorm in eval mode is folded into the weights above and runs as the norm-free topology
        gw.norm = 1 if self.unet_norm == "instance_norm" else (2 if (self.unet_norm == "batch_norm" and self.training) else 0)
        gw.last_act = _LAST[self.last_layer]
        self._packed = (gw, keep)
        self._pack_key = key
        return self._packed

    def _workspace(self, n, chunk, keep_act, dev, slot=0):
        lib = _hip.lib()
        code = self._dtype_code()
        nbytes = lib.uncl_gen_workspace_bytes_ex(n, chunk, code, int(keep_act), 1 if (self.unet_norm == "instance_norm" or self._bn_train()) else 0)
        k = (slot, dev)
        ws = self._ws.get(k)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws[k] = ws
        return ws, nbytes

    def _bn_train(self):
        return self.unet_norm == "batch_norm" and self.training

    def _bn_arrays(self, grads=None):
        """ctypes pointer arrays for uncl_gen_set_bn: the eighteen BatchNorm2d layers in state_dict order (= the library's order);
        `grads` = (list of 18 weight-gradient tensors, list of 18 bias-gradient tensors) or None"""
        from .state_spec import batch_norm_layers
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        names = [q for _, q in batch_norm_layers()]
        keep = []

        def arr(key):
            ts = []
            for q in names:
                t = sd[q + key]
                if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                    raise _hip.HipError("batch_norm tensors must be contiguous fp32 device tensors (%s%s)" % (q, key))
                ts.append(t)
            keep.extend(ts)
            return (C.c_void_p * 18)(*[t.data_ptr() for t in ts])

        a = [arr(".weight"), arr(".bias"), arr(".running_mean"), arr(".running_var")]
        if grads is not None:
            keep.extend(grads[0] + grads[1])
            a += [(C.c_void_p * 18)(*[t.data_ptr() for t in grads[0]]), (C.c_void_p * 18)(*[t.data_ptr() for t in grads[1]])]
        else:
            a += [None, None]
        return a, keep

    def _drop_scale(self, n, dev):
        """(2, N) multipliers keep/keep_prob of the two DropPath sites, or None in eval mode."""
        if not self.training or self.drop_path_prob == 0.0:
            return None
        keep_prob = 1.0 - self.drop_path_prob
        if self.forced_drop_keep is not None:
            keep = torch.as_tensor(self.forced_drop_keep, dtype=torch.float32, device=dev).reshape(2, n)
        else:
            keep = torch.empty(2, n, device=dev).bernoulli_(keep_prob)
        return (keep / keep_prob).contiguous()

    def _drop_scales(self, T, n, dev):
        """`_drop_scale` for the T frame calls of a clip: one draw of (T, 2, N) instead of T draws of (2, N) (the reference's
        DropPath draws per call, timm/layers/drop.py; the flags of different frames are independent either way).  A list of T
        (2, N) tensors, or of T `None`s in eval mode."""
        if not self.training or self.drop_path_prob == 0.0:
            return [None] * T
        if self.forced_drop_keep is not None:
            return [self._drop_scale(n, dev) for _ in range(T)]
        keep_prob = 1.0 - self.drop_path_prob
        allk = torch.empty(T, 2, n, device=dev).bernoulli_(keep_prob) / keep_prob
        return [allk[t] for t in range(T)]

    def _run(self, x_flat, need_feat, prev_ws=None, keep_act=False, slot=0, want_knn=False, save_preact=False,
             return_drop=False, clip=None, tiles_in_place=None, drop="draw", out=None):
        """x_flat: (N,256,256) fp32 on the GPU.  Returns (out (N,1,256,256) fp32, up_x NHWC or None, knn or None, ws).
        clip = (T, t): frame t of a T-frame clip in ONE workspace laid out for T * N samples (uncl_gen_run.clip_T; the previous
        frame is that workspace's slice t - 1, prev_ws stays None)."""
        lib = _hip.lib()
        gw, _keep = self._packed_weights()
        n = x_flat.shape[0]
        if tiles_in_place is not None:
            # x_flat is a stack of whole frames (F, H, W); tiles_in_place = device int32 offsets of the 256 x 256 tiles inside it
            n = int(tiles_in_place.numel())
        dev = x_flat.device
        chunk = self.chunk if self.chunk and self.chunk > 0 else 0
        bn_train = self._bn_train()
        if bn_train:
            # batch statistics (unet_parts.py:72-73 in training mode): the whole batch in one chunk with its normalised
            # pre-activations kept, also under no_grad -- the reference's module updates its running statistics there too
            keep_act, chunk = True, 0
            arrs, _bnkeep = self._bn_arrays()
            _hip.check(lib.uncl_gen_set_bn(arrs[0], arrs[1], arrs[2], arrs[3], 0.1, None, None), "uncl_gen_set_bn")
        if clip is not None:
            if prev_ws is not None or not keep_act or bn_train or self.unet_norm != "none":
                raise _hip.HipError("clip layout: keep_act, no prev_ws (implied), unet_norm 'none'")
            chunk = 0
        ws, nbytes = self._workspace(n * clip[0] if clip is not None else n, chunk, keep_act, dev, slot)
        if out is None:
            out = torch.empty(n, 1, 256, 256, dtype=torch.float32, device=dev)
        elif out.shape != (n, 1, 256, 256) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise _hip.HipError("generator: `out` must be a contiguous fp32 (N,1,256,256) tensor on the input's device")
        up = torch.empty(n, 256, 256, 32, dtype=_hip.torch_dtype(gw.dtype), device=dev) if need_feat else None
        knn = torch.empty(n, 144, 9, dtype=torch.int32, device=dev) if want_knn else None
        # drop: the caller's (2, N) DropPath multipliers (a clip draws its frames' flags at once: _drop_scales), None for none
        ds = self._drop_scale(n, dev) if isinstance(drop, str) else drop
        run = _hip.GenRun()
        run.N, run.chunk, run.keep_activations = n, chunk, int(keep_act)
        run.x, run.out = x_flat.data_ptr(), out.data_ptr()
        run.up_x = up.data_ptr() if up is not None else None
        run.knn_idx = knn.data_ptr() if knn is not None else None
        run.drop_scale = ds.data_ptr() if ds is not None else None
        run.workspace, run.workspace_bytes = ws.data_ptr(), nbytes
        run.prev_workspace = prev_ws.data_ptr() if prev_ws is not None else None
        if tiles_in_place is not None:
            run.x_tile_off, run.x_pitch, run.x_rows = tiles_in_place.data_ptr(), int(x_flat.shape[2]), int(x_flat.shape[0] * x_flat.shape[1])
        run.save_preact = int(save_preact)
        if clip is not None:
            run.clip_T, run.clip_t = int(clip[0]), int(clip[1])
        _hip.check(lib.uncl_gen_forward(C.byref(gw), C.byref(run), _hip.stream_ptr()), "uncl_gen_forward")
        if bn_train:
            from .state_spec import batch_norm_layers
            bufs = dict(self.named_buffers())
            for _, q in batch_norm_layers():
                bufs[q + ".num_batches_tracked"].add_(1)
        if return_drop:
            return out, up, knn, ws, ds
        return out, up, knn, ws

    @staticmethod
    def _check_input(x, hdim):
        if x.shape[hdim] != params.input_size or x.shape[hdim + 1] != params.input_size:
            # the reference fails with a RuntimeError at `inputs + self.pos_embed` (Unet_singleFrame.py:94)
            raise ValueError("the generator only accepts %dx%d inputs (12x12 pos_embed); got %s. Use "
                             "uncltmo_amd.tiler.test_big_size_image2 for larger frames."
                             % (params.input_size, params.input_size, tuple(x.shape)))
        if not x.is_cuda:
            raise _hip.HipError("generator input is on %s; the HIP path needs it on the MI355X" % x.device)

    @staticmethod
    def _crop(x_out, diffY, diffX):
        """utils/data_loader_util.py:165-172 (centre crop)."""
        h, w = x_out.shape[-2], x_out.shape[-1]
        th, tw = h - diffY, w - diffX
        i, j = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return x_out[..., i:i + th, j:j + tw]


class UNet(_GeneratorBase):
    """Image generator.  forward(x[N,1,256,256]) -> (x_out[N,1,256,256] fp32, up_x[N,32,256,256])
    (reference: Unet_singleFrame.py:177-213).  `up_x` is returned as a channels-last view in the compute dtype."""

    def forward(self, x, apply_crop=True, diffY=0, diffX=0):
        self._check_input(x, 2)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self._needs_autograd(x):
            from .autograd import generator_image_apply
            x_out, up_x = generator_image_apply(self, x)
        else:
            xf = x.detach().reshape(-1, 256, 256).float().contiguous()
            x_out, up, _, _ = self._run(xf, need_feat=True)
            up_x = up.permute(0, 3, 1, 2)
        if apply_crop and self.to_crop:
            x_out = self._crop(x_out, diffY, diffX)
        return x_out, up_x

    @torch.no_grad()
    def forward_detached(self, x, apply_crop=True, diffY=0, diffX=0):
        """forward() for a caller that wants x_out only and no graph -- the discriminator step's `fake` (GanTrainerImg.py:206-211
        builds the generator's graph and detaches it): same arithmetic in the module's current mode (DropPath draws in train()),
        but the 32-channel feature map is neither written nor returned, and the last layer takes the one-channel form."""
        self._check_input(x, 2)
        x_out, _, _, _ = self._run(x.detach().reshape(-1, 256, 256).float().contiguous(), need_feat=False)
        if apply_crop and self.to_crop:
            x_out = self._crop(x_out, diffY, diffX)
        return x_out

    def _needs_autograd(self, x):
        # batch_norm in EVAL mode is an inference configuration: the folded weights have no backward pass of their own
        if self.unet_norm == "batch_norm" and not self.training:
            self._warn_detached(x)
            return False
        return True

    @torch.no_grad()
    def infer_frames(self, frames):
        """The tiler's gather folded into the first layer's loader: frames (F,H,W) fp32 on the device -> the tone-mapped 256 x 256 overlap
        tiles (F*T,1,256,256) in uncl_tile_gather's order, read in place (model_save_util.py:409-486's crops are never copied out).
        None where this configuration's first layer is not rebuilt in the second layer's loader (fp32, a norm between them): the
        caller gathers then.  OFF unless UNCL_TILES_IN_PLACE=1: measured at 8 x 1024^2, the first layer's launch takes 0.416 instead
        of 0.351 ms when its 16-row patches come from 4 KB-pitch frame rows instead of a cache-warm 256 KB tile (the gather it
        saves is 0.016 ms), the whole forward is a tie at best."""
        gw, _ = self._packed_weights()
        if self._dtype_code() not in (_hip.BF16, _hip.F16) or gw.norm != 0 or os.environ.get("UNCL_TILES_IN_PLACE", "0") != "1":
            return None
        F, H, W = frames.shape
        key = (F, H, W, frames.device)
        cache = self.__dict__.setdefault("_tile_off_cache", {})
        off = cache.get(key)
        if off is None:
            lib = _hip.lib()
            T = lib.uncl_tile_count(int(H), int(W))
            if T <= 0:
                raise ValueError("tiler needs H > 256 and W > 256 (got %dx%d)" % (H, W))
            host = (C.c_int32 * (F * T))()
            if lib.uncl_tile_offsets(int(F), int(H), int(W), host) != 0:
                return None                                  # (more than 2^31 pixels in the stack: gather instead)
            off = cache[key] = torch.tensor(list(host), dtype=torch.int32).to(frames.device)
        out, _, _, _ = self._run(frames.float().contiguous(), need_feat=False, tiles_in_place=off)
        return out

    @torch.no_grad()
    def infer(self, x, want_knn=False):
        """Inference entry used by the tiler: (N,1,256,256) -> (N,1,256,256); skips writing up_x to HBM."""
        self._check_input(x, 2)
        xf = x.reshape(-1, 256, 256).float().contiguous()
        out, _, knn, _ = self._run(xf, need_feat=False, want_knn=want_knn)
        return (out, knn) if want_knn else out


def _batch_major(out_all):
    """(T,B,...) frame-major outputs -> a fresh contiguous (B,T,...) tensor (never an alias of the frames' own buffers, which a
    backward pass still reads): one transposing copy."""
    v = out_all.transpose(0, 1)
    return v.clone() if v.is_contiguous() else v.contiguous()


def gauss_stats(x_nhwc, n, h, w, c):
    """[mean(x), mean(Gaussian local variance)] per (sample, channel): fp32 (n, 2, c).  x_nhwc is a contiguous
    NHWC tensor (or an (n,h,w) fp32 image when c == 1).  Reference: Unet.py:112-123,274-278."""
    lib = _hip.lib()
    code = _hip.BF16 if x_nhwc.dtype == torch.bfloat16 else _hip.F32
    ws = torch.empty(lib.uncl_gauss_stats_workspace_bytes(n, h, c), dtype=torch.uint8, device=x_nhwc.device)
    out = torch.empty(n, 2, c, dtype=torch.float32, device=x_nhwc.device)
    _hip.check(lib.uncl_gauss_stats(_hip.ptr(x_nhwc), code, out.data_ptr(), n, h, w, c, ws.data_ptr(), _hip.stream_ptr()),
               "uncl_gauss_stats")
    return out


class UNetVideo(_GeneratorBase):
    """Video generator (reference class `UNet` of models/unet_multi_filters/Unet.py:136-289).
    forward(x[B,T,1,256,256]) -> (frames[B,T,1,256,256] fp32, feats[B,T,64,1,1] fp32).

    Frames of a clip are sequential: from the second frame on, the first C/32 channels entering every down / up
    stage are read from the previous frame's activations (kept in that frame's workspace), Unet.py:244,270.
    feats = [mean(up_x), mean(Gaussian local variance of up_x)] per channel (Unet.py:274-278)."""

    @torch.no_grad()
    def forward_detached(self, x, apply_crop=True, diffY=0, diffX=0):
        """forward() for a caller that wants the frames only and no graph -- the discriminator step's `fake` clip (GanTrainer.py:206-
        211): the same frame loop in the module's current mode (recurrent hand-off through the kept workspaces, DropPath draws in
        train()), but neither the 32-channel feature map nor its per-frame Gaussian statistics are computed, and the last layer
        takes the one-channel form.  Returns frames (B,T,1,H,W)."""
        if x.dim() != 5:
            raise ValueError("video generator expects (B,T,1,H,W)")
        self._check_input(x, 3)
        B, T = x.shape[0], x.shape[1]
        xs = x.detach().reshape(B, T, 256, 256).transpose(0, 1).float().contiguous()
        # the frames' outputs one behind the other (frame-major), handed back with ONE transposing copy instead of a T-way cat
        out_all = torch.empty(T, B, 1, 256, 256, dtype=torch.float32, device=x.device)
        drops = self._drop_scales(T, B, x.device)
        prev_ws = None
        for t in range(T):
            _, _, _, ws = self._run(xs[t], need_feat=False, prev_ws=prev_ws, keep_act=True, slot=t, drop=drops[t], out=out_all[t])
            prev_ws = ws
        x_out = _batch_major(out_all)
        if apply_crop and self.to_crop:
            x_out = self._crop(x_out, diffY, diffX)
        return x_out

    def forward(self, x, apply_crop=True, diffY=0, diffX=0):
        if x.dim() != 5:
            raise ValueError("video generator expects (B,T,1,H,W)")
        self._check_input(x, 3)
        bn_train = self._bn_train()        # batch statistics per frame call, like the reference's modules inside its frame loop
        if self.unet_norm == "batch_norm" and not bn_train:
            self._warn_detached(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and (self.unet_norm != "batch_norm" or bn_train):
            from .autograd import generator_video_apply
            x_out, feats = generator_video_apply(self, x)
            if apply_crop and self.to_crop:
                x_out = self._crop(x_out, diffY, diffX)
            return x_out, feats
        B, T = x.shape[0], x.shape[1]
        feats = []
        prev_ws = None
        # the clip's frames one behind the other (one copy instead of T strided ones)
        xs = x.detach().reshape(B, T, 256, 256).transpose(0, 1).float().contiguous()
        out_all = torch.empty(T, B, 1, 256, 256, dtype=torch.float32, device=x.device)
        drops = self._drop_scales(T, B, x.device)
        for t in range(T):
            xf = xs[t]
            _, up, _, ws = self._run(xf, need_feat=True, prev_ws=prev_ws, keep_act=True, slot=t, drop=drops[t], out=out_all[t])
            st = gauss_stats(up, B, 256, 256, 32)                      # (B,2,32)
            feats.append(st.reshape(B, 1, 64, 1, 1))
            prev_ws = ws
        x_out = _batch_major(out_all)
        if apply_crop and self.to_crop:
            x_out = self._crop(x_out, diffY, diffX)
        return x_out, torch.cat(feats, 1)
