"""Autograd bridge of the image and video generators: forward = uncl_gen_forward with activations kept, backward = ONE call of
uncl_gen_backward (hand-written HIP dgrad / wgrad / element-wise kernels), then the packed weight gradients are
re-laid-out into the reference parameter layout.  compute_dtype 'bf16' is the training path (bf16 activations and activation
gradients, fp32 accumulation, fp32 parameter gradients); 'fp32' is the PARITY mode: everything in fp32, weight gradients by
deterministic fixed-order kernels (csrc/bwd_f32.hip) -- slower, used to check a trainer step against the CPU oracle at fp32
tolerances (tests/test_gpu_trainer.py, tests/test_gpu_backward.py)."""
import ctypes as C
import os

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
        # every other gradient (biases, first layer, outconv, pos_embed) is a view of ONE flat buffer, the unpacked weights of
        # another: a data-parallel step all-reduces two or three flat tensors in place instead of concatenating 57 of them
        bsz = [int(torch.tensor(self.spec[nm + ".bias"][0]).prod()) for nm in self.names]
        extra = [288, 32, 32, 1, 144 * 256]
        # unet_norm='batch_norm' (training): gradients of the eighteen BatchNorm2d weights and biases, in state_dict order
        self.bn_names, bn_sz = [], []
        if getattr(module, "unet_norm", "none") == "batch_norm":
            from .state_spec import batch_norm_layers
            sdp = dict(module.named_parameters())
            self.bn_names = [q for _, q in batch_norm_layers()]
            bn_sz = [sdp[q + ".weight"].numel() for q in self.bn_names]
        self.small = torch.empty(sum(bsz) + sum(extra) + 2 * sum(bn_sz), dtype=torch.float32, device=dev)
        off, self.gb = 0, []
        for n_ in bsz:
            self.gb.append(self.small[off:off + n_])
            off += n_
        self.g_inc_w = self.small[off:off + 288].view(32, 1, 3, 3); off += 288
        self.g_inc_b = self.small[off:off + 32]; off += 32
        self.g_oc_w = self.small[off:off + 32]; off += 32
        self.g_oc_b = self.small[off:off + 1]; off += 1
        self.g_pe = self.small[off:off + 144 * 256].view(144, 256); off += 144 * 256
        self.g_bn_w, self.g_bn_b = [], []
        for n_ in bn_sz:
            self.g_bn_w.append(self.small[off:off + n_]); off += n_
        for n_ in bn_sz:
            self.g_bn_b.append(self.small[off:off + n_]); off += n_
        self.flat = torch.empty(sum(self.sizes), dtype=torch.float32, device=dev)      # unpacked weights, layer order
        self.w_off = [sum(self.sizes[:i]) for i in range(len(self.sizes) + 1)]
        self.DEC0 = 14            # packed weights 14..25 are the decoder's (uncl_gen_layer_name order)
        self.ev_half = None
        self.gws = None

    def run(self, xf, out, up, ws, ds, g_out, gup, accumulate=False, prev_ws=None, carry_in=None, carry_out=None, ev_half=None,
            clip=None):
        """One uncl_gen_backward call over the n samples of (xf, out, up, ws).  clip = (T, t): `ws` is a clip workspace, the
        gradient arena spans the clip too and the 3x3 / 2x2 weight gradients are taken once, by the call for frame 0
        (uncl_gen_bwd.clip_T)."""
        lib = _hip.lib()
        module = self.module
        gwts, _keep = module._packed_weights()
        n = xf.shape[0]
        gbytes = lib.uncl_gen_backward_workspace_bytes_dt(n * clip[0] if clip is not None else n, module._dtype_code())
        # the gradient arena (gigabytes at training batch sizes) lives with the module: one backward pass runs at a time, and
        # allocating it per pass sends the caching allocator through its slow paths once other large blocks have come and gone
        cache = module.__dict__.setdefault("_gws_cache", {})
        self.gws = cache.get(self.dev)
        if self.gws is None or self.gws.numel() < gbytes:
            self.gws = cache[self.dev] = torch.empty(gbytes, dtype=torch.uint8, device=self.dev)
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
        b.ssr_fused = int(getattr(module, "_ssr_fused", 0))
        b.prev_workspace = prev_ws.data_ptr() if prev_ws is not None else None
        if clip is not None:
            b.clip_T, b.clip_t, b.prev_workspace = int(clip[0]), int(clip[1]), None
        b.carry_in = carry_in.data_ptr() if carry_in is not None else None
        b.carry_out = carry_out.data_ptr() if carry_out is not None else None
        b.ev_decoder_done = None
        if ev_half is not None:
            h = ev_half.cuda_event
            b.ev_decoder_done = h.value if hasattr(h, "value") else int(h)
            if not b.ev_decoder_done:
                raise _hip.HipError("the decoder-done event has no HIP handle yet (record it once before the call)")
        if self.bn_names:
            arrs, _bnkeep = module._bn_arrays((self.g_bn_w, self.g_bn_b))
            _hip.check(lib.uncl_gen_set_bn(arrs[0], arrs[1], arrs[2], arrs[3], 0.1, arrs[4], arrs[5]), "uncl_gen_set_bn")
        _hip.check(lib.uncl_gen_backward(C.byref(gwts), C.byref(b), _hip.stream_ptr()), "uncl_gen_backward")

    def unpack(self, lo=0, hi=None):
        """packed [tap][Cout][Cin] -> reference parameter layout for the layers lo <= i < hi (views of self.flat), one launch"""
        lib = _hip.lib()
        hi = len(self.names) if hi is None else hi
        items = (_hip.UnpackItem * (hi - lo))()
        for j, i in enumerate(range(lo, hi)):
            shape, kind = self.spec[self.names[i] + ".weight"]
            transposed = kind == "convT"
            k = shape[2]
            cout, cin = (shape[1], shape[0]) if transposed else (shape[0], shape[1])
            it = items[j]
            it.packed, it.dst = self.gw_flat.data_ptr() + self.w_off[i] * 4, self.flat.data_ptr() + self.w_off[i] * 4
            it.Cout, it.Cin, it.k, it.transposed, it.flip, it.accumulate = cout, cin, k, int(transposed), \
                (1 if (transposed and k == 3) else 0), 0
        _hip.check(lib.uncl_unpack_conv_wgrads(items, hi - lo, _hip.stream_ptr()), "uncl_unpack_conv_wgrads")

    def grads(self, published=False):
        """state_dict name -> gradient tensor (views of self.flat / self.small); `published`: in the PUBLISHED layout the kernels
        write (a variant configuration's own parameters are slices / sums of these: generator._variant_grads)"""
        g = {"inc.conv.conv.weight": self.g_inc_w, "inc.conv.conv.bias": self.g_inc_b,
             "outc.conv.weight": self.g_oc_w.reshape(1, 32, 1, 1), "outc.conv.bias": self.g_oc_b,
             "gcn.pos_embed": self.g_pe.t().reshape(1, 256, 12, 12).contiguous()}
        for i, nm in enumerate(self.names):
            g[nm + ".weight"] = self.flat[self.w_off[i]:self.w_off[i + 1]].view(self.spec[nm + ".weight"][0])
            g[nm + ".bias"] = self.gb[i]
        for q, gw_, gb_ in zip(self.bn_names, self.g_bn_w, self.g_bn_b):
            g[q + ".weight"], g[q + ".bias"] = gw_, gb_
        return g if published else self.module._variant_grads(g)

    def finish(self, last_run):
        """Unpack everything after the last uncl_gen_backward call of a pass (`last_run(ev)` makes that call).  With a
        data-parallel reducer attached to the module the decoder half is unpacked and all-reduced on a side stream as soon as
        its event fires, under the graph block's and the encoder's kernels; the rest follows on the caller's stream."""
        red = getattr(self.module, "_grad_reducer", None)
        self.reduced = False
        if red is None or not red.active():
            last_run(None)
            self.unpack()
            return self.grads()
        # A variant configuration's gradients (skip-operator sub-sets, bilinear / parameter-free `up`) are slices, concatenations and
        # tap sums of the published-layout buffers (generator._variant_grads): copies made before the collectives have finished
        # would hold this rank's values only.  So the reducer is handed the PUBLISHED layout and the module's re-layout as a
        # function it applies in finish(), after the collectives and the division by the world size (round 6; before: refused).
        post = self.module._variant_grads if self.module._is_variant() else None
        self.reduced = True     # the buffers now belong to the reducer: the caller hands autograd NO parameter gradients
        if red.in_stream or torch.cuda.is_current_stream_capturing():
            # everything on the caller's stream: the pass, the re-layout, then the collectives in order (a captured step: one
            # stream, no cross-stream edge; the exchange is exposed, 19.7 MB over xGMI)
            last_run(None)
            self.unpack()
            g = self.grads(published=True)
            for t in (self.flat, self.small, g["gcn.pos_embed"]):
                red.launch_in_stream(t, self)
            red.keep(self, g, post)
            return g
        ev = torch.cuda.Event()
        ev.record()             # torch creates the hipEvent lazily: record once so that the raw handle exists, the library
        last_run(ev)            # records it again where the decoder's gradients are final
        cut = self.w_off[self.DEC0]
        with torch.cuda.stream(red.stream(self.dev)):
            red.stream(self.dev).wait_event(ev)
            self.unpack(self.DEC0, len(self.names))
            red.launch(self.flat[cut:], self)
        self.unpack(0, self.DEC0)
        g = self.grads(published=True)      # pos_embed's gradient is a transposed COPY of its slot in `small`, made before the reductions
        red.launch(self.flat[:cut], self)
        red.launch(self.small, self)
        red.launch(g["gcn.pos_embed"], self)
        red.keep(self, g, post)
        return g


class _WsLease:
    """A training forward's activation workspace (about a gigabyte at N = 32) must stay untouched until its backward pass has
    run, but allocating one per step sends the caching allocator to hipMalloc once other large blocks have come and gone
    (measured: 8 - 14 device allocations in ten steps, 8.4 -> 14 - 25 ms per step).  The lease parks the tensor in the module's
    free list when the autograd node that owns it dies, and the next forward takes it from there."""

    def __init__(self, module, dev, key):
        self.pool = module.__dict__.setdefault("_ws_pool", {}).setdefault((dev, key), [])
        self.module, self.dev, self.key, self.ws = module, dev, key, None

    def install(self, slot):
        """put a pooled tensor (if any) where module._workspace will look for it"""
        if self.pool:
            self.module._ws[(slot, self.dev)] = self.pool.pop()

    def take(self, slot):
        self.ws = self.module._ws.pop((slot, self.dev), None)

    def __del__(self):
        if self.ws is not None and len(self.pool) < 4:
            self.pool.append(self.ws)


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        n = x.shape[0]
        xf = x.detach().reshape(n, 256, 256).float().contiguous()
        # a private workspace per call: it must survive until backward (two generator passes may be alive in a step)
        lease = _WsLease(module, xf.device, "img")
        lease.install("train")
        out, up, _, ws, ds = module._run(xf, need_feat=True, keep_act=True, slot="train", save_preact=True, return_drop=True)
        lease.take("train")
        ctx.lease = lease
        ctx.module = module
        # `out` is returned: the node must not hold the returned OBJECT (its grad_fn is this node -- a reference cycle that only
        # the cyclic garbage collector breaks, which kept ~1.2 GB per step alive for several steps); an alias of its storage does
        ctx.saved = (xf, out.detach(), up, ws, ds)
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
        grads = gs.finish(lambda ev: gs.run(xf, out, up, ws, ds, g_out, gup, ev_half=ev))
        if gs.reduced:      # data-parallel: .grad is assigned by GradReducer.finish after the collectives (distributed.py)
            return (None, None) + (None,) * len(ctx.pnames)
        return (None, None) + tuple(grads.get(k) for k in ctx.pnames)


class _VideoGeneratorFn(torch.autograd.Function):
    """Clip forward (frames sequential, Unet.py:213-289) and backward through time: frames are visited last to first, the
    head-channel gradients of the eight recurrent hand-offs travel between consecutive frames in a small carry arena and
    the parameter gradients of all frames accumulate in one packed set.

    Clip layout (`module.clip_wgrad`, default: on for bf16, T > 1, unet_norm 'none'): the frames' activations live in ONE workspace
    laid out for T * B samples and their activation gradients in one arena of the same shape, so that the 3x3 / 2x2 weight and
    bias gradients are taken ONCE per clip over all T * B samples after the data-gradient chain has reached frame 0 -- 26
    launches at T * B samples each instead of 26 per frame at B (a frame's weight gradients depend on nothing later in the
    pass; at B = 8 every per-frame launch sat on its 27 - 55 us floor)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        from .generator import gauss_stats
        B, T = x.shape[0], x.shape[1]
        dev = x.device
        frames, feats, leases = [], [], []
        prev_ws = None
        use_clip = getattr(module, "clip_wgrad", None)
        if use_clip is None:     # UNCL_CLIP_WGRAD=0: the per-frame form (A/B timing)
            use_clip = module._dtype_code() == _hip.BF16 and os.environ.get("UNCL_CLIP_WGRAD", "1") != "0"
        use_clip = bool(use_clip) and T > 1 and getattr(module, "unet_norm", "none") == "none"
        ctx.clip = use_clip
        if use_clip:
            lease = _WsLease(module, dev, ("clipws", T))
            lease.install(("clipws", T))
            # the clip's frames one behind the other in ONE array (one copy instead of T): the deferred pass takes the first layer's
            # weight gradient over all T * B samples from frame 0's pointer (uncl_gen_bwd.clip_T)
            xs = x.detach().reshape(B, T, 256, 256).transpose(0, 1).float().contiguous().reshape(T * B, 256, 256)
        # the frames' outputs frame-major in one array (one transposing copy at the end instead of a T-way cat), their DropPath
        # flags drawn at once
        out_all = torch.empty(T, B, 1, 256, 256, dtype=torch.float32, device=dev)
        drops = module._drop_scales(T, B, dev)
        for t in range(T):
            xf = xs[t * B:(t + 1) * B] if use_clip else x[:, t].detach().reshape(B, 256, 256).float().contiguous()
            if use_clip:
                out, up, _, ws, ds = module._run(xf, need_feat=True, keep_act=True, slot=("clipws", T), save_preact=True,
                                                 return_drop=True, clip=(T, t), drop=drops[t], out=out_all[t])
                frames.append((xf, out, up, ws, ds))
                feats.append(gauss_stats(up, B, 256, 256, 32).reshape(B, 1, 64, 1, 1))
                continue
            lease = _WsLease(module, dev, ("clip", t))
            lease.install(("clip", t))
            out, up, _, ws, ds = module._run(xf, need_feat=True, prev_ws=prev_ws, keep_act=True, slot=("clip", t),
                                             save_preact=True, return_drop=True, drop=drops[t], out=out_all[t])
            lease.take(("clip", t))
            leases.append(lease)
            feats.append(gauss_stats(up, B, 256, 256, 32).reshape(B, 1, 64, 1, 1))
            frames.append((xf, out, up, ws, ds))
            prev_ws = ws
        if use_clip:
            lease.take(("clipws", T))
            leases.append(lease)
        ctx.module, ctx.frames, ctx.leases = module, frames, leases
        from .generator import _batch_major
        return _batch_major(out_all), torch.cat(feats, 1)

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
        # the incoming gradients frame-major, one copy each instead of one strided copy per frame
        go_all = None if g_out is None else g_out.reshape(B, T, 256, 256).transpose(0, 1).float().contiguous()
        gf_all = None if g_feats is None else g_feats.reshape(B, T, 2, 32).transpose(0, 1).float().contiguous()
        for t in range(T - 1, -1, -1):
            xf, out, up, ws, ds = frames[t]
            go = torch.zeros_like(out) if go_all is None else go_all[t].reshape(B, 1, 256, 256)
            gup = None
            if gf_all is not None:
                gst = gf_all[t]
                gup = torch.empty_like(up)
                _hip.check(lib.uncl_gauss_stats_backward(up.data_ptr(), code, gst.data_ptr(), gup.data_ptr(), B, 256, 256, 32, 0,
                                                         _hip.stream_ptr()), "uncl_gauss_stats_backward")
            carry_in = carries[(t + 1) % 2] if t < T - 1 else None      # written by frame t+1
            carry_out = carries[t % 2] if t > 0 else None               # read by frame t-1
            if ctx.clip:
                call = lambda ev, a_=(xf, out, up, ws, ds, go, gup), t_=t, ci=carry_in, co=carry_out: gs.run(
                    *a_, accumulate=(t_ != T - 1), carry_in=ci, carry_out=co, ev_half=ev, clip=(T, t_))
            else:
                call = lambda ev, a_=(xf, out, up, ws, ds, go, gup), t_=t, ci=carry_in, co=carry_out: gs.run(
                    *a_, accumulate=(t_ != T - 1), prev_ws=frames[t_ - 1][3] if t_ > 0 else None, carry_in=ci, carry_out=co,
                    ev_half=ev)
            if t > 0:
                call(None)
        grads = gs.finish(call)         # frame 0: the last call of the pass
        # the saved activations stay with the graph node: the reference calls backward twice on it
        # (errG_d.backward(retain_graph=True), then the structural loss; GanTrainer.py:338,461)
        if gs.reduced:
            return (None, None) + (None,) * len(ctx.pnames)
        return (None, None) + tuple(grads.get(k) for k in ctx.pnames)


def _warn_bf16_norm(module):
    if module._dtype_code() == _hip.BF16 and getattr(module, "unet_norm", "none") in ("instance_norm", "batch_norm"):
        import warnings
        warnings.warn("uncltmo_amd: training with unet_norm='instance_norm' / 'batch_norm' in bf16 loses the gradient below the per-channel "
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
