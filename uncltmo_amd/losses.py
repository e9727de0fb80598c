"""Loss heads of the GanTrainer step as autograd Functions over the fused loss+gradient HIP kernels
(csrc/loss_heads.hip, csrc/stats_kernels.hip).  Names follow the trainer methods they replace
(GanTrainerImg.py:219-229 contrastive_D_loss, :410-439 nce, :341-408 pseudo_label_loss / infoNCE2,
GanTrainer.py:669-682 L_TV)."""
import ctypes as C

import torch

from . import _hip
from .generator import gauss_stats


def _scalar(dev):
    # every loss kernel WRITES its result when called with accumulate_loss = 0 (loss_heads.hip: `loss[0] = (accumulate ? loss[0] :
    # 0) + ...`), which is how every call below calls them: no zero fill (it was ten 4-us launches per optimisation step)
    return torch.empty(1, dtype=torch.float32, device=dev)


class _CganFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real, fake, weight=1.0):
        n = real.numel()
        r, f = real.detach().reshape(n).float().contiguous(), fake.detach().reshape(n).float().contiguous()
        loss, grf = _scalar(r.device), torch.empty(2, n, dtype=torch.float32, device=r.device)    # both gradients in ONE tensor:
        _hip.check(_hip.lib().uncl_cgan_loss(r.data_ptr(), f.data_ptr(), n, float(weight), loss.data_ptr(), grf[0].data_ptr(),
                                             grf[1].data_ptr(), 0, _hip.stream_ptr()), "uncl_cgan_loss")   # one multiply in backward
        ctx.save_for_backward(grf)
        ctx.shapes = (real.shape, fake.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grf,) = ctx.saved_tensors
        t = grf * g
        return t[0].reshape(ctx.shapes[0]), t[1].reshape(ctx.shapes[1]), None


def contrastive_D_loss(real_logits, fake_logits, weight=1.0):
    """GanTrainerImg.py:219-229; `weight` (python float): the caller's loss weight applied inside the kernel -- loss and gradients
    come out scaled, with no scalar-multiply launch (and autograd node) around the head"""
    return _CganFn.apply(real_logits, fake_logits, float(weight))


def _dense(t):
    """A view of `t` whose memory is one dense block (the NCE sums run over all elements of a sample, so any common layout
    of anchor / positive / negative will do) and the permutation that maps a gradient in that layout back to t's axes."""
    if t.is_contiguous():
        return t, None
    if t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous():          # channel-last features (the generator's up_x)
        return t.permute(0, 2, 3, 1), (0, 3, 1, 2)
    return t.contiguous(), None


def _row_of(row, whole):
    """index i if `row` (leading dim 1) is whole[i:i+1] viewing the same memory, else -1"""
    if row.shape[0] != 1 or row.shape[1:] != whole.shape[1:] or row.stride()[1:] != whole.stride()[1:]:
        return -1
    if row.untyped_storage().data_ptr() != whole.untyped_storage().data_ptr() or whole.stride(0) == 0:
        return -1
    off = row.storage_offset() - whole.storage_offset()
    if off < 0 or off % whole.stride(0) != 0 or off // whole.stride(0) >= whole.shape[0]:
        return -1
    return off // whole.stride(0)


class _NceFn(torch.autograd.Function):
    """anchor / pos / neg: same shape (N, ...) in ANY common layout; `hw` = spatial positions averaged.  The loss is computed
    in forward, the gradients in backward (uncl_nce_backward) in the feature dtype and layout, scaled by the upstream
    gradient on the device.  pos_row / neg_row >= 0: positive / negative are that row of the anchor tensor itself
    (infoNCE2): they are then not separate autograd inputs and the anchor receives the complete gradient."""

    @staticmethod
    def forward(ctx, anchor, pos, neg, hw, k, c, pos_shared, neg_shared, pos_row, neg_row, rows_dev=None, lmcl=False):
        lib = _hip.lib()
        n = anchor.shape[0]
        E = anchor.numel() // n
        if anchor.dtype not in (torch.bfloat16, torch.float32):
            # the kernels read bf16 or fp32 elements: anything else (fp16 is an accepted INFERENCE dtype) would be read as fp32
            raise TypeError("uncltmo_amd nce: features must be bfloat16 or float32, got %s" % anchor.dtype)
        code = _hip.BF16 if anchor.dtype == torch.bfloat16 else _hip.F32
        a, perm = _dense(anchor.detach())
        if perm is None:
            p, q = pos.detach().contiguous(), neg.detach().contiguous()
        else:
            back = [perm.index(i) for i in range(4)]
            p, q = pos.detach().permute(*back).contiguous(), neg.detach().permute(*back).contiguous()
        dev = a.device
        ws = torch.empty(lib.uncl_nce_workspace_bytes(n), dtype=torch.uint8, device=dev)
        loss = _scalar(dev)
        need = ctx.needs_input_grad
        vec = 8 if code == _hip.BF16 else 4
        aligned = all(t.data_ptr() % 16 == 0 for t in (a, p, q))
        ctx.deferred = E % vec == 0 and aligned
        if rows_dev is not None:
            if not ctx.deferred:
                raise ValueError("uncltmo_amd: row-shared NCE needs 16-byte aligned rows")
            rows_dev = rows_dev.detach().to(torch.int32).contiguous()
            p = q = a[0:1]          # placeholders: the kernels take the rows from `rows_dev`
        if pos_row >= 0 or neg_row >= 0:
            if not ctx.deferred:
                raise ValueError("uncltmo_amd: row-shared NCE needs 16-byte aligned rows")
            if pos_row >= 0:
                p = a[pos_row:pos_row + 1]
            if neg_row >= 0:
                q = a[neg_row:neg_row + 1]
        ga = gp = gq = None
        if not ctx.deferred:      # odd sizes (the discriminator's two-element features): fp32 gradients from the fused call
            ga = torch.empty(a.shape, dtype=torch.float32, device=dev) if need[0] else None
            gp = torch.empty(p.shape, dtype=torch.float32, device=dev) if need[1] else None
            gq = torch.empty(q.shape, dtype=torch.float32, device=dev) if need[2] else None
        P = lambda t: t.data_ptr() if t is not None else None
        _hip.check(lib.uncl_nce_loss(a.data_ptr(), p.data_ptr(), q.data_ptr(), code, n, E, hw, int(pos_shared), int(neg_shared),
                                     float(k), float(c), 1.0, loss.data_ptr(), P(ga), P(gp), P(gq), 2 if lmcl else 0, 0, ws.data_ptr(),
                                     P(rows_dev), _hip.stream_ptr()), "uncl_nce_loss")
        ctx.rows_dev = rows_dev
        ctx.g = (ga, gp, gq)
        ctx.dt = (anchor.dtype, pos.dtype, neg.dtype)
        ctx.saved = (a, p, q, ws, perm)
        ctx.args = (code, n, E, hw, int(pos_shared), int(neg_shared), float(k), float(c), int(pos_row), int(neg_row))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if not ctx.deferred:
            out = [None if t is None else (t * g).to(dt) for t, dt in zip(ctx.g, ctx.dt)]
            return (out[0], out[1], out[2]) + (None,) * 9
        lib = _hip.lib()
        a, p, q, ws, perm = ctx.saved
        code, n, E, hw, ps, qs, k, c, pos_row, neg_row = ctx.args
        need = ctx.needs_input_grad
        gs = g.detach().float().reshape(1).contiguous()
        ga = torch.empty_like(a) if need[0] else None
        rows_dev = ctx.rows_dev
        gp = torch.empty_like(p) if need[1] and pos_row < 0 and rows_dev is None else None
        gq = torch.empty_like(q) if need[2] and neg_row < 0 and rows_dev is None else None
        P = lambda t: t.data_ptr() if t is not None else None
        _hip.check(lib.uncl_nce_backward(a.data_ptr(), p.data_ptr(), q.data_ptr(), code, n, E, hw, ps, qs, k, c, ws.data_ptr(),
                                         gs.data_ptr(), P(ga), P(gp), P(gq), code, pos_row, neg_row, P(rows_dev),
                                         _hip.stream_ptr()),
                   "uncl_nce_backward")
        if perm is not None:
            ga, gp, gq = [None if t is None else t.permute(*perm) for t in (ga, gp, gq)]
        return (ga, gp, gq) + (None,) * 9


def nce_rows(anchor, rows, k, c, hw=None):
    """infoNCE2 (GanTrainerImg.py:398-402): positive / negative are rows rows[0] / rows[1] of `anchor`, where `rows` is a
    DEVICE int tensor (the arg-max / arg-min tmqi_naturalness returns): the selection never goes through the host, and the
    anchor receives the complete gradient (its own plus the two shared rows')."""
    if hw is None:
        hw = anchor.shape[-1] * anchor.shape[-2]
    dummy = anchor.detach()[0:1]
    return _NceFn.apply(anchor, dummy, dummy, hw, k, c, True, True, -1, -1, rows)


def nce(anchor, positive, negative, k, c, hw=None, form="InfoNCE"):
    """2-way InfoNCE with s(a,b) = mean_hw sum_c a b / (c + k|a-b|).  Tensors are (N,C,H,W)-shaped (any memory layout
    shared by the three); `positive` / `negative` may have a leading dim of 1 (one row shared by all samples), and may be
    rows of `anchor` itself (infoNCE2, GanTrainerImg.py:398-402): the anchor then gets their gradient as well."""
    if hw is None:
        hw = anchor.shape[-1] * anchor.shape[-2]
    n = anchor.shape[0]
    pos_shared = positive.shape[0] == 1 and n != 1
    neg_shared = negative.shape[0] == 1 and n != 1
    E = anchor.numel() // n
    vec_ok = E % (8 if anchor.dtype == torch.bfloat16 else 4) == 0 and _dense(anchor.detach())[0].data_ptr() % 16 == 0 and \
        (E * anchor.element_size()) % 16 == 0
    pos_row = _row_of(positive, anchor) if (pos_shared and vec_ok and positive.requires_grad == anchor.requires_grad) else -1
    neg_row = _row_of(negative, anchor) if (neg_shared and vec_ok and negative.requires_grad == anchor.requires_grad) else -1
    # a row of the anchor is passed as a detached tensor: its gradient is folded into the anchor's inside the kernel
    pos_in = positive.detach() if pos_row >= 0 else positive
    neg_in = negative.detach() if neg_row >= 0 else negative
    return _NceFn.apply(anchor, pos_in, neg_in, hw, k, c, pos_shared, neg_shared, pos_row, neg_row, None, form == "LMCL")


class _SimPairFn(torch.autograd.Function):
    """(anchor, x1, x2) -> (s(anchor, x1), s(anchor, x2)), each (N,), s(a,b) = mean_hw sum_c a b / (c + k|a-b|)
    (GanTrainerImg.py:421-429): the building block of nce() with longer positive / negative lists.  x1 / x2: the anchor's shape, or
    a leading dim of 1 (one row for every sample).  fp32 gradients from uncl_nce_similarity_backward, cast to the inputs' dtypes."""

    @staticmethod
    def forward(ctx, anchor, x1, x2, hw, k, c):
        lib = _hip.lib()
        if anchor.dtype not in (torch.bfloat16, torch.float32):
            raise TypeError("uncltmo_amd nce: features must be bfloat16 or float32, got %s" % anchor.dtype)
        if x1.dtype != anchor.dtype or x2.dtype != anchor.dtype:
            raise TypeError("uncltmo_amd nce: anchor, positives and negatives must share one dtype")
        n = anchor.shape[0]
        a = anchor.detach().contiguous()
        E = a.numel() // n
        xs, shared = [], []
        for x in (x1, x2):
            sh = x.shape[0] == 1 and n != 1
            if (not sh and x.shape != anchor.shape) or (sh and x.shape[1:] != anchor.shape[1:]):
                raise ValueError("uncltmo_amd nce: a positive / negative must have the anchor's shape or a leading dimension of 1")
            xs.append(x.detach().contiguous())
            shared.append(sh)
        code = _hip.BF16 if a.dtype == torch.bfloat16 else _hip.F32
        ws = torch.empty(lib.uncl_nce_workspace_bytes(n), dtype=torch.uint8, device=a.device)
        sims = torch.empty(n, 2, dtype=torch.float32, device=a.device)
        _hip.check(lib.uncl_nce_similarity(a.data_ptr(), xs[0].data_ptr(), xs[1].data_ptr(), code, n, E, hw, int(shared[0]),
                                           int(shared[1]), float(k), float(c), sims.data_ptr(), ws.data_ptr(), _hip.stream_ptr()),
                   "uncl_nce_similarity")
        ctx.saved = (a, xs[0], xs[1])
        ctx.args = (code, n, E, hw, int(shared[0]), int(shared[1]), float(k), float(c))
        ctx.dt = (anchor.dtype, x1.dtype, x2.dtype)
        ctx.shapes = (anchor.shape, x1.shape, x2.shape)
        return sims[:, 0], sims[:, 1]

    @staticmethod
    def backward(ctx, g1, g2):
        lib = _hip.lib()
        a, p, q = ctx.saved
        code, n, E, hw, ps, qs, k, c = ctx.args
        need = ctx.needs_input_grad
        gs = torch.stack([g1.detach().float(), g2.detach().float()], 1).contiguous()
        out = [torch.empty(t.shape, dtype=torch.float32, device=a.device) if nd else None for t, nd in zip((a, p, q), need[:3])]
        P = lambda t: t.data_ptr() if t is not None else None
        _hip.check(lib.uncl_nce_similarity_backward(a.data_ptr(), p.data_ptr(), q.data_ptr(), code, n, E, hw, ps, qs, k, c,
                                                    gs.data_ptr(), P(out[0]), P(out[1]), P(out[2]), 0, _hip.stream_ptr()),
                   "uncl_nce_similarity_backward")
        res = [None if t is None else t.to(dt).reshape(sh) for t, dt, sh in zip(out, ctx.dt, ctx.shapes)]
        return (res[0], res[1], res[2], None, None, None)


def nce_lists(anchor, positives, negatives, k, c, hw=None, form="InfoNCE"):
    """nce() of the trainers with ANY number of positives and negatives (GanTrainerImg.py:410-439): per positive p the logits
    [s(a,p), s(a,n_1), ..., s(a,n_Q)] go through the 2..(Q+1)-way cross-entropy with class 0 ('InfoNCE', :431-433) or lmcl_loss
    ('LMCL', :441-450: -log(exp(s_p) / sum_j exp(s_nj))), averaged over the positives.  The similarities come from the HIP kernels
    two at a time (uncl_nce_similarity); the logits are (N, Q+1) device tensors.  The published call sites (one positive, one
    negative) use the fused nce() above instead -- THIS form is not on the published step path: its last reduction over the
    (N, Q+1) logits (a few hundred floats) is torch's cross_entropy / logsumexp, the only PyTorch arithmetic kernels in the package."""
    if form not in ("InfoNCE", "LMCL"):
        raise TypeError("%s is not found in loss/adversarial.py" % form)
    if len(positives) == 0 or len(negatives) == 0:
        raise ValueError("uncltmo_amd nce: at least one positive and one negative")
    if hw is None:
        hw = anchor.shape[-1] * anchor.shape[-2]
    xs = list(negatives) + list(positives)
    sims = []
    for i in range(0, len(xs), 2):
        pair = xs[i:i + 2]
        s1, s2 = _SimPairFn.apply(anchor, pair[0], pair[-1], hw, k, c)
        sims += [s1, s2][:len(pair)]
    neg, pos = sims[:len(negatives)], sims[len(negatives):]
    n = anchor.shape[0]
    loss = None
    for sp in pos:
        if form == "InfoNCE":
            logits = torch.stack([sp] + neg, 1)
            l = torch.nn.functional.cross_entropy(logits, torch.zeros(n, dtype=torch.long, device=logits.device))
        else:
            l = -(sp - torch.logsumexp(torch.stack(neg, 1), 1)).mean()
        loss = l if loss is None else loss + l
    return loss / len(pos)


class _FrameStatsFn(torch.autograd.Function):
    """(N,1,H,W) fp32 -> (mean (N,), mean Gaussian local variance (N,)) with analytic backward."""

    @staticmethod
    def forward(ctx, x):
        n, _, h, w = x.shape
        xf = x.detach().reshape(n, h, w).float().contiguous()
        st = gauss_stats(xf, n, h, w, 1)
        ctx.save_for_backward(xf)
        return st[:, 0, 0], st[:, 1, 0]           # strided views of (n, 2, 1): the L1 kernel takes a stride, no copies

    @staticmethod
    def backward(ctx, g_mean, g_var):
        (xf,) = ctx.saved_tensors
        n, h, w = xf.shape
        lib = _hip.lib()
        gx = torch.empty_like(xf)
        _hip.check(lib.uncl_add_per_sample_const(gx.data_ptr(), g_mean.float().contiguous().data_ptr(), h * w, n, 1.0 / (h * w), 0,
                                                 _hip.stream_ptr()), "uncl_add_per_sample_const")
        _hip.check(lib.uncl_gauss_var_backward(xf.data_ptr(), g_var.float().contiguous().data_ptr(), gx.data_ptr(), n, h, w, 1,
                                               _hip.stream_ptr()), "uncl_gauss_var_backward")
        return gx.reshape(n, 1, h, w)


class _PseudoTermsFn(torch.autograd.Function):
    """(4N,1,128,128) patches + the device index of the pseudo label -> the two L1 terms of pseudo_label_loss (GanTrainerImg.py:
    360-367): |patch mean - label's mean| and |patch contrast - label's contrast| averaged over the patches.  One statistics pass and
    one launch for both terms forward; backward: one launch that applies the two upstream scalars, then the two statistics
    backward kernels (before: index_select / expand / L1 as autograd nodes -- seven launches forward, twelve backward)."""

    @staticmethod
    def forward(ctx, patches, row):
        n, _, h, w = patches.shape
        xf = patches.detach().reshape(n, h, w).float().contiguous()
        st = gauss_stats(xf, n, h, w, 1)                       # (n, 2, 1)
        loss2 = torch.empty(2, dtype=torch.float32, device=xf.device)
        grad = torch.empty(n, 2, dtype=torch.float32, device=xf.device)
        _hip.check(_hip.lib().uncl_l1_to_row(st.data_ptr(), n, row.data_ptr(), loss2.data_ptr(), grad.data_ptr(), _hip.stream_ptr()),
                   "uncl_l1_to_row")
        ctx.save_for_backward(xf, grad)
        return loss2[0], loss2[1]

    @staticmethod
    def backward(ctx, g_mean_term, g_var_term):
        xf, grad = ctx.saved_tensors
        n, h, w = xf.shape
        lib = _hip.lib()
        gs = torch.empty(2, n, dtype=torch.float32, device=xf.device)
        _hip.check(lib.uncl_l1_to_row_backward(grad.data_ptr(), n, g_mean_term.detach().float().contiguous().data_ptr(),
                                               g_var_term.detach().float().contiguous().data_ptr(), gs.data_ptr(), _hip.stream_ptr()),
                   "uncl_l1_to_row_backward")
        gx = torch.empty_like(xf)
        _hip.check(lib.uncl_add_per_sample_const(gx.data_ptr(), gs[0].data_ptr(), h * w, n, 1.0 / (h * w), 0, _hip.stream_ptr()),
                   "uncl_add_per_sample_const")
        _hip.check(lib.uncl_gauss_var_backward(xf.data_ptr(), gs[1].data_ptr(), gx.data_ptr(), n, h, w, 1, _hip.stream_ptr()),
                   "uncl_gauss_var_backward")
        return gx.reshape(n, 1, h, w), None


def pseudo_label_pair(patches, best_worst):
    """the (mean term, contrast term) of pseudo_label_loss for `patches` with the label row best_worst[0] (device int32)"""
    if best_worst.dtype != torch.int32 or not best_worst.is_cuda:
        raise TypeError("uncltmo_amd: the pseudo label's index must be a device int32 tensor (tmqi_naturalness's best_worst)")
    return _PseudoTermsFn.apply(patches, best_worst.detach())


def frame_stats(x):
    return _FrameStatsFn.apply(x)


class _L1PairsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        n = a.numel()

        def flat(t):
            """(fp32 tensor, element stride): a 1-D view where the layout allows it (strided slices, expanded rows: stride 0) --
            the kernel reads through a stride, so per-sample statistics and their broadcast partners are not copied"""
            t = t.detach()
            if t.dtype != torch.float32:
                t = t.float()
            if t.dim() == 1 and t.shape[0] == n:
                return t, t.stride(0)
            return t.reshape(n).contiguous(), 1

        (af, sa), (bf, sb) = flat(a), flat(b)
        loss, gab = _scalar(af.device), torch.empty(2, n, dtype=torch.float32, device=af.device)
        _hip.check(_hip.lib().uncl_l1_pairs(af.data_ptr(), sa, bf.data_ptr(), sb, n, 1.0, loss.data_ptr(), gab[0].data_ptr(),
                                            gab[1].data_ptr(), 0, _hip.stream_ptr()), "uncl_l1_pairs")
        ctx.save_for_backward(gab)
        ctx.shapes = (a.shape, b.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (gab,) = ctx.saved_tensors
        t = gab * g
        return t[0].reshape(ctx.shapes[0]), t[1].reshape(ctx.shapes[1])


def l1_mean(a, b):
    """nn.L1Loss()(a, b) for per-sample scalars."""
    return _L1PairsFn.apply(a, b)


def tmqi_naturalness(frames, patch=None):
    """fp64 naturalness scores of (N,1,H,W) frames scaled by 255 (TMQI.py:210-242), per frame or per patch x patch tile
    (row-major within a frame).  Returns (scores float64 (N*tiles,), best_worst int32 (2,))."""
    n, _, h, w = frames.shape
    ph, pw = (h, w) if patch is None else (patch, patch)
    xf = frames.detach().reshape(n, h, w).float().contiguous()
    cnt = n * (h // ph) * (w // pw)
    scores = torch.empty(cnt, dtype=torch.float64, device=xf.device)
    bw = torch.empty(2, dtype=torch.int32, device=xf.device)
    _hip.check(_hip.lib().uncl_tmqi_naturalness(xf.data_ptr(), n, h, w, ph, pw, 255.0, scores.data_ptr(), bw.data_ptr(),
                                                _hip.stream_ptr()), "uncl_tmqi_naturalness")
    return scores, bw


class _WeightedSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        lib = _hip.lib()
        n = len(terms)
        ts = [t.detach().reshape(1).float().contiguous() for t in terms]
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        w = (C.c_float * n)(*weights)
        out = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        _hip.check(lib.uncl_weighted_sum(ptrs, w, n, out.data_ptr(), _hip.stream_ptr()), "uncl_weighted_sum")
        ctx.weights = tuple(weights)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.weights)
        w = (C.c_float * n)(*ctx.weights)
        gs = g.detach().reshape(1).float().contiguous()
        out = torch.empty(n, dtype=torch.float32, device=gs.device)
        _hip.check(_hip.lib().uncl_weighted_sum_backward(gs.data_ptr(), w, n, out.data_ptr(), _hip.stream_ptr()),
                   "uncl_weighted_sum_backward")
        return (None,) + tuple(out[i] for i in range(n))


def weighted_sum(pairs):
    """sum_i w_i * t_i for (python float w_i, 0-dim device tensor t_i) pairs: the loss weighting of the trainers
    (GanTrainerImg.py:285-313) as one launch forward and one backward instead of a chain of scalar tensor ops."""
    pairs = [(float(w), t) for w, t in pairs]
    out = None
    for i in range(0, len(pairs), 16):
        chunk = pairs[i:i + 16]
        s = _WeightedSumFn.apply([w for w, _ in chunk], *[t for _, t in chunk])
        out = s if out is None else out + s
    return out


class _TvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        xf = x.detach().reshape(n * c, h, w).float().contiguous()
        loss, gx = _scalar(xf.device), torch.empty_like(xf)
        ws = torch.empty(1024, dtype=torch.float32, device=xf.device)
        # L_TV divides by the batch size (dim 0) only
        _hip.check(_hip.lib().uncl_tv_loss(xf.data_ptr(), n * c, h, w, float(c), loss.data_ptr(), gx.data_ptr(), 0, 0,
                                           ws.data_ptr(), _hip.stream_ptr()), "uncl_tv_loss")
        ctx.save_for_backward(gx)
        ctx.shape = x.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (gx,) = ctx.saved_tensors
        return (gx * g).reshape(ctx.shape)


def tv_loss(x):
    return _TvFn.apply(x)
