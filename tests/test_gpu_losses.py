"""GPU parity of the loss kernels (forward values and input gradients) against the oracle and the reference's
goldens, through the Python host that binds the C ABI."""
import numpy as np
import pytest
import torch

from hip_util import rel_l2
from oracle import losses as OL
from uncltmo_amd import _hip, synth
from uncltmo_amd.struct_loss import StructLoss

pytestmark = pytest.mark.gpu


def test_struct_loss_matches_reference_golden(golden):
    g = golden("losses")
    fake = synth.ldr_frames(2, 64, 64, salt="slf").cuda().requires_grad_(True)
    hdr = synth.smooth_hdr_frames(2, 64, 64, salt="slh").cuda()
    sl = StructLoss([1.0, 1.0, 1.0], crop_input=0)
    one = StructLoss([1.0], crop_input=0)
    np.testing.assert_allclose(one(fake, None, hdr, [1.0]).item(), g["loss.struct_level"], rtol=1e-5)
    lp = sl(fake, None, hdr, [1.0, 1.0, 1.0])
    np.testing.assert_allclose(lp.item(), g["loss.struct_pyr"], rtol=1e-5)
    lp.backward()
    assert rel_l2(fake.grad.cpu(), torch.from_numpy(g["loss.struct_pyr.gfake"])) < 1e-4


def test_struct_loss_full_size_vs_oracle():
    fake = synth.smooth_hdr_frames(3, salt="sf").clamp(0, 1).cuda().requires_grad_(True)
    hdr = synth.smooth_hdr_frames(3, salt="sh").cuda()
    w = [1.0, 0.5, 2.0]
    l = StructLoss(w, crop_input=0)(fake, None, hdr, w) * 3.0
    l.backward()
    fc = fake.detach().cpu().requires_grad_(True)
    lo = OL.struct_loss_pyramid(fc, hdr.cpu(), w) * 3.0
    lo.backward()
    np.testing.assert_allclose(l.item(), lo.item(), rtol=2e-5)
    assert rel_l2(fake.grad.cpu(), fc.grad) < 2e-4


def test_struct_loss_edge_cases():
    # identical images: zero loss; constant images: var = 0 everywhere (the [var > 0] branch), finite gradient
    x = synth.ldr_frames(1, 32, 48, salt="e1").cuda()
    sl = StructLoss([1.0, 1.0], crop_input=0)
    assert abs(sl(x, None, x.clone(), [1.0, 1.0]).item()) < 1e-10
    c = torch.full((1, 1, 32, 48), 0.3, device="cuda", requires_grad=True)
    l = sl(c, None, x, [1.0, 1.0])
    l.backward()
    cc = torch.full((1, 1, 32, 48), 0.3, requires_grad=True)
    lo = OL.struct_loss_pyramid(cc, x.cpu(), [1.0, 1.0])
    lo.backward()
    np.testing.assert_allclose(l.item(), lo.item(), rtol=1e-4)
    assert torch.isfinite(c.grad).all()


def test_bicubic_half_matches_reference_golden(golden):
    g = golden("losses")
    ramp = (torch.arange(16.0)[:, None] * 3 + torch.arange(16.0)[None, :] ** 2).reshape(1, 16, 16).cuda()
    out = torch.empty(1, 8, 8, device="cuda")
    _hip.check(_hip.lib().uncl_bicubic_half(ramp.data_ptr(), out.data_ptr(), 1, 16, 16, _hip.stream_ptr()), "bicubic")
    np.testing.assert_allclose(out.cpu().numpy().reshape(1, 1, 8, 8), g["bicubic_half_ramp"], rtol=1e-6, atol=1e-5)


# ---- discriminator ------------------------------------------------------------------------------------------
from conftest import synth_state                                         # noqa: E402
from oracle import discriminator as OD                                   # noqa: E402
from uncltmo_amd import losses as HL                                     # noqa: E402
from uncltmo_amd import state_spec                                       # noqa: E402
from uncltmo_amd.discriminator import SimpleDiscriminator                # noqa: E402


def make_d():
    d = SimpleDiscriminator(256, 1, 16, "none", "none", 0, 0)
    synth.fill_state_dict(d, "d0")
    return d.cuda()


def test_simple_d_state_dict_keys_and_forward_golden(golden):
    g = golden("disc")
    d = make_d()
    assert [k for k, _, _ in state_spec.simple_d_spec()] == list(d.state_dict().keys())
    x = torch.cat([synth.ldr_frames(2, salt="dA"), synth.smooth_hdr_frames(1, salt="dB")], 0).cuda()
    with torch.no_grad():
        o, f = d(x)
    np.testing.assert_allclose(o.cpu().numpy(), g["d.output"], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(f.cpu().numpy(), g["d.fea_final"], rtol=2e-4, atol=1e-7)


def test_simple_d_backward_vs_oracle_autograd():
    d = make_d()
    x = torch.cat([synth.ldr_frames(2, salt="dA"), synth.smooth_hdr_frames(2, salt="dC")], 0)
    xg = x.cuda().requires_grad_(True)
    o, f = d(xg)
    wo = torch.tensor([0.3, -1.0, 0.7, 0.2], device="cuda").reshape(4, 1)
    wf = torch.tensor([[1.0, -2.0], [0.5, 3.0], [-1.0, 1.0], [2.0, 0.1]], device="cuda").reshape(4, 2, 1, 1)
    ((o * wo).sum() + (f * wf).sum() * 10.0).backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    xc = x.clone().requires_grad_(True)
    oo, ff = OD.simple_d_forward(sd, xc)
    ((oo * wo.cpu()).sum() + (ff * wf.cpu()).sum() * 10.0).backward()
    assert rel_l2(xg.grad.cpu(), xc.grad) < 1e-4
    for k, p in d.named_parameters():
        assert rel_l2(p.grad.cpu(), sd[k].grad) < 1e-4, k


# ---- loss heads ---------------------------------------------------------------------------------------------
def test_contrastive_d_loss_golden(golden):
    g = golden("losses")
    a = torch.tensor(synth.hash_uniform("la", 4) * 4 - 2).reshape(4, 1).cuda().requires_grad_(True)
    b = torch.tensor(synth.hash_uniform("lb", 4) * 4 - 2).reshape(4, 1).cuda().requires_grad_(True)
    l = HL.contrastive_D_loss(a, b)
    (l * 2.0).backward()
    np.testing.assert_allclose(l.item(), g["loss.cgan"], rtol=1e-6)
    np.testing.assert_allclose(a.grad.cpu().numpy() / 2, g["loss.cgan.ga"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b.grad.cpu().numpy() / 2, g["loss.cgan.gb"], rtol=1e-5, atol=1e-7)
    # larger batch vs the oracle
    r, f = torch.randn(37, generator=torch.Generator().manual_seed(1)), torch.randn(37, generator=torch.Generator().manual_seed(2))
    rc, fc = r.clone().requires_grad_(True), f.clone().requires_grad_(True)
    OL.contrastive_d_loss(rc, fc).backward()
    rg, fg = r.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
    HL.contrastive_D_loss(rg, fg).backward()
    assert rel_l2(rg.grad.cpu(), rc.grad) < 1e-5 and rel_l2(fg.grad.cpu(), fc.grad) < 1e-5


@pytest.mark.parametrize("n,row,ties", [(8, 3, False), (128, 0, False), (128, 127, True), (300, 17, True), (1, 0, False)])
def test_pseudo_label_l1_pair_against_torch_autograd(n, row, ties):
    """uncl_l1_to_row (the two L1 terms of pseudo_label_loss against the label row, GanTrainerImg.py:360-367) and its backward
    against nn.L1Loss on the index_select / expand form the trainer used to build: losses, gradients of every row, the label row's
    own (minus the sum of the others' signs), ties (sign 0), more rows than one workgroup has threads"""
    import ctypes as C
    g = torch.Generator().manual_seed(11 + n)
    st = torch.rand(n, 2, generator=g)
    if ties:
        st[n // 3] = st[row]
        st[n - 1 - (row == n - 1), 0] = st[row, 0]
    ref = st.clone().requires_grad_(True)
    idx = torch.tensor([row])
    l0 = torch.nn.L1Loss()(ref[:, 0], ref[:, 0].index_select(0, idx).expand(n))
    l1 = torch.nn.L1Loss()(ref[:, 1], ref[:, 1].index_select(0, idx).expand(n))
    (0.7 * l0 - 1.3 * l1).backward()
    lib = _hip.lib()
    sd, rd = st.cuda().contiguous(), torch.tensor([row, 0], dtype=torch.int32, device="cuda")
    loss2 = torch.empty(2, device="cuda")
    grad = torch.empty(n, 2, device="cuda")
    _hip.check(lib.uncl_l1_to_row(sd.data_ptr(), n, rd.data_ptr(), loss2.data_ptr(), grad.data_ptr(), _hip.stream_ptr()), "l1_to_row")
    np.testing.assert_allclose(loss2.cpu().numpy(), [l0.item(), l1.item()], rtol=1e-6, atol=1e-8)
    out = torch.empty(2, n, device="cuda")
    g0, g1 = torch.tensor([0.7], device="cuda"), torch.tensor([-1.3], device="cuda")
    _hip.check(lib.uncl_l1_to_row_backward(grad.data_ptr(), n, g0.data_ptr(), g1.data_ptr(), out.data_ptr(), _hip.stream_ptr()), "l1_to_row_bwd")
    np.testing.assert_allclose(out.t().cpu().numpy(), ref.grad.numpy(), rtol=1e-5, atol=1e-7)


def test_nce_with_longer_lists_golden_and_trainer_dispatch(golden):
    """nce() with several positives / negatives (GanTrainerImg.py:410-439) and lmcl_loss (:441-450): losses.nce_lists against the
    reference's own outputs, through the trainer method that used to refuse them; bf16 features against the oracle"""
    from nce_cases import NCE_LISTS_CASES, nce_lists_inputs
    from uncltmo_amd.trainer_img import GanTrainer
    g = golden("nce_lists")
    tr = GanTrainer.__new__(GanTrainer)
    for tag, shape, n_pos, n_neg, shared, k, c in NCE_LISTS_CASES:
        for form in ("InfoNCE", "LMCL"):
            an, pos, neg = [t.cuda() if torch.is_tensor(t) else [x.cuda() for x in t] for t in nce_lists_inputs(tag, shape, n_pos, n_neg, shared)]
            an.requires_grad_(True); pos[0].requires_grad_(True); neg[-1].requires_grad_(True)
            l = tr.nce(an, pos, neg, form, k, c)           # shared negatives stay single rows: no repeat needed here
            l.backward()
            key = "%s.%s" % (tag, form)
            np.testing.assert_allclose(l.item(), g[key], rtol=2e-5)
            assert rel_l2(an.grad.cpu(), torch.from_numpy(g[key + ".ga"])) < 1e-4, key
            assert rel_l2(pos[0].grad.cpu(), torch.from_numpy(g[key + ".gp0"])) < 1e-4, key
            assert rel_l2(neg[-1].grad.cpu(), torch.from_numpy(g[key + ".gn_last"])) < 1e-4, key
    # bf16 channel-last features, two positives and two negatives, one of each a row of the anchor itself
    fea = torch.rand(4, 8, 12, 12, generator=torch.Generator().manual_seed(7))
    oth = torch.rand(2, 4, 8, 12, 12, generator=torch.Generator().manual_seed(8))
    fb = fea.cuda().to(torch.bfloat16).requires_grad_(True)
    ob = oth.cuda().to(torch.bfloat16)
    lb = HL.nce_lists(fb, [ob[0], fb[2:3]], [ob[1], fb[1:2]], 1, 1e-2)
    fr = fb.detach().float().cpu().requires_grad_(True)
    orf = ob.float().cpu()
    lr_ = OL.nce_lists(fr, [orf[0], fr[2:3].repeat(4, 1, 1, 1)], [orf[1], fr[1:2].repeat(4, 1, 1, 1)], 1, 1e-2)
    np.testing.assert_allclose(lb.item(), lr_.item(), rtol=1e-4)
    lb.backward(); lr_.backward()
    assert fb.grad.dtype == torch.bfloat16 and rel_l2(fb.grad.float().cpu(), fr.grad) < 6e-3
    with pytest.raises(TypeError):
        tr.nce(fb, [ob[0], ob[1]], [ob[1]], "nope", 1, 1e-2)


def test_nce_golden_and_shared_rows(golden):
    g = golden("losses")
    for tag, shape, (k, c) in [("nce_d", (4, 2, 1, 1), (1, 1e-2)), ("nce_d2", (4, 2, 1, 1), (1e3, 2)),
                               ("nce_map", (3, 32, 16, 16), (1, 1e-2))]:
        n = int(np.prod(shape))
        an = torch.tensor(synth.hash_uniform(tag + "a", n)).reshape(shape).cuda().requires_grad_(True)
        po = torch.tensor(synth.hash_uniform(tag + "p", n)).reshape(shape).cuda()
        ne = torch.tensor(synth.hash_uniform(tag + "n", n)).reshape(shape).cuda()
        l = HL.nce(an, po, ne, k, c)
        l.backward()
        np.testing.assert_allclose(l.item(), g["loss." + tag], rtol=1e-5)
        assert rel_l2(an.grad.cpu(), torch.from_numpy(g["loss.%s.ga" % tag])) < 1e-4
    # infoNCE2 pattern: positive / negative are rows of the anchor tensor itself (gradient flows into those rows too)
    fea = torch.rand(4, 8, 12, 12, generator=torch.Generator().manual_seed(5))
    fc = fea.clone().requires_grad_(True)
    lo = OL.nce(fc, fc[2].unsqueeze(0).repeat(4, 1, 1, 1), fc[1].unsqueeze(0).repeat(4, 1, 1, 1), 1, 1e-2)
    lo.backward()
    fg = fea.cuda().requires_grad_(True)
    l = HL.nce(fg, fg[2:3], fg[1:2], 1, 1e-2)
    l.backward()
    np.testing.assert_allclose(l.item(), lo.item(), rtol=1e-5)
    assert rel_l2(fg.grad.cpu(), fc.grad) < 1e-4
    # bf16 channel-last features (what the generator hands over)
    fb = fea.cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    lb = HL.nce(fb, fb[2:3], fb[1:2], 1, 1e-2)
    fr = fb.detach().float().cpu().requires_grad_(True)
    lr_ = OL.nce(fr, fr[2].unsqueeze(0).repeat(4, 1, 1, 1), fr[1].unsqueeze(0).repeat(4, 1, 1, 1), 1, 1e-2)
    np.testing.assert_allclose(lb.item(), lr_.item(), rtol=1e-4)
    # ... and its gradient: one bf16 channel-last tensor holding the anchor's and the two shared rows' contributions,
    # scaled by the upstream factor on the device
    (0.37 * lb).backward()
    (0.37 * lr_).backward()
    assert fb.grad.dtype == torch.bfloat16 and fb.grad.permute(0, 2, 3, 1).is_contiguous()
    assert rel_l2(fb.grad.float().cpu(), fr.grad) < 6e-3
    # the same with the two row indices left on the device (what the trainer does with the naturalness arg-max / arg-min)
    fd = fb.detach().clone().requires_grad_(True)
    ld = HL.nce_rows(fd, torch.tensor([2, 1], dtype=torch.int32, device="cuda"), 1, 1e-2)
    (0.37 * ld).backward()
    assert ld.item() == lb.item() and torch.equal(fd.grad, fb.grad)
    # positive and negative the SAME row; separate shared rows that are not part of the anchor; per-sample rows
    for pr, qr in [(3, 3), (0, 3)]:
        f1 = fea.clone().requires_grad_(True)
        OL.nce(f1, f1[pr].unsqueeze(0).repeat(4, 1, 1, 1), f1[qr].unsqueeze(0).repeat(4, 1, 1, 1), 1, 1e-2).backward()
        f2 = fea.cuda().requires_grad_(True)
        HL.nce(f2, f2[pr:pr + 1], f2[qr:qr + 1], 1, 1e-2).backward()
        if pr == qr:       # the two logits are equal: loss = log 2 and every gradient cancels to rounding noise
            assert f2.grad.abs().max().item() < 1e-5 and f1.grad.abs().max().item() < 1e-5
        else:
            assert rel_l2(f2.grad.cpu(), f1.grad) < 1e-4
    a1, p1, q1 = [torch.rand(s_, 8, 12, 12, generator=torch.Generator().manual_seed(60 + i)).requires_grad_(True)
                  for i, s_ in enumerate((4, 1, 4))]
    OL.nce(a1, p1.repeat(4, 1, 1, 1), q1, 2.0, 0.5).backward()
    a2, p2, q2 = [t.detach().cuda().requires_grad_(True) for t in (a1, p1, q1)]
    HL.nce(a2, p2, q2, 2.0, 0.5).backward()
    for t2, t1 in ((a2, a1), (p2, p1), (q2, q1)):
        assert rel_l2(t2.grad.cpu(), t1.grad) < 1e-4


def test_tmqi_naturalness_golden_and_selection(golden):
    g = golden("losses")
    fr = torch.cat([synth.smooth_hdr_frames(3, salt="tmq"), synth.ldr_frames(1, salt="tmq2")], 0).cuda()
    s, bw = HL.tmqi_naturalness(fr)
    np.testing.assert_allclose(s.cpu().numpy(), g["tmqi_n"][0::2], rtol=1e-6, atol=1e-12)
    sp, bwp = HL.tmqi_naturalness(fr, patch=128)
    np.testing.assert_allclose(sp.cpu().numpy()[1::4], g["tmqi_n"][1::2], rtol=1e-6, atol=1e-12)   # patch (0,1) of each frame
    ref = [float(v) for v in s.cpu()]
    assert bw.cpu().tolist() == [ref.index(max(ref)), ref.index(min(ref))]


def test_frame_stats_l1_and_tv_vs_oracle(golden):
    g = golden("losses")
    fk = synth.smooth_hdr_frames(2, salt="plf").cuda().requires_grad_(True)
    ld = synth.ldr_frames(2, salt="pll").cuda()
    m_f, v_f = HL.frame_stats(fk)
    m_p, v_p = HL.frame_stats(ld)
    lc = HL.l1_mean(v_f, v_p)
    (lc + 0.5 * HL.l1_mean(m_f, m_p)).backward()
    np.testing.assert_allclose(lc.item(), g["loss.contrast_l1"], rtol=2e-4)
    fc = synth.smooth_hdr_frames(2, salt="plf").requires_grad_(True)
    lm, lcon = OL.brightness_contrast_l1(fc, ld.cpu())
    (lcon + 0.5 * lm).backward()
    assert rel_l2(fk.grad.cpu(), fc.grad) < 2e-4
    f2 = synth.ldr_frames(2, 32, 48, salt="tv").cuda().requires_grad_(True)
    l = HL.tv_loss(f2)
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss.tv"], rtol=1e-5)
    np.testing.assert_allclose(f2.grad.cpu().numpy(), g["loss.tv.g"], rtol=1e-4, atol=1e-9)


def test_adam_step_matches_torch():
    import ctypes as C
    gen = torch.Generator().manual_seed(0)
    shapes = [(5,), (3, 7), (2, 4, 3, 3)]
    ps = [torch.randn(s, generator=gen) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3, betas=(0.5, 0.999))
    dev = [p.cuda() for p in ps]
    m = [torch.zeros_like(p) for p in dev]
    v = [torch.zeros_like(p) for p in dev]
    for step in range(1, 4):
        gs = [torch.randn(s, generator=gen) for s in shapes]
        for r, gg in zip(ref, gs):
            r.grad = gg.clone()
        opt.step()
        gd = [gg.cuda() for gg in gs]
        arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        n = (C.c_int * len(dev))(*[t.numel() for t in dev])
        _hip.check(_hip.lib().uncl_adam_step(arr(dev), arr(gd), arr(m), arr(v), n, len(dev), 1e-3, 0.5, 0.999, 1e-8, step,
                                             _hip.stream_ptr()), "adam")
        torch.cuda.synchronize()
    for r, d in zip(ref, dev):
        np.testing.assert_allclose(d.cpu().numpy(), r.detach().numpy(), rtol=1e-5, atol=1e-7)


# ---- PatchGAN (forward-parity module) ------------------------------------------------------
def test_patch_d_forward_matches_reference_golden(golden):
    from uncltmo_amd import model_factory
    g = golden("disc")
    p = model_factory.create_D_net(1, 16, torch.device("cuda"), False, "instance_norm", False, "patchD", 3, "none", 1, 0, 0, 0)
    assert [k for k, _, _ in state_spec.patch_d_spec()] == list(p.state_dict().keys())
    synth.fill_state_dict(p, "p0")
    x = torch.cat([synth.ldr_frames(2, salt="dA"), synth.smooth_hdr_frames(1, salt="dB")], 0).cuda()
    with torch.no_grad():
        o = p(x)
    assert o.shape == (3, 1, 30, 30)
    assert rel_l2(o.cpu(), torch.from_numpy(g["patchd.output"])) < 1e-4
    # other sizes / batch: against the oracle
    x2 = synth.smooth_hdr_frames(2, 144, 144, salt="dP").cuda()
    with torch.no_grad():
        o2 = p(x2)
    ref = OD.patch_d_forward({k: v.cpu() for k, v in p.state_dict().items()}, x2.cpu())
    assert o2.shape == ref.shape and rel_l2(o2.cpu(), ref) < 1e-4


def _patch_d():
    from uncltmo_amd import model_factory
    p = model_factory.create_D_net(1, 16, torch.device("cuda"), False, "instance_norm", False, "patchD", 3, "none", 1, 0, 0, 0)
    synth.fill_state_dict(p, "p0")
    return p


def test_patch_d_backward_matches_reference_golden(golden):
    """Least-squares GAN loss through the PatchGAN: loss, every parameter gradient and the input gradient against vectors
    captured from the reference's module under torch autograd (make_golden.py capture_patchd_grad)."""
    g = golden("patchd_grad")
    p = _patch_d()
    x = torch.cat([synth.ldr_frames(1, salt="dA"), synth.smooth_hdr_frames(1, salt="dB")], 0).cuda().requires_grad_(True)
    o = p(x)
    loss = ((o - 1.0) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    named = dict(p.named_parameters())
    for key, t in [("input", x.grad)] + [(k, v.grad) for k, v in named.items()]:
        assert t is not None, key
        flat = t.detach().reshape(-1).cpu()
        pos, val = g["g.%s.pos" % key], g["g.%s.val" % key]
        assert list(g["g.%s.shape" % key]) == list(t.shape)
        got = flat[torch.from_numpy(pos)].numpy()
        scale = np.abs(val).max() + 1e-30
        assert np.abs(got - val).max() / scale < 2e-4, key
        assert abs(flat.double().sum().item() - float(g["g.%s.sum" % key])) < 2e-4 * float(g["g.%s.abssum" % key]) + 1e-12, key


def test_patch_d_backward_matches_oracle_other_shapes():
    p = _patch_d()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in p.state_dict().items()}
    for n, h, salt in [(1, 64, "pA"), (3, 144, "pB")]:
        x = synth.smooth_hdr_frames(n, h, h, salt=salt)
        xg = x.cuda().requires_grad_(True)
        for q in p.parameters():
            q.grad = None
        o = p(xg)
        w = torch.linspace(-0.5, 1, o.numel(), device="cuda").reshape(o.shape)
        (o * w).sum().backward()
        xc = x.clone().requires_grad_(True)
        for v in sd.values():
            v.grad = None
        oc = OD.patch_d_forward(sd, xc)
        (oc * w.cpu()).sum().backward()
        assert rel_l2(o.detach().cpu(), oc.detach()) < 1e-4
        assert rel_l2(xg.grad.cpu(), xc.grad) < 1e-3
        for k, q in p.named_parameters():
            assert rel_l2(q.grad.cpu(), sd[k].grad) < 1e-3, (k, n, h)
    # deterministic: a second backward of the same input gives the same bits
    a = [q.grad.clone() for q in p.parameters()]
    for q in p.parameters():
        q.grad = None
    o = p(xg)
    (o * w).sum().backward()
    assert all(torch.equal(u, q.grad) for u, q in zip(a, p.parameters()))
    # x without grad and frozen parameters: plain forward path
    for q in p.parameters():
        q.requires_grad_(False)
    assert not p(x.cuda()).requires_grad
