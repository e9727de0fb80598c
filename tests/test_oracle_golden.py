"""Pin the CPU oracle against golden vectors captured from the upstream reference (no GPU needed)."""
import numpy as np
import pytest
import torch

from conftest import check_summary, synth_state
from oracle import discriminator as OD
from oracle import generator as OG
from oracle import losses as OL
from oracle import tiler as OT
from oracle import tmqi as OTM
from uncltmo_amd import state_spec, synth


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def sdG():
    return synth_state(state_spec.generator_spec(), "g0")


@pytest.fixture(scope="module")
def sdD():
    return synth_state(state_spec.simple_d_spec(), "d0")


def test_relative_pos_and_windows(golden):
    g = golden("generator")
    np.testing.assert_allclose(OG.sincos_relative_pos().numpy(), g["relative_pos"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(OG.gauss_window().numpy(), golden("losses")["gauss_window"])


def test_generator_eval(golden, sdG):
    g = golden("generator")
    x = torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)
    want = {}
    with torch.no_grad():
        y, up = OG.unet_image_forward(sdG, x, want=want)
    assert rel_l2(y, g["g_eval.x_out"]) < 1e-5
    np.testing.assert_array_equal(want["knn_idx"].numpy(), g["g_eval.knn_idx"])
    check_summary(up, g, "g_eval.up_x")
    for name in ["inc", "down0", "down1", "down2", "down3", "gcn", "up0", "up1", "up2", "up3"]:
        check_summary(want[name], g, "g_eval." + name)


def test_generator_train_mask(golden, sdG):
    g = golden("generator")
    x = torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)
    keep = torch.tensor([[1.0, 0.0], [1.0, 0.0]])
    with torch.no_grad():
        y, up = OG.unet_image_forward(sdG, x, training=True, drop_keep=keep)
    check_summary(y, g, "g_train.x_out")
    check_summary(up, g, "g_train.up_x")


def test_generator_rejects_other_sizes(golden, sdG):
    g = golden("generator")
    for hw in [(268, 268), (512, 512), (256, 512)]:
        assert g["g_err.%dx%d" % hw] == 1
        with pytest.raises(ValueError):
            OG.unet_image_forward(sdG, torch.zeros(1, 1, *hw))


def test_video_generator(golden, sdG):
    g = golden("video")
    x = torch.cat([synth.smooth_hdr_frames(1, salt="v%d" % t) for t in range(3)], 0).unsqueeze(0)
    with torch.no_grad():
        y, f = OG.unet_video_forward(sdG, x)
    np.testing.assert_allclose(f.numpy(), g["v_eval.feats"], rtol=1e-4, atol=1e-6)
    for t in range(3):
        check_summary(y[:, t], g, "v_eval.frame%d" % t)


def test_discriminators(golden, sdD):
    g = golden("disc")
    x = torch.cat([synth.ldr_frames(2, salt="dA"), synth.smooth_hdr_frames(1, salt="dB")], 0)
    with torch.no_grad():
        o, f = OD.simple_d_forward(sdD, x)
        po = OD.patch_d_forward(synth_state(state_spec.patch_d_spec(), "p0"), x)
    np.testing.assert_allclose(o.numpy(), g["d.output"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f.numpy(), g["d.fea_final"], rtol=1e-5, atol=1e-7)
    assert rel_l2(po, g["patchd.output"]) < 1e-5


def test_losses(golden):
    g = golden("losses")
    a = torch.tensor(synth.hash_uniform("la", 4) * 4 - 2).reshape(4, 1).requires_grad_(True)
    b = torch.tensor(synth.hash_uniform("lb", 4) * 4 - 2).reshape(4, 1).requires_grad_(True)
    l = OL.contrastive_d_loss(a, b)
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss.cgan"], rtol=1e-6)
    np.testing.assert_allclose(a.grad.numpy(), g["loss.cgan.ga"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b.grad.numpy(), g["loss.cgan.gb"], rtol=1e-5, atol=1e-7)
    for tag, shape, (k, c) in [("nce_d", (4, 2, 1, 1), (1, 1e-2)), ("nce_d2", (4, 2, 1, 1), (1e3, 2)),
                               ("nce_map", (3, 32, 16, 16), (1, 1e-2))]:
        n = int(np.prod(shape))
        an = torch.tensor(synth.hash_uniform(tag + "a", n)).reshape(shape).requires_grad_(True)
        po = torch.tensor(synth.hash_uniform(tag + "p", n)).reshape(shape)
        ne = torch.tensor(synth.hash_uniform(tag + "n", n)).reshape(shape)
        l = OL.nce(an, po, ne, k, c)
        l.backward()
        np.testing.assert_allclose(l.item(), g["loss." + tag], rtol=1e-6)
        np.testing.assert_allclose(an.grad.numpy(), g["loss.%s.ga" % tag], rtol=1e-4, atol=1e-8)
    fake = synth.ldr_frames(2, 64, 64, salt="slf").requires_grad_(True)
    hdr = synth.smooth_hdr_frames(2, 64, 64, salt="slh")
    np.testing.assert_allclose(OL.struct_loss_level(fake, hdr).item(), g["loss.struct_level"], rtol=1e-6)
    lp = OL.struct_loss_pyramid(fake, hdr, [1.0, 1.0, 1.0])
    lp.backward()
    np.testing.assert_allclose(lp.item(), g["loss.struct_pyr"], rtol=1e-6)
    np.testing.assert_allclose(fake.grad.numpy(), g["loss.struct_pyr.gfake"], rtol=1e-4, atol=1e-7)
    f2 = synth.ldr_frames(2, 32, 48, salt="tv").requires_grad_(True)
    l = OL.tv_loss(f2)
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss.tv"], rtol=1e-6)
    np.testing.assert_allclose(f2.grad.numpy(), g["loss.tv.g"], rtol=1e-5, atol=1e-9)


def test_tmqi_naturalness(golden):
    g = golden("losses")
    fr = torch.cat([synth.smooth_hdr_frames(3, salt="tmq"), synth.ldr_frames(1, salt="tmq2")], 0).numpy()
    sc = []
    for i in range(4):
        sc.append(OTM.naturalness(fr[i, 0] * 255))
        sc.append(OTM.naturalness(fr[i, 0, :128, 128:] * 255))
    np.testing.assert_allclose(np.array(sc), g["tmqi_n"], rtol=1e-6, atol=1e-12)


def test_pseudo_label_and_contrast(golden):
    g = golden("losses")
    fk = synth.smooth_hdr_frames(2, salt="plf").requires_grad_(True)
    l = OL.pseudo_label_loss(fk)
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss.pseudo"], rtol=1e-5)
    check_summary(fk.grad, g, "loss.pseudo.g", rtol=1e-4, atol=1e-10)
    fk2 = synth.smooth_hdr_frames(2, salt="plf").requires_grad_(True)
    _, lc = OL.brightness_contrast_l1(fk2, synth.ldr_frames(2, salt="pll"))
    lc.backward()
    np.testing.assert_allclose(lc.item(), g["loss.contrast_l1"], rtol=1e-5)
    check_summary(fk2.grad, g, "loss.contrast_l1.g", rtol=1e-4, atol=1e-10)


def _standin(p, **kw):
    yy = torch.arange(256.0).reshape(*([1] * (p.dim() - 2)), 256, 1) / 255.0
    xx = torch.arange(256.0).reshape(*([1] * (p.dim() - 1)), 256) / 255.0
    return p * (0.5 + xx + 2.0 * yy), None


def test_tiler(golden, sdG):
    g = golden("tiler")
    x = synth.hdr_frames(1, 272, 272, salt="tile272")
    np.testing.assert_allclose(OT.tiled_forward(x, _standin).numpy(), g["tiler.standin.272x272"], rtol=1e-6, atol=1e-7)
    x = synth.hdr_frames(1, 400, 528, salt="tile400")
    check_summary(OT.tiled_forward(x, _standin), g, "tiler.standin.400x528", rtol=1e-6, atol=1e-7)
    x5 = synth.hdr_frames(2, 300, 272, salt="tile5").reshape(1, 2, 1, 300, 272)
    check_summary(OT.tiled_forward(x5, _standin), g, "tiler.standin5.300x272", rtol=1e-6, atol=1e-7)
    x = synth.smooth_hdr_frames(1, 272, 272, salt="tileG")
    y = OT.tiled_forward(x, lambda p: OG.unet_image_forward(sdG, p))
    assert rel_l2(y, g["tiler.realG.272"]) < 1e-5
    with pytest.raises(ValueError):
        OT.axis_plan(256)
    assert OT.tile_count(1024, 1024) == 25 and OT.tile_count(2160, 3840) == 220


def test_generator_variants_state_dict_and_oracle_forward_backward(golden):
    """skip operators original_unet / square / square_root (unet_parts.py:311-332) and the bilinear decoder path (:256-259):
    state_spec's keys and shapes are the reference's state_dict, the oracle's forward and parameter gradients are the reference's"""
    from generator_variants import GENERATOR_VARIANTS
    from uncltmo_amd import params
    g = golden("generator_variants")
    x = torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)
    wy = 0.5 + synth.smooth_hdr_frames(2, salt="bwy")
    for tag, op, bil, upm in GENERATOR_VARIANTS:
        spec = state_spec.generator_spec(32, params.get_layer_factor(op), "none", bil, upm)
        assert [k for k, _, _ in spec] == list(g[tag + ".keys"])
        assert [",".join(str(d) for d in s_) for _, s_, _ in spec] == list(g[tag + ".shapes"])
        sd = {k: v.clone().requires_grad_(not k.endswith("relative_pos")) for k, v in synth_state(spec, "g0").items()}
        y, up = OG.unet_image_forward(sd, x, con_operator=op)
        check_summary(y, g, tag + ".x_out", rtol=2e-4, atol=1e-6)
        check_summary(up, g, tag + ".up_x", rtol=2e-4, atol=1e-5)
        ((y * wy).sum() + 1e-3 * up.sum()).backward()
        for k, v in sd.items():
            if v.grad is None:
                continue
            gr = v.grad.double().reshape(-1)
            ref_n = float(g["%s.grad.%s" % (tag, k)])
            assert abs(gr.norm().item() - ref_n) <= 2e-3 * ref_n + 1e-9, (tag, k, gr.norm().item(), ref_n)


def test_nce_with_longer_lists_matches_reference_golden(golden):
    """nce() with several positives / negatives and lmcl_loss (GanTrainerImg.py:410-450): oracle/losses.py:nce_lists against the
    reference's own method (tests/golden/make_golden.py nce_lists)"""
    from nce_cases import NCE_LISTS_CASES, nce_lists_inputs
    g = golden("nce_lists")
    for tag, shape, n_pos, n_neg, shared, k, c in NCE_LISTS_CASES:
        for form in ("InfoNCE", "LMCL"):
            an, pos, neg = nce_lists_inputs(tag, shape, n_pos, n_neg, shared)
            an.requires_grad_(True); pos[0].requires_grad_(True); neg[-1].requires_grad_(True)
            l = OL.nce_lists(an, pos, [f.repeat(shape[0], 1, 1, 1) if shared else f for f in neg], k, c, form)
            l.backward()
            key = "%s.%s" % (tag, form)
            np.testing.assert_allclose(l.item(), g[key], rtol=1e-6)
            np.testing.assert_allclose(an.grad.numpy(), g[key + ".ga"], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(pos[0].grad.numpy(), g[key + ".gp0"], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(neg[-1].grad.numpy(), g[key + ".gn_last"], rtol=1e-4, atol=1e-8)


def inference_inputs():
    """Same synthetic frame / stand-in generator output as tests/golden/make_golden.py:inference_inputs."""
    rgb = torch.from_numpy(synth.hash_uniform("inf_rgb", 3 * 300 * 280).reshape(3, 300, 280).copy()).float() ** 4 * 1000 - 0.01
    fake = torch.from_numpy(synth.hash_uniform("inf_fake", 304 * 288).reshape(1, 1, 304, 288).copy()).float() ** 2
    return rgb, 1275.0, fake


def test_inference_pre_and_post_processing(golden):
    from oracle import inference as OI
    g = golden("inference")
    rgb, f, fake = inference_inputs()
    rgb_s, gray = OI.hdr_log_gray(rgb, f)
    check_summary(gray, g, "inf.gray_log", rtol=1e-6, atol=1e-7)
    rgb_p, dY, dX = OI.resize_im(rgb_s)
    gray_p, _, _ = OI.resize_im(gray)
    assert [dY, dX] == list(g["inf.diff"])
    check_summary(rgb_p, g, "inf.rgb_padded", rtol=1e-6, atol=1e-7)
    check_summary(gray_p, g, "inf.gray_padded", rtol=1e-6, atol=1e-7)
    col = OI.finish(rgb_p, fake, dY, dX)
    check_summary(col, g, "inf.color", rtol=1e-6, atol=1e-7)
    im = OI.to_uint8(col)
    assert int(im.astype(np.int64).sum()) == int(g["inf.uint8.sum"])
    assert np.array_equal(im.reshape(-1)[g["inf.uint8.pos"]], g["inf.uint8.val"])


def tmqi_inputs(h, w, salt):
    """Same synthetic HDR / LDR pair as tests/golden/make_golden.py:tmqi_inputs."""
    hdr = synth.smooth_hdr_frames(1, h, w, salt="tmqi_h" + salt)[0, 0].double().numpy() ** 3 * 4000.0 + 0.05
    noise = synth.hash_uniform("tmqi_n" + salt, h * w).reshape(h, w).astype(np.float64)
    ldr = 255.0 * np.clip((np.log10(hdr) - np.log10(hdr.min())) / (np.log10(hdr.max()) - np.log10(hdr.min())) * 0.9 + 0.04 * noise, 0, 1)
    return hdr, ldr


def test_full_tmqi_matches_reference_class(golden):
    from oracle import tmqi as OT
    g = golden("tmqi")
    for (h, w), salt in (((256, 256), "a"), ((200, 176), "b")):
        hdr, ldr = tmqi_inputs(h, w, salt)
        Q, S, N, sl = OT.tmqi(hdr, ldr)
        np.testing.assert_allclose([Q, S, N], g["tmqi.%s.QSN" % salt], rtol=1e-9)
        np.testing.assert_allclose(sl, g["tmqi.%s.s_local" % salt], rtol=1e-9)


def test_tmqi_per_level_maps_match_reference_class(golden):
    """the reference's fifth return value, `s_maps` (TMQI.py:152-157): shapes, sums and hashed samples of the five maps"""
    from oracle import tmqi as OT
    g = golden("tmqi_maps")
    for (h, w), salt in (((256, 256), "a"), ((200, 176), "b")):
        hdr, ldr = tmqi_inputs(h, w, salt)
        maps = []
        _, _, _, sl = OT.tmqi(hdr, ldr, maps=maps)
        assert len(maps) == 5
        for l, m in enumerate(maps):
            assert m.shape == ((h >> l) - 10, (w >> l) - 10)
            check_summary(torch.from_numpy(np.ascontiguousarray(m)), g, "tmqi.%s.map%d" % (salt, l), rtol=1e-6, atol=1e-7)   # (samples are stored as fp32)
            np.testing.assert_allclose(m.mean(), sl[l], rtol=1e-12)


def test_loader_hdr_branch_oracle_vs_reference_golden(golden):
    """oracle/data_loader.py's HDR branch against the reference's own npy_loader output (make_golden.py loader)"""
    from oracle import data_loader as OD
    from uncltmo_amd import synth
    g = golden("loader")
    u = synth.hash_uniform("loader_hdr", 256 * 256 * 3).astype(np.float64) ** 4
    arr = (u * 4000.0 + 0.01).astype(np.float32).reshape(256, 256, 3)
    bf = float(g["loader.hdr.frame0.bf"])
    assert abs(bf - 0.37 * 255 * 0.1) < 1e-9          # get_f (ProcessedDatasetFolderImg.py:27-36)
    out = OD.frame(arr, (256, 256, 0, 0), True, brightness_factor=bf)
    check_summary(torch.from_numpy(out["input"]), g, "loader.hdr.frame0.input", rtol=2e-5, atol=2e-6)
    check_summary(torch.from_numpy(out["gray_norm"]), g, "loader.hdr.frame0.gray_norm", rtol=2e-6, atol=1e-7)
    check_summary(torch.from_numpy(out["gray"]), g, "loader.hdr.frame0.gray", rtol=2e-6, atol=1e-4)
    check_summary(torch.from_numpy(out["color"]), g, "loader.hdr.frame0.color", rtol=0, atol=0)


def clip_inputs():
    """the frames of tests/golden/make_golden.py:tester_inputs (same hash generator)"""
    base = synth.smooth_hdr_frames(1, 300, 340, salt="tst_base")[0, 0].numpy().astype(np.float32)
    frames = []
    for t in range(3):
        tex = synth.hash_uniform("tst_f%d" % t, 300 * 340 * 3).reshape(300, 340, 3).astype(np.float32)
        frames.append(((base[:, :, None] * (1.0 + 0.1 * t)) ** 3 * 40.0 * (0.7 + 0.3 * tex)).astype(np.float32))
    return frames, 0.5


def clip_standin(p, apply_crop=True, diffY=0, diffX=0):
    c = p.clamp_min(0) ** 0.6
    prev = torch.cat([c[:, :1], c[:, :-1]], 1)
    return 0.05 + 0.75 * c + 0.15 * prev, None


def check_ldr_frames(results, g, tag, max_bad=0.0):
    for i, im in enumerate(results):
        im = np.asarray(im)
        assert tuple(im.shape) == tuple(g["tester.%s.ldr%d.shape" % (tag, i)])
        bad = (im.reshape(-1)[g["tester.%s.ldr%d.pos" % (tag, i)]].astype(np.int64) - g["tester.%s.ldr%d.val" % (tag, i)].astype(np.int64))
        assert np.abs(bad).max() <= (0 if max_bad == 0.0 else 1), (tag, i, np.abs(bad).max())
        assert (bad != 0).mean() <= max_bad, (tag, i, (bad != 0).mean())
        if max_bad == 0.0:
            assert int(im.astype(np.int64).sum()) == int(g["tester.%s.ldr%d.sum" % (tag, i)])


def test_tester_eval_on_video_oracle_vs_reference_golden(golden, sdG):
    """oracle/tester.py against the fixture captured through the reference's own Tester.eval_on_video (Tester.py:314-391): the
    8-bit frames bit for bit, the warp errors, and -- with the structure-preserving stand-in generator -- the scene's TMQI."""
    from oracle import tester as OTS
    import oracle.tester
    g = golden("tester")
    frames, lam = clip_inputs()
    f_factor = float(g["tester.f_factor"][0])
    assert f_factor == lam * 255 * 0.1
    ident = lambda f1, f0: f1
    # stand-in generator: monkey-patch the oracle's model call through its tiler hook
    orig = oracle.tester.unet_video_forward
    try:
        oracle.tester.unet_video_forward = lambda sd, x: clip_standin(x)
        scene, res, scores, mse, rel = OTS.eval_on_video(None, frames, f_factor, align=ident)
    finally:
        oracle.tester.unet_video_forward = orig
    check_ldr_frames(res, g, "tone")
    np.testing.assert_allclose([scene, mse, rel], g["tester.tone.scores"], rtol=1e-6)
    scene, res, scores, mse, rel = OTS.eval_on_video(sdG, frames, f_factor, align=ident)
    check_ldr_frames(res, g, "G")
    assert np.isnan(scene) and np.isnan(g["tester.G.scores"][0])     # the reference's own TMQI is NaN on this pairing (fixture note)
    np.testing.assert_allclose([mse, rel], g["tester.G.scores"][1:], rtol=1e-6)


# ---- unet_norm='batch_norm' (unet_parts.py:20-21, 34-35): goldens from the reference built with that flag --------------------------
@pytest.fixture(scope="module")
def sdG_bn():
    return synth.bnorm_state(synth_state(state_spec.generator_spec(unet_norm="batch_norm"), "g0"))


def test_batch_norm_state_dict_layout(golden):
    """key names, order and shapes of the reference's state_dict with nn.BatchNorm2d behind every 3x3 convolution: the checkpoint
    contract (strict load) -- state_spec, and the HIP module built from it"""
    g = golden("generator_bnorm")
    spec = state_spec.generator_spec(unet_norm="batch_norm")
    assert [k for k, _, _ in spec] == list(g["bnorm.keys"])
    assert [",".join(str(d) for d in s) for _, s, _ in spec] == list(g["bnorm.shapes"])
    from uncltmo_amd.generator import UNet
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "batch_norm", "none", "relu", 1, "replicate", 2, 0)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["bnorm.keys"])
    assert [",".join(str(d) for d in v.shape) for v in sd.values()] == list(g["bnorm.shapes"])
    assert sd["inc.conv.norm.num_batches_tracked"].dtype == torch.long
    assert len(state_spec.batch_norm_layers()) == 18


def test_generator_batch_norm_eval(golden, sdG_bn):
    g = golden("generator_bnorm")
    x = torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)
    with torch.no_grad():
        y, up = OG.unet_image_forward(sdG_bn, x, unet_norm="batch_norm")
    assert rel_l2(y, g["bnorm.x_out"]) < 1e-5
    check_summary(up, g, "bnorm.up_x")


def test_generator_batch_norm_training_statistics(golden):
    """training mode: batch statistics in the forward, running statistics updated with momentum 0.1 (unbiased variance) and the
    batch counter advanced -- the oracle leaves the same state behind as the reference module"""
    g = golden("generator_bnorm")
    sd = synth.bnorm_state(synth_state(state_spec.generator_spec(unet_norm="batch_norm"), "g0"))
    x = torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)
    keep = torch.tensor([[1.0, 0.0], [1.0, 0.0]])
    with torch.no_grad():
        y, up = OG.unet_image_forward(sd, x, unet_norm="batch_norm", training=True, drop_keep=keep)
    check_summary(y, g, "bnorm_train.x_out")
    check_summary(up, g, "bnorm_train.up_x")
    n = 0
    for k in list(g):
        if k.startswith("bnorm_train.after."):
            np.testing.assert_allclose(sd[k[len("bnorm_train.after."):]].double().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)
            n += 1
    assert n == 54
