"""GPU parity of the inference pre- / post-processing against goldens captured from the
reference's own functions, numpy, and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from hip_util import rel_l2
from uncltmo_amd import frame_util, inference, synth

pytestmark = pytest.mark.gpu


def inference_inputs():
    rgb = torch.from_numpy(synth.hash_uniform("inf_rgb", 3 * 300 * 280).reshape(3, 300, 280).copy()).float() ** 4 * 1000 - 0.01
    fake = torch.from_numpy(synth.hash_uniform("inf_fake", 304 * 288).reshape(1, 1, 304, 288).copy()).float() ** 2
    return rgb, 1275.0, fake


def test_pre_processing_matches_reference_golden(golden):
    g = golden("inference")
    rgb, f, _ = inference_inputs()
    rgb_s, gray = frame_util.hdr_log_gray(rgb.cuda(), f)
    assert float(rgb_s.min()) == 0.0 and gray.shape == (1, 300, 280)
    check_summary(gray, g, "inf.gray_log", rtol=2e-6, atol=2e-7)      # device log10f vs torch's CPU log10: last-bit differences
    rgb_p, dY, dX = frame_util.resize_im(rgb_s, 1, 0)
    gray_p, _, _ = frame_util.resize_im(gray, 1, 0)
    assert [dY, dX] == list(g["inf.diff"])
    check_summary(rgb_p, g, "inf.rgb_padded", rtol=0, atol=0)         # shift + replicate padding: bit-exact
    check_summary(gray_p, g, "inf.gray_padded", rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 304 * 288, 3 * 1024 * 1024 + 5])
def test_percentile_is_exactly_numpy(n):
    x = torch.from_numpy(synth.hash_uniform("pct%d" % n, n).copy()).float()
    x = (x - 0.3) * 7.0                                      # negative and positive values
    if n > 100:
        x[::3] = x[1]                                        # many ties
        x[5] = float("inf")
        x[6] = -0.0
    qs = [0.0, 0.1, 0.5, 37.3, 50.0, 99.0, 99.5, 100.0]
    got = frame_util.percentile(x.cuda(), qs)
    ref = [np.percentile(x.numpy(), q) for q in qs]
    for q, a, b in zip(qs, got, ref):
        assert a == b or (np.isnan(a) and np.isnan(b)), (n, q, a, b)
    # the same interpolation finished on the device (no value crosses to the host): bit-identical
    dev = frame_util.percentile(x.cuda(), qs, on_device=True)
    assert dev.is_cuda and dev.dtype == torch.float32
    for q, a, b in zip(qs, dev.cpu().numpy(), ref):
        assert a == np.float32(b) or (np.isnan(a) and np.isnan(b)), (n, q, a, b)


def test_post_processing_matches_reference_golden(golden):
    g = golden("inference")
    rgb, f, fake = inference_inputs()
    rgb_s, gray = frame_util.hdr_log_gray(rgb.cuda(), f)
    rgb_p, dY, dX = frame_util.resize_im(rgb_s, 1, 0)
    min_p, max_p = frame_util.percentile(fake.cuda(), [0.5, 99.5])
    assert [float(min_p), float(max_p)] == [float(v) for v in g["inf.percentiles"]]
    col = frame_util.back_to_color_and_crop(rgb_p, fake.cuda(), min_p, max_p, dY, dX)
    check_summary(col, g, "inf.color", rtol=1e-6, atol=1e-7)
    im = frame_util.to_uint8_outlier(col).cpu().numpy()
    # device-resident percentiles through the same two stages: identical results
    lohi = frame_util.percentile(fake.cuda(), [0.5, 99.5], on_device=True)
    col_d = frame_util.back_to_color_and_crop(rgb_p, fake.cuda(), lohi, None, dY, dX)
    assert torch.equal(col_d, col)
    assert np.array_equal(frame_util.to_uint8_outlier(col, on_device=True).cpu().numpy(), im)
    assert im.shape == (300, 280, 3) and im.dtype == np.uint8
    d = im.reshape(-1)[g["inf.uint8.pos"]].astype(np.int64) - g["inf.uint8.val"].astype(np.int64)
    # truncation to 8 bits: a last-bit difference of the fp32 stretch can move a value across an integer boundary
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 2e-3, (np.abs(d).max(), (d != 0).mean())
    assert abs(int(im.astype(np.int64).sum()) - int(g["inf.uint8.sum"])) <= 2e-3 * im.size


def test_whole_frame_vs_oracle_pipeline():
    """rgb -> log luminance -> pad -> tiled generator (fp32 parity mode) -> percentile clamp -> colour -> 8 bit, against the
    CPU oracle running the same stages (4 tiles)."""
    from oracle import inference as OI
    from oracle import tiler as OT
    from oracle.generator import unet_image_forward
    from uncltmo_amd.generator import UNet
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
               "replicate", 2, 0, compute_dtype="fp32")
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    rgb = synth.smooth_hdr_frames(3, 300, 280, salt="inf_e2e").reshape(3, 300, 280) * 50.0 + 0.01
    col, im = inference.run_model_on_frame(net, rgb.cuda(), 1275.0)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    rgb_s, gray = OI.hdr_log_gray(rgb, 1275.0)
    rgb_p, dY, dX = OI.resize_im(rgb_s)
    gray_p, _, _ = OI.resize_im(gray)
    with torch.no_grad():
        fake = OT.tiled_forward(gray_p.unsqueeze(0), lambda t: unet_image_forward(sd, t))
    ref = OI.finish(rgb_p, fake, dY, dX)
    assert col.shape == ref.shape == (3, 300, 280)
    assert rel_l2(col.cpu(), ref) < 1e-4
    ref_im = OI.to_uint8(ref)
    d = im.cpu().numpy().astype(np.int64) - ref_im.astype(np.int64)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 2e-2


def test_arguments_are_checked():
    from uncltmo_amd import _hip
    with pytest.raises(_hip.HipError):
        frame_util.hdr_log_gray(torch.zeros(3, 8, 8), 10.0)          # host tensor: no CPU path
    with pytest.raises(_hip.HipError):
        frame_util.hdr_log_gray(torch.zeros(3, 8, 8).cuda(), -1.0)   # UNCL_ERR_ARG


# ---- full TMQI on device ----------------------------------------------------------
def tmqi_inputs(h, w, salt):
    hdr = synth.smooth_hdr_frames(1, h, w, salt="tmqi_h" + salt)[0, 0].double().numpy() ** 3 * 4000.0 + 0.05
    noise = synth.hash_uniform("tmqi_n" + salt, h * w).reshape(h, w).astype(np.float64)
    ldr = 255.0 * np.clip((np.log10(hdr) - np.log10(hdr.min())) / (np.log10(hdr.max()) - np.log10(hdr.min())) * 0.9 + 0.04 * noise, 0, 1)
    return hdr, ldr


@pytest.mark.parametrize("shape,salt", [((256, 256), "a"), ((200, 176), "b")])
def test_full_tmqi_matches_reference_golden(golden, shape, salt):
    from uncltmo_amd.tmqi import TMQI
    g = golden("tmqi")
    hdr, ldr = tmqi_inputs(shape[0], shape[1], salt)
    Q, S, N, sl, maps = TMQI()(torch.from_numpy(hdr).float().cuda(), torch.from_numpy(ldr).float().cuda())
    # the device takes fp32 images (the goldens were computed from the float64 arrays): 1e-5 covers the input rounding
    np.testing.assert_allclose([Q, S, N], g["tmqi.%s.QSN" % salt], rtol=2e-5)
    np.testing.assert_allclose(sl, g["tmqi.%s.s_local" % salt], rtol=2e-5)
    # the per-level maps (the reference's `s_maps`): against the reference's own maps, and consistent with the level means
    from conftest import check_summary
    gm = golden("tmqi_maps")
    assert len(maps) == 5
    for l, m in enumerate(maps):
        assert m.dtype == torch.float64 and tuple(m.shape) == ((shape[0] >> l) - 10, (shape[1] >> l) - 10)
        check_summary(m, gm, "tmqi.%s.map%d" % (salt, l), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(float(m.mean()), sl[l], rtol=1e-12)
    assert TMQI()(torch.from_numpy(hdr).float().cuda(), torch.from_numpy(ldr).float().cuda(), with_maps=False)[4] is None


def test_full_tmqi_large_frame_vs_oracle():
    from oracle import tmqi as OT
    from uncltmo_amd.tmqi import TMQI
    hdr, ldr = tmqi_inputs(400, 528, "c")
    h32, l32 = torch.from_numpy(hdr).float(), torch.from_numpy(ldr / 255.0).float()
    Q, S, N, sl, _ = TMQI()(h32.cuda(), l32.cuda(), ldr_scale=255.0, with_maps=False)
    rQ, rS, rN, rsl = OT.tmqi(h32.double().numpy(), (l32 * 255.0).double().numpy())
    np.testing.assert_allclose([Q, S, N], [rQ, rS, rN], rtol=1e-6)
    np.testing.assert_allclose(sl, rsl, rtol=1e-6)
    with pytest.raises(Exception):
        TMQI()(h32[:100, :100].cuda(), l32[:100, :100].cuda())           # pyramid would not hold an 11x11 window


def test_eval_on_video_clip_metrics():
    """Tester.eval_on_video's tensor path: the clip goes through the recurrent video generator tile by tile; the scene score is
    the mean of the per-frame TMQI of the 8-bit results; the warp-error formulas against numpy."""
    from uncltmo_amd import model_factory, tester
    from uncltmo_amd.tmqi import TMQI
    G = model_factory.create_G_net("unet", torch.device("cuda"), False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none",
                                   "none", "relu", True, 1, 1, 0, "replicate", 2, 0)
    synth.fill_state_dict(G, "g0")
    G.eval()
    g = torch.Generator().manual_seed(9)
    frames = [(torch.rand(3, 300, 340, generator=g) ** 4 * 50).cuda() for _ in range(3)]
    score, ldr = tester.eval_on_video(G, frames, 127.5 * 0.1, final_shape_addition=0, add_frame=False)
    assert len(ldr) == 3 and all(t.shape == (300, 340, 3) and t.dtype == torch.uint8 for t in ldr)
    per = [TMQI()(frames[i].permute(1, 2, 0).contiguous(), ldr[i].float())[0] for i in range(3)]
    assert abs(score - sum(per) / 3) < 1e-12 and 0.0 < score < 1.0
    # the recurrent hand-off is live: frame 1 evaluated alone differs from frame 1 inside the clip
    _, solo = tester.eval_on_video(G, [frames[1]], 127.5 * 0.1)
    assert not torch.equal(solo[0], ldr[1])
    # warp errors (Tester.py:385-389) with a caller-side alignment (identity here)
    s2, ldr2, mse, rel = tester.eval_on_video(G, frames, 127.5 * 0.1, align_fn=lambda f1, f0: f1)
    a = ldr[1].cpu().numpy().astype(np.float32)[32:-32, 32:-32] / 255.0
    b = ldr[0].cpu().numpy().astype(np.float32)[32:-32, 32:-32] / 255.0
    np.testing.assert_allclose(mse, np.mean((a - b) ** 2), rtol=1e-5)
    np.testing.assert_allclose(rel, np.mean(np.abs(a - b) / (1e-8 + a + b)), rtol=1e-5)
    assert s2 == score and all(torch.equal(u, v) for u, v in zip(ldr, ldr2))
    # the reference's own route (Tester.py:379-389): the flow is ESTIMATED on a frame pair of another method and the result's
    # frame 1 is warped with it.  With the pair = the results themselves and frame 1 = frame 0 shifted by three pixels the aligned
    # error must fall well below the unaligned one
    shifted = torch.roll(ldr[0], shifts=(0, 3), dims=(0, 1))
    s3, ldr3, mse_a, rel_a = tester.eval_on_video(G, frames, 127.5 * 0.1, flow_images=(ldr[1], ldr[0]))
    assert s3 == score and mse_a >= 0.0 and rel_a >= 0.0
    from uncltmo_amd import frame_util
    f = frame_util.compute_flow(shifted, ldr[0])            # shifted(p + f) ~ frame0(p): f = (+3, 0)
    inner = f[40:-40, 40:-40]
    assert (inner[..., 0] - 3.0).abs().median() < 0.25 and inner[..., 1].abs().median() < 0.25
    mse_al, _ = tester.warp_errors(ldr[0], frame_util.align_frames(shifted, f))
    mse_un, _ = tester.warp_errors(ldr[0], shifted)
    assert mse_al < 0.5 * mse_un


def test_eval_on_video_vs_reference_golden(golden):
    """uncltmo_amd.tester.eval_on_video against the fixture captured through the reference's own Tester.eval_on_video
    (Tester.py:314-391; tests/golden/make_golden.py:capture_tester): the stand-in generator case pins the whole tensor path --
    log compression, padding, the 5-D tiler, percentile clamp, colour, the 8-bit stretch, TMQI of every frame, both warp-error
    formulas -- and the recurrent generator case (fp32) the clip through the real network."""
    from test_oracle_golden import check_ldr_frames, clip_inputs, clip_standin
    from uncltmo_amd import model_factory, tester
    g = golden("tester")
    frames_np, lam = clip_inputs()
    f_factor = float(g["tester.f_factor"][0])
    frames = [torch.from_numpy(f.transpose(2, 0, 1).copy()).cuda() for f in frames_np]
    ident = lambda f1, f0: f1
    # stand-in generator (elementwise torch ops on the device): every stage but the network itself
    score, ldr, mse, rel = tester.eval_on_video(clip_standin, frames, f_factor, align_fn=ident)
    # the device path evaluates the same fp32 operations; pow / log10 may differ from the host libm in the last ulp, which can
    # move a value across an 8-bit rounding boundary: at most 1 level on at most 0.5 % of the samples
    check_ldr_frames([t.cpu().numpy() for t in ldr], g, "tone", max_bad=5e-3)
    np.testing.assert_allclose(score, g["tester.tone.scores"][0], rtol=2e-4)
    np.testing.assert_allclose([mse, rel], g["tester.tone.scores"][1:], rtol=5e-3)
    # the recurrent generator, fp32 parity mode
    G = model_factory.create_G_net("unet", torch.device("cuda"), False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none",
                                   "none", "relu", True, 1, 1, 0, "replicate", 2, 0, compute_dtype="fp32")
    synth.fill_state_dict(G, "g0")
    G.eval()
    score, ldr, mse, rel = tester.eval_on_video(G, frames, f_factor, align_fn=ident)
    check_ldr_frames([t.cpu().numpy() for t in ldr], g, "G", max_bad=2e-2)
    np.testing.assert_allclose([mse, rel], g["tester.G.scores"][1:], rtol=2e-2)
    assert np.isnan(score) == bool(np.isnan(g["tester.G.scores"][0]))
