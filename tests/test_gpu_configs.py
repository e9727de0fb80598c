"""The BASELINE.json configurations as they are specified, each against the CPU oracle or a golden captured from the reference:

  C2  8 x 1024^2 -> 200 tiles, bf16: sampled tiles of the REAL bench batch (four parts on four streams) vs the oracle
  C3  full image-trainer step at N = 32 frames (loader batch 16 x 2) vs the oracle's step on the same inputs
  C4  video step on a 512 x 512 clip of T = 5 cut into four 256 x 256 crops vs a golden from the reference's GanTrainer
  C5  one 2160 x 3840 frame -> 220 tiles (fp16 / bf16): sampled tiles vs the oracle, cross-fade properties on the full frame
"""
import types

import numpy as np
import pytest
import torch

from conftest import synth_state
from uncltmo_amd import model_factory, state_spec, synth, tiler
from uncltmo_amd.generator import UNet
from uncltmo_amd.optim import Adam

pytestmark = pytest.mark.gpu

G_ARGS = (1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _oracle_tiles(x):
    from oracle.generator import unet_image_forward
    from oracle.state import generator_state
    torch.set_num_threads(min(64, torch.get_num_threads() or 1) or 1)
    with torch.no_grad():
        y, _ = unet_image_forward(generator_state("g0"), x.cpu().float())
    return y


def test_c2_bench_batch_sampled_tiles_vs_oracle():
    """bench.py's own step: 200 tiles in ONE generator call, split into four parts on four streams before the last decoder
    stage.  One tile of each part (and the last tile of the batch) is checked against the fp32 oracle."""
    net = UNet(*G_ARGS, compute_dtype="bf16")
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    frames = synth.hdr_frames(8, 1024, 1024, salt="bench0").cuda()
    tiles = tiler.gather_tiles(frames.reshape(8, 1024, 1024))
    assert tiles.shape[0] == 200
    with torch.no_grad():
        out = net.infer(tiles)
    pick = [3, 63, 137, 160, 199]                     # parts are tiles [0,50) [50,100) [100,150) [150,200)
    want = _oracle_tiles(tiles[pick])
    got = out[pick].float().cpu()
    for i, t in enumerate(pick):
        assert _rel(got[i], want[i]) < 3e-2, t
    # and the cross-faded frames are finite and inside the sigmoid's range
    full = tiler.test_big_size_image2(frames, net, 0, 0, 0)
    assert full.shape == (8, 1, 1024, 1024) and torch.isfinite(full).all() and 0 <= full.min() and full.max() <= 1


def _trainer(video, dtype="bf16"):
    from uncltmo_amd.trainer_img import GanTrainer as ImgTrainer
    from uncltmo_amd.trainer_vid import GanTrainer as VidTrainer
    dev = torch.device("cuda")
    make = model_factory.create_G_net if video else model_factory.create_G_net2
    G = make("unet", dev, False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none", "none", "relu", True, 1, 1, 0,
             "replicate", 2, 0, compute_dtype=dtype)
    D = model_factory.create_D_net(1, 16, dev, False, "none", True, "simpleD", 3, "none", 3, 0, 0, 0)
    synth.fill_state_dict(G, "g0")
    synth.fill_state_dict(D, "d0")
    G.train()
    G.drop_path_prob = 0.0
    opt = types.SimpleNamespace(device=dev, pyramid_weight_list=torch.tensor([1.0, 1.0, 1.0]), ssim_loss_factor=1.0,
                                ssim_window_size=5, struct_method="gamma_ssim", add_frame=0, final_shape_addition=0,
                                loss_g_d_factor=0.1, adv_weight_list=torch.tensor([0.2, 0.2, 0.2]))
    cls = VidTrainer if video else ImgTrainer
    return cls(opt, G, D, Adam(G.parameters(), lr=1e-5, betas=(0.5, 0.999)), Adam(D.parameters(), lr=1.5e-5, betas=(0.5, 0.999)),
               None, None), G, D


_C3 = {}


def _c3_oracle():
    """the oracle's N = 32 step (and, for the fp32-mode gate, the same step with every weight moved by at most one fp32 ulp, two
    seeds): computed once per session, shared by the bf16 and the fp32-mode tests"""
    if _C3:
        return _C3
    from oracle import trainer as OTR
    B, T = 16, 2
    hdr = synth.smooth_hdr_frames(B * T, salt="c3hdr").reshape(B, T, 1, 256, 256)
    pos = synth.ldr_frames(B * T, salt="c3pos").reshape(B, T, 1, 256, 256)
    neg = (synth.ldr_frames(B * T, salt="c3neg") ** 2).reshape(B, T, 1, 256, 256)
    torch.set_num_threads(min(64, torch.get_num_threads()))

    def run(perturb):
        sdG = synth_state(state_spec.generator_spec(), "g0")
        if perturb:
            gen = torch.Generator().manual_seed(perturb)
            sdG = {k: (v * (1 + 6e-8 * (2 * torch.rand(v.shape, generator=gen) - 1))).float()
                   if not k.endswith("relative_pos") else v for k, v in sdG.items()}
        st = OTR.StepState(sdG, synth_state(state_spec.simple_d_spec(), "d0"), video=False)
        errD = OTR.train_d(st, hdr, pos, 0, training=False)
        want = {}
        errGd, errGs = OTR.train_g(st, hdr, pos, neg, 0, training=False, want=want)
        return errD.item(), errGd.item(), errGs.item(), want["grad_total"]

    _C3.update(hdr=hdr, pos=pos, neg=neg, base=run(0), run=run)
    return _C3


def test_c3_image_step_n32_vs_oracle():
    """configs[2] at its own size: loader batch 16 x 2 frames = N 32.  The oracle (pinned to the reference's whole-step goldens at
    N = 4, tests/test_oracle_step.py) runs the same step on the host; compared: the three loss scalars and every gradient
    tensor's direction and length (rel-L2), bf16 generator vs fp32 oracle."""
    c3 = _c3_oracle()
    hdr, pos, neg = c3["hdr"], c3["pos"], c3["neg"]
    errD_o, errGd_o, errGs_o, grads_o = c3["base"]
    want = {"grad_total": grads_o}

    tr, G, D = _trainer(False)
    tr.train_D(hdr.cuda(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errD.item(), errD_o, rtol=2e-2)
    tr.optimizerG = types.SimpleNamespace(step=lambda: None)
    tr.train_G(hdr.cuda(), hdr.cuda().clone(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errG_d.item(), errGd_o, rtol=3e-2)
    np.testing.assert_allclose(tr.errG_struct.item(), errGs_o, rtol=2e-2)
    # bf16 activations and activation gradients end to end: the error grows with the depth of the backward path, so the gate
    # is per LEVEL of the network (a regression in one decoder level cannot hide under the encoder's bound).  Bounds = 1.5 x
    # the worst tensor of the level measured at round 3 (printed with -s), rounded up.
    def level(k):
        if k == "gcn.pos_embed":
            return k
        parts = k.split(".")
        return ".".join(parts[:2]) if parts[0] in ("down_path", "up_path") else parts[0]

    BOUND = C3_LEVEL_BOUNDS
    worst = {}
    for k, p in G.named_parameters():
        if p.grad is None:              # the fixed relative_pos table
            continue
        r = _rel(p.grad.cpu(), want["grad_total"][k])
        lv = level(k)
        if r > worst.get(lv, (0.0, ""))[0]:
            worst[lv] = (r, k)
    print("C3 bf16 gradient rel-L2, worst tensor per level:", {lv: round(v[0], 4) for lv, v in sorted(worst.items())})
    bad = {lv: v for lv, v in worst.items() if v[0] > BOUND[lv]}
    assert not bad, bad
    assert set(worst) == set(BOUND)


def test_c3_image_step_n32_fp32_mode_vs_oracle():
    """configs[2] at its own size in the PARITY mode (compute_dtype='fp32': fp32 activations, gradients and deterministic
    reductions; GanTrainerImg.py:200-339): the three loss scalars within 1e-4 of the oracle's, every gradient tensor within
    max(floor, 3 s) of the oracle's, s = how far the oracle's own gradient moves when every weight moves by at most one fp32 ulp
    (the gate of tests/test_gpu_trainer.py::test_fp32_step_vs_reference_golden_and_oracle, which runs at N = 4; two perturbation
    seeds here: an N = 32 oracle step takes a minute of host time)."""
    c3 = _c3_oracle()
    hdr, pos, neg = c3["hdr"], c3["pos"], c3["neg"]
    errD_o, errGd_o, errGs_o, grads_o = c3["base"]
    tr, G, D = _trainer(False, dtype="fp32")
    tr.train_D(hdr.cuda(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errD.item(), errD_o, rtol=1e-4)
    tr.optimizerG = types.SimpleNamespace(step=lambda: None)
    tr.train_G(hdr.cuda(), hdr.cuda().clone(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errG_d.item(), errGd_o, rtol=1e-4)
    np.testing.assert_allclose(tr.errG_struct.item(), errGs_o, rtol=1e-4)
    perturbed = [c3["run"](seed)[3] for seed in (1, 2)]
    bad, tight, levels = {}, 0, {}
    for k, p in G.named_parameters():
        if p.grad is None:
            continue
        ref = grads_o[k].double()
        sens = max(((wp[k].double() - ref).norm() / ref.norm().clamp_min(1e-30)).item() for wp in perturbed)
        r = ((p.grad.double().cpu() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        tight += r <= 1e-3
        levels[k] = (round(r, 6), round(sens, 6))
        floor = 2e-2 if k == "gcn.pos_embed" else 2e-3
        if r > max(floor, 3.0 * sens):
            bad[k] = (r, sens)
    print("C3 fp32-mode gradient rel-L2 (and one-ulp sensitivity) per tensor:", levels)
    assert not bad, bad
    assert tight >= 20, tight          # the decoder half meets 1e-3 outright


# per-level gates of test_c3_*: 1.5 x the measured worst tensor of the level, rounded up
C3_LEVEL_BOUNDS = {"inc": 0.055, "down_path.0": 0.075, "down_path.1": 0.12, "down_path.2": 0.13, "down_path.3": 0.05, "gcn": 0.05,
                   "gcn.pos_embed": 0.2, "up_path.0": 0.05, "up_path.1": 0.07, "up_path.2": 0.065, "up_path.3": 0.05, "outc": 0.07}
# measured (round 3, MI355X): inc 0.034, down_path.0 0.049, .1 0.080, .2 0.084, .3 0.032, gcn 0.031, gcn.pos_embed 0.134,
# up_path.0 0.033, .1 0.046, .2 0.042, .3 0.031, outc 0.046


def c4_inputs():
    from uncltmo_amd.frame_util import clip_to_crops
    hdr = clip_to_crops(synth.hdr_frames(5, 512, 512, salt="c4hdr").reshape(1, 5, 1, 512, 512))
    pos = clip_to_crops(synth.ldr_frames(5, 512, 512, salt="c4pos").reshape(1, 5, 1, 512, 512))
    neg = clip_to_crops(synth.ldr_frames(5, 512, 512, salt="c4neg").reshape(1, 5, 1, 512, 512)) ** 2
    return hdr, pos, neg


def test_c4_video_step_t5_crops_vs_reference_golden(golden):
    """configs[3]: a 512 x 512 clip of T = 5 -> clip_to_crops -> four clips of 256 x 256 -> one GanTrainer (video) step.  The
    golden was captured from the reference's own trainer on the same tensors (tests/golden/make_golden.py vid_c4)."""
    g = golden("vid_c4")
    tag = "vid_c4_e0"
    hdr, pos, neg = c4_inputs()
    assert hdr.shape == (4, 5, 1, 256, 256)
    # crop k of the clip is window (k // 2, k % 2) of every frame
    full = synth.hdr_frames(5, 512, 512, salt="c4hdr")
    assert torch.equal(hdr[3, 2, 0], full[2, 0, 256:, 256:]) and torch.equal(hdr[1, 4, 0], full[4, 0, :256, 256:])
    tr, G, D = _trainer(True)
    tr.train_D(hdr.cuda(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errD.item(), g[tag + ".errD"], rtol=2e-2)
    tr.optimizerG = types.SimpleNamespace(step=lambda: None)
    tr.train_G(hdr.cuda(), hdr.cuda().clone(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errG_d.item(), g[tag + ".errG_d"], rtol=3e-2)
    np.testing.assert_allclose(tr.errG_struct.item(), g[tag + ".errG_struct"], rtol=2e-2)
    bad = {}
    for k, p in G.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.double().reshape(-1).cpu()
        ref_n = float(g[tag + ".gradG." + k])
        tol = 0.3 if k == "gcn.pos_embed" else 0.1
        if abs(gr.norm().item() - ref_n) > tol * ref_n + 1e-12:
            bad[k] = ("norm", gr.norm().item(), ref_n)
        # direction: the 64 sampled elements of the reference gradient against ours, relative to the tensor's rms
        idx = torch.from_numpy(g[tag + ".gradGpos." + k])
        ref_v = torch.from_numpy(g[tag + ".gradGval." + k])
        rms = ref_n / max(gr.numel(), 1) ** 0.5
        err = (gr[idx] - ref_v).norm().item() / (len(idx) ** 0.5 * rms + 1e-30)
        if err > (0.5 if k == "gcn.pos_embed" else 0.25):      # 64 samples of a bf16-path gradient: ~2x the tensor-level rel-L2
            bad[k] = ("samples", err)
    assert not bad, bad


def test_c4_video_step_t5_crops_fp32_mode_vs_reference_golden(golden):
    """configs[3] in the parity mode (compute_dtype='fp32'; GanTrainer.py:202-338): losses within 1e-4 of the reference's golden,
    every gradient tensor's norm within 5e-3 and the 64 sampled elements of the reference's gradient within 1e-2 of the tensor's
    rms (5e-2 on the encoder levels, whose gradients are ill-conditioned in fp32 itself: tests/test_gpu_trainer.py measures that
    with one-ulp perturbations of the oracle) -- the bf16 test above allows 10 - 30 % and 25 - 50 %.  Measured (round 5, MI355X):
    norms 1e-6 ... 1.1e-3 (outc.conv.bias, one element: 2.9e-3), samples 4e-5 ... 2.9e-3 outside the encoder, 1.7e-3 ... 2.4e-2 in it."""
    g = golden("vid_c4")
    tag = "vid_c4_e0"
    hdr, pos, neg = c4_inputs()
    tr, G, D = _trainer(True, dtype="fp32")
    tr.train_D(hdr.cuda(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errD.item(), g[tag + ".errD"], rtol=1e-4)
    tr.optimizerG = types.SimpleNamespace(step=lambda: None)
    tr.train_G(hdr.cuda(), hdr.cuda().clone(), pos.cuda(), neg.cuda(), 0)
    np.testing.assert_allclose(tr.errG_d.item(), g[tag + ".errG_d"], rtol=1e-4)
    np.testing.assert_allclose(tr.errG_struct.item(), g[tag + ".errG_struct"], rtol=1e-4)
    loose = lambda k: k.startswith(("inc.", "down_path.0.", "down_path.1.", "down_path.2."))
    bad, levels = {}, {}
    for k, p in G.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.double().reshape(-1).cpu()
        ref_n = float(g[tag + ".gradG." + k])
        nerr = abs(gr.norm().item() - ref_n) / (ref_n + 1e-30)
        idx = torch.from_numpy(g[tag + ".gradGpos." + k])
        ref_v = torch.from_numpy(g[tag + ".gradGval." + k])
        rms = ref_n / max(gr.numel(), 1) ** 0.5
        serr = (gr[idx] - ref_v).norm().item() / (len(idx) ** 0.5 * rms + 1e-30)
        levels[k] = (round(nerr, 6), round(serr, 6))
        if nerr > 5e-3:
            bad[k] = ("norm", nerr)
        if serr > (5e-2 if loose(k) else 1e-2):
            bad[k] = ("samples", serr)
    print("C4 fp32-mode gradient (norm error, sampled-element error / rms) per tensor:", levels)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_c5_4k_frame_sampled_tiles_vs_oracle(dtype):
    """configs[4]: one 2160 x 3840 frame -> 220 overlap tiles -> generator -> cross-fade.  Three tiles (first, an interior one, the
    edge-aligned last) are checked against the fp32 oracle; the blended frame must equal the tile outputs wherever only
    one tile covers a pixel."""
    from uncltmo_amd import _hip
    if dtype == "fp16" and not hasattr(_hip, "F16"):
        pytest.skip("fp16 compute dtype not built")
    net = UNet(*G_ARGS, compute_dtype=dtype)
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    frame = synth.hdr_frames(1, 2160, 3840, salt="c5").cuda()
    tiles = tiler.gather_tiles(frame.reshape(1, 2160, 3840))
    assert tiles.shape[0] == 220
    with torch.no_grad():
        out = net.infer(tiles)
    pick = [0, 107, 219]
    want = _oracle_tiles(tiles[pick])
    for i, t in enumerate(pick):
        assert _rel(out[pick[i]].float().cpu(), want[i]) < (1e-2 if dtype == "fp16" else 3e-2), (dtype, t)
    full = tiler.test_big_size_image2(frame, net, 0, 0, 0)
    assert full.shape == (1, 1, 2160, 3840) and torch.isfinite(full).all()
    # top-left 192 x 192 block is covered by tile 0 only (stride 192, overlap 64)
    assert torch.equal(full[0, 0, :192, :192], out[0, 0, :192, :192].float())
