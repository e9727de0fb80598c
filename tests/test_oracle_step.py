"""Pin the oracle's full optimisation step (image + video trainers) against the reference's goldens."""
import numpy as np
import pytest
import torch

from conftest import synth_state
from oracle import trainer as OTR
from uncltmo_amd import state_spec, synth


def step_inputs():
    B, T = 2, 2
    hdr = torch.stack([torch.cat([synth.smooth_hdr_frames(1, salt="s%d_%d" % (b, t)) for t in range(T)], 0)
                       for b in range(B)], 0)
    pos = synth.ldr_frames(B * T, salt="spos").reshape(B, T, 1, 256, 256)
    neg = (synth.ldr_frames(B * T, salt="sneg") ** 2).reshape(B, T, 1, 256, 256)
    return hdr, pos, neg


@pytest.mark.parametrize("video,epoch", [(False, 0), (False, 7), (False, 10), (True, 0), (True, 7), (True, 10)])
def test_step(golden, video, epoch):
    g = golden("vid_step" if video else "img_step")
    tag = "%s_step_e%d" % ("vid" if video else "img", epoch)
    st = OTR.StepState(synth_state(state_spec.generator_spec(), "g0"), synth_state(state_spec.simple_d_spec(), "d0"),
                       video=video)
    hdr, pos, neg = step_inputs()
    # DropPath was disabled for the capture (third-party RNG is unpinned), i.e. eval-mode arithmetic
    errD = OTR.train_d(st, hdr, pos, epoch, training=False)
    np.testing.assert_allclose(errD.item(), g[tag + ".errD"], rtol=1e-5)
    for k, v in st.sdD.items():
        np.testing.assert_allclose(v.grad.double().norm().item(), g[tag + ".gradD." + k], rtol=2e-4, err_msg=k)
    np.testing.assert_allclose(st.sdD["tail.1.weight"].detach().numpy()[:, :64], g[tag + ".D_after.tail"], rtol=1e-5,
                               atol=1e-7)
    if not video and epoch > 9:
        assert g[tag + ".nameerror"] == 1
        with pytest.raises(NameError):
            OTR.train_g(st, hdr, pos, neg, epoch, training=False)
        return
    want = {}
    errG_d, errG_s = OTR.train_g(st, hdr, pos, neg, epoch, training=False, want=want)
    np.testing.assert_allclose(errG_d.item(), g[tag + ".errG_d"], rtol=1e-4)
    np.testing.assert_allclose(errG_s.item(), g[tag + ".errG_struct"], rtol=1e-5)
    for k, gr in want["grad_total"].items():
        np.testing.assert_allclose(gr.double().norm().item(), g[tag + ".gradG." + k], rtol=2e-3, err_msg=k)
    for k, v in st.sdG.items():
        np.testing.assert_allclose(v.detach().double().sum().item(), g[tag + ".G_after." + k], rtol=1e-5, atol=1e-4,
                                   err_msg=k)


def test_step_c4_clip_of_five_frames_in_four_crops(golden):
    """BASELINE configs[3] at test size: the oracle's video step on a 512 x 512 clip of T = 5 cut into four 256 x 256 crops
    against the reference's own GanTrainer on the same tensors (make_golden.py vid_c4), incl. sampled gradient elements."""
    from uncltmo_amd.frame_util import clip_to_crops
    g = golden("vid_c4")
    tag = "vid_c4_e0"
    hdr = clip_to_crops(synth.hdr_frames(5, 512, 512, salt="c4hdr").reshape(1, 5, 1, 512, 512))
    pos = clip_to_crops(synth.ldr_frames(5, 512, 512, salt="c4pos").reshape(1, 5, 1, 512, 512))
    neg = clip_to_crops(synth.ldr_frames(5, 512, 512, salt="c4neg").reshape(1, 5, 1, 512, 512)) ** 2
    st = OTR.StepState(synth_state(state_spec.generator_spec(), "g0"), synth_state(state_spec.simple_d_spec(), "d0"), video=True)
    errD = OTR.train_d(st, hdr, pos, 0, training=False)
    np.testing.assert_allclose(errD.item(), g[tag + ".errD"], rtol=1e-5)
    want = {}
    errG_d, errG_s = OTR.train_g(st, hdr, pos, neg, 0, training=False, want=want)
    np.testing.assert_allclose(errG_d.item(), g[tag + ".errG_d"], rtol=1e-4)
    np.testing.assert_allclose(errG_s.item(), g[tag + ".errG_struct"], rtol=1e-5)
    for k, gr in want["grad_total"].items():
        np.testing.assert_allclose(gr.double().norm().item(), g[tag + ".gradG." + k], rtol=2e-3, err_msg=k)
        v = gr.double().reshape(-1)[torch.from_numpy(g[tag + ".gradGpos." + k])].numpy()
        scale = float(g[tag + ".gradG." + k]) / max(gr.numel(), 1) ** 0.5
        np.testing.assert_allclose(v, g[tag + ".gradGval." + k], rtol=2e-3, atol=2e-3 * scale, err_msg=k)
