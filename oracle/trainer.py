"""ORACLE (test infrastructure): CPU restatement of one GanTrainerImg / GanTrainer optimisation step.

Restates GanTrainerImg.py:200-339,452-461 and GanTrainer.py:202-338,453-462.  State is held as plain
dicts of leaf tensors; Adam comes from torch.optim (betas (0.5, 0.999), main_train_image.py:29-32).
Host-side syncs, printing and `detect_anomaly` of the reference are not part of the arithmetic and are
omitted.  See oracle/__init__.py for the usage rules.
"""
import torch

from . import losses as L
from .discriminator import simple_d_forward
from .generator import unet_image_forward, unet_video_forward

EPOCH_STEP1, EPOCH_STEP2 = 6, 9      # GanTrainerImg.py:111-112


class StepState:
    def __init__(self, sdG, sdD, g_lr=1e-5, d_lr=1.5e-5, video=False):
        self.sdG = {k: v.clone().requires_grad_(not k.endswith("relative_pos")) for k, v in sdG.items()}
        self.sdD = {k: v.clone().requires_grad_(True) for k, v in sdD.items()}
        self.optG = torch.optim.Adam([v for v in self.sdG.values() if v.requires_grad], lr=g_lr, betas=(0.5, 0.999))
        self.optD = torch.optim.Adam(list(self.sdD.values()), lr=d_lr, betas=(0.5, 0.999))
        self.video = video

    def zero(self, sd):
        for v in sd.values():
            v.grad = None


def _g(state, hdr, training, drop_keep):
    """hdr is the loader tensor (B,2|T,1,256,256).  Image: frames are flattened into the batch
    (GanTrainerImg.py:240,272); video: the clip goes in whole and outputs are flattened after."""
    if state.video:
        out, feat = unet_video_forward(state.sdG, hdr, training=training, drop_keep=drop_keep)
        return out.reshape(-1, *out.shape[2:]), feat.reshape(-1, *feat.shape[2:])
    x = hdr.reshape(-1, *hdr.shape[2:])
    return unet_image_forward(state.sdG, x, training=training, drop_keep=drop_keep)


def train_d(state, hdr, ldr_pos, epoch, adv_w0=0.2, training=True, drop_keep=None):
    """GanTrainerImg.py:200-260.  (The reference also runs netD on the negatives and discards it.)"""
    state.zero(state.sdD)
    d_real, _ = simple_d_forward(state.sdD, ldr_pos.reshape(-1, *ldr_pos.shape[2:]))
    with torch.no_grad():
        fake, _ = _g(state, hdr, training, drop_keep)
    d_fake, _ = simple_d_forward(state.sdD, fake.detach())
    scale = 1.0 if epoch <= EPOCH_STEP1 else 1e-6
    errD = adv_w0 * scale * L.contrastive_d_loss(d_real, d_fake)
    errD.backward()
    state.optD.step()
    return errD.detach()


def g_d_loss(state, fake, fea_fake, hdr_flat, pos_flat, neg_flat, epoch, factor=0.1, want=None):
    """update_g_d_loss's scalar (before backward), three epoch regimes.  The image trainer's last regime
    references an undefined L_TV (NameError upstream); the video trainer's is implemented."""
    d_fake, f_fake = simple_d_forward(state.sdD, fake)
    d_pos, f_pos = simple_d_forward(state.sdD, pos_flat)
    _, f_neg = simple_d_forward(state.sdD, neg_flat)
    _, f_in = simple_d_forward(state.sdD, hdr_flat)
    cgan = L.contrastive_d_loss(d_fake, d_pos)
    if epoch <= EPOCH_STEP2:
        first = epoch <= EPOCH_STEP1
        err = factor * (1.0 if first else 1e-6) * cgan
        err = err + factor * 0.5 * L.nce(f_fake, f_pos, f_in, 1, 1e-2)
        err = err + factor * 0.5 * (0.2 * L.nce(f_fake, f_pos, f_neg, 1e3, 2))
        n2 = L.info_nce2(fea_fake, fake, 1, 1e-2, want=want)
        err = err + (factor * 1e-6 * n2 if first else factor * 0.1 * (5 * n2))
        l_mean, l_con = L.brightness_contrast_l1(fake, pos_flat)
        err = err + (factor * 1e-6 * l_mean if first else factor * 0.5 * (1e2 * l_mean))
        err = err + (factor * 1e-6 * l_con if first else factor * 0.5 * (2 * l_con))
        err = err + factor * 1e-6 * L.pseudo_label_loss(fake, want=want)
        return err
    if not state.video:
        raise NameError("name 'L_TV' is not defined")      # GanTrainerImg.py:335, reproduced on purpose
    l_mean, _ = L.brightness_contrast_l1(fake, pos_flat)
    err = factor * 1e-6 * cgan
    err = err + factor * 0.5 * (1e2 * l_mean)
    err = err + factor * 0.5 * (1e2 * L.pseudo_label_loss(fake, want=want))
    err = err + factor * 0.2 * (1e5 * L.tv_loss(fake))
    return err


def train_g(state, hdr, ldr_pos, ldr_neg, epoch, factor=0.1, struct_factor=1.0, pyramid=(1.0, 1.0, 1.0),
            training=True, drop_keep=None, want=None):
    """GanTrainerImg.py:262-292: two backward passes into G (adversarial/contrastive, then structural),
    gradients accumulate, one Adam step."""
    state.zero(state.sdG)
    state.zero(state.sdD)
    fake, fea = _g(state, hdr, training, drop_keep)
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    errG_d = g_d_loss(state, fake, fea, flat(hdr), flat(ldr_pos), flat(ldr_neg), epoch, factor, want)
    errG_d.backward(retain_graph=True)
    if want is not None:
        want["grad_after_first"] = {k: v.grad.clone() for k, v in state.sdG.items() if v.grad is not None}
    errG_s = struct_factor * L.struct_loss_pyramid(fake, flat(hdr), pyramid)
    errG_s.backward()
    if want is not None:
        want["grad_total"] = {k: v.grad.clone() for k, v in state.sdG.items() if v.grad is not None}
    state.optG.step()
    return errG_d.detach(), errG_s.detach()
