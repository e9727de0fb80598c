"""Video GAN trainer, MI355X-native.  Same constructor / method signatures and side-effect attributes as the reference's
`GanTrainer.GanTrainer` (GanTrainer.py:58-339, 341-462, 669-682).  It differs from the image trainer in three places only:
the generator receives the clip whole (B,T,1,H,W) and runs its frames sequentially with the recurrent channel hand-off
(Unet.py:213-289), its per-frame features are (B*T,64,1,1) statistics instead of a feature map (GanTrainer.py:273-275), and
the epoch > 9 regime is implemented (total-variation term L_TV, GanTrainer.py:332-337, 669-682).

The generator backward runs through time in hand-written kernels (uncltmo_amd/autograd.py:_VideoGeneratorFn); see
uncltmo_amd/trainer_img.py for the stated differences from the reference (no host syncs, one summed backward pass).
"""
from . import losses as L
from .trainer_img import GanTrainer as _ImageTrainer


class GanTrainer(_ImageTrainer):
    def _generate(self, hdr_input):
        fake, fea = self.netG(hdr_input, diffY=self.final_shape_addition, diffX=self.final_shape_addition)
        return (fake.reshape(-1, fake.shape[2], fake.shape[3], fake.shape[4]),
                fea.reshape(-1, fea.shape[2], fea.shape[3], fea.shape[4]))

    def _generate_detached(self, hdr_input):
        """the discriminator step's fake clip: no graph, no feature map, no per-frame feature statistics"""
        import os
        fd = getattr(self.netG, "forward_detached", None)
        if fd is None or os.environ.get("UNCL_D_DETACHED", "1") == "0":          # (0: the full forward, A/B)
            return self._generate(hdr_input)[0]
        fake = fd(hdr_input, diffY=self.final_shape_addition, diffX=self.final_shape_addition)
        return fake.reshape(-1, fake.shape[2], fake.shape[3], fake.shape[4])

    def _last_regime_loss(self, cgan, fake, ldr_pos, hdr_input):
        """GanTrainer.py:332-337: adversarial term switched off, brightness + pseudo-label + total variation."""
        f = self.loss_g_d_factor
        m_f, _ = L.frame_stats(fake)
        m_p, _ = L.frame_stats(ldr_pos)
        f = float(f)
        terms = [(f * 1e-6, cgan), (f * 0.5 * 1e2, L.l1_mean(m_f, m_p.detach()))]
        terms += [(f * 0.5 * 1e2 * w, t) for w, t in self.pseudo_label_terms(fake, hdr_input)]
        terms.append((f * 0.2 * 1e5, L.tv_loss(fake)))
        return L.weighted_sum(terms)
