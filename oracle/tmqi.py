"""ORACLE (test infrastructure): TMQI statistical naturalness in numpy float64.

Restates TMQI.py:210-242 (`_StatisticalNaturalness`, "original" branch) — the only part of TMQI whose
value the training step consumes (argmax / argmin sample selection, GanTrainerImg.py:341-408).
scipy.stats beta/norm pdfs are written out in closed form.
"""
import math

import numpy as np

PHAT1, PHAT2 = 4.4, 10.1
MUHAT, SIGMAHAT = 115.94, 27.99


def _beta_pdf(x, a, b):
    if x <= 0.0 or x >= 1.0:
        return 0.0
    lg = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    return math.exp(lg + (a - 1.0) * math.log(x) + (b - 1.0) * math.log1p(-x))


def naturalness(L_ldr):
    """L_ldr: 2-D array already scaled to [0,255] (the trainers pass fake*255)."""
    # The reference keeps its float32 input dtype through np.mean / np.std / the division by 64.29, so its
    # own value carries ~1e-7 relative float32 noise that depends on the numpy version's scalar promotion.
    # The oracle (and the device kernel) accumulate in float64 instead; goldens are matched to 1e-6.
    L = np.asarray(L_ldr, dtype=np.float64)
    u = float(np.mean(L))
    W, H = L.shape
    w_extra = 11 - W % 11          # always 1..11: a full extra zero block when W % 11 == 0
    h_extra = 11 - H % 11
    t = np.pad(L, pad_width=((0, w_extra), (0, h_extra)), mode="constant")
    blocks = t.reshape(t.shape[0] // 11, 11, t.shape[1] // 11, 11).transpose(0, 2, 1, 3)
    sig = float(np.mean(np.std(blocks, axis=(-1, -2))))
    mode = (PHAT1 - 1.0) / (PHAT1 + PHAT2 - 2.0)
    pc = _beta_pdf(sig / 64.29, PHAT1, PHAT2) / _beta_pdf(mode, PHAT1, PHAT2)
    pb = math.exp(-0.5 * ((u - MUHAT) / SIGMAHAT) ** 2)
    return pb * pc
