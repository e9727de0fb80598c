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


# ---- full TMQI (structural fidelity S, naturalness N, quality Q): TMQI.py:107-207 ------------------------------
A_Q, ALPHA, BETA = 0.8012, 0.3046, 0.7088
LEVEL_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss2d():
    """np.outer(gaussian(11, 1.5), gaussian(11, 1.5)), normalised (TMQI.py:119-121, 180)."""
    k = np.arange(11) - 5.0
    g = np.exp(-0.5 * (k / 1.5) ** 2)
    w = np.outer(g, g)
    return w / w.sum()


def _norm_cdf(x, loc, scale):
    from math import sqrt
    import numpy as _np
    try:
        from scipy.special import erfc
        return 0.5 * erfc(-(x - loc) / (scale * sqrt(2.0)))
    except ImportError:                      # pragma: no cover
        return 0.5 * _np.vectorize(math.erfc)(-(x - loc) / (scale * sqrt(2.0)))


def _conv_valid(img, win):
    """'valid' 2-D convolution, float64, direct summation (the window is symmetric, so correlation == convolution)."""
    from numpy.lib.stride_tricks import sliding_window_view
    v = sliding_window_view(img, win.shape)
    return np.einsum("ijkl,kl->ij", v, win[::-1, ::-1])


def s_local(img1, img2, sf, C1=0.01, C2=10.0, want_map=False):
    """TMQI.py:178-207: one pyramid level's local structural fidelity map mean (and, on request, the map: the reference returns both)."""
    win = _gauss2d()
    mu1, mu2 = _conv_valid(img1, win), _conv_valid(img2, win)
    s1 = np.sqrt(np.maximum(_conv_valid(img1 * img1, win) - mu1 * mu1, 0))
    s2 = np.sqrt(np.maximum(_conv_valid(img2 * img2, win) - mu2 * mu2, 0))
    s12 = _conv_valid(img1 * img2, win) - mu1 * mu2
    csf = 100.0 * 2.6 * (0.0192 + 0.114 * sf) * np.exp(-(0.114 * sf) ** 1.1)
    u = 128 / (1.4 * csf)
    sig = u / 3.0
    p1, p2 = _norm_cdf(s1, u, sig), _norm_cdf(s2, u, sig)
    s_map = ((2 * p1 * p2 + C1) / (p1 ** 2 + p2 ** 2 + C1)) * ((s12 + C2) / (s1 * s2 + C2))
    return (float(np.mean(s_map)), s_map) if want_map else float(np.mean(s_map))


def structural_fidelity(L_hdr, L_ldr, levels=5, maps=None):
    """TMQI.py:149-172: five dyadic levels (2x2 mean, keep every second sample), product of weighted level means."""
    f = 32.0
    out = []
    for _ in range(levels):
        f = f / 2
        if maps is None:
            out.append(s_local(L_hdr, L_ldr, f))
        else:
            sl, sm = s_local(L_hdr, L_ldr, f, want_map=True)
            out.append(sl)
            maps.append(sm)
        k = np.ones((2, 2)) / 4.0
        L_hdr = _conv_valid(L_hdr, k)[::2, ::2]
        L_ldr = _conv_valid(L_ldr, k)[::2, ::2]
    S = float(np.prod(np.power(out, LEVEL_WEIGHTS)))
    return S, out


def tmqi(hdr, ldr, maps=None):
    """TMQI.py:107-146 (`original` branch): grayscale hdr (any range) and ldr ([0,255]) -> (Q, S, N, s_local[5]); `maps`: a list
    that receives the five per-level maps (the reference's fifth return value `s_maps`, :152-157)."""
    hdr = np.asarray(hdr, dtype=np.float64)
    ldr = np.asarray(ldr, dtype=np.float64)
    N = naturalness(ldr)
    factor = float(2 ** 32 - 1.0)
    L_hdr = factor * (hdr - hdr.min()) / (hdr.max() - hdr.min())
    S, sl = structural_fidelity(L_hdr, ldr, maps=maps)
    Q = A_Q * (S ** ALPHA) + (1.0 - A_Q) * (N ** BETA)
    return Q, S, N, sl
