"""ORACLE (test infrastructure): CPU restatement of the adversarial / contrastive / structural losses.

See oracle/__init__.py for the usage rules.  Differentiable through torch autograd so the parity tests
can compare input gradients of the HIP kernels.
"""
import torch
import torch.nn.functional as F

from .generator import gauss_window, local_variance
from .tmqi import naturalness

EPS2 = 1e-05  # utils/params.py:49


def contrastive_d_loss(real_logits, fake_logits):
    """GanTrainerImg.py:219-229.  half(t1, t2) = CE(rows [t1_i, t2_0 .. t2_{N-1}], class 0);
    total = half(real, fake) + half(-fake, -real)."""
    r, f = real_logits.reshape(-1), fake_logits.reshape(-1)

    def half(t1, t2):
        t = torch.cat((t1[:, None], t2[None, :].expand(t1.shape[0], -1)), dim=-1)
        return F.cross_entropy(t, torch.zeros(t1.shape[0], dtype=torch.long))

    return half(r, f) + half(-f, -r)


def nce_similarity(a, b, k, c):
    """s(a,b) = mean_hw sum_ch a*b / (c + k*|a-b|) -> (N,1).  GanTrainerImg.py:421-429."""
    return torch.sum((a * b) * (1 / (c + k * torch.abs(a - b))), dim=1).mean(dim=[-1, -2]).unsqueeze(1)


def nce(anchor, positive, negative, k, c):
    """2-way InfoNCE with the custom similarity.  GanTrainerImg.py:410-439 (one positive, one negative)."""
    logits = torch.cat([nce_similarity(anchor, positive, k, c), nce_similarity(anchor, negative, k, c)], dim=1)
    return F.cross_entropy(logits, torch.zeros(anchor.shape[0], dtype=torch.long))


def lmcl_loss(logits):
    """GanTrainerImg.py:441-450 on a list of (N,1) similarities, the first one positive: -log(exp(s_pos) / sum_j exp(s_neg_j))."""
    neg = torch.cat(logits[1:], dim=1)
    return -torch.log(logits[0].exp() / neg.exp().sum(dim=1, keepdim=True)).mean()


def nce_lists(anchor, positives, negatives, k, c, form="InfoNCE"):
    """GanTrainerImg.py:410-439 with any number of positives and negatives: every positive meets ALL negatives in one
    (Q+1)-way cross-entropy against class 0 (:431-433) or in lmcl_loss (:434-435); the mean over the positives is returned (:439)."""
    neg = [nce_similarity(anchor, f, k, c) for f in negatives]
    loss = 0
    for f in positives:
        pos = [nce_similarity(anchor, f, k, c)]
        if form == "InfoNCE":
            loss = loss + F.cross_entropy(torch.cat(pos + neg, dim=1), torch.zeros(anchor.shape[0], dtype=torch.long))
        elif form == "LMCL":
            loss = loss + lmcl_loss(pos + neg)
        else:
            raise TypeError("%s is not found in loss/adversarial.py" % form)
    return loss / len(positives)


def tmqi_scores_frames(fake):
    """Naturalness of each (N,1,H,W) frame scaled by 255, as the trainers do (GanTrainerImg.py:388-397)."""
    imgs = fake.permute(0, 2, 3, 1).detach().cpu().numpy()
    return [naturalness(imgs[i, :, :, 0] * 255) for i in range(imgs.shape[0])]


def select_best_worst(scores):
    """Index of the max and of the min score, first occurrence (sorted + list.index, :398-402)."""
    s = sorted(scores)
    return scores.index(s[-1]), scores.index(s[0])


def info_nce2(fea_fake, fake, k=1, c=1e-2, want=None):
    """GanTrainerImg.py:384-408: the frame with the best TMQI naturalness is everyone's positive, the worst
    everyone's negative; features are the generator's (N,32,256,256) maps (image) or (N,64,1,1) (video)."""
    scores = tmqi_scores_frames(fake)
    best, worst = select_best_worst(scores)
    if want is not None:
        want["nce2_scores"], want["nce2_best"], want["nce2_worst"] = scores, best, worst
    n = fea_fake.shape[0]
    pos = fea_fake[best].unsqueeze(0).repeat(n, 1, 1, 1)
    neg = fea_fake[worst].unsqueeze(0).repeat(n, 1, 1, 1)
    return nce(fea_fake, pos, neg, k, c)


def pseudo_label_loss(fake, want=None):
    """GanTrainerImg.py:341-368: split every frame 2x2 into 128^2 patches, score each by naturalness, the
    best patch is the pseudo label; L1 between patch means and its mean + L1 between the means of the
    Gaussian local variance."""
    imgs = fake.permute(0, 2, 3, 1).detach().cpu().numpy()
    ps = 256 // 2
    patches, scores = [], []
    for i in range(fake.shape[0]):
        for j in range(2):
            for kk in range(2):
                scores.append(naturalness(imgs[i, j * ps:(j + 1) * ps, kk * ps:(kk + 1) * ps, 0] * 255))
                patches.append(fake[i:i + 1, 0:1, j * ps:(j + 1) * ps, kk * ps:(kk + 1) * ps])
    best = scores.index(sorted(scores)[-1])
    if want is not None:
        want["pl_scores"], want["pl_best"] = scores, best
    label = patches[best].repeat(len(patches), 1, 1, 1)
    patches = torch.cat(patches, 0)
    loss = F.l1_loss(patches.mean(dim=[-1, -2]), label.mean(dim=[-1, -2]))
    win = gauss_window()
    loss = loss + F.l1_loss(local_variance(patches, win).mean(dim=[-1, -2]),
                            local_variance(label, win).mean(dim=[-1, -2]))
    return loss


def brightness_contrast_l1(fake, ldr_pos):
    """GanTrainerImg.py:308-313: (L1 of per-frame means, L1 of per-frame mean Gaussian local variance)."""
    win = gauss_window()
    l_mean = F.l1_loss(fake.mean(dim=[-1, -2]), ldr_pos.mean(dim=[-1, -2]))
    l_con = F.l1_loss(local_variance(fake, win).mean(dim=[-1, -2]), local_variance(ldr_pos, win).mean(dim=[-1, -2]))
    return l_mean, l_con


def tv_loss(x):
    """GanTrainer.py:669-682."""
    n, _, h, w = x.shape
    h_tv = torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum()
    w_tv = torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum()
    return 2 * (h_tv / ((h - 1) * w) + w_tv / (h * (w - 1))) / n


def struct_loss_level(img1, img2, window_size=5):
    """models/struct_loss.py:57-87: every 5x5 window is normalised by its own box-filter mean / std and the
    two normalised window stacks are compared by MSE."""
    ws = window_size
    win = torch.ones((1, 1, ws, ws), dtype=img1.dtype) / (ws * ws)     # dtype follows the inputs (fp64 ground-truth runs)
    win = win / win.sum()

    def stats(img):
        mu = F.conv2d(img, win)
        var = F.conv2d(img * img, win) - mu.pow(2)
        std = torch.pow(torch.max(var, torch.zeros_like(var)) + EPS2, 0.5)
        return mu, std

    def windows(img):
        w = img.unfold(2, ws, 1).unfold(3, ws, 1)
        return w.reshape(w.shape[0], w.shape[1], w.shape[2], w.shape[3], ws * ws)

    mu1, std1 = stats(img1)
    mu2, std2 = stats(img2)
    n1 = (windows(img1) - mu1.unsqueeze(4)) / (std1.unsqueeze(4) + EPS2)
    n2 = (windows(img2) - mu2.unsqueeze(4)) / (std2.unsqueeze(4) + EPS2)
    return F.mse_loss(n1, n2)


def struct_loss_pyramid(fake, hdr_input, pyramid_weight_list, window_size=5):
    """models/struct_loss.py:46-54: levels weighted, images halved by bicubic between levels."""
    total = []
    a, b = fake, hdr_input
    for wgt in pyramid_weight_list:
        total.append(wgt * struct_loss_level(a, b, window_size))
        a = F.interpolate(a, scale_factor=0.5, mode="bicubic", align_corners=False)
        b = F.interpolate(b, scale_factor=0.5, mode="bicubic", align_corners=False)
    return torch.sum(torch.stack(total))
