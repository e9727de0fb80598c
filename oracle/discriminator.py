"""ORACLE (test infrastructure): CPU restatement of the discriminators.  See oracle/__init__.py."""
import torch
import torch.nn.functional as F

from .generator import gauss_window, local_variance


def simple_d_forward(sd, x):
    """SimpleDiscriminator, published config (input 256, dim 16, no pad, no maxpool, no sigmoid).

    models/Discriminator.py:97-126: conv4x4 s2 -> LeakyReLU(0.2) -> conv4x4 s2 -> LeakyReLU -> conv1x1 gives
    `fea` (N,1,62,62); logit = Linear(3844->1, no bias)(flatten(fea)); the second output is
    [mean(fea), mean(Gaussian local variance of fea)] as (N,2,1,1).
    """
    h = F.leaky_relu(F.conv2d(x, sd["model.0.weight"], sd["model.0.bias"], stride=2), 0.2)
    h = F.leaky_relu(F.conv2d(h, sd["model.2.weight"], sd["model.2.bias"], stride=2), 0.2)
    fea = F.conv2d(h, sd["model.4.weight"], sd["model.4.bias"])
    out = F.linear(fea.reshape(fea.shape[0], -1), sd["tail.1.weight"])
    f1 = fea.mean(dim=(2, 3), keepdim=True)
    f2 = local_variance(fea, gauss_window()).mean(dim=(2, 3), keepdim=True)
    return out, torch.cat([f1, f2], dim=1)


def patch_d_forward(sd, x, n_layers=3):
    """NLayerDiscriminator (PatchGAN) with instance norm.  models/Discriminator.py:129-167 and
    models/Blocks.py:6-36: conv(4,2,1)+LReLU, then (n_layers-1) x [conv(4,2,1,no bias)+IN+LReLU],
    one [conv(4,1,1,no bias)+IN+LReLU], and a final conv(4,1,1) with bias."""
    h = F.leaky_relu(F.conv2d(x, sd["model.0.weight"], sd["model.0.bias"], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers):
        h = F.conv2d(h.float(), sd["model.%d.conv.weight" % idx], None, stride=2, padding=1)
        h = F.leaky_relu(F.instance_norm(h, eps=1e-5), 0.2)
        idx += 1
    h = F.conv2d(h.float(), sd["model.%d.conv.weight" % idx], None, stride=1, padding=1)
    h = F.leaky_relu(F.instance_norm(h, eps=1e-5), 0.2)
    idx += 1
    return F.conv2d(h, sd["model.%d.weight" % idx], sd["model.%d.bias" % idx], stride=1, padding=1)
