"""Deterministic synthetic weights and HDR inputs.

There is no network, so neither trained checkpoints nor datasets exist here.  Both the parity tests
and bench.py therefore rebuild identical tensors from a counter-based integer hash that only needs
numpy: every element of every state_dict tensor is a pure function of (tensor name, flat index).
The capture script (tests/golden/make_golden.py) loads the same tensors into the reference model, so
golden vectors, the oracle and the HIP path all see bit-identical weights.

Input statistics follow the reference's data path: HDR luminance is log-compressed to [0, 1]
(`log10(x / max * lambda + 1)` then `/ max`, ProcessedDatasetFolder.py:147-149, model_save_util.py:
219-240); LDR images are plain [0, 1] (`bugy_max_normalization`, ProcessedDatasetFolder.py:18-19).
"""
import math

import numpy as np
import torch

_M64 = (1 << 64) - 1


def _fnv1a64(text):
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def hash_uniform(key, n, offset=0):
    """n floats in [0, 1): splitmix64 of (fnv1a(key) + index), top 24 bits.  float32, reproducible."""
    with np.errstate(over="ignore"):
        z = np.arange(offset, offset + n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(_fnv1a64(key))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def _fan_in(name, shape):
    if len(shape) == 4:
        # Conv2d weight is (Cout, Cin/g, kh, kw); ConvTranspose2d is (Cin, Cout, kh, kw).  Transposed
        # layers of the generator: `*.up.weight`, `up_path.*.conv.conv*.weight`,
        # `down_path.3.mpconv.1.conv1.weight` (unet_parts.py:112,149,162,269).
        transposed = (name.endswith(".up.weight") or name.startswith("up_path.") and ".conv.conv" in name
                      or name == "down_path.3.mpconv.1.conv1.weight")
        cin = shape[0] if transposed else shape[1]
        if name.endswith(".up.weight"):
            return cin  # stride-2 2x2: each output pixel sees one tap
        return cin * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    return 1


def synth_tensor(name, shape, salt="w0"):
    """Deterministic value for one state_dict entry (He-uniform weights, small biases)."""
    n = int(np.prod(shape))
    u = hash_uniform(salt + ":" + name, n)
    if name.endswith("relative_pos"):
        raise ValueError("relative_pos is a fixed buffer, not synthesised")
    if name.endswith("pos_embed"):
        v = (u - 0.5) * 0.2
    elif name.endswith(".bias"):
        v = (u - 0.5) * 0.1
    else:
        a = math.sqrt(6.0 / _fan_in(name, shape))
        v = (u * 2.0 - 1.0) * np.float32(a)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state_dict(module, salt="w0"):
    """Overwrite every parameter/buffer of `module` in place, except fixed geometric buffers."""
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith("relative_pos") or k.endswith("num_batches_tracked"):
                continue
            v.copy_(synth_tensor(k, tuple(v.shape), salt).to(v.dtype))
    return module


def bnorm_state(sd):
    """The BatchNorm entries of a synth-filled state dict made plausible IN PLACE, the same way in the fixture script
    (tests/golden/make_golden.py) and in the tests: running_var in [0.5, 1.5], running_mean small, gamma around one
    (fill_state_dict / synth_tensor write zero-mean values into every tensor)."""
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith("running_var"):
                v.copy_(0.5 + v.abs() * 0.4)          # synth values of a 1-D tensor lie in +-2.45
            elif k.endswith("running_mean"):
                v.mul_(0.1)
            elif (".norm." in k or ".norm1." in k) and k.endswith(".weight"):
                v.copy_(1.0 + 0.1 * v)
    return sd


def hdr_frames(n, h=256, w=256, salt="hdr0", lam=255.0 * 0.1 * 50.0):
    """(n,1,h,w) float32 log-compressed HDR luminance in [0,1] with a heavy-tailed radiance prior."""
    u = hash_uniform(salt, n * h * w).astype(np.float64) ** 4
    x = np.log10(u * lam + 1.0) / math.log10(lam + 1.0)
    return torch.from_numpy(x.astype(np.float32).reshape(n, 1, h, w))


def smooth_hdr_frames(n, h=256, w=256, salt="hdrs0"):
    """Like hdr_frames but spatially correlated (sum of a few hashed low-frequency cosines + noise),
    so that window statistics (struct loss, TMQI naturalness) are not pure white noise."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    out = np.empty((n, 1, h, w), dtype=np.float32)
    for i in range(n):
        p = hash_uniform("%s:%d" % (salt, i), 24).astype(np.float64)
        img = np.zeros((h, w))
        for t in range(6):
            fy, fx, ph, amp = p[4 * t] * 0.08, p[4 * t + 1] * 0.08, p[4 * t + 2] * 6.283, 0.3 + p[4 * t + 3]
            img += amp * np.cos(fy * yy + fx * xx + ph)
        img = (img - img.min()) / (img.max() - img.min() + 1e-12)
        noise = hash_uniform("%s:n%d" % (salt, i), h * w).reshape(h, w).astype(np.float64)
        rad = (0.85 * img + 0.15 * noise) ** 4
        lam = 255.0 * 0.1 * 50.0
        out[i, 0] = (np.log10(rad * lam + 1.0) / math.log10(lam + 1.0)).astype(np.float32)
    return torch.from_numpy(out)


def ldr_frames(n, h=256, w=256, salt="ldr0"):
    """(n,1,h,w) float32 in [0,1): stand-in for the DIV2K / SICE LDR crops."""
    return torch.from_numpy(hash_uniform(salt, n * h * w).reshape(n, 1, h, w).copy())
