"""ORACLE (test infrastructure): CPU restatement of the inference pre- / post-processing around the tiler.

Follows utils/model_save_util.py:203-217 (load_inference arithmetic), :389-401 (percentile clamp, stretch, colour),
utils/data_loader_util.py:135-185 (resize_im, add_frame_to_im), utils/hdr_image_util.py:68-74 (to_gray_tensor),
:93-103 (to_0_1_range_outlier), :120-131 (back_to_color_tensor), :237-241 (save_gray_tensor_as_numpy_stretch).
File reading / writing (imageio, cv2.resize) is outside the path.  See oracle/__init__.py for the usage rules.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPSILON = 1e-08      # utils/params.py:48


def to_gray_tensor(rgb):
    """hdr_image_util.py:68-74."""
    return (0.299 * rgb[0] + 0.587 * rgb[1] + 0.114 * rgb[2])[None, :, :]


def hdr_log_gray(rgb, f_factor):
    """model_save_util.py:209-217: (3,H,W) linear radiance -> (rgb shifted to >= 0, log-compressed luminance (1,H,W))."""
    if rgb.min() < 0:
        rgb = rgb - rgb.min()
    gray = to_gray_tensor(rgb)
    gray = gray - gray.min()
    gray = torch.log10((gray / gray.max()) * f_factor + 1)
    gray = gray / gray.max()
    return rgb, gray


def add_frame_to_im(im, diffX, diffY):
    """data_loader_util.py:175-179."""
    return F.pad(im.unsqueeze(0), (diffX // 2, diffX - diffX // 2, diffY // 2, diffY - diffY // 2), mode="replicate").squeeze(0)


def resize_im(im, add_frame=True, final_shape_addition=0):
    """data_loader_util.py:135-158: replicate-pad (C,H,W) to 16*floor(H/16)+16 (the `add_frame` argument is overridden to
    True upstream)."""
    h, w = im.shape[1], im.shape[2]
    h1, w1 = int(16 * int(h / 16.)) + 16, int(16 * int(w / 16.)) + 16
    diffY, diffX = abs(h - h1), abs(w - w1)
    return add_frame_to_im(im, diffX=diffX, diffY=diffY), diffY, diffX


def back_to_color_tensor(im_hdr, fake):
    """hdr_image_util.py:120-131."""
    if im_hdr.min() < 0:
        im_hdr = im_hdr - im_hdr.min()
    g = to_gray_tensor(im_hdr)
    norm = torch.zeros(im_hdr.shape)
    for c in range(3):
        norm[c] = im_hdr[c] / (g + EPSILON)
    return torch.pow(norm, 0.5) * fake


def finish(rgb_padded, fake, diffY, diffX):
    """model_save_util.py:389-401: fake (1,1,H1,W1) -> colour image (3,H,W), padding removed."""
    max_p = np.percentile(fake.numpy(), 99.5)
    min_p = np.percentile(fake.numpy(), 0.5)
    fake2 = fake.clamp(min_p, max_p)
    stretch = (fake2 - fake2.min()) / (fake2.max() - fake2.min())
    col = back_to_color_tensor(rgb_padded, stretch[0])
    im_max = col.max()
    col = col[:, diffY // 2:-(diffY - diffY // 2), diffX // 2:-(diffX - diffX // 2)]
    return col.clamp(min=0, max=im_max)


def to_uint8(col):
    """hdr_image_util.py:237-241 with :93-103; `np.float32` scalars keep the arithmetic in float32 (numpy 1.x semantics of
    the reference's era; under numpy 2 the float32 percentile scalars behave the same way)."""
    t = col.clamp(0, 1).permute(1, 2, 0).numpy()
    im_max = np.percentile(t, 99.0)
    im_min = np.percentile(t, 0.1)
    if np.max(t) - np.min(t) == 0:
        t = (t - im_min) / (im_max - im_min + EPSILON)
    else:
        t = (t - im_min) / (im_max - im_min)
    return (np.clip(t, 0, 1) * 255).astype("uint8")
