"""ORACLE (test infrastructure): numpy restatement of `npy_loader` (utils/ProcessedDatasetFolderImg.py:43-206), given the random
choices explicitly.  cv2.resize (INTER_LINEAR) and cv2.cvtColor(RGB2YUV)[..., :1] are restated from OpenCV's published rules
(cv2 is absent from the reference tree and from this image: that part is unpinned); the HDR branch follows the reference line
by line and is pinned by tests/golden/loader.npz.  See oracle/__init__.py for the usage rules."""
import numpy as np

from .hdr_io import _lin_coord


def resize_linear(img, rh, rw):
    """cv2.resize(img, (rw, rh)) for float32 (H, W, C): horizontal pass, then vertical, float32 products and sums"""
    H, W = img.shape[:2]
    if (rh, rw) == (H, W):
        return img.astype(np.float32).copy()
    y0, y1, fy = _lin_coord(rh, H)
    x0, x1, fx = _lin_coord(rw, W)
    img = img.astype(np.float32)
    ax, ay = (np.float32(1) - fx)[None, :, None], (np.float32(1) - fy)[:, None, None]
    rows = img[:, x0] * ax + img[:, x1] * fx[None, :, None]
    return (rows[y0] * ay + rows[y1] * fy[:, None, None]).astype(np.float32)


def frame(arr, draw, hdr, normalization="bugy_max_normalization", max_stretch=1.0, min_stretch=0.0, brightness_factor=1.0):
    """One frame of npy_loader for the choices draw = (rh, rw, yy, xx) -> dict of float32 arrays (C, 256, 256)."""
    rh, rw, yy, xx = draw
    f32 = np.float32
    color = resize_linear(np.asarray(arr, f32), rh, rw)[yy:yy + 256, xx:xx + 256]
    out = {"color": color.transpose(2, 0, 1).copy()}
    if not hdr:
        y = ((color[..., 0] * f32(0.299) + color[..., 1] * f32(0.587)) + color[..., 2] * f32(0.114)).astype(f32)
        if normalization == "bugy_max_normalization":
            y = (y / f32(255)).astype(f32)   # ProcessedDatasetFolder.py:18-19: a division
        elif normalization == "max_normalization":
            y = y / y.max()
        elif normalization == "stretch":
            y = np.clip(((y - y.min()) / y.max()) * f32(max_stretch) - f32(min_stretch), 0, 1).astype(f32)
        out["input"] = y[None]
        return out
    c = out["color"]
    gray = ((f32(0.299) * c[0] + f32(0.587) * c[1]) + f32(0.114) * c[2]).astype(f32)       # hdr_image_util.py:68-74
    out["gray_norm"] = (gray / gray.max())[None]
    g = (gray - gray.min()).astype(f32)
    out["gray"] = g[None]
    a = np.log10((g / g.max()) * f32(brightness_factor) + f32(1)).astype(f32)
    out["input"] = (a / a.max())[None]
    return out
