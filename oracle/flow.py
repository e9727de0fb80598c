"""ORACLE (test infrastructure): dense inverse optical flow for the warp-error metric of the video evaluator.

The reference estimates the flow between two 8-bit frames with `cv2.optflow.createOptFlow_DeepFlow().calc(img1, img0, None)`
(GanTrainer.py:597-646 `estimate_invflow` / `compute_flow`; callers Tester.py:379-384, metrics/compute_wrap_error.py:105-114) and
warps with it (`align_frames`).  OpenCV (and its contrib module `optflow`) is a third-party dependency that is absent from the
reference tree and from this image (README.md lists `opencv-python`, no version), and DeepFlow is a variational method on top of a
learned-free deep matching whose published description does not fix an implementation: **parity with cv2 DeepFlow unpinned**.
What is restated here is the CONTRACT of `compute_flow(img_to_align, img_source)` -- an (H, W, 2) float32 field f with
img_to_align(p + f(p)) ~ img_source(p), ready for `warp_flow` -- by a published, deterministic algorithm that a kernel can follow
operation for operation: coarse-to-fine iterative Lucas-Kanade (Lucas & Kanade 1981; Bouguet's pyramidal form 2001) with box windows:

  pyramid     5-tap binomial blur [1 4 6 4 1] / 16 (edge replicate), every second pixel; levels while min(H, W) >= 2 * MIN_SIDE
  per level   f <- 2 * bilinear-upsampled f of the coarser level (0 at the coarsest);  Ix, Iy = central differences of the
              SOURCE image (edge replicate);  ITERS times:  It = bilinear(img_to_align, p + f) - img_source;
              b = box_(2R+1)(Ix It, Iy It),  G = box_(2R+1)(Ix Ix, Ix Iy, Iy Iy) + LAM I;  d = -G^-1 b, clamped to |d| <= 1 per
              component;  f <- f + d;  f <- box_3(f) / box_3(1)   (a 3 x 3 mean keeps the field regular where G is weak)

It is pinned by synthetic motions with KNOWN flow (tests/test_flow.py: sub-pixel and multi-pixel translations, a rotation + scale
about the image centre), not by cv2.  Only tests/ and the smoke / cpu-baseline legs may import this module (oracle/__init__.py)."""
import numpy as np

MIN_SIDE = 16      # the coarsest level keeps at least this many pixels per side
ITERS = 4
RADIUS = 7         # 15 x 15 window
LAM = 1e-2         # Tikhonov term on G (images in [0, 255]: gradients ~1 ... 50, window sums ~1e2 ... 1e5)


def _blur_down(img):
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0
    p = np.pad(img, ((0, 0), (2, 2)), mode="edge")
    h = sum(k[i] * p[:, i:i + img.shape[1]] for i in range(5))
    p = np.pad(h, ((2, 2), (0, 0)), mode="edge")
    v = sum(k[i] * p[i:i + img.shape[0], :] for i in range(5))
    return v[::2, ::2]


def _grad(img):
    p = np.pad(img, 1, mode="edge")
    return 0.5 * (p[1:-1, 2:] - p[1:-1, :-2]), 0.5 * (p[2:, 1:-1] - p[:-2, 1:-1])


def _bilinear(img, x, y):
    H, W = img.shape
    x = np.clip(x, 0.0, W - 1.0)
    y = np.clip(y, 0.0, H - 1.0)
    x0 = np.minimum(np.floor(x).astype(np.int64), W - 2) if W > 1 else np.zeros_like(x, np.int64)
    y0 = np.minimum(np.floor(y).astype(np.int64), H - 2) if H > 1 else np.zeros_like(y, np.int64)
    fx, fy = x - x0, y - y0
    return ((1 - fy) * ((1 - fx) * img[y0, x0] + fx * img[y0, x0 + 1]) + fy * ((1 - fx) * img[y0 + 1, x0] + fx * img[y0 + 1, x0 + 1]))


def _box(a, r):
    """zero-padded (2r+1)^2 box SUM"""
    p = np.pad(a, ((0, 0), (r, r)))
    c = np.cumsum(np.pad(p, ((0, 0), (1, 0))), axis=1)
    h = c[:, 2 * r + 1:] - c[:, :-(2 * r + 1)]
    p = np.pad(h, ((r, r), (0, 0)))
    c = np.cumsum(np.pad(p, ((1, 0), (0, 0))), axis=0)
    return c[2 * r + 1:, :] - c[:-(2 * r + 1), :]


def _upsample2(f, H, W):
    """the coarser level's field at this level's pixel centres: pixel (y, x) here sits at (y / 2, x / 2) there; values x 2"""
    ys, xs = np.arange(H)[:, None] * 0.5, np.arange(W)[None, :] * 0.5
    return 2.0 * _bilinear(f, np.broadcast_to(xs, (H, W)), np.broadcast_to(ys, (H, W)))


def pyramid(img):
    levels = [np.asarray(img, np.float64)]
    while min(levels[-1].shape) >= 2 * MIN_SIDE:
        levels.append(_blur_down(levels[-1]))
    return levels


def compute_flow(img_to_align, img_source):
    """(H, W) or (H, W, C) images in [0, 255] (channel 0 is used, like GanTrainer.compute_flow:640-641) -> (H, W, 2) float32 field f,
    f[..., 0] along x, with img_to_align(p + f(p)) ~ img_source(p)."""
    a0 = np.asarray(img_to_align, np.float64)
    s0 = np.asarray(img_source, np.float64)
    if a0.ndim == 3:
        a0, s0 = a0[:, :, 0], s0[:, :, 0]
    pa, ps = pyramid(a0), pyramid(s0)
    fx = fy = None
    for lv in range(len(pa) - 1, -1, -1):
        A, S = pa[lv], ps[lv]
        H, W = S.shape
        if fx is None:
            fx, fy = np.zeros((H, W)), np.zeros((H, W))
        else:
            fx, fy = _upsample2(fx, H, W), _upsample2(fy, H, W)
        Ix, Iy = _grad(S)
        gxx, gxy, gyy = _box(Ix * Ix, RADIUS) + LAM, _box(Ix * Iy, RADIUS), _box(Iy * Iy, RADIUS) + LAM
        det = gxx * gyy - gxy * gxy
        ones3 = _box(np.ones((H, W)), 1)
        xs, ys = np.broadcast_to(np.arange(W)[None, :], (H, W)), np.broadcast_to(np.arange(H)[:, None], (H, W))
        for _ in range(ITERS):
            It = _bilinear(A, xs + fx, ys + fy) - S
            bx, by = _box(Ix * It, RADIUS), _box(Iy * It, RADIUS)
            dx = np.clip(-(gyy * bx - gxy * by) / det, -1.0, 1.0)
            dy = np.clip(-(gxx * by - gxy * bx) / det, -1.0, 1.0)
            fx, fy = _box(fx + dx, 1) / ones3, _box(fy + dy, 1) / ones3
    return np.stack([fx, fy], -1).astype(np.float32)
