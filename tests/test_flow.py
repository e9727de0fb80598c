"""The optical-flow substitute of the video evaluator's warp error (Tester.py:379-389, GanTrainer.py:597-646, metrics/
compute_wrap_error.py:91-125: cv2 DeepFlow, absent from the reference tree and this image -- parity with cv2 unpinned).  The oracle
(oracle/flow.py, pyramidal Lucas-Kanade) is pinned HERE by synthetic motions whose flow is known; the HIP kernels must follow the
oracle (tests/test_gpu_flow.py)."""
import numpy as np

from oracle import flow as OF
from oracle.tester import warp_flow as o_warp


def texture(h, w, seed=0):
    """a smooth random texture in [0, 255] with structure at several scales (flow needs gradients everywhere)"""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w))
    for s, amp in ((4, 1.0), (9, 1.0), (21, 1.5)):
        g = rng.standard_normal((h // s + 3, w // s + 3))
        ys, xs = np.arange(h)[:, None] / s, np.arange(w)[None, :] / s
        img += amp * OF._bilinear(g, np.broadcast_to(xs + 1, (h, w)), np.broadcast_to(ys + 1, (h, w)))
    img -= img.min()
    return 255.0 * img / img.max()


def sample(img, fx, fy):
    h, w = img.shape
    xs, ys = np.broadcast_to(np.arange(w)[None, :], (h, w)), np.broadcast_to(np.arange(h)[:, None], (h, w))
    return OF._bilinear(img, xs + fx, ys + fy)


def epe(f, fx, fy, border):
    d = np.hypot(f[..., 0] - fx, f[..., 1] - fy)
    return float(d[border:-border, border:-border].mean()), float(d[border:-border, border:-border].max())


def test_translations_are_recovered():
    """img_source(p) = img_to_align(p + t): the field must be t everywhere (away from the frame)"""
    A = texture(96, 128, 1)
    for tx, ty in ((0.0, 0.0), (0.4, -0.3), (2.5, 1.25), (-5.0, 3.0), (7.5, -6.0)):
        S = sample(A, tx, ty)
        f = OF.compute_flow(A, S)
        assert f.shape == (96, 128, 2) and f.dtype == np.float32
        mean, worst = epe(f, tx, ty, 16)
        assert mean < 0.05 and worst < 0.3, (tx, ty, mean, worst)


def test_rotation_and_scale_are_recovered():
    A = texture(120, 120, 2)
    h, w = A.shape
    ys, xs = np.broadcast_to(np.arange(h)[:, None], (h, w)) - (h - 1) / 2, np.broadcast_to(np.arange(w)[None, :], (h, w)) - (w - 1) / 2
    th, sc = np.deg2rad(2.0), 1.02
    fx = sc * (np.cos(th) * xs - np.sin(th) * ys) - xs
    fy = sc * (np.sin(th) * xs + np.cos(th) * ys) - ys
    S = sample(A, fx, fy)
    f = OF.compute_flow(A, S)
    mean, worst = epe(f, fx, fy, 16)
    assert mean < 0.08 and worst < 0.5, (mean, worst)


def test_alignment_reduces_the_warp_error():
    """the use the evaluator makes of it: warp frame 1 onto frame 0 with the estimated field (GanTrainer.align_frames)"""
    A = texture(96, 128, 3)
    S = sample(A, 3.0, -2.0)
    a8 = np.clip(np.rint(A), 0, 255).astype(np.uint8)[..., None].repeat(3, -1)
    s8 = np.clip(np.rint(S), 0, 255).astype(np.uint8)[..., None].repeat(3, -1)
    f = OF.compute_flow(a8, s8)
    aligned = o_warp(a8, f)
    before = np.abs(a8.astype(float) - s8)[16:-16, 16:-16].mean()
    after = np.abs(aligned.astype(float) - s8)[16:-16, 16:-16].mean()
    assert after < 0.15 * before and after < 1.5, (before, after)


def test_three_channel_input_uses_channel_zero_and_sizes_are_free():
    A = texture(70, 53, 4)
    S = sample(A, 1.0, 0.5)
    a3 = np.stack([A, 255 - A, A * 0], -1)
    s3 = np.stack([S, S * 0, 255 - S], -1)
    f3, f1 = OF.compute_flow(a3, s3), OF.compute_flow(A, S)
    assert np.array_equal(f3, f1) and f3.shape == (70, 53, 2)
