"""ORACLE (test infrastructure): CPU restatement of the tensor path of `Tester.eval_on_video` (Tester.py:314-391).

Per frame: load_inference's arithmetic (Tester.py:229-251 = model_save_util.py:209-217), resize_im (data_loader_util.py:135-158),
the whole clip through the 5-D overlap tiler with the recurrent video generator (Tester.py:150-227 = the tiler of
model_save_util.py:488-565), percentile clamp / stretch / colour / crop (:343-358), tensor_to_numpy + to_0_1_range_outlier
(:393-410), TMQI of (original RGB, 8-bit result) (:373), scene score = mean; the two warp-error formulas (:387-389) on a frame
pair the caller has aligned (the reference aligns with cv2 DeepFlow on images of ANOTHER method read from disk: outside the
path; the golden fixture is captured with an identity alignment).  See oracle/__init__.py for the usage rules.
"""
import numpy as np
import torch

from . import inference as OI
from . import tiler as OT
from . import tmqi as OTM
from .generator import unet_video_forward


def rgb_to_y(rgb):
    """TMQI.py:46-49."""
    return 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]


def warp_errors(img0_target_u8, img1_aligned_u8, border=32):
    a = img1_aligned_u8.astype(np.float32) / 255.0
    b = img0_target_u8.astype(np.float32) / 255.0
    a, b = a[border:-border, border:-border, :], b[border:-border, border:-border, :]
    return float(np.mean(np.power(a - b, 2))), float(np.mean(np.abs(a - b) / (1e-8 + a + b)))


def warp_flow(img_u8, flow):
    """GanTrainer.warp_flow (GanTrainer.py:584-595): cv2.remap(img, flow + pixel grid, None, cv2.INTER_LINEAR).  cv2 is a
    third-party dependency that is absent from the reference tree and from this image (README.md lists `opencv-python`, no version):
    **parity with cv2 unpinned**.  Restated from OpenCV's published algorithm for CV_8U / INTER_LINEAR / BORDER_CONSTANT(0)
    (imgproc/src/imgwarp.cpp, remap + remapBilinear, 4.x): map coordinates rounded to 1/32 pixel with cvRound (half to even),
    integer weights (32-fy)(32-fx)*32 ... of 2^15, result (sum + 2^14) >> 15, taps outside the image contribute the border value 0.
    Pinned by hand-computed vectors in tests/test_warp_flow.py.  img_u8 (H,W,C) uint8, flow (Hf,Wf,2) float32 -> (Hf,Wf,C) uint8;
    `flow` is not modified (the reference adds the grid in place)."""
    img = np.asarray(img_u8)
    assert img.dtype == np.uint8 and img.ndim == 3 and flow.ndim == 3 and flow.shape[-1] == 2
    H, W, C = img.shape
    hf, wf = flow.shape[:2]
    mx = (flow[:, :, 0].astype(np.float32) + np.arange(wf, dtype=np.float32)).astype(np.float32)
    my = (flow[:, :, 1].astype(np.float32) + np.arange(hf, dtype=np.float32)[:, None]).astype(np.float32)
    sxq = np.rint(np.nan_to_num(mx * np.float32(32.0), nan=0.0).clip(-2147483000.0, 2147483000.0)).astype(np.int64)
    syq = np.rint(np.nan_to_num(my * np.float32(32.0), nan=0.0).clip(-2147483000.0, 2147483000.0)).astype(np.int64)
    ax, ay = sxq & 31, syq & 31
    sx, sy = np.clip(sxq >> 5, -32768, 32767), np.clip(syq >> 5, -32768, 32767)
    w = [(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32]
    out = np.zeros((hf, wf, C), np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        pix = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64) * ok[..., None]
        out += pix * w[k][..., None]
    return ((out + (1 << 14)) >> 15).astype(np.uint8)


def eval_on_video(sd, rgb_frames_hwc, f_factor, align=None):
    """sd: video generator state dict; rgb_frames_hwc: list of (H,W,3) float32 numpy frames of one scene.
    Returns (tmqi_scene, [uint8 (H,W,3)], per-frame TMQI[, warp_mse, warp_rel])."""
    padded, grays = [], []
    diffY = diffX = 0
    for im in rgb_frames_hwc:
        rgb = torch.from_numpy(im.transpose(2, 0, 1)).float()                   # tranforms.ToTensor (tranforms.py:35-44)
        rgb_s, gray = OI.hdr_log_gray(rgb, f_factor)
        rgb_p, diffY, diffX = OI.resize_im(rgb_s)
        gray_p, diffY, diffX = OI.resize_im(gray)
        padded.append(rgb_p)
        grays.append(gray_p.unsqueeze(0).unsqueeze(0))
    clip = torch.cat(grays, 1)                                                   # (1,T,1,H1,W1)
    # apply_crop=False (Tester.py:342): the generator's output crop is off, the padding comes off after the colour step
    fakes = OT.tiled_forward(clip, lambda x: unet_video_forward(sd, x))
    results, scores = [], []
    for i, im in enumerate(rgb_frames_hwc):
        col = OI.finish(padded[i], fakes[:, i], diffY, diffX)
        ldr = OI.to_uint8(col)
        results.append(ldr)
        q = OTM.tmqi(rgb_to_y(im.astype(np.float64)), rgb_to_y(ldr.astype(np.float32).astype(np.float64)))[0]
        scores.append(q)
    scene = float(np.mean(scores))
    if align is None or len(results) < 2:
        return scene, results, scores
    mse, rel = warp_errors(results[0], align(results[1], results[0]))
    return scene, results, scores, mse, rel
