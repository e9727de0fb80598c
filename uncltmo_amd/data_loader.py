"""GPU-side data path: `npy_loader` of the reference (utils/ProcessedDatasetFolderImg.py:43-206; the video variant
utils/ProcessedDatasetFolder.py:43-236 runs the same per-frame arithmetic) with the same name, arguments and return tuple.

What the reference does per sample on the host with cv2 / numpy and then copies to the GPU frame by frame
(`.cuda()` inside `__getitem__`), runs here as device kernels on the uploaded array: random resize (cv2.resize, INTER_LINEAR),
random 256 x 256 crop, RGB -> Y, LDR normalisation or HDR log compression.  The random choices are drawn on the host from
numpy's global generator in the reference's order (mode, size, crop x, crop y -- per frame), so a seeded run picks the same
windows.  cv2 is absent from the reference tree and this image: the resize / cvtColor steps follow OpenCV's published rules
(parity with cv2 itself unpinned); the HDR branch is pinned by a golden captured from the reference."""
import os

import numpy as np
import torch

from . import _hip, frame_util

PATCH = 256


def get_f(f_dict_path, im_name, factor_coeff):
    """ProcessedDatasetFolderImg.py:27-36: lambda table lookup -> brightness factor."""
    data = np.load(f_dict_path, allow_pickle=True)
    if im_name in data[()]:
        return data[()][im_name] * 255 * factor_coeff
    raise Exception("no lambda found for file %s in %s" % (im_name, f_dict_path))


def _draw(h, w, always):
    """The reference's random choices for one frame, in its order of np.random calls (:62-79 / :103-121)."""
    rh, rw = h, w
    if always or h != PATCH:
        mode = np.random.randint(0, 2)
        rh = PATCH if mode == 0 else int(np.random.uniform(256, 512))
        rw = rh
    xx = yy = 0
    if rh != PATCH:
        xx = np.random.randint(0, rw - PATCH)
        yy = np.random.randint(0, rh - PATCH)
    return rh, rw, yy, xx


def _frame(src, h, w, draw, y_scale, want_y):
    lib = _hip.lib()
    rh, rw, yy, xx = draw
    color = torch.empty(3, PATCH, PATCH, dtype=torch.float32, device=src.device)
    y = torch.empty(1, PATCH, PATCH, dtype=torch.float32, device=src.device) if want_y else None
    _hip.check(lib.uncl_loader_resize_crop(src.data_ptr(), h, w, rh, rw, yy, xx, PATCH, float(y_scale), color.data_ptr(),
                                           y.data_ptr() if want_y else None, _hip.stream_ptr()), "uncl_loader_resize_crop")
    return color, y


def npy_loader(path, addFrame, hdrMode, ldrNegMode, normalization, min_stretch, max_stretch, factor_coeff, use_contrast_ratio_f,
               use_hist_fit, f_dict_path, final_shape_addition, real_video=False, device="cuda", brightness_factor=None):
    """-> (input_im_frames (2,1,256,256), color_im_frames (2,3,256,256), gray_norm, gray, brightness_factor); LDR modes return the
    input frames in the third and fourth slot and 0, like the reference.  `path` may also be an (H,W,3) array / tensor."""
    lib = _hip.lib()
    if isinstance(path, (str, os.PathLike)):
        arr = np.load(path, allow_pickle=True)
    else:
        arr = path
    src = torch.as_tensor(np.asarray(arr, dtype=np.float32) if not torch.is_tensor(arr) else arr, dtype=torch.float32).to(device)
    src = src.contiguous()
    if src.dim() != 3 or src.shape[2] != 3:
        raise ValueError("npy_loader expects an (H, W, 3) array, got %s" % (tuple(src.shape),))
    if not src.is_cuda:
        raise _hip.HipError("npy_loader runs on the MI355X; device=%s" % device)
    h, w = int(src.shape[0]), int(src.shape[1])
    inputs, colors, gnorms, grays = [], [], [], []
    bf = 0
    for _ in range(2):
        # the LDR-negative branch always draws a mode; the other branch only for inputs that are not already 256 high
        draw = _draw(h, w, always=bool(ldrNegMode))
        if not hdrMode or ldrNegMode:
            y_scale = -255.0 if normalization == "bugy_max_normalization" else 1.0      # negative: divide (the reference's `/ 255`)
            color, y = _frame(src, h, w, draw, y_scale, True)
            if normalization in ("max_normalization", "stretch"):
                ws = torch.empty(2050, dtype=torch.float32, device=src.device)
                _hip.check(lib.uncl_loader_ldr_normalize(y.data_ptr(), y.numel(), 0 if normalization == "max_normalization" else 1,
                                                         float(max_stretch), float(min_stretch), ws.data_ptr(), _hip.stream_ptr()),
                           "uncl_loader_ldr_normalize")
            inputs.append(y.unsqueeze(0))
            colors.append(color.unsqueeze(0))
            continue
        color, _ = _frame(src, h, w, draw, 1.0, False)
        if brightness_factor is None:
            bf = get_f(f_dict_path, os.path.splitext(os.path.basename(str(path)))[0], factor_coeff)
        else:
            bf = brightness_factor
        gray_log = torch.empty(1, PATCH, PATCH, dtype=torch.float32, device=src.device)
        stats = torch.empty(4, dtype=torch.float32, device=src.device)
        ws = torch.empty(lib.uncl_frame_workspace_bytes(), dtype=torch.uint8, device=src.device)
        _hip.check(lib.uncl_hdr_log_gray(color.data_ptr(), PATCH, PATCH, float(bf), None, gray_log.data_ptr(), stats.data_ptr(),
                                         ws.data_ptr(), _hip.stream_ptr()), "uncl_hdr_log_gray")
        gn = torch.empty(1, PATCH, PATCH, dtype=torch.float32, device=src.device)
        gs = torch.empty(1, PATCH, PATCH, dtype=torch.float32, device=src.device)
        _hip.check(lib.uncl_loader_gray_outputs(color.data_ptr(), PATCH, PATCH, stats.data_ptr(), gn.data_ptr(), gs.data_ptr(),
                                                _hip.stream_ptr()), "uncl_loader_gray_outputs")
        inp = gray_log
        if addFrame:
            inp = frame_util.add_frame_to_im(inp, final_shape_addition, final_shape_addition)
        inputs.append(inp.unsqueeze(0))
        colors.append(color.unsqueeze(0))
        gnorms.append(gn.unsqueeze(0))
        grays.append(gs.unsqueeze(0))
    inp, col = torch.cat(inputs, 0), torch.cat(colors, 0)
    if not hdrMode or ldrNegMode:
        return inp, col, inp, inp, 0
    return inp.float(), col.float(), torch.cat(gnorms, 0).float(), torch.cat(grays, 0).float(), bf
