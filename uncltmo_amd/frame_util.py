"""Frame helpers either side of the tiler, MI355X-native: same names and argument meaning as the reference's
utils/data_loader_util.py (resize_im, add_frame_to_im, add_frame_to_im_batch, crop_input_hdr_batch :135-185) and
utils/hdr_image_util.py (to_gray_tensor is folded into hdr_log_gray; back_to_color_tensor :120-131; to_0_1_range_outlier
:93-103), all on device tensors through csrc/frame_ops.hip.  There is no CPU path."""
import ctypes as C

import numpy as np
import torch

from . import _hip


def _ws(dev):
    return torch.empty(_hip.lib().uncl_frame_workspace_bytes(), dtype=torch.uint8, device=dev)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise _hip.HipError("%s is on %s; the HIP path needs it on the MI355X" % (what, t.device))


def add_frame_to_im_batch(images_batch, diffX, diffY):
    """(B,C,H,W) -> replicate-padded (B,C,H+diffY,W+diffX); data_loader_util.py:182-185."""
    _need_gpu(images_batch, "images_batch")
    b, c, h, w = images_batch.shape
    x = images_batch.float().contiguous()
    y = torch.empty(b, c, h + diffY, w + diffX, dtype=torch.float32, device=x.device)
    _hip.check(_hip.lib().uncl_replicate_pad(x.data_ptr(), y.data_ptr(), b * c, h, w, diffY // 2, diffX // 2, h + diffY, w + diffX,
                                             _hip.stream_ptr()), "uncl_replicate_pad")
    return y


def add_frame_to_im(input_im, diffX, diffY):
    """(C,H,W) -> replicate-padded; data_loader_util.py:175-179."""
    return add_frame_to_im_batch(input_im.unsqueeze(0), diffX, diffY).squeeze(0)


def resize_im(im, add_frame, final_shape_addition):
    """data_loader_util.py:135-158: pad (C,H,W) with replicated borders to 16*floor(H/16)+16 so that the generator's tiles
    need no further padding (upstream overrides `add_frame` to True).  Returns (im, diffY, diffX)."""
    h, w = im.shape[1], im.shape[2]
    h1, w1 = int(16 * int(h / 16.)) + 16, int(16 * int(w / 16.)) + 16
    diffY, diffX = abs(h - h1), abs(w - w1)
    return add_frame_to_im(im, diffX=diffX, diffY=diffY), diffY, diffX


def crop_input_hdr_batch(input_hdr_batch, diffY, diffX):
    """data_loader_util.py:165-172 (a view)."""
    b, c, h, w = input_hdr_batch.shape
    th, tw = h - diffY, w - diffX
    i, j = int(round((h - th) / 2.)), int(round((w - tw) / 2.))
    return input_hdr_batch[:, :, i:i + th, j:j + tw]


def clip_to_crops(clips, crop=256):
    """(B,T,C,H,W) clips with H, W multiples of `crop` -> (B*(H/crop)*(W/crop), T, C, crop, crop): every spatial window of a
    clip becomes a clip of its own (row-major window order inside each source clip), so that a T-frame sequence stays one
    recurrent unit.  The published generator only accepts 256x256 inputs (gcn.pos_embed is (1,256,12,12): Unet.py:66,94),
    so the 512x512 training clips of BASELINE configs[3] enter the video trainer (GanTrainer.py:263-291) as four crops each."""
    B, T, Cc, H, W = clips.shape
    if H % crop or W % crop:
        raise ValueError("clip_to_crops needs H and W to be multiples of %d (got %dx%d)" % (crop, H, W))
    ny, nx = H // crop, W // crop
    x = clips.reshape(B, T, Cc, ny, crop, nx, crop).permute(0, 3, 5, 1, 2, 4, 6)
    return x.reshape(B * ny * nx, T, Cc, crop, crop).contiguous()


def crops_to_clip(crops, ny, nx):
    """Inverse of clip_to_crops: (B*ny*nx, T, C, h, w) -> (B, T, C, ny*h, nx*w)."""
    n, T, Cc, h, w = crops.shape
    B = n // (ny * nx)
    x = crops.reshape(B, ny, nx, T, Cc, h, w).permute(0, 3, 4, 1, 5, 2, 6)
    return x.reshape(B, T, Cc, ny * h, nx * w).contiguous()


def hdr_log_gray(rgb_img, f_factor):
    """The tensor arithmetic of load_inference / load_inference2 (model_save_util.py:209-217): (3,H,W) linear radiance ->
    (rgb shifted to be non-negative (3,H,W), log-compressed luminance in [0,1] (1,H,W))."""
    _need_gpu(rgb_img, "rgb_img")
    lib = _hip.lib()
    x = rgb_img.float().contiguous()
    _, h, w = x.shape
    rgb = torch.empty_like(x)
    gray = torch.empty(1, h, w, dtype=torch.float32, device=x.device)
    stats = torch.empty(4, dtype=torch.float32, device=x.device)
    ws = _ws(x.device)
    _hip.check(lib.uncl_hdr_log_gray(x.data_ptr(), h, w, float(f_factor), rgb.data_ptr(), gray.data_ptr(), stats.data_ptr(),
                                     ws.data_ptr(), _hip.stream_ptr()), "uncl_hdr_log_gray")
    return rgb, gray


def _numpy_percentile_plan(n, q):
    """(previous index, next index, finish(prev_value, next_value)) exactly as the installed numpy's np.percentile (method
    'linear') treats a float32 array of n values.  numpy 2 derives the fractional rank in the array's dtype (float32), older
    releases in float64: the plan is built from numpy's own helpers when they are importable so that the result matches
    `np.percentile(x.cpu().numpy(), q)` (model_save_util.py:389-390) on this host bit for bit; otherwise float64 rules."""
    try:
        from numpy.lib import _function_base_impl as npf
        q_ = np.asanyarray(np.true_divide(q, np.float32(100)))
        props = npf._QuantileMethods["linear"]
        virt = np.asanyarray(props["get_virtual_index"](n, q_))
        if np.issubdtype(virt.dtype, np.integer):
            i = int(virt)
            return i, i, (lambda a, b: np.float32(a)), (0.0, False)
        prev, nxt = npf._get_indexes(np.empty(1, dtype=np.float32), virt, n)
        gamma = npf._get_gamma(virt, prev, props)
        fin = lambda a, b: npf._lerp(np.float32(a), np.float32(b), gamma)
        # (gamma, arithmetic in float64?) for the device form of the same interpolation (uncl_percentile_lerp)
        return int(prev) % n, int(nxt) % n, fin, (float(gamma), np.asarray(gamma).dtype == np.float64)
    except (ImportError, AttributeError, KeyError, TypeError):
        pos = q / 100.0 * (n - 1)
        lo = int(np.floor(pos))
        t = pos - lo

        def fin(a, b):
            a, b = np.float32(a), np.float32(b)
            d = b - a
            return np.float32(a + d * t if t < 0.5 else b - d * (1 - t))
        return lo, min(lo + 1, n - 1), fin, None


def percentile(x, qs, on_device=False):
    """np.percentile(x.cpu().numpy(), q) for each q (linear interpolation), from exact order statistics selected on the
    device: a few floats cross to the host instead of the image (model_save_util.py:389-390, hdr_image_util.py:93-97).
    on_device=True: the interpolation runs on the device as well and a (len(qs),) fp32 device tensor is returned -- nothing
    crosses to the host (same values bit for bit: numpy's `_lerp` restated with separately rounded operations).
    NaNs are not supported (the reference would return NaN)."""
    _need_gpu(x, "x")
    lib = _hip.lib()
    xf = x.float().contiguous()
    n = xf.numel()
    plans = [_numpy_percentile_plan(n, q) for q in qs]
    ranks = []
    for lo, hi, _, _ in plans:
        ranks += [lo, hi]
    out = torch.empty(len(ranks), dtype=torch.float32, device=xf.device)
    ws = _ws(xf.device)
    for i0 in range(0, len(ranks), 8):           # the select resolves up to eight ranks per sweep over the data
        part = ranks[i0:i0 + 8]
        arr = (C.c_ulonglong * len(part))(*part)
        _hip.check(lib.uncl_order_stats(xf.data_ptr(), n, arr, len(part), out.data_ptr() + 4 * i0, ws.data_ptr(),
                                        _hip.stream_ptr()), "uncl_order_stats")
    if on_device and all(p[3] is not None for p in plans) and len(plans) <= 8:
        k = len(plans)
        gam = (C.c_double * k)(*[p[3][0] for p in plans])
        f64 = (C.c_int * k)(*[int(p[3][1]) for p in plans])
        res = torch.empty(k, dtype=torch.float32, device=xf.device)
        _hip.check(lib.uncl_percentile_lerp(out.data_ptr(), gam, f64, k, res.data_ptr(), _hip.stream_ptr()), "uncl_percentile_lerp")
        return res
    v = out.cpu().numpy()
    vals = [fin(v[2 * i], v[2 * i + 1]) for i, (_, _, fin, _) in enumerate(plans)]
    if on_device:       # plans without numpy's helpers: finish on the host, hand back a device tensor all the same
        return torch.tensor([float(np.float32(t)) for t in vals], dtype=torch.float32, device=xf.device)
    return vals


def back_to_color_and_crop(rgb_padded, fake, min_p, max_p, diffY, diffX):
    """model_save_util.py:391-400: clamp to the percentiles, min-max stretch, back_to_color_tensor (hdr_image_util.py:120-131)
    and removal of the resize_im padding, fused.  rgb_padded (3,H1,W1) non-negative, fake (..,H1,W1) -> (3,H,W).  The final
    `clamp(0, im_max)` of the reference cannot change a value (the maximum is taken before the crop) and is dropped."""
    _need_gpu(fake, "fake")
    h1, w1 = rgb_padded.shape[1], rgb_padded.shape[2]
    h, w = h1 - diffY, w1 - diffX
    rgb = rgb_padded.float().contiguous()
    f = fake.reshape(h1, w1).float().contiguous()
    out = torch.empty(3, h, w, dtype=torch.float32, device=f.device)
    if torch.is_tensor(min_p):        # (2,) device tensor [min_p, max_p] from percentile(..., on_device=True); max_p unused
        lohi = min_p.float().contiguous()
        _hip.check(_hip.lib().uncl_color_finish_dev(rgb.data_ptr(), f.data_ptr(), out.data_ptr(), h1, w1, diffY // 2, diffX // 2, h, w,
                                                    lohi.data_ptr(), _hip.stream_ptr()), "uncl_color_finish_dev")
        return out
    _hip.check(_hip.lib().uncl_color_finish(rgb.data_ptr(), f.data_ptr(), out.data_ptr(), h1, w1, diffY // 2, diffX // 2, h, w,
                                            float(np.float32(min_p)), float(np.float32(max_p)), _hip.stream_ptr()),
               "uncl_color_finish")
    return out


def to_uint8_outlier(color, on_device=False):
    """hdr_image_util.save_gray_tensor_as_numpy_stretch up to the file write (:237-241): clamp(0,1), to_0_1_range_outlier
    (percentiles 0.1 / 99.0 over all channels, :93-103), *255, truncate.  (C,H,W) fp32 -> (H,W,C) uint8 on the device."""
    _need_gpu(color, "color")
    lib = _hip.lib()
    x = color.float().contiguous()
    c, h, w = x.shape
    cl = torch.empty_like(x)
    _hip.check(lib.uncl_clamp01(x.data_ptr(), cl.data_ptr(), x.numel(), _hip.stream_ptr()), "uncl_clamp01")
    out = torch.empty(h, w, c, dtype=torch.uint8, device=x.device)
    if on_device:
        lohi = percentile(cl, [0.1, 99.0], on_device=True)
        _hip.check(lib.uncl_to_uint8_dev(cl.data_ptr(), out.data_ptr(), c, h, w, lohi.data_ptr(), _hip.stream_ptr()), "uncl_to_uint8_dev")
        return out
    im_min, im_max = percentile(cl, [0.1, 99.0])
    if float(im_max) - float(im_min) == 0:
        im_max = np.float32(im_max) + np.float32(1e-08)
    _hip.check(lib.uncl_to_uint8(cl.data_ptr(), out.data_ptr(), c, h, w, float(im_min), float(im_max), _hip.stream_ptr()),
               "uncl_to_uint8")
    return out


def warp_flow(img_u8, flow):
    """GanTrainer.warp_flow (GanTrainer.py:584-595): cv2.remap(img, flow + pixel grid, None, cv2.INTER_LINEAR) on the device.
    img_u8: (H,W,C) uint8, flow: (Hf,Wf,2) fp32 displacements (x, y) -> (Hf,Wf,C) uint8.  Unlike the reference this does not add
    the grid into `flow` in place."""
    _need_gpu(img_u8, "img")
    _need_gpu(flow, "flow")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3:
        raise TypeError("warp_flow expects an (H,W,C) uint8 image")
    if flow.dim() != 3 or flow.shape[-1] != 2:
        raise ValueError("flow must be (Hf,Wf,2)")          # the reference asserts the same
    img_u8 = img_u8.contiguous()
    flow = flow.float().contiguous()
    h, w, c = img_u8.shape
    hf, wf = flow.shape[:2]
    out = torch.empty(hf, wf, c, dtype=torch.uint8, device=img_u8.device)
    _hip.check(_hip.lib().uncl_warp_flow(img_u8.data_ptr(), flow.data_ptr(), out.data_ptr(), h, w, c, hf, wf, _hip.stream_ptr()),
               "uncl_warp_flow")
    return out


def compute_flow(img_to_align, img_source, unit_range=None):
    """GanTrainer.compute_flow (GanTrainer.py:620-646; the evaluator's optical flow, Tester.py:379-384): both frames (H,W,C) or
    (H,W), in [0, 255] (or [0, 1] floats, scaled like the reference does), on the GPU -> (H,W,2) fp32 field f with
    img_to_align(p + f(p)) ~ img_source(p), i.e. what `align_frames` / `warp_flow` take.  Channel 0 of each frame is used, as in the
    reference.  The reference's estimator is cv2 DeepFlow (absent here: parity unpinned); this is pyramidal Lucas-Kanade on the
    device (csrc/flow.hip, oracle/flow.py): motions beyond about one pixel per pyramid level and iteration (x 4 iterations) are
    clamped, and warp errors computed through it are NOT numerically comparable with the reference's published (DeepFlow) ones.
    `unit_range`: None = the reference's heuristic (a float frame whose maximum is <= 1 is taken as [0, 1] and brought to 8 bits:
    one host synchronisation per frame, and a dark frame already on the 0 .. 255 scale is misread, as in the reference); True /
    False state the range and skip both."""
    _need_gpu(img_to_align, "img_to_align")
    _need_gpu(img_source, "img_source")

    def plane(t):
        t = t if t.dim() == 2 else t[:, :, 0]
        if t.dtype == torch.uint8:
            return t.float().contiguous()
        t = t.float()
        if unit_range if unit_range is not None else float(t.max()) <= 1.0:    # GanTrainer.py:632-637: [0, 1] images -> 8 bits first
            t = (t * 255).clamp(0, 255)
        return torch.floor(t).contiguous()              # astype(np.uint8) truncates

    a, s = plane(img_to_align), plane(img_source)
    if a.shape != s.shape:
        raise ValueError("compute_flow: the two frames differ in size: %s vs %s" % (tuple(a.shape), tuple(s.shape)))
    h, w = a.shape
    lib = _hip.lib()
    nb = lib.uncl_optical_flow_workspace_bytes(h, w)
    if nb == 0:
        raise ValueError("compute_flow needs frames of at least 2 x 2 pixels")
    ws = torch.empty(nb, dtype=torch.uint8, device=a.device)
    flow = torch.empty(h, w, 2, dtype=torch.float32, device=a.device)
    _hip.check(lib.uncl_optical_flow(a.data_ptr(), s.data_ptr(), h, w, flow.data_ptr(), ws.data_ptr(), nb, _hip.stream_ptr()),
               "uncl_optical_flow")
    return flow


def align_frames(img_to_align, flow, unit_range=None):
    """GanTrainer.align_frames (GanTrainer.py:648-666): 8-bit conversion of a [0, 1] image, then warp_flow.  `unit_range` as in
    compute_flow (None: the reference's max() <= 1 test, one host synchronisation)."""
    _need_gpu(img_to_align, "img_to_align")
    t = img_to_align
    if t.dtype != torch.uint8:
        t = t.float()
        if unit_range if unit_range is not None else float(t.max()) <= 1.0:
            t = (t * 255).clamp(0, 255)
        t = t.to(torch.uint8)
    return warp_flow(t if t.dim() == 3 else t.unsqueeze(-1), flow)

