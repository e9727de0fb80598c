"""Single-frame inference entry, MI355X-native: the tensor path of the reference's `run_model_on_single_image2`
(utils/model_save_util.py:293-404) between its file read and its file write -- log compression, replicate padding to the
tile grid, overlap-tiled generator (uncltmo_amd.tiler), percentile clamp + stretch, colour restoration, 8-bit stretch -- with
every stage on the device.  `run_model_on_single_image2` adds the file read for Radiance .hdr input (uncltmo_amd.hdr_io) and
the reference's 1/4 down-scale; .exr / .dng input and writing the .png stay with the caller."""
import os

import numpy as np
import torch

from . import frame_util, hdr_io
from .tiler import test_big_size_image2


@torch.no_grad()
def run_model_on_frame(G_net, rgb_img, f_factor, model_params=None, final_shape_addition=0):
    """rgb_img: (3,H,W) linear radiance on the GPU; f_factor: lambda * 255 * factor_coeff of that frame
    (model_save_util.py:221-222).  Returns (colour image (3,H,W) fp32, 8-bit image (H,W,3) uint8)."""
    add_frame = (model_params or {}).get("add_frame", 1)
    rgb_img, gray_im_log = frame_util.hdr_log_gray(rgb_img, f_factor)
    rgb_img, diffY, diffX = frame_util.resize_im(rgb_img, add_frame, final_shape_addition)
    gray_im_log, diffY, diffX = frame_util.resize_im(gray_im_log, add_frame, final_shape_addition)
    fake = test_big_size_image2(input_data=gray_im_log.unsqueeze(0), model=G_net, apply_crop=add_frame, diffY=diffY, diffX=diffX)
    # the four percentiles stay on the device: nothing between the file read and the 8-bit image waits for the host
    lohi = frame_util.percentile(fake, [0.5, 99.5], on_device=True)
    color = frame_util.back_to_color_and_crop(rgb_img, fake, lohi, None, diffY, diffX)
    return color, frame_util.to_uint8_outlier(color, on_device=True)


def load_inference2(im_path, f_factor_path, factor_coeff, device, scale=4):
    """model_save_util.py:219-240 for '.hdr' (Radiance) and '.npy' input: -> (rgb (3,h,w), log-compressed luminance (1,h,w),
    f_factor).  f_factor_path: the reference's lambda table (a pickled dict name -> lambda in an .npy file) or a plain number."""
    name = os.path.splitext(os.path.basename(im_path))[0]
    if isinstance(f_factor_path, (int, float)):
        lam = float(f_factor_path)
    else:
        lam = float(np.load(f_factor_path, allow_pickle=True)[()][name])
    f_factor = lam * 255 * factor_coeff
    ext = os.path.splitext(im_path)[1]
    if ext == ".hdr":
        rgb = hdr_io.read_hdr(im_path, device=device, scale=scale)
    elif ext == ".npy":
        if scale != 1:
            raise NotImplementedError("the down-scale is fused into the Radiance reader; pass scale=1 for .npy input")
        rgb = torch.from_numpy(np.load(im_path, allow_pickle=True).astype("float32")).permute(2, 0, 1).contiguous().to(device)
    else:
        raise Exception("invalid hdr file format: {}".format(ext))      # hdr_image_util.py:52 (.exr / .dng need FreeImage)
    rgb, gray = frame_util.hdr_log_gray(rgb, f_factor)
    return rgb, gray, f_factor


@torch.no_grad()
def run_model_on_single_image2(G_net, im_path, device, im_name, output_path, model_params, f_factor_path, final_shape_addition,
                               scale=4):
    """model_save_util.py:293-404 with the reference's argument list: read `im_path`, tone-map it tile by tile and return
    (colour (3,h,w) fp32, 8-bit (h,w,3) uint8); with `output_path` the 8-bit image is also written as
    <output_path>/<im_name>.png when PIL is importable (the reference uses imageio)."""
    name = os.path.splitext(os.path.basename(im_path))[0]
    if isinstance(f_factor_path, (int, float)):
        lam = float(f_factor_path)
    else:
        lam = float(np.load(f_factor_path, allow_pickle=True)[()][name])
    f_factor = lam * 255 * model_params["factor_coeff"]
    ext = os.path.splitext(im_path)[1]
    if ext != ".hdr":
        raise Exception("invalid hdr file format: {}".format(ext))
    rgb = hdr_io.read_hdr(im_path, device=device, scale=scale)
    color, u8 = run_model_on_frame(G_net, rgb, f_factor, model_params, final_shape_addition)
    if output_path:
        from PIL import Image
        os.makedirs(output_path, exist_ok=True)
        Image.fromarray(u8.cpu().numpy()).save(os.path.join(output_path, im_name + ".png"))
    return color, u8
