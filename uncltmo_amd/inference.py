"""Single-frame inference entry, MI355X-native: the tensor path of the reference's `run_model_on_single_image2`
(utils/model_save_util.py:293-404) between its file read and its file write -- log compression, replicate padding to the
tile grid, overlap-tiled generator (uncltmo_amd.tiler), percentile clamp + stretch, colour restoration, 8-bit stretch -- with
every stage on the device.  Reading the .hdr/.exr file, the optional cv2.resize and writing the .png stay with the caller."""
import torch

from . import frame_util
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
    min_p, max_p = frame_util.percentile(fake, [0.5, 99.5])
    color = frame_util.back_to_color_and_crop(rgb_img, fake, min_p, max_p, diffY, diffX)
    return color, frame_util.to_uint8_outlier(color)
