"""Video evaluation entry of the in-training evaluator, MI355X-native: the tensor path of `Tester.eval_on_video`
(Tester.py:314-391) -- per-frame log compression (load_inference, :229-251), replicate padding to the tile grid, the 5-D
overlap tiler over the whole clip with the recurrent video generator, percentile clamp + stretch, colour restoration, the 8-bit
stretch (tensor_to_numpy / to_0_1_range_outlier, :393-410) and the clip's mean TMQI -- with every stage on the device.

The warp error of the reference needs an optical flow between two frames (cv2 DeepFlow on tone-mapped images of ANOTHER method
read from disk, :379-386).  cv2 does not exist here (parity with DeepFlow unpinned): `frame_util.compute_flow` is a pyramidal
Lucas-Kanade estimator on the device with the same contract (round 5), `flow_images=(frame1, frame0)` hands eval_on_video the pair
the flow is estimated on (the reference reads it from the other method's output directory), `flow=` a ready field, `align_fn` the
caller's own alignment; `warp_errors` evaluates the reference's two formulas (:387-389) on the aligned pair."""
import torch

from . import frame_util
from .tiler import test_big_size_image
from .tmqi import TMQI


def warp_errors(img0_target, img1_aligned, border=32):
    """Tester.py:385-389: both (H,W,3) uint8 (or [0,255] float) images -> (mean squared error, mean relative absolute error) of
    the [0,1]-scaled images without a `border`-pixel frame."""
    a = img1_aligned.float() / 255.0
    b = img0_target.float() / 255.0
    a, b = a[border:-border, border:-border, :], b[border:-border, border:-border, :]
    d = a - b
    return float((d * d).mean()), float((d.abs() / (1e-8 + a + b)).mean())


@torch.no_grad()
def eval_on_video(G_net, rgb_frames, f_factor, final_shape_addition=0, add_frame=False, align_fn=None, flow=None, flow_images=None):
    """rgb_frames: list of (3,H,W) linear-radiance frames of ONE scene on the GPU (read_hdr_image + hdr_im_transform of the
    reference); f_factor: lambda * 255 * factor_coeff of the scene.  Returns (tmqi_scene, ldr_results[, warp_mse, warp_rel]):
    the mean TMQI over the frames, the tone-mapped 8-bit frames (H,W,3) and, when `align_fn(frame1_u8, frame0_u8)` (the
    caller's optical-flow alignment of frame 1 onto frame 0) is given, the reference's two warp errors.  `flow` (H,W,2 fp32 on the
    GPU: the inverse flow the reference gets from cv2 DeepFlow, Tester.py:379-384) instead of `align_fn` aligns on the device with
    frame_util.warp_flow (= align_frames / warp_flow, GanTrainer.py:584-595, 652-666): no host round trip.
    `flow_images=(frame1, frame0)`: the flow is ESTIMATED here by frame_util.compute_flow -- pyramidal Lucas-Kanade, not the
    reference's cv2 DeepFlow: warp_mse / warp_rel obtained this way are LK-based and NOT numerically comparable with warp errors
    published for the reference (pass the reference's own field as `flow=`, or its alignment as `align_fn`, for that)."""
    if len(rgb_frames) == 0:
        raise ValueError("eval_on_video needs at least one frame")
    originals, padded, grays = [], [], []
    diffY = diffX = 0
    for rgb in rgb_frames:
        if not rgb.is_cuda:
            raise TypeError("eval_on_video expects frames on the GPU")
        originals.append(rgb.float())
        rgb_s, gray = frame_util.hdr_log_gray(rgb, f_factor)            # shift for exr, luminance, log10, normalise
        rgb_p, diffY, diffX = frame_util.resize_im(rgb_s, add_frame, final_shape_addition)
        gray_p, diffY, diffX = frame_util.resize_im(gray, add_frame, final_shape_addition)
        padded.append(rgb_p)
        grays.append(gray_p.unsqueeze(0).unsqueeze(0))                   # (1,1,1,H1,W1)
    clip = torch.cat(grays, 1)                                            # (1,T,1,H1,W1)
    fakes = test_big_size_image(input_data=clip, model=G_net, apply_crop=False, diffY=diffY, diffX=diffX)
    tmqi = TMQI()
    total, results = 0.0, []
    for i in range(len(rgb_frames)):
        fake = fakes[:, i]
        lohi = frame_util.percentile(fake, [0.5, 99.5], on_device=True)
        color = frame_util.back_to_color_and_crop(padded[i], fake, lohi, None, diffY, diffX)
        ldr = frame_util.to_uint8_outlier(color, on_device=True)          # (H,W,3) uint8
        results.append(ldr)
        score = tmqi(originals[i].permute(1, 2, 0).contiguous(), ldr.float(), with_maps=False)[0]
        total += score
    tmqi_scene = total / len(rgb_frames)
    if (align_fn is None and flow is None and flow_images is None) or len(results) < 2:
        return tmqi_scene, results
    if flow is None and flow_images is not None:
        # Tester.py:379-384: flow = compute_flow(img1, img0) on the OTHER method's frames 1 and 0 of the scene
        flow = frame_util.compute_flow(flow_images[0], flow_images[1])
    aligned = frame_util.warp_flow(results[1], flow) if flow is not None else align_fn(results[1], results[0])
    mse, rel = warp_errors(results[0], aligned)
    return tmqi_scene, results, mse, rel
