"""Overlap-tile inference for frames larger than the generator's 256x256 window.

Drop-in for `utils/model_save_util.py:test_big_size_image2` (4-D, :409-486) and `test_big_size_image` (5-D,
:488-565): 256^2 patches, stride 192, linear cross-fade along x then y, edge-aligned final tile per axis.
Instead of one batch-1 model call per tile and 64 tiny blend kernels per seam, all tiles of all frames are
gathered by one kernel, run through the generator as ONE batch, and cross-faded by one kernel.
"""
import torch

from . import _hip


def _check_geometry(patch_h, patch_w, patch_h_overlap, patch_w_overlap):
    if (patch_h, patch_w, patch_h_overlap, patch_w_overlap) != (256, 256, 64, 64):
        raise NotImplementedError("the HIP tiler is built for 256x256 patches with overlap 64 (the published "
                                  "setting, model_save_util.py:305)")


def tile_count(H, W):
    n = _hip.lib().uncl_tile_count(int(H), int(W))
    if n < 0:
        # the reference's loops leave w_end / h_end undefined for H or W <= 256 and crash (model_save_util.py:417-441)
        raise ValueError("tiler needs H > 256 and W > 256 (got %dx%d)" % (H, W))
    return n


def gather_tiles(frames):
    """frames (F,H,W) fp32 cuda -> tiles (F*T, 1, 256, 256)."""
    F, H, W = frames.shape
    T = tile_count(H, W)
    tiles = torch.empty(F * T, 1, 256, 256, dtype=torch.float32, device=frames.device)
    _hip.check(_hip.lib().uncl_tile_gather(_hip.ptr(frames), _hip.ptr(tiles), F, H, W, _hip.stream_ptr()),
               "uncl_tile_gather")
    return tiles


def blend_tiles(tiles, F, H, W):
    """tiles (F*T,1,256,256) fp32 cuda -> frames (F,H,W)."""
    out = torch.empty(F, H, W, dtype=torch.float32, device=tiles.device)
    _hip.check(_hip.lib().uncl_tile_blend(_hip.ptr(tiles), _hip.ptr(out), F, H, W, _hip.stream_ptr()),
               "uncl_tile_blend")
    return out


def _run_model(model, tiles, apply_crop, diffY, diffX):
    with torch.no_grad():
        if hasattr(model, "infer"):
            return model.infer(tiles)
        out, _ = model(tiles, apply_crop=apply_crop, diffY=diffY, diffX=diffX)
        return out


def test_big_size_image2(input_data, model, apply_crop, diffY, diffX, patch_h=256, patch_w=256,
                         patch_h_overlap=64, patch_w_overlap=64):
    """input_data (N,1,H,W) -> (N,1,H,W).  Reference: utils/model_save_util.py:409-486."""
    _check_geometry(patch_h, patch_w, patch_h_overlap, patch_w_overlap)
    N, Cc, H, W = input_data.shape
    if Cc != 1:
        raise ValueError("tiler expects single-channel frames")
    frames = input_data.reshape(N, H, W).float().contiguous()
    out = None
    if hasattr(model, "infer_frames"):
        tile_count(H, W)                 # (the reference's own failure for H or W <= 256)
        with torch.no_grad():
            out = model.infer_frames(frames)            # tiles read in place by the first layer's loader (16-bit inference)
    if out is None:
        out = _run_model(model, gather_tiles(frames), apply_crop, diffY, diffX).float().contiguous()
    return blend_tiles(out, N, H, W).reshape(N, 1, H, W)


def test_big_size_image(input_data, model, apply_crop, diffY, diffX, patch_h=256, patch_w=256,
                        patch_h_overlap=64, patch_w_overlap=64):
    """input_data (B,T,1,H,W) -> (B,T,1,H,W).  Reference: utils/model_save_util.py:488-565.  Every tile
    position is a clip of T frames that the (recurrent) video generator consumes in order."""
    _check_geometry(patch_h, patch_w, patch_h_overlap, patch_w_overlap)
    B, T, Cc, H, W = input_data.shape
    frames = input_data.reshape(B * T, H, W).float().contiguous()
    tiles = gather_tiles(frames)                                   # (B*T*nt, 1, 256, 256), frame-major
    nt = tiles.shape[0] // (B * T)
    clips = tiles.reshape(B, T, nt, 1, 256, 256).permute(0, 2, 1, 3, 4, 5).reshape(B * nt, T, 1, 256, 256).contiguous()
    with torch.no_grad():
        out, _ = model(clips, apply_crop=apply_crop, diffY=diffY, diffX=diffX)
    out = out.reshape(B, nt, T, 1, 256, 256).permute(0, 2, 1, 3, 4, 5).reshape(B * T * nt, 1, 256, 256).float().contiguous()
    return blend_tiles(out, B * T, H, W).reshape(B, T, 1, H, W)


class TiledGraph:
    """The whole tiled forward (gather -> generator -> cross-fade) of a FIXED frame batch shape as one hipGraph.

    For streams of single frames (25 tiles at 1024^2) the ~35 launches of a forward are latency-, not throughput-bound;
    replaying them as a graph removes the per-launch host cost and the gaps between dependent launches.  The graph owns
    its input and output buffers: `graph(frames)` copies the frames in, replays, and returns the (re-used) output tensor --
    clone it if it has to outlive the next call.  Weights are read through the pointers captured at construction: rebuild
    the object after changing the model's parameters."""

    def __init__(self, model, n_frames, H, W):
        if not hasattr(model, "infer"):
            raise TypeError("TiledGraph needs the HIP image generator (uncltmo_amd.generator.UNet)")
        dev = next(model.parameters()).device
        self.model = model
        self.static_in = torch.zeros(n_frames, 1, H, W, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up: kernel attributes, packed weights and workspaces exist before capture
            for _ in range(2):
                test_big_size_image2(self.static_in, model, 0, 0, 0)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._keep = model._packed_weights()    # the captured launches read these buffers
        self._pack_key = model._pack_key        # ... of THIS parameter version
        self.graph = torch.cuda.CUDAGraph()
        from .step_graph import capture_error_mode      # a process group's watchdog thread must not trip the capture
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode()):
            self.static_out = test_big_size_image2(self.static_in, model, 0, 0, 0)
        # the captured launches also hold raw pointers into the generator's workspace: keep those tensors alive even if a later
        # call with a larger batch makes the model replace its cached workspace
        self._ws_keep = [t for k, t in model._ws.items() if isinstance(k, tuple) and k[-1] == dev]

    def __call__(self, frames):
        if frames.shape != self.static_in.shape:
            raise ValueError("TiledGraph was captured for frames of shape %s, got %s" % (tuple(self.static_in.shape), tuple(frames.shape)))
        self.model._packed_weights()
        if self.model._pack_key != self._pack_key:
            raise RuntimeError("TiledGraph: the model's parameters changed after capture (optimizer step / load_state_dict); "
                               "the captured launches still read the old packed weights -- build a new TiledGraph")
        self.static_in.copy_(frames)
        self.graph.replay()
        return self.static_out
