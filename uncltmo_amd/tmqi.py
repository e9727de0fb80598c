"""TMQI (tone-mapped image quality index), MI355X-native: same call shape and return order as the reference's
`TMQI()(hdrImage, ldrImage)` (TMQI.py:92-146) -- (Q, S, N, s_local) with s_local the five per-level structural fidelity
means -- computed in fp64 on the device by csrc/tmqi.hip.  The per-level maps (`s_maps`) the reference also returns are not
materialised.  Inputs are device tensors: (H,W) luminance or (H,W,3) RGB (converted with TMQI.py:46-49's weights)."""
import torch

from . import _hip


class TMQI:
    name = "TMQI"

    def __call__(self, hdrImage, ldrImage, ldr_scale=1.0):
        """hdrImage: any range; ldrImage: [0,255] (or pass ldr_scale=255 for a [0,1] image)."""
        if not hdrImage.is_cuda or not ldrImage.is_cuda:
            raise _hip.HipError("TMQI needs CUDA(HIP) tensors; there is no CPU path")
        if hdrImage.shape != ldrImage.shape:
            raise AssertionError("images must have same dimensions")          # TMQI.py:94
        if hdrImage.dim() == 3:
            w = torch.tensor([0.2126, 0.7152, 0.0722], dtype=torch.float32, device=hdrImage.device)
            hdrImage, ldrImage = hdrImage.float() @ w, ldrImage.float() @ w
        h, wd = hdrImage.shape
        lib = _hip.lib()
        hdr, ldr = hdrImage.float().contiguous(), ldrImage.float().contiguous()
        out = torch.empty(8, dtype=torch.float64, device=hdr.device)
        ws = torch.empty(lib.uncl_tmqi_workspace_bytes(h, wd), dtype=torch.uint8, device=hdr.device)
        _hip.check(lib.uncl_tmqi(hdr.data_ptr(), ldr.data_ptr(), h, wd, float(ldr_scale), out.data_ptr(), ws.data_ptr(),
                                 _hip.stream_ptr()), "uncl_tmqi")
        o = out.cpu()
        return float(o[0]), float(o[1]), float(o[2]), [float(v) for v in o[3:8]]
