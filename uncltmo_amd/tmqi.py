"""TMQI (tone-mapped image quality index), MI355X-native: same call shape and return order as the reference's
`TMQI()(hdrImage, ldrImage)` (TMQI.py:92-146) -- (Q, S, N, s_local, s_maps) with s_local the five per-level structural fidelity
means and s_maps the five per-level maps they are the means of (fp64 device tensors, (H_l - 10, W_l - 10)) -- computed in fp64 on
the device by csrc/tmqi.hip.  Inputs are device tensors: (H,W) luminance or (H,W,3) RGB (converted with TMQI.py:46-49's weights)."""
import ctypes as C

import torch

from . import _hip


class TMQI:
    name = "TMQI"

    def __call__(self, hdrImage, ldrImage, ldr_scale=1.0, with_maps=True):
        """hdrImage: any range; ldrImage: [0,255] (or pass ldr_scale=255 for a [0,1] image).  with_maps=False leaves the fifth
        element of the result None (the metric loops that only read Q / S / N, Tester.py:373, do not pay for 2.7 H W doubles)."""
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
        maps, ptrs = None, None
        if with_maps and h >= 176 and wd >= 176:
            maps = [torch.empty((h >> l) - 10, (wd >> l) - 10, dtype=torch.float64, device=hdr.device) for l in range(5)]
            ptrs = (C.c_void_p * 5)(*[m.data_ptr() for m in maps])
        _hip.check(lib.uncl_tmqi_maps(hdr.data_ptr(), ldr.data_ptr(), h, wd, float(ldr_scale), out.data_ptr(), ptrs, ws.data_ptr(),
                                      _hip.stream_ptr()), "uncl_tmqi_maps")
        o = out.cpu()
        return float(o[0]), float(o[1]), float(o[2]), [float(v) for v in o[3:8]], maps
