"""Radiance .hdr (RGBE) reader for the inference entry: replaces `hdr_image_util.read_hdr_image` for '.hdr' files
(utils/hdr_image_util.py:35-39, imageio's FreeImage plugin in the reference) and the cv2.resize of `load_inference2`
(utils/model_save_util.py:225-226).  The header is parsed here, the run-length scanlines are decoded by the library's host
function, and the RGBE -> fp32 conversion (with cv2's INTER_LINEAR down-scale to (W//scale, H//scale), any H and W) runs on the MI355X: the image never exists
as host floats."""
import ctypes as C

import numpy as np
import torch

from . import _hip


def parse_header(buf):
    """-> (H, W, offset of the first scanline byte).  '#?RADIANCE' / '#?RGBE' magic, header lines up to the empty one,
    FORMAT=32-bit_rle_rgbe, orientation '-Y H +X W' (the only one FreeImage's reader accepts as well)."""
    if not (buf.startswith(b"#?RADIANCE") or buf.startswith(b"#?RGBE")):
        raise ValueError("not a Radiance picture (missing '#?RADIANCE' magic)")
    pos = 0
    while True:
        end = buf.find(b"\n", pos)
        if end < 0:
            raise ValueError("truncated Radiance header")
        line = buf[pos:end]
        pos = end + 1
        if line == b"":
            break
        if line.startswith(b"FORMAT=") and line.strip() != b"FORMAT=32-bit_rle_rgbe":
            raise ValueError("unsupported Radiance FORMAT: %r" % line)
    end = buf.find(b"\n", pos)
    parts = buf[pos:end].split() if end >= 0 else []
    if len(parts) != 4 or parts[0] != b"-Y" or parts[2] != b"+X":
        raise ValueError("unsupported Radiance resolution line %r" % buf[pos:max(end, pos)])
    return int(parts[1]), int(parts[3]), end + 1


def decode_rgbe(buf):
    """file bytes -> (H, W, 4) uint8 RGBE on the host (uncl_rgbe_decode)."""
    H, W, off = parse_header(buf)
    out = np.empty((H, W, 4), np.uint8)
    data = np.frombuffer(buf, np.uint8, len(buf) - off, off)
    rc = _hip.lib().uncl_rgbe_decode(data.ctypes.data_as(C.c_void_p), data.size, H, W, out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("malformed or truncated Radiance scanline data (uncl_rgbe_decode: %d)" % rc)
    return out


def read_hdr(path_or_bytes, device="cuda", scale=1):
    """-> (3, H//scale, W//scale) fp32 linear radiance on `device` (what `tranforms.hdr_im_transform(read_hdr_image(p))`
    hands to load_inference; with scale=4 what load_inference2 gets after its cv2.resize)."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    rgbe = torch.from_numpy(decode_rgbe(bytes(buf))).to(device)
    H, W = rgbe.shape[0], rgbe.shape[1]
    if scale < 1:
        raise ValueError("scale must be >= 1")
    out = torch.empty(3, H // scale, W // scale, dtype=torch.float32, device=rgbe.device)
    _hip.check(_hip.lib().uncl_rgbe_to_planes(_hip.ptr(rgbe), out.data_ptr(), H, W, scale, _hip.stream_ptr()), "uncl_rgbe_to_planes")
    return out
