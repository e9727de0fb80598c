"""ORACLE (test infrastructure, CPU only -- never imported by the product path).

Radiance .hdr (RGBE) reading as the reference gets it from `imageio.imread(path, format="HDR-FI")`
(utils/hdr_image_util.py:35-39).  imageio / FreeImage are third-party dependencies that are NOT in the reference tree and
not installed here (README.md lists no pinned version), so this is a restatement of the published format -- Radiance
`color.c` scanline coding and Bruce Walter's rgbe.c conversion, which FreeImage's PluginHDR follows: value = mantissa *
2^(E-136), zero when E == 0 -- and **parity with FreeImage is unpinned**: what is pinned is decode(encode(x)) on synthetic
images, the hand-written byte vectors in tests/test_hdr_io.py, and the self-consistency of the product decoder with this
one on the upstream sample image when it is present.  The writer exists for the tests only.

`downscale_linear` restates cv2.resize(img, (W//s, H//s)) (INTER_LINEAR, no anti-aliasing) of load_inference2
(utils/model_save_util.py:225-226) for even integer s; cv2 is absent too, same caveat."""
import numpy as np


def parse_header(buf):
    """-> (H, W, offset of the first scanline byte).  Accepts the '#?RADIANCE' / '#?RGBE' magic, any header lines up to
    the empty one, FORMAT=32-bit_rle_rgbe and the standard orientation '-Y H +X W'."""
    if not (buf.startswith(b"#?RADIANCE") or buf.startswith(b"#?RGBE")):
        raise ValueError("not a Radiance picture (magic)")
    pos = 0
    fmt_ok = False
    while True:
        end = buf.index(b"\n", pos)
        line = buf[pos:end]
        pos = end + 1
        if line == b"":
            break
        if line.startswith(b"FORMAT="):
            fmt_ok = line.strip() == b"FORMAT=32-bit_rle_rgbe"
            if not fmt_ok:
                raise ValueError("unsupported FORMAT: %r" % line)
    end = buf.index(b"\n", pos)
    parts = buf[pos:end].split()
    if len(parts) != 4 or parts[0] != b"-Y" or parts[2] != b"+X":
        raise ValueError("unsupported resolution line %r" % buf[pos:end])
    return int(parts[1]), int(parts[3]), end + 1


def decode_rgbe_bytes(buf, off, H, W):
    """scanlines -> (H, W, 4) uint8"""
    out = np.zeros((H, W, 4), np.uint8)
    p = off
    for y in range(H):
        if 8 <= W <= 0x7fff and buf[p] == 2 and buf[p + 1] == 2 and not (buf[p + 2] & 0x80):
            assert (buf[p + 2] << 8 | buf[p + 3]) == W
            p += 4
            for c in range(4):
                x = 0
                while x < W:
                    cnt = buf[p]
                    p += 1
                    if cnt > 128:
                        cnt -= 128
                        out[y, x:x + cnt, c] = buf[p]
                        p += 1
                    else:
                        out[y, x:x + cnt, c] = np.frombuffer(buf, np.uint8, cnt, p)
                        p += cnt
                    x += cnt
        else:
            rest = np.frombuffer(buf, np.uint8, (H - y) * W * 4, p).reshape(H - y, W, 4)
            out[y:] = rest
            break
    return out


def rgbe_to_float(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e == 0, np.float32(0), np.ldexp(np.float32(1.0), e - 136).astype(np.float32))
    return (rgbe[..., :3].astype(np.float32) * f[..., None]).astype(np.float32)


def read_hdr(buf):
    """file bytes -> (H, W, 3) float32"""
    H, W, off = parse_header(buf)
    return rgbe_to_float(decode_rgbe_bytes(buf, off, H, W))


def float_to_rgbe(img):
    """rgbe.c float2rgbe: (H, W, 3) float -> (H, W, 4) uint8"""
    v = img.max(-1)
    m, e = np.frexp(v)
    scale = np.where(v < 1e-32, 0.0, m * 256.0 / np.where(v < 1e-32, 1.0, v))
    out = np.zeros(img.shape[:2] + (4,), np.uint8)
    out[..., :3] = (img * scale[..., None]).astype(np.int64).clip(0, 255).astype(np.uint8)
    out[..., 3] = np.where(v < 1e-32, 0, e + 128).astype(np.uint8)
    return out


def write_hdr(rgbe, rle=True, extra_header=b"EXPOSURE=          1.0000000000000\n"):
    """(H, W, 4) uint8 -> file bytes (tests only).  rle: new-style run-length scanlines, else flat pixels."""
    H, W, _ = rgbe.shape
    head = b"#?RADIANCE\n# oracle test writer\nFORMAT=32-bit_rle_rgbe\n" + extra_header + b"\n" + ("-Y %d +X %d\n" % (H, W)).encode()
    if not rle or W < 8 or W > 0x7fff:      # such widths are never run-length coded (Radiance color.c fwritecolrs)
        return head + rgbe.tobytes()
    body = bytearray()
    for y in range(H):
        body += bytes([2, 2, W >> 8, W & 255])
        for c in range(4):
            row = rgbe[y, :, c]
            x = 0
            while x < W:
                run = 1
                while x + run < W and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 4:
                    body += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    lit = 0
                    while x + lit < W and lit < 128:
                        r = 1
                        while x + lit + r < W and r < 4 and row[x + lit + r] == row[x + lit]:
                            r += 1
                        if r >= 4:
                            break
                        lit += 1
                    lit = max(lit, 1)
                    body += bytes([lit]) + row[x:x + lit].tobytes()
                    x += lit
    return head + bytes(body)


def _lin_coord(n_dst, n_src):
    """cv2 INTER_LINEAR sample positions along one axis (imgproc/resize.cpp): fx = float32((d + 0.5) * scale - 0.5) with
    scale = 1 / (n_dst / n_src) in double, s = floor(fx), weight of the right / lower neighbour fx - s, clamped to the ends."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    w = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= n_src - 1
    s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    w = np.where(lo | hi, np.float32(0), w).astype(np.float32)
    return s, np.minimum(s + 1, n_src - 1), w


def downscale_linear(img, s):
    """cv2.resize(img, (W//s, H//s)) with the default INTER_LINEAR (utils/model_save_util.py:225-226): (H, W, C) float32 ->
    (H//s, W//s, C).  The ratio per axis is W / (W // s), not s: for sizes that are not multiples of s (the reference's own
    sample belgium.hdr is 769 x 1025) the sample point drifts by up to a source pixel across the image."""
    H, W = img.shape[:2]
    Ho, Wo = H // s, W // s
    y0, y1, fy = _lin_coord(Ho, H)
    x0, x1, fx = _lin_coord(Wo, W)
    img = img.astype(np.float32)
    ax, ay = (np.float32(1) - fx)[None, :, None], (np.float32(1) - fy)[:, None, None]
    fxb, fyb = fx[None, :, None], fy[:, None, None]
    rows = img[:, x0] * ax + img[:, x1] * fxb           # horizontal pass on every source row (float32 products and sum)
    return (rows[y0] * ay + rows[y1] * fyb).astype(np.float32)
