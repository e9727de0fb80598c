"""Radiance .hdr (RGBE) input: oracle restatement against hand-written byte vectors, the library's host
scanline decoder against the oracle, and -- on the GPU -- the device conversion / down-scale and the file-to-8-bit entry."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import hdr_io as OH

HEAD = b"#?RADIANCE\n# comment\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n"


def synthetic(h, w, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((h, w, 3)) ** 6 * 3000).astype(np.float32)
    x[h // 3, w // 5: w - 3] = x[h // 3, w // 5]          # long runs
    x[0, :3] = 0.0                                         # exponent 0
    x[-1] = 1e-3
    return x


def test_oracle_known_byte_vectors():
    # one run-length scanline of width 8: R = 8 x 0x80, G = literals 1..8, B = run of 5 x 0x10 + 3 literals, E = 8 x 0x81
    scan = bytes([2, 2, 0, 8]) + bytes([128 + 8, 0x80]) + bytes([8, 1, 2, 3, 4, 5, 6, 7, 8]) + \
        bytes([128 + 5, 0x10, 3, 9, 9, 7]) + bytes([128 + 8, 0x81])
    buf = HEAD + b"-Y 1 +X 8\n" + scan
    img = OH.read_hdr(buf)
    assert img.shape == (1, 8, 3)
    # E = 0x81 = 129: factor 2^(129-136) = 1/128
    np.testing.assert_array_equal(img[0, :, 0], np.full(8, 128 / 128, np.float32))
    np.testing.assert_array_equal(img[0, :, 1], np.arange(1, 9, dtype=np.float32) / 128)
    np.testing.assert_array_equal(img[0, :, 2], np.array([16] * 5 + [9, 9, 7], np.float32) / 128)
    # flat pixels (width < 8 can never be run-length coded), exponent 0 -> black, exponent 136 -> mantissa itself
    flat = HEAD + b"-Y 1 +X 2\n" + bytes([200, 100, 50, 0, 200, 100, 50, 136])
    np.testing.assert_array_equal(OH.read_hdr(flat), np.array([[[0, 0, 0], [200, 100, 50]]], np.float32))
    with pytest.raises(ValueError):
        OH.parse_header(b"P6\n")
    with pytest.raises(ValueError):
        OH.parse_header(HEAD + b"+Y 1 +X 8\n")


def test_oracle_round_trip_and_downscale():
    x = synthetic(13, 70, 1)
    rg = OH.float_to_rgbe(x)
    for rle in (True, False):
        back = OH.read_hdr(OH.write_hdr(rg, rle))
        np.testing.assert_array_equal(back, OH.rgbe_to_float(rg))
        assert np.abs(back - x).max() <= x.max() / 128            # 8-bit mantissa shared by the three channels
    y = OH.downscale_linear(np.arange(8 * 12 * 1, dtype=np.float32).reshape(8, 12, 1), 4)
    np.testing.assert_array_equal(y[..., 0], np.array([[19.5, 23.5, 27.5], [67.5, 71.5, 75.5]], np.float32))


def _cv2_axis_by_hand(n_src, n_dst):
    """cv2's INTER_LINEAR taps along one axis in exact rational arithmetic (independent of oracle/hdr_io.py's float code):
    source position (d + 1/2) * n_src / n_dst - 1/2 -> (left index, right index, weight of the right tap)."""
    from fractions import Fraction
    taps = []
    for d in range(n_dst):
        f = (Fraction(2 * d + 1, 2)) * Fraction(n_src, n_dst) - Fraction(1, 2)
        s = f.numerator // f.denominator
        w = f - s
        if s < 0:
            s, w = 0, Fraction(0)
        if s >= n_src - 1:
            s, w = n_src - 1, Fraction(0)
        taps.append((s, min(s + 1, n_src - 1), w))
    return taps


def test_oracle_downscale_not_divisible_sizes_belgium():
    """The reference's own sample, belgium.hdr, is 769 x 1025: cv2.resize(img, (1025 // 4, 769 // 4)) samples at
    (d + 0.5) * 1025 / 256 - 0.5, not at 4 d + 1.5 (utils/model_save_util.py:225-226).  Expected values of the last column / row
    and of a mid-image pixel are computed here by hand from the rational tap positions."""
    H, W, s = 769, 1025, 4
    xt, yt = _cv2_axis_by_hand(W, W // s), _cv2_axis_by_hand(H, H // s)
    # the drift: last column sits at 1022.498..., i.e. between source 1022 and 1023 (the integer-scale rule said 1021 / 1022)
    assert xt[-1][:2] == (1022, 1023) and abs(float(xt[-1][2]) - 0.498046875) < 1e-12
    assert yt[-1][:2] == (766, 767) and abs(float(yt[-1][2]) - (191.5 * 769 / 192 - 0.5 - 766)) < 1e-9
    assert xt[0][:2] == (1, 2) and abs(float(xt[0][2]) - 0.501953125) < 1e-12      # 0.5 * 1025 / 256 - 0.5 = 1.50195...
    img = synthetic(H, W, 11)[..., :1]
    y = OH.downscale_linear(img, s)
    assert y.shape == (192, 256, 1)
    for oy, ox in [(0, 0), (191, 255), (191, 0), (0, 255), (100, 128), (57, 200)]:
        (x0, x1, fx), (y0, y1, fy) = xt[ox], yt[oy]
        fx, fy = float(fx), float(fy)
        top = float(img[y0, x0, 0]) * (1 - fx) + float(img[y0, x1, 0]) * fx
        bot = float(img[y1, x0, 0]) * (1 - fx) + float(img[y1, x1, 0]) * fx
        want = top * (1 - fy) + bot * fy
        # cv2 rounds the source coordinate to float32 (half an ulp at 766 is 3e-5 of a pixel): the taps agree to that, and a
        # sample point off by a whole pixel (the integer-scale rule) would be an O(1) error on this noise image
        big = max(abs(float(img[yy, xx, 0])) for yy in (y0, y1) for xx in (x0, x1))
        assert abs(float(y[oy, ox, 0]) - want) <= 1e-4 * big + 1e-12, (oy, ox)


def test_host_decoder_equals_oracle_and_rejects_bad_streams():
    from uncltmo_amd import hdr_io
    for h, w, seed in [(13, 70, 2), (5, 7, 3), (3, 300, 4)]:
        rg = OH.float_to_rgbe(synthetic(h, w, seed))
        for rle in (True, False):
            buf = OH.write_hdr(rg, rle)
            assert hdr_io.parse_header(buf) == OH.parse_header(buf)
            np.testing.assert_array_equal(hdr_io.decode_rgbe(buf), rg)
    good = OH.write_hdr(OH.float_to_rgbe(synthetic(9, 40, 5)), True)
    with pytest.raises(ValueError):
        hdr_io.decode_rgbe(good[:-7])                              # truncated
    H, W, off = OH.parse_header(good)
    bad = bytearray(good)
    bad[off + 3] = 41                                              # scanline width disagrees with the header
    with pytest.raises(ValueError):
        hdr_io.decode_rgbe(bytes(bad))
    with pytest.raises(ValueError):
        hdr_io.parse_header(good.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"))
    sample = "/root/reference/activate_trained_model/input_images/belgium.hdr"
    if os.path.exists(sample):                                     # the upstream sample image, when the tree is mounted
        buf = open(sample, "rb").read()
        H, W, off = OH.parse_header(buf)
        np.testing.assert_array_equal(hdr_io.decode_rgbe(buf), OH.decode_rgbe_bytes(buf, off, H, W))


@pytest.mark.gpu
@pytest.mark.parametrize("scale,h,w", [(1, 37, 91), (2, 37, 91), (4, 37, 91), (4, 769, 1025), (3, 50, 64), (4, 64, 128)])
def test_device_conversion_and_downscale(scale, h, w):
    from uncltmo_amd import hdr_io
    x = synthetic(h, w, 6)
    rg = OH.float_to_rgbe(x)
    buf = OH.write_hdr(rg, True)
    got = hdr_io.read_hdr(buf, scale=scale).cpu().numpy()
    want = OH.rgbe_to_float(rg)
    if scale > 1:
        want = OH.downscale_linear(want, scale)
    np.testing.assert_array_equal(got, want.transpose(2, 0, 1))
    if (h, w, scale) == (769, 1025, 4):
        # belgium.hdr's size: last column / row against the hand-computed taps, so that oracle and kernel cannot share a mistake
        xt, yt = _cv2_axis_by_hand(w, w // scale), _cv2_axis_by_hand(h, h // scale)
        src = OH.rgbe_to_float(rg).astype(np.float64)
        for oy, ox in [(191, 255), (0, 255), (191, 0), (96, 255)]:
            (x0, x1, fx), (y0, y1, fy) = xt[ox], yt[oy]
            fx, fy = float(fx), float(fy)
            top = src[y0, x0] * (1 - fx) + src[y0, x1] * fx
            bot = src[y1, x0] * (1 - fx) + src[y1, x1] * fx
            # float32 source coordinates move a tap weight by up to 3e-5: bound relative to the largest tap, not to the result
            tol = 1e-4 * np.abs(np.stack([src[y0, x0], src[y0, x1], src[y1, x0], src[y1, x1]])).max(0) + 1e-12
            assert (np.abs(got[:, oy, ox] - (top * (1 - fy) + bot * fy)) <= tol).all(), (oy, ox)
    with pytest.raises(ValueError):
        hdr_io.read_hdr(buf, scale=0)


@pytest.mark.gpu
def test_hdr_file_to_8bit_image_end_to_end():
    """run_model_on_single_image2 (model_save_util.py:293-404) from a Radiance file: same result as the frame entry on the
    oracle-decoded, oracle-down-scaled image, and the colour output follows the oracle's whole pipeline."""
    from oracle import inference as OI
    from oracle import generator as OG
    from oracle import tiler as OT
    from uncltmo_amd import inference, synth
    from uncltmo_amd.generator import UNet
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0,
               compute_dtype="fp32")
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    frame = synth.hdr_frames(1, 1088, 1216, salt="hdrfile")[0, 0].numpy()           # log-compressed synthetic luminance
    lin = (10.0 ** (frame * 3.0) - 1.0)[..., None] * np.array([0.9, 1.0, 0.7], np.float32)
    rg = OH.float_to_rgbe(lin.astype(np.float32))
    params = {"factor_coeff": 0.1, "add_frame": 1}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synthetic.hdr")
        open(path, "wb").write(OH.write_hdr(rg, True))
        color, u8 = inference.run_model_on_single_image2(net, path, "cuda", "synthetic", d, params, 25.0, 0, scale=4)
        assert os.path.exists(os.path.join(d, "synthetic.png"))
        rgb2, gray2, f2 = inference.load_inference2(path, 25.0, 0.1, "cuda", scale=4)
    small = OH.downscale_linear(OH.rgbe_to_float(rg), 4).transpose(2, 0, 1)
    assert color.shape == (3, 272, 304) and u8.shape == (272, 304, 3) and u8.dtype == torch.uint8
    c2, u2 = inference.run_model_on_frame(net, torch.from_numpy(small).cuda(), 25.0 * 255 * 0.1, params, 0)
    assert torch.equal(color, c2) and torch.equal(u8, u2)
    rgb_o, gray_o = OI.hdr_log_gray(torch.from_numpy(small), 25.0 * 255 * 0.1)
    assert f2 == 25.0 * 255 * 0.1
    np.testing.assert_allclose(gray2.cpu().numpy(), gray_o.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_array_equal(rgb2.cpu().numpy(), rgb_o.numpy())
