"""warp_flow (GanTrainer.py:584-595) = cv2.remap(img, flow + grid, None, INTER_LINEAR) on 8-bit images.  cv2 is absent from the
reference tree and this image, so the oracle restates OpenCV's fixed-point bilinear remap and is pinned HERE by vectors computed
by hand from that algorithm (1/32-pixel coordinates, weights of 2^15, +2^14 >> 15, constant-0 border); the HIP kernel must equal the
oracle bit for bit."""
import numpy as np
import pytest
import torch

from oracle.tester import warp_flow as o_warp


def _img():
    # 3 x 4, one channel: rows [10 20 30 40], [50 60 70 80], [90 100 110 120]
    return (np.arange(12, dtype=np.uint8).reshape(3, 4, 1) + 1) * 10


def test_oracle_hand_vectors():
    img = _img()
    flow = np.zeros((3, 4, 2), np.float32)
    assert np.array_equal(o_warp(img, flow), img)                           # identity
    f = flow.copy(); f[..., 0] = 0.5                                        # half a pixel to the right
    # (a + b) / 2 exactly, last column: (40*16*32*32 + 0) -> 20; rounding (x*32768/2+16384)>>15 = x/2 (+0.5 floor)
    want = np.array([[15, 25, 35, 20], [55, 65, 75, 40], [95, 105, 115, 60]], np.uint8).reshape(3, 4, 1)
    assert np.array_equal(o_warp(img, f), want)
    f = flow.copy(); f[..., 1] = 0.25                                       # a quarter pixel down: (3a + b) / 4
    want = np.array([[20, 30, 40, 50], [60, 70, 80, 90], [68, 75, 83, 90]], np.uint8).reshape(3, 4, 1)
    # last row: (3 * 90 + 0) / 4 = 67.5 -> (90*24576 + 16384) >> 15 = 68; 100 -> 75; 110 -> 82.5 -> 83; 120 -> 90
    assert np.array_equal(o_warp(img, f), want)
    f = flow.copy(); f[..., 0] = 1.0 / 64.0                                 # cvRound(x + 1/64)*32 = 32x + 0.5 -> half to even
    got = o_warp(img, f)[0, :, 0]
    # x = 0: 0.5 -> 0 (even): 10; x = 1: 32.5 -> 32: 20; x = 2: 64.5 -> 64: 30; x = 3: 96.5 -> 96: 40
    assert got.tolist() == [10, 20, 30, 40]
    f = flow.copy(); f[..., 0] = 3.0 / 64.0                                 # 1.5 -> 2 (even): weights 30 / 2 of 32
    got = o_warp(img, f)[0, :, 0]
    # (10*30 + 20*2)/32 = 10.625 -> (10*30720 + 20*2048 + 16384) >> 15 = 11; 20.625 -> 21; 30.625 -> 31; (40*30)/32 = 37.5 -> 38
    assert got.tolist() == [11, 21, 31, 38]
    f = flow.copy(); f[..., 0] = -1.5; f[..., 1] = -0.5                     # up-left, partly outside
    got = o_warp(img, f)[:, :, 0]
    # out(0,0): taps (-1,-2),(-1,-1),(0,-2),(0,-1) all outside -> 0; out(0,1): x=-0.5: taps x=-1 (out), 0: y=-0.5: rows -1 (out), 0:
    # 10 * 16*16*32 / 32768 = 2.5 -> 3;  out(1,2): x = 0.5, y = 0.5: (10+20+50+60)/4 = 35
    assert got[0, 0] == 0 and got[0, 1] == 3 and got[1, 2] == 35
    f = flow.copy(); f[..., 0] = 1e9; f[0, 0, 1] = np.nan                   # saturating coordinates: everything outside
    assert (o_warp(img, f)[:, 1:, :] == 0).all()


def test_oracle_flow_is_not_modified_and_other_sizes():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    flow = (rng.standard_normal((11, 29, 2)) * 3).astype(np.float32)
    keep = flow.copy()
    out = o_warp(img, flow)
    assert out.shape == (11, 29, 3) and np.array_equal(flow, keep)


@pytest.mark.gpu
def test_hip_warp_flow_equals_oracle():
    from uncltmo_amd import frame_util
    rng = np.random.default_rng(5)
    for (h, w, c, hf, wf) in ((3, 4, 1, 3, 4), (64, 96, 3, 64, 96), (37, 53, 3, 41, 29), (128, 128, 4, 128, 128)):
        img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        flow = (rng.standard_normal((hf, wf, 2)) * 4).astype(np.float32)
        flow[::7, ::5] = np.round(flow[::7, ::5] * 64) / 64                 # exact ties of the 1/32 rounding
        flow[0, 0] = (np.nan, 1e30)
        flow[-1, -1] = (-1e30, np.inf)
        want = o_warp(img, flow)
        got = frame_util.warp_flow(torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda()).cpu().numpy()
        assert np.array_equal(got, want), (h, w, c, np.abs(got.astype(int) - want.astype(int)).max())
    for f in (np.zeros((3, 4, 2), np.float32),):
        got = frame_util.warp_flow(torch.from_numpy(_img()).cuda(), torch.from_numpy(f).cuda()).cpu().numpy()
        assert np.array_equal(got, _img())
