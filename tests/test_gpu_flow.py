"""csrc/flow.hip (the evaluator's optical flow, GanTrainer.py:597-646 / Tester.py:379-389) against oracle/flow.py, which
tests/test_flow.py pins with synthetic motions of known flow; and the same synthetic motions directly on the device."""
import numpy as np
import pytest
import torch

from oracle import flow as OF
from oracle.tester import warp_errors as o_warp_errors
from test_flow import epe, sample, texture
from uncltmo_amd import frame_util, tester

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("h,w,tx,ty", [(96, 128, 2.5, 1.25), (70, 53, -1.0, 0.5), (200, 333, 7.5, -6.0), (33, 40, 0.4, -0.3)])
def test_device_flow_follows_the_oracle(h, w, tx, ty):
    A = np.floor(texture(h, w, 10 + h))
    S = np.floor(sample(texture(h, w, 10 + h), tx, ty))
    want = OF.compute_flow(A, S)
    got = frame_util.compute_flow(dev(A.astype(np.float32)), dev(S.astype(np.float32))).cpu().numpy()
    assert got.shape == want.shape
    # fp32 kernels against the fp64 oracle: the field itself is a few pixels, agreement to a few thousandths of a pixel
    assert np.abs(got - want).max() < 2e-2 and np.abs(got - want).mean() < 2e-3, (np.abs(got - want).max(), np.abs(got - want).mean())


def test_device_flow_recovers_known_motions_and_aligns():
    A = texture(160, 224, 21)
    for tx, ty in ((0.0, 0.0), (3.25, -2.5), (-6.0, 4.0)):
        S = sample(A, tx, ty)
        a8 = np.clip(np.rint(A), 0, 255).astype(np.uint8)[..., None].repeat(3, -1)
        s8 = np.clip(np.rint(S), 0, 255).astype(np.uint8)[..., None].repeat(3, -1)
        f = frame_util.compute_flow(dev(a8), dev(s8))
        mean, worst = epe(f.cpu().numpy(), tx, ty, 20)
        assert mean < 0.06 and worst < 0.35, (tx, ty, mean, worst)
        aligned = frame_util.align_frames(dev(a8), f)
        mse, rel = tester.warp_errors(dev(s8), aligned)
        mse0, rel0 = tester.warp_errors(dev(s8), dev(a8))
        if tx or ty:
            assert mse < 0.05 * mse0, (mse, mse0)
        o_mse, o_rel = o_warp_errors(s8, aligned.cpu().numpy())
        assert abs(mse - o_mse) < 1e-6 and abs(rel - o_rel) < 1e-5


def test_unit_range_inputs_are_scaled_like_the_reference():
    A = texture(64, 80, 31)
    S = sample(A, 1.5, 0.0)
    f255 = frame_util.compute_flow(dev(np.floor(A).astype(np.float32)), dev(np.floor(S).astype(np.float32)))
    f1 = frame_util.compute_flow(dev((np.floor(A) / 255.0).astype(np.float32)), dev((np.floor(S) / 255.0).astype(np.float32)))
    assert (f255 - f1).abs().max() < 0.15      # x / 255 * 255 re-truncates a few pixels by one level
    with pytest.raises(ValueError):
        frame_util.compute_flow(dev(np.zeros((10, 12), np.float32)), dev(np.zeros((10, 13), np.float32)))
