"""GPU-side data path (uncltmo_amd/data_loader.py:npy_loader) against the numpy oracle on every branch, and its HDR branch against
a golden captured from the reference's own npy_loader (utils/ProcessedDatasetFolderImg.py:43-160)."""
import os

import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import data_loader as OD
from uncltmo_amd import data_loader, synth

pytestmark = pytest.mark.gpu


def hdr_array():
    u = synth.hash_uniform("loader_hdr", 256 * 256 * 3).astype(np.float64) ** 4
    return (u * 4000.0 + 0.01).astype(np.float32).reshape(256, 256, 3)


def test_hdr_branch_vs_reference_golden(golden, tmp_path):
    g = golden("loader")
    np.save(tmp_path / "sample.npy", hdr_array())
    np.save(tmp_path / "lambdas.npy", {"sample": np.float64(0.37)}, allow_pickle=True)
    for add_frame in (0, 1):
        inp, col, gnorm, gray, bf = data_loader.npy_loader(str(tmp_path / "sample.npy"), add_frame, True, False,
                                                           "bugy_max_normalization", 0.0, 1.0, 0.1, False, True,
                                                           str(tmp_path / "lambdas.npy"), 16, False)
        tag = "loader.hdr.frame%d" % add_frame
        assert abs(bf - float(g[tag + ".bf"])) < 1e-12
        assert inp.shape == ((2, 1, 272, 272) if add_frame else (2, 1, 256, 256)) and torch.equal(inp[0], inp[1])
        check_summary(inp[0].cpu(), g, tag + ".input", rtol=2e-5, atol=2e-6)
        if not add_frame:
            check_summary(gnorm[0].cpu(), g, tag + ".gray_norm", rtol=2e-6, atol=1e-7)
            check_summary(gray[0].cpu(), g, tag + ".gray", rtol=2e-6, atol=1e-4)
            check_summary(col[0].cpu(), g, tag + ".color", rtol=0, atol=0)


@pytest.mark.parametrize("h,w,norm", [(512, 512, "bugy_max_normalization"), (300, 412, "max_normalization"), (640, 480, "stretch")])
def test_ldr_branches_vs_oracle_with_the_references_random_order(h, w, norm):
    arr = (synth.hash_uniform("ldr%d" % h, h * w * 3) * 255).astype(np.float32).reshape(h, w, 3)
    for neg in (False, True):
        np.random.seed(7)
        inp, col, a, b, z = data_loader.npy_loader(arr, 0, False, neg, norm, 0.1, 1.2, 0.1, False, True, None, 0, False)
        assert z == 0 and a is inp and b is inp and inp.shape == (2, 1, 256, 256) and col.shape == (2, 3, 256, 256)
        np.random.seed(7)        # replay the reference's draws: mode, size, crop x, crop y per frame
        for k in range(2):
            mode = np.random.randint(0, 2)
            rh = 256 if mode == 0 else int(np.random.uniform(256, 512))
            xx = yy = 0
            if rh != 256:
                xx = np.random.randint(0, rh - 256)
                yy = np.random.randint(0, rh - 256)
            want = OD.frame(arr, (rh, rh, yy, xx), False, norm, 1.2, 0.1)
            np.testing.assert_array_equal(col[k].cpu().numpy(), want["color"])
            np.testing.assert_allclose(inp[k].cpu().numpy(), want["input"], rtol=3e-7, atol=1e-7)


def test_hdr_branch_with_resize_and_crop_vs_oracle():
    arr = (synth.hash_uniform("hdrbig", 400 * 400 * 3).astype(np.float64) ** 4 * 3000 + 0.02).astype(np.float32).reshape(400, 400, 3)
    np.random.seed(11)
    inp, col, gn, gs, bf = data_loader.npy_loader(arr, 0, True, False, "bugy_max_normalization", 0, 1, 0.1, False, True, None, 0,
                                                  False, brightness_factor=9.4)
    np.random.seed(11)
    for k in range(2):
        mode = np.random.randint(0, 2)
        rh = 256 if mode == 0 else int(np.random.uniform(256, 512))
        xx = yy = 0
        if rh != 256:
            xx = np.random.randint(0, rh - 256)
            yy = np.random.randint(0, rh - 256)
        want = OD.frame(arr, (rh, rh, yy, xx), True, brightness_factor=9.4)
        np.testing.assert_array_equal(col[k].cpu().numpy(), want["color"])
        np.testing.assert_allclose(gn[k].cpu().numpy(), want["gray_norm"], rtol=3e-7)
        np.testing.assert_allclose(gs[k].cpu().numpy(), want["gray"], rtol=3e-7, atol=1e-4)
        np.testing.assert_allclose(inp[k].cpu().numpy(), want["input"], rtol=2e-5, atol=2e-6)
