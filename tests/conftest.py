import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no GPU is visible so that a bare `pytest tests/` works here.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _poison_free_device_memory(request):
    """UNCL_POISON_GB=8 python -m pytest tests -m gpu: every GPU test starts with the allocator's free memory full of 0x7F bytes
    (uncltmo_amd/debug_poison.py), so that a read of unwritten workspace memory shows instead of seeing a fresh process's zeros"""
    if os.environ.get("UNCL_POISON_GB") and "gpu" in request.keywords and torch.cuda.is_available():
        from uncltmo_amd.debug_poison import poison_free_memory
        poison_free_memory()
    yield


@pytest.fixture(autouse=True)
def _checked_library_reports_nothing(request):
    """UNCL_HIP_LIB=uncltmo_amd/libuncltmo_hip_checked.so python -m pytest tests -m gpu: with the CHECKED library (DESIGN.md section 5,
    __graft_entry__.build_checked()) every GPU test also asserts that no 3x3 convolution / weight-gradient / up-conv launch touched
    memory outside the tensors it was given -- the tests' ragged shapes are where an index would stray.  A no-op with the product
    library (uncl_checked_report returns UNCL_ERR_ARG there)."""
    yield
    if not os.environ.get("UNCL_HIP_LIB") or "gpu" not in request.keywords or not torch.cuda.is_available():
        return
    import ctypes
    from uncltmo_amd import _hip
    out = (ctypes.c_ulonglong * 4)()
    if _hip.lib().uncl_checked_report(out, 1) == 0:
        assert out[0] == 0, "checked library: %d accesses outside the launch's tensors, first at 0x%x (source line %d, %d bytes)" % (
            out[0], out[1], out[2], out[3])


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


def synth_state(spec, salt):
    """state dict (plain tensors) for a spec from uncltmo_amd.state_spec, via the hash generator."""
    from oracle.state import synth_state as _s
    return _s(spec, salt)


def check_summary(t, g, key, rtol=1e-4, atol=1e-5):
    """Compare a tensor with a golden summary written by make_golden.summarize()."""
    assert tuple(t.shape) == tuple(g[key + ".shape"]), (key, t.shape, g[key + ".shape"])
    f = t.detach().double().reshape(-1).cpu()
    vals = f[torch.from_numpy(g[key + ".pos"])].numpy()
    np.testing.assert_allclose(vals, g[key + ".val"], rtol=rtol, atol=atol, err_msg=key)
    np.testing.assert_allclose(f.abs().sum().item(), g[key + ".abssum"], rtol=rtol, err_msg=key + ".abssum")
