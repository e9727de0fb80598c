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
