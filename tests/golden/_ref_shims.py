"""Import shims for running the UPSTREAM reference (cao-cong/UnCLTMO) in THIS container only.

The reference needs eight third-party packages that are not installed here (timm, cv2, imageio,
skimage, torchvision, torchsummary, contracts, fid's inception deps) and two APIs that were removed
from numpy/scipy.  This file fabricates just enough of them, in memory, for the hot-path modules to
import.  It is used ONLY by tests/golden/make_golden.py to capture golden vectors; it never travels
into the product path and nothing here is copied from the reference.

`DropPath` follows timm's documented behaviour (per-sample Bernoulli(keep) / keep in train mode,
identity in eval).  timm is unpinned by the reference (README.md:23-24), so train-mode parity is only
pinned through the explicit `forced_mask` hook below.
"""
import sys
import types

import numpy as np
import scipy.signal
import scipy.signal.windows
import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


class DropPath(nn.Module):
    forced_mask = None  # class-level hook: (B,) tensor of 0/1 keep flags, set by the capture script

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        if DropPath.forced_mask is not None:
            m = DropPath.forced_mask.to(x).reshape(-1, *([1] * (x.dim() - 1)))
        else:
            m = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            m = m / keep
        return x * m


def _view_as_blocks(arr, block_shape):
    bh, bw = block_shape
    h, w = arr.shape
    assert h % bh == 0 and w % bw == 0
    return arr.reshape(h // bh, bh, w // bw, bw).transpose(0, 2, 1, 3)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(scipy.signal, "gaussian"):
        scipy.signal.gaussian = scipy.signal.windows.gaussian

    ident = lambda *a, **k: (lambda f: f)
    # timm
    _mod("timm")
    _mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _mod("timm.models")
    _mod("timm.models.helpers", load_pretrained=lambda *a, **k: None)
    _mod("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x),
         trunc_normal_=lambda t, **k: t)
    _mod("timm.models.registry", register_model=lambda f: f)
    # cv2 / imageio / skimage / torchvision / torchsummary / contracts
    _mod("cv2")
    _mod("imageio")
    sk = _mod("skimage")
    sk.util = _mod("skimage.util", view_as_blocks=_view_as_blocks)
    sk.transform = _mod("skimage.transform")
    sk.color = _mod("skimage.color")
    sk.exposure = _mod("skimage.exposure")
    sk.io = _mod("skimage.io")
    tv = _mod("torchvision")
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("torchvision shim: not callable in the capture environment")

    tv.transforms = _mod("torchvision.transforms", Compose=_Any, ToTensor=_Any, Normalize=_Any)
    tv.utils = _mod("torchvision.utils")
    tv.datasets = _mod("torchvision.datasets", DatasetFolder=_Any)
    tv.models = _mod("torchvision.models")
    _mod("torchsummary", summary=lambda *a, **k: None)
    _mod("contracts", contract=ident)
    _mod("wget")
    # `fid` is imported by both trainers but every call site is commented out (GanTrainerImg.py:570-572)
    fid = _mod("fid")
    fid.fid_score = _mod("fid.fid_score")
    # matplotlib-free stand-in for utils/plot_util (imported for loss plots only)
    cv2 = sys.modules["cv2"]
    cv2.optflow = _mod("cv2.optflow")
    return DropPath
