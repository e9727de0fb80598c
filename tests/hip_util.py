"""Thin helpers for the GPU parity tests: drive single kernels through the C ABI (ctypes), exactly as the
Python host does."""
import ctypes as C

import torch

from uncltmo_amd import _hip


def to_nhwc(t, code):
    """(N,C,H,W) cpu fp32 -> (N,H,W,C) cuda in the compute dtype."""
    return t.permute(0, 2, 3, 1).contiguous().to(_hip.torch_dtype(code)).cuda()


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_weight(w, code, transposed=False, flip=False):
    """Reference-layout fp32 weight -> packed device tensor."""
    k = w.shape[2]
    cout, cin = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    src = w.float().contiguous().cuda()
    dst = torch.empty(k * k * cout * cin, dtype=_hip.torch_dtype(code), device="cuda")
    _hip.check(_hip.lib().uncl_pack_conv_weight(src.data_ptr(), dst.data_ptr(), code, cout, cin, k, int(transposed),
                                                int(flip), _hip.stream_ptr()), "pack")
    return dst


def run_conv(**kw):
    d = _hip.ConvDesc()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, k, v)
    _hip.check(_hip.lib().uncl_conv_igemm(C.byref(d), _hip.stream_ptr()), "uncl_conv_igemm")
    torch.cuda.synchronize()


def run_pipe(pool_out=None, **kw):
    d = _hip.ConvDesc()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, k, v)
    _hip.check(_hip.lib().uncl_conv3x3_pipe(C.byref(d), pool_out.data_ptr() if pool_out is not None else None,
                                            _hip.stream_ptr()), "uncl_conv3x3_pipe")
    torch.cuda.synchronize()


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def assert_elementwise(out, ref, dtype="bf16", what=""):
    """Element-wise gate for a 16-bit kernel output against an fp32 / fp64 evaluation of the SAME rounded operands:
    |out - ref| <= 2^-7 |ref| + 2^-7 rms(ref) for bf16 (2^-10 for fp16) -- four times the relative spacing of the format plus a floor
    tied to the tensor's own scale (cancellation near zero, one-ulp differences of the rounded concat members).  A rel-L2 gate of
    1.5e-2 passes with 2e-4 of the elements entirely wrong; this one fails on a single wrong, missing (NaN) or shifted element."""
    out, ref = out.double().cpu(), ref.double().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all(), "%s: %d non-finite (unwritten?) elements" % (what, int((~torch.isfinite(out)).sum()))
    eps = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
    tol = eps * ref.abs() + eps * ref.pow(2).mean().sqrt()
    bad = (out - ref).abs() > tol
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError("%s: %d of %d elements outside the element-wise gate; first at %s: got %.6g want %.6g (tol %.3g)"
                             % (what, int(bad.sum()), bad.numel(), idx, out[tuple(idx)].item(), ref[tuple(idx)].item(),
                                tol[tuple(idx)].item()))
