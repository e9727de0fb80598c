"""GPU parity of the backward kernels (weight / data gradients) against torch CPU autograd of the same op."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hip_util import from_nhwc, pack_weight, rel_l2, run_pipe, to_nhwc
from uncltmo_amd import _hip

pytestmark = pytest.mark.gpu
BF = _hip.BF16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def q(t):
    return t.to(torch.bfloat16).float()


def wgrad(gy, dw_shape_packed, **kw):
    d = _hip.ConvDesc()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, k, v)
    dw = torch.zeros(dw_shape_packed, dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().uncl_conv_wgrad(C.byref(d), gy.data_ptr(), dw.data_ptr(), _hip.stream_ptr()), "uncl_conv_wgrad")
    torch.cuda.synchronize()
    return dw


def unpack(dw, cout, cin, k, transposed, flip):
    dst = torch.zeros((cin, cout, k, k) if transposed else (cout, cin, k, k), dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().uncl_unpack_conv_wgrad(dw.data_ptr(), dst.data_ptr(), cout, cin, k, int(transposed), int(flip), 0,
                                                 _hip.stream_ptr()), "unpack")
    return dst.cpu()


@pytest.mark.parametrize("cin,cout,h,w,n", [(32, 32, 20, 37, 2), (64, 96, 33, 40, 3), (32, 64, 70, 70, 1)])
def test_wgrad3x3_valid(cin, cout, h, w, n):
    x = q(rnd(n, cin, h, w, seed=1)).requires_grad_(False)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=0.1).requires_grad_(True)
    gy = q(rnd(n, cout, h - 2, w - 2, seed=3))
    F.conv2d(x, wt).backward(gy)
    dw = wgrad(to_nhwc(gy, BF), (9, cout, cin), dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin,
               Cout=cout, src0=to_nhwc(x, BF), src0_H=h, src0_W=w, src0_C=cin)
    assert rel_l2(unpack(dw, cout, cin, 3, False, False), wt.grad) < 2e-3


def test_wgrad3x3_transposed_concat_ssr():
    c, cout, h, w, n = 32, 64, 19, 21, 2
    x2 = q(rnd(n, c, h, w, seed=4).abs())
    x1 = q(rnd(n, c, h - 1, w - 1, seed=5))
    wt = rnd(4 * c, cout, 3, 3, seed=6, scale=0.05).requires_grad_(True)
    cat = torch.cat([x2, F.pad(x1, (0, 1, 0, 1), mode="replicate"), q(x2 ** 2), q((x2 + 1e-8) ** 0.5)], 1)
    gy = q(rnd(n, cout, h + 2, w + 2, seed=7))
    F.conv_transpose2d(cat, wt).backward(gy)
    dw = wgrad(to_nhwc(gy, BF), (9, cout, 4 * c), dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=n, H=h, W=w,
               Cin=4 * c, Cout=cout, src0=to_nhwc(x2, BF), src0_H=h, src0_W=w, src0_C=c, src1=to_nhwc(x1, BF), src1_H=h - 1,
               src1_W=w - 1, src1_C=c)
    assert rel_l2(unpack(dw, cout, 4 * c, 3, True, True), wt.grad) < 3e-3


@pytest.mark.parametrize("n", [2, 3])
def test_wgrad1x1(n):
    cin, cout = 256, 128
    x = q(rnd(n, cin, 12, 12, seed=8))
    wt = rnd(cout, cin, 1, 1, seed=9, scale=0.1).requires_grad_(True)
    gy = q(rnd(n, cout, 12, 12, seed=10))
    F.conv2d(x, wt).backward(gy)
    dw = wgrad(to_nhwc(gy, BF), (1, cout, cin), dtype=BF, ksize=1, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=12, W=12, Cin=cin,
               Cout=cout, src0=to_nhwc(x, BF), src0_H=12, src0_W=12, src0_C=cin)
    assert rel_l2(unpack(dw, cout, cin, 1, False, False), wt.grad) < 2e-3


def test_dgrad_via_pipe_kernel_is_conv_with_transposed_weights():
    """d/dx of a valid 3x3 conv = full (pad 2) correlation of gy with the in/out-swapped, flipped weights; the forward
    kernel computes it when handed the weight as if it were a ConvTranspose2d weight of shape (Cout, Cin, 3, 3)."""
    cin, cout, h, w, n = 64, 32, 23, 35, 2
    x = q(rnd(n, cin, h, w, seed=11)).requires_grad_(True)
    wt = q(rnd(cout, cin, 3, 3, seed=12, scale=0.1))
    gy = q(rnd(n, cout, h - 2, w - 2, seed=13))
    F.conv2d(x, wt).backward(gy)
    gx = torch.zeros(n, h, w, cin, dtype=torch.bfloat16, device="cuda")
    # conv weight (Cout, Cin, 3, 3) read as a transposed-conv weight (Cin'=Cout, Cout'=Cin): exactly the dgrad operator
    run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=n, H=h - 2, W=w - 2, Cin=cout, Cout=cin,
             src0=to_nhwc(gy, BF), src0_H=h - 2, src0_W=w - 2, src0_C=cout,
             weight=pack_weight(wt, BF, transposed=True, flip=True), act=_hip.ACT_NONE, out=gx, out_H=h, out_W=w, out_C=cin)
    assert rel_l2(from_nhwc(gx), x.grad) < 1.5e-2


def test_colsum_bias_grad():
    x = q(rnd(5000, 512, seed=14)).to(torch.bfloat16).cuda()
    out = torch.zeros(512, device="cuda")
    ws = torch.empty(_hip.lib().uncl_colsum_workspace_bytes(512), dtype=torch.uint8, device="cuda")
    _hip.check(_hip.lib().uncl_colsum_bf16(x.data_ptr(), 5000, 512, 512, out.data_ptr(), 0, ws.data_ptr(), _hip.stream_ptr()), "colsum")
    assert rel_l2(out.cpu(), x.float().cpu().sum(0)) < 1e-5
