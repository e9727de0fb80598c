"""Flat M-tiles (csrc/conv3x3_flat.hip) against the rectangular tiles of csrc/conv3x3_pc.hip: the output pixels of a sample are
linearised on the pitch of the padded input and cut into M-tiles of 32, tiles run across sample borders.  Same accumulation order per
output element, so every stored element must be IDENTICAL to the rectangular kernel's -- which tests/test_gpu_conv.py and
tests/test_gpu_elementwise.py pin to torch.  Layers: unet_parts.py:56-87 (double_conv), 98-112 / 149-162 (transposed pairs),
311-332 (skip operator: concat [x2, x1, x2^2, sqrt(x2 + 1e-8)]), their data gradients.  Outputs start as NaN (forward) so that an
element no store reached fails; every forced M-tiles-per-wave setting (2 / 3 / 4) and the launcher's own choice are run."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hip_util import assert_elementwise, from_nhwc, pack_weight, run_pipe, to_nhwc
from uncltmo_amd import _hip

pytestmark = pytest.mark.gpu
BF = _hip.BF16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def q(t, code=BF):
    return t.to(_hip.torch_dtype(code)).float()


def flat_vs_rect(fn, settings=(1, 2, 3, 4)):
    """fn() under rectangular tiles, then under every flat setting; returns (rect, [flat...])"""
    lib = _hip.lib()
    old = lib.uncl_conv3x3_set_flat(0)
    try:
        ref = fn()
        got = []
        for sset in settings:
            lib.uncl_conv3x3_set_flat(sset)
            got.append(fn())
    finally:
        lib.uncl_conv3x3_set_flat(old)
    return ref, got


PLAIN = [  # cin, cout, input h, w, n, pad
    (64, 128, 61, 61, 5, 0),        # down_path.1 first conv: two cout tiles, 59 x 59 (113 M-tiles per sample)
    (128, 256, 28, 28, 9, 0),       # down_path.2 first conv: four cout tiles, 23 M-tiles per sample: every tile crosses a sample border
    (128, 128, 26, 26, 7, 2),       # up_path.0 second conv (transposed): zero borders, 27 M-tiles per sample
    (64, 64, 59, 59, 3, 2),         # up_path.1 second conv: resident weights (one cout tile, two chunks)
    (256, 256, 26, 26, 4, 0),       # eight chunks, four cout tiles, 24 x 24 output (20 M-tiles per sample)
    (96, 64, 17, 40, 3, 0),         # three chunks: odd chunk count flips the stage parity per tile; ragged 15 x 38
    (64, 64, 7, 9, 11, 0),          # 5 x 7 outputs: 2 M-tiles per sample, many sample borders per tile
    (64, 64, 3, 3, 70, 0),          # 1 x 1 outputs: one M-tile per sample, one valid pixel each
    (64, 64, 5, 34, 2, 2),          # 7 x 36 output, pitch 38: a row is longer than an M-tile
    (32, 64, 30, 30, 3, 0),         # single chunk
    (256, 256, 12, 12, 9, 0),       # down_path.3 first conv: 10 x 10 outputs, 4 M-tiles per sample (whole-sample tiles otherwise)
    (256, 256, 10, 10, 7, 2),       # down_path.3 second conv (transposed): 12 x 12 outputs, 6 M-tiles per sample
]


@pytest.mark.parametrize("cin,cout,h,w,n,pad", PLAIN)
@pytest.mark.parametrize("code", [_hip.BF16, _hip.F16])
def test_flat_plain_bitwise(cin, cout, h, w, n, pad, code):
    dt = _hip.torch_dtype(code)
    x, b = q(rnd(n, cin, h, w, seed=301), code), rnd(cout, seed=303)
    wt = q(rnd(cout, cin, 3, 3, seed=302, scale=0.1), code) if pad == 0 else q(rnd(cin, cout, 3, 3, seed=302, scale=0.1), code)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    xd, bd = to_nhwc(x, code), b.cuda()
    wd = pack_weight(wt, code) if pad == 0 else pack_weight(wt, code, transposed=True, flip=True)

    def run():
        out = torch.full((n, ho, wo, cout), float("nan"), dtype=dt, device="cuda")
        run_pipe(dtype=code, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout, src0=xd, src0_H=h,
                 src0_W=w, src0_C=cin, weight=wd, bias=bd, act=_hip.ACT_RELU, out=out, out_H=ho, out_W=wo, out_C=cout)
        return out

    ref, got = flat_vs_rect(run)
    assert torch.isfinite(ref.float()).all()
    for g, sset in zip(got, (1, 2, 3, 4)):
        assert torch.equal(ref, g), "flat setting %d differs from the rectangular tiles" % sset
    y = F.relu(F.conv2d(x, wt, b) if pad == 0 else F.conv_transpose2d(x, wt, b))
    assert_elementwise(from_nhwc(got[0]), y, "bf16" if code == _hip.BF16 else "fp16", "flat plain")


@pytest.mark.parametrize("c,cout,h,w,n,short", [(256, 128, 24, 24, 5, 0),      # up_path.0.conv.conv: 32 chunks, two cout tiles
                                                (128, 64, 57, 57, 3, 1),       # up_path.1.conv.conv: x1 is 56 x 56, replicate-padded
                                                (64, 64, 9, 13, 4, 0), (32, 64, 20, 20, 3, 1)])
def test_flat_concat_ssr_bitwise(c, cout, h, w, n, short):
    x2 = q(rnd(n, c, h, w, seed=311).abs() * (rnd(n, c, h, w, seed=315) > -0.5))       # ReLU outputs: exact zeros too
    x1 = q(rnd(n, c, h - short, w - short, seed=312))
    wt, b = q(rnd(4 * c, cout, 3, 3, seed=313, scale=0.05)), rnd(cout, seed=314)
    x2d, x1d, wd, bd = to_nhwc(x2, BF), to_nhwc(x1, BF), pack_weight(wt, BF, transposed=True, flip=True), b.cuda()

    def run():
        out = torch.full((n, h + 2, w + 2, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=n, H=h, W=w, Cin=4 * c, Cout=cout, src0=x2d, src0_H=h,
                 src0_W=w, src0_C=c, src1=x1d, src1_H=h - short, src1_W=w - short, src1_C=c, weight=wd, bias=bd,
                 act=_hip.ACT_RELU, out=out, out_H=h + 2, out_W=w + 2, out_C=cout)
        return out

    ref, got = flat_vs_rect(run)
    for g, sset in zip(got, (1, 2, 3, 4)):
        assert torch.equal(ref, g), "flat setting %d differs from the rectangular tiles" % sset
    cat = torch.cat([x2, F.pad(x1, (0, short, 0, short), mode="replicate"), q(x2 ** 2), q(torch.sqrt(x2 + 1e-8))], 1)
    assert_elementwise(from_nhwc(got[0]), F.relu(F.conv_transpose2d(cat, wt, b)), "bf16", "flat concat-ssr")


@pytest.mark.parametrize("gc,cd,gh,pad_d,n,accumulate,use_mask", [(128, 64, 59, 2, 3, 1, True),      # dgrad of down_path.1 first conv
                                                                  (256, 128, 26, 2, 6, 0, True),     # dgrad of down_path.2 first conv
                                                                  (128, 128, 28, 0, 5, 1, False),    # dgrad of a transposed layer: valid
                                                                  (64, 64, 61, 0, 2, 0, False)])
def test_flat_dgrad_store_bitwise(gc, cd, gh, pad_d, n, accumulate, use_mask):
    """gradient mode of the epilogue (ReLU mask of the producing layer, accumulation into an existing gradient)"""
    oh = gh + 2 * pad_d - 2
    gy = to_nhwc(q(rnd(n, gc, gh, gh, seed=331)), BF)
    wd = pack_weight(q(rnd(cd, gc, 3, 3, seed=332, scale=0.1)), BF)
    mask = to_nhwc(q(rnd(n, cd, oh, oh, seed=333)), BF)
    init = to_nhwc(q(rnd(n, cd, oh, oh, seed=334)), BF)

    def run():
        out = init.clone()
        d = _hip.ConvDesc()
        for k_, v in dict(dtype=BF, ksize=3, pad=pad_d, src_mode=_hip.SRC_PLAIN, N=n, H=gh, W=gh, Cin=gc, Cout=cd,
                          src0=gy.data_ptr(), src0_H=gh, src0_W=gh, src0_C=gc, weight=wd.data_ptr(), act=_hip.ACT_NONE,
                          out=out.data_ptr(), out_H=oh, out_W=oh, out_C=cd).items():
            setattr(d, k_, v)
        _hip.check(_hip.lib().uncl_conv3x3_dgrad(C.byref(d), mask.data_ptr() if use_mask else None, 0.0, accumulate,
                                                 _hip.stream_ptr()), "dgrad")
        torch.cuda.synchronize()
        return out

    ref, got = flat_vs_rect(run)
    for g, sset in zip(got, (1, 2, 3, 4)):
        assert torch.equal(ref, g), "flat setting %d differs from the rectangular tiles" % sset


def test_flat_is_taken_at_bench_size():
    """the launcher's cost figure must actually choose the flat tiles on the levels they were built for (100 samples per launch: the
    two-stream forward of the bench): a poisoned rectangular path would otherwise go unnoticed"""
    lib = _hip.lib()
    if not hasattr(lib, "uncl_conv3x3_flat_count"):
        pytest.skip("no launch counter in this build")
    n, c, h = 100, 64, 61
    x = to_nhwc(q(rnd(n, c, h, h, seed=341)), BF)
    wd, bd = pack_weight(q(rnd(128, c, 3, 3, seed=342, scale=0.1)), BF), rnd(128, seed=343).cuda()
    out = torch.empty(n, h - 2, h - 2, 128, dtype=torch.bfloat16, device="cuda")
    before = lib.uncl_conv3x3_flat_count()
    run_pipe(dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=h, Cin=c, Cout=128, src0=x, src0_H=h, src0_W=h,
             src0_C=c, weight=wd, bias=bd, act=_hip.ACT_RELU, out=out, out_H=h - 2, out_W=h - 2, out_C=128)
    assert lib.uncl_conv3x3_flat_count() == before + 1
