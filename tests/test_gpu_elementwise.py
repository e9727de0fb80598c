"""Element-wise gates for the 16-bit convolution kernels (the fast path of the bench and of training).

tests/test_gpu_conv.py gates these kernels by rel-L2 (1.5e-2 ... 2e-2), which a tensor passes with 2e-4 of its elements entirely
wrong: an edge lane, a tile-border row, one dropped store.  Here every output element is checked on its own,
    |out - ref| <= 2^-7 |ref| + 2^-7 rms(ref)        (hip_util.assert_elementwise; 2^-10 for fp16)
against an fp32 convolution of the SAME rounded operands, and every output buffer starts as NaN so that an element no store reached
fails the gate.  Cases: (1) every 3x3 / 2x2 layer of the generator at its real size (unet_parts.py:19-33, 98-112, 149-162, 212, 233,
269, 292-298, 311-332; Unet_singleFrame.py:200-209), through the launch configuration the forward uses (concat-ssr loader with the
56 -> 57 replicate pad, fused up-conv, fused first layer, pooled copies, 10x10 / 12x12 whole-sample tiles); (2) every border class of
the tilings: output widths = 0, 1, 31 (mod 32), heights = 0, 1, 7 (mod 8) and 0, 1, 15 (mod 16), for 32- and 64-channel tiles, valid
and transposed; (3) the data- and weight-gradient kernels at full layer sizes."""
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


def nan_out(*shape, dtype=torch.bfloat16):
    return torch.full(shape, float("nan"), dtype=dtype, device="cuda")


def conv_ref(x, wt, b, pad):
    """fp32 reference on the CPU (fp64 accumulation is not needed against a 2^-7 gate)"""
    y = F.conv2d(x, wt, b) if pad == 0 else F.conv_transpose2d(x, wt, b)
    return F.relu(y)


def run_plain(x, wt, b, pad, pool=False, code=BF):
    n, cin, h, w = x.shape
    cout = wt.shape[0] if pad == 0 else wt.shape[1]
    ho, wo = (h - 2, w - 2) if pad == 0 else (h + 2, w + 2)
    dt = _hip.torch_dtype(code)
    out = nan_out(n, ho, wo, cout, dtype=dt)
    pl = nan_out(n, ho // 2, wo // 2, cout, dtype=dt) if pool else None
    packed = pack_weight(wt, code) if pad == 0 else pack_weight(wt, code, transposed=True, flip=True)
    run_pipe(pool_out=pl, dtype=code, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout,
             src0=to_nhwc(x, code), src0_H=h, src0_W=w, src0_C=cin, weight=packed, bias=b.cuda(), act=_hip.ACT_RELU, out=out,
             out_H=ho, out_W=wo, out_C=cout)
    return out, pl


# ---- (2) border classes of the tilings ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (128, 64), (32, 64)])
@pytest.mark.parametrize("pad", [0, 2])
def test_tile_border_classes(cin, cout, pad):
    """output extents that end exactly on, one past, and one short of a tile border in both directions"""
    for it, (ho, wo) in enumerate([(32, 32), (33, 33), (31, 63), (16, 65), (17, 31), (15, 95), (8, 64), (9, 1 + 32), (7, 30),
                                   (24, 24), (26, 26), (28, 28), (57, 59)]):
        h, w = (ho + 2, wo + 2) if pad == 0 else (ho - 2, wo - 2)
        n = 2 if ho * wo > 1500 else 3
        x = q(rnd(n, cin, h, w, seed=11 + it))
        b = rnd(cout, seed=13 + it)
        wt = q(rnd(cout, cin, 3, 3, seed=12 + it, scale=0.06)) if pad == 0 else q(rnd(cin, cout, 3, 3, seed=12 + it, scale=0.06))
        ref = conv_ref(x, wt, b, pad)
        pool = pad == 0 and cout >= 32
        out, pl = run_plain(x, wt, b, pad, pool=pool)
        assert_elementwise(from_nhwc(out), ref, "bf16", "cin %d cout %d pad %d out %dx%d" % (cin, cout, pad, ho, wo))
        if pool:
            # the pooled copy is a selection of stored values: exact against the max-pool of what was stored
            assert torch.equal(from_nhwc(pl), F.max_pool2d(from_nhwc(out), 2)), (cin, cout, ho, wo)


# ---- (1) every layer of the generator at its real size -------------------------------------------------------------------------
ENCODER = [  # (name, cin, cout, input extent, pad, pooled copy)
    ("down_path.0.conv", 32, 64, 126, 0, False), ("down_path.0.conv1", 64, 64, 124, 0, True),
    ("down_path.1.conv", 64, 128, 61, 0, False), ("down_path.1.conv1", 128, 128, 59, 0, True),
    ("down_path.2.conv", 128, 256, 28, 0, False), ("down_path.2.conv1", 256, 256, 26, 0, True),
    ("down_path.3.conv", 256, 256, 12, 0, False), ("down_path.3.conv1", 256, 256, 10, 2, False),
    ("up_path.0.conv.conv1", 128, 128, 26, 2, False), ("up_path.1.conv.conv1", 64, 64, 59, 2, False),
    ("up_path.2.conv.conv1", 32, 32, 124, 2, False), ("up_path.3.conv.conv1", 32, 32, 254, 2, False),
    ("inc.conv.conv1", 32, 32, 254, 0, True)]


@pytest.mark.parametrize("name,cin,cout,h,pad,pool", ENCODER)
def test_plain_layers_at_full_size(name, cin, cout, h, pad, pool):
    n = 2 if h > 100 else 5
    x = q(rnd(n, cin, h, h, seed=21).abs())
    wt = q(rnd(cout, cin, 3, 3, seed=22, scale=0.05)) if pad == 0 else q(rnd(cin, cout, 3, 3, seed=22, scale=0.05))
    b = rnd(cout, seed=23, scale=0.3)
    ref = conv_ref(x, wt, b, pad)
    out, pl = run_plain(x, wt, b, pad, pool=pool)
    assert_elementwise(from_nhwc(out), ref, "bf16", name)
    if pool:
        assert torch.equal(from_nhwc(pl), F.max_pool2d(from_nhwc(out), 2)), name


@pytest.mark.parametrize("name,c,cout,hs,h1", [("up_path.0.conv.conv", 256, 128, 24, 24), ("up_path.1.conv.conv", 128, 64, 57, 56),
                                               ("up_path.2.conv.conv", 64, 32, 122, 122), ("up_path.3.conv.conv", 32, 32, 252, 252)])
def test_concat_ssr_layers_at_full_size(name, c, cout, hs, h1):
    """skip (hs x hs) and up-sampled map (h1 x h1, replicate-padded to the skip's extent: 56 -> 57 on the second level)"""
    n = 2
    x2 = q(rnd(n, c, hs, hs, seed=31).abs() * (rnd(n, c, hs, hs, seed=32) > -0.4))        # ReLU outputs: exact zeros too
    x1 = q(rnd(n, c, h1, h1, seed=33))
    wt, b = q(rnd(4 * c, cout, 3, 3, seed=34, scale=0.03)), rnd(cout, seed=35, scale=0.3)
    d = hs - h1
    x1p = F.pad(x1, (d // 2, d - d // 2, d // 2, d - d // 2), mode="replicate")
    cat = torch.cat([x2, x1p, q(x2 ** 2), q(torch.sqrt(x2 + 1e-8))], 1)
    ref = F.relu(F.conv_transpose2d(cat, wt, b))
    out = nan_out(n, hs + 2, hs + 2, cout)
    run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=n, H=hs, W=hs, Cin=4 * c, Cout=cout, src0=to_nhwc(x2, BF),
             src0_H=hs, src0_W=hs, src0_C=c, src1=to_nhwc(x1, BF), src1_H=h1, src1_W=h1, src1_C=c,
             weight=pack_weight(wt, BF, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_RELU, out=out, out_H=hs + 2,
             out_W=hs + 2, out_C=cout)
    assert_elementwise(from_nhwc(out), ref, "bf16", name)


FLAT_PLAIN = [e for e in ENCODER if e[3] <= 61 and e[2] >= 64 and not e[5]]      # the <= 61-pixel levels without a pooled copy


@pytest.mark.parametrize("mpw", [2, 3])
@pytest.mark.parametrize("name,cin,cout,h,pad,pool", FLAT_PLAIN)
def test_plain_layers_at_full_size_on_forced_flat_tiles(name, cin, cout, h, pad, pool, mpw):
    """The flat M-tiles (csrc/conv3x3_flat.hip) are the product's default on these levels at inference batch sizes, but at a test's
    batch the launcher's cost figure keeps the rectangles -- so far the flat kernel met torch only through `flat == rectangular, bit
    for bit`.  Here it is FORCED (uncl_conv3x3_set_flat(mpw): that many M-tiles per multiplying wave wherever the kernel applies) and
    gated element by element against the fp32 convolution of the same operands, at the real layer sizes."""
    lib = _hip.lib()
    n = 5
    x = q(rnd(n, cin, h, h, seed=121).abs())
    wt = q(rnd(cout, cin, 3, 3, seed=122, scale=0.05)) if pad == 0 else q(rnd(cin, cout, 3, 3, seed=122, scale=0.05))
    b = rnd(cout, seed=123, scale=0.3)
    ref = conv_ref(x, wt, b, pad)
    old = lib.uncl_conv3x3_set_flat(mpw)
    try:
        out, _ = run_plain(x, wt, b, pad, pool=False)
    finally:
        lib.uncl_conv3x3_set_flat(old)
    assert_elementwise(from_nhwc(out), ref, "bf16", "%s, flat tiles x%d" % (name, mpw))


@pytest.mark.parametrize("mpw", [2, 3])
@pytest.mark.parametrize("name,c,cout,hs,h1", [("up_path.0.conv.conv", 256, 128, 24, 24), ("up_path.1.conv.conv", 128, 64, 57, 56)])
def test_concat_ssr_layers_at_full_size_on_forced_flat_tiles(name, c, cout, hs, h1, mpw):
    """the two skip-concat layers the forward runs on flat tiles, forced onto them at a test's batch size, against torch"""
    lib = _hip.lib()
    n = 3
    x2 = q(rnd(n, c, hs, hs, seed=131).abs() * (rnd(n, c, hs, hs, seed=132) > -0.4))
    x1 = q(rnd(n, c, h1, h1, seed=133))
    wt, b = q(rnd(4 * c, cout, 3, 3, seed=134, scale=0.03)), rnd(cout, seed=135, scale=0.3)
    d = hs - h1
    x1p = F.pad(x1, (d // 2, d - d // 2, d // 2, d - d // 2), mode="replicate")
    cat = torch.cat([x2, x1p, q(x2 ** 2), q(torch.sqrt(x2 + 1e-8))], 1)
    ref = F.relu(F.conv_transpose2d(cat, wt, b))
    out = nan_out(n, hs + 2, hs + 2, cout)
    old = lib.uncl_conv3x3_set_flat(mpw)
    try:
        run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=n, H=hs, W=hs, Cin=4 * c, Cout=cout, src0=to_nhwc(x2, BF),
                 src0_H=hs, src0_W=hs, src0_C=c, src1=to_nhwc(x1, BF), src1_H=h1, src1_W=h1, src1_C=c,
                 weight=pack_weight(wt, BF, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_RELU, out=out, out_H=hs + 2,
                 out_W=hs + 2, out_C=cout)
    finally:
        lib.uncl_conv3x3_set_flat(old)
    assert_elementwise(from_nhwc(out), ref, "bf16", "%s, flat tiles x%d" % (name, mpw))


@pytest.mark.parametrize("c,h,w,n", [(32, 126, 126, 2), (64, 61, 61, 3), (64, 9, 23, 2), (64, 17, 40, 2)])
def test_fused_upconv_concat_layer_at_full_size(c, h, w, n):
    """up_path.3 / up_path.2: the 2x2 stride-2 transposed conv (32 -> 32, 64 -> 64 channels) recomputed in the concat layer's
    loader (UNCL_SRC_CONCAT_SSR_UP; unet_parts.py:269,288,311-332), also at sizes whose tiles hang over every border"""
    H, W = 2 * h, 2 * w
    x2 = q(rnd(n, c, H, W, seed=41).abs() * (rnd(n, c, H, W, seed=42) > -0.4))
    xs = q(rnd(n, c, h, w, seed=43).abs())
    wu, bu = q(rnd(c, c, 2, 2, seed=44, scale=0.15)), rnd(c, seed=45, scale=0.2)
    wt, b = q(rnd(4 * c, 32, 3, 3, seed=46, scale=0.05)), rnd(32, seed=47, scale=0.3)
    x1 = q(F.conv_transpose2d(xs, wu, bu, stride=2))
    cat = torch.cat([x2, x1, q(x2 ** 2), q(torch.sqrt(x2 + 1e-8))], 1)
    ref = F.relu(F.conv_transpose2d(cat, wt, b))
    out = nan_out(n, H + 2, W + 2, 32)
    run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR_UP, N=n, H=H, W=W, Cin=4 * c, Cout=32, src0=to_nhwc(x2, BF),
             src0_H=H, src0_W=W, src0_C=c, src1=to_nhwc(xs, BF), src1_H=h, src1_W=w, src1_C=c,
             up_w=pack_weight(wu, BF, transposed=True), up_b=bu.cuda(), weight=pack_weight(wt, BF, transposed=True, flip=True),
             bias=b.cuda(), act=_hip.ACT_RELU, out=out, out_H=H + 2, out_W=W + 2, out_C=32)
    assert_elementwise(from_nhwc(out), ref, "bf16", "concat-ssr + fused up (%d channels)" % c)
    # the launch is deterministic: a difference between repeated launches is a race (round 5: the staging waves once read the
    # LDS-resident up-conv weights before every wave had finished writing them -- one run in three showed it)
    args = dict(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR_UP, N=n, H=H, W=W, Cin=4 * c, Cout=32, src0=to_nhwc(x2, BF),
                src0_H=H, src0_W=W, src0_C=c, src1=to_nhwc(xs, BF), src1_H=h, src1_W=w, src1_C=c,
                up_w=pack_weight(wu, BF, transposed=True), up_b=bu.cuda(), weight=pack_weight(wt, BF, transposed=True, flip=True),
                bias=b.cuda(), act=_hip.ACT_RELU, out_H=H + 2, out_W=W + 2, out_C=32)
    for _ in range(6):
        again = nan_out(n, H + 2, W + 2, 32)
        run_pipe(out=again, **args)
        assert torch.equal(out, again), "repeated launches differ"


def test_fused_first_layer_at_full_size():
    """inc: conv(1 -> 32) rebuilt from the fp32 image inside inc.conv.conv1's loader (UNCL_SRC_IMAGE1), with the pooled copy"""
    n, h = 3, 256
    x = rnd(n, 1, h, h, seed=51).abs()
    w0, b0 = rnd(32, 1, 3, 3, seed=52, scale=0.3), rnd(32, seed=53, scale=0.2)
    w1, b1 = q(rnd(32, 32, 3, 3, seed=54, scale=0.06)), rnd(32, seed=55, scale=0.2)
    # the loader splits the fp32 image into a bf16 head + tail (exact to 2^-17) and rounds the first layer's weights and output
    mid = q(F.relu(F.conv2d(x, q(w0), b0)))
    ref = F.relu(F.conv2d(mid, w1, b1))
    out, pooled = nan_out(n, h - 4, h - 4, 32), nan_out(n, (h - 4) // 2, (h - 4) // 2, 32)
    run_pipe(pool_out=pooled, dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_IMAGE1, N=n, H=h - 2, W=h - 2, Cin=32, Cout=32,
             src0=x.reshape(n, h, h).cuda().contiguous(), src0_H=h, src0_W=h, src0_C=1, pre_w=w0.cuda().contiguous(),
             pre_b=b0.cuda(), weight=pack_weight(w1, BF), bias=b1.cuda(), act=_hip.ACT_RELU, out=out, out_H=h - 4, out_W=h - 4,
             out_C=32)
    assert_elementwise(from_nhwc(out), ref, "bf16", "inc (fused first layer)")
    assert torch.equal(from_nhwc(pooled), F.max_pool2d(from_nhwc(out), 2))


@pytest.mark.parametrize("c,h", [(256, 12), (128, 28), (64, 61), (32, 126)])
def test_upconv2x2_at_full_size(c, h):
    n = 3
    x, wt, b = q(rnd(n, c, h, h, seed=61)), q(rnd(c, c, 2, 2, seed=62, scale=0.1)), rnd(c, seed=63, scale=0.3)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    out = nan_out(n, 2 * h, 2 * h, c)
    xd, wd, bd = to_nhwc(x, BF), pack_weight(wt, BF, transposed=True), b.cuda()
    _hip.check(_hip.lib().uncl_upconv2x2(xd.data_ptr(), None, 0, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, h, c, c,
                                         _hip.stream_ptr()), "upconv")
    torch.cuda.synchronize()
    assert_elementwise(from_nhwc(out), ref, "bf16", "up %d" % c)


# ---- (3) gradients at full layer sizes -------------------------------------------------------------------------------------
def _unpack(dw, cout, cin, k, transposed, flip):
    dst = torch.empty((cin, cout, k, k) if transposed else (cout, cin, k, k), dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().uncl_unpack_conv_wgrad(dw.data_ptr(), dst.data_ptr(), cout, cin, k, int(transposed), int(flip), 0,
                                                 _hip.stream_ptr()), "unpack")
    return dst.cpu()


@pytest.mark.parametrize("cin,cout,h,pad,n", [(32, 32, 254, 0, 2), (64, 64, 124, 0, 3), (128, 128, 59, 0, 4), (256, 256, 26, 0, 6),
                                              (32, 32, 254, 2, 2), (128, 128, 26, 2, 5)])
def test_dgrad_at_full_size_with_mask(cin, cout, h, pad, n):
    """data gradient of a valid / transposed 3x3 layer (forward kernel over re-packed weights), multiplied by the ReLU mask of the
    layer that produced its input, element by element against autograd"""
    x = q(rnd(n, cin, h, h, seed=71))
    x.requires_grad_(True)
    wt = q(rnd(cout, cin, 3, 3, seed=72, scale=0.05)) if pad == 0 else q(rnd(cin, cout, 3, 3, seed=72, scale=0.05))
    y = F.conv2d(x, wt) if pad == 0 else F.conv_transpose2d(x, wt)
    gy = q(rnd(*y.shape, seed=73))
    y.backward(gy)
    mask = q(rnd(n, cin, h, h, seed=74))                    # the producing layer's (signed) pre-ReLU output stands in
    want = x.grad * (mask > 0)
    hy = y.shape[2]
    if pad == 0:
        # gradient of a valid Conv2d: pad-2 correlation of gy with the weight read as a ConvTranspose2d weight (Cin' = cout)
        packed, dpad = pack_weight(wt.detach(), BF, transposed=True, flip=True), 2
    else:
        # gradient of a ConvTranspose2d: valid correlation of gy with the weight read as a Conv2d weight (Cout' = cin)
        packed, dpad = pack_weight(wt.detach(), BF, transposed=False), 0
    d = _hip.ConvDesc()
    gx = nan_out(n, h, h, cin)
    gyd, md = to_nhwc(gy, BF), to_nhwc(mask, BF)
    for k_, v in dict(dtype=BF, ksize=3, pad=dpad, src_mode=_hip.SRC_PLAIN, N=n, H=hy, W=hy, Cin=cout, Cout=cin, src0=gyd.data_ptr(),
                      src0_H=hy, src0_W=hy, src0_C=cout, weight=packed.data_ptr(), act=_hip.ACT_NONE, out=gx.data_ptr(), out_H=h,
                      out_W=h, out_C=cin).items():
        setattr(d, k_, v)
    _hip.check(_hip.lib().uncl_conv3x3_dgrad(C.byref(d), md.data_ptr(), 0.0, 0, _hip.stream_ptr()), "dgrad")
    torch.cuda.synchronize()
    assert_elementwise(from_nhwc(gx), want, "bf16", "dgrad %d->%d %d pad %d" % (cin, cout, h, pad))


@pytest.mark.parametrize("cin,cout,h,pad,n", [(32, 32, 254, 0, 2), (64, 64, 124, 0, 3), (128, 128, 59, 0, 4), (32, 32, 254, 2, 2),
                                              (128, 128, 26, 2, 5)])
def test_wgrad_at_full_size(cin, cout, h, pad, n):
    """weight gradient (fp32 atomics over bf16 products): every one of the 9 Cin Cout elements within 2^-10 of its own value plus
    2^-10 of the tensor's rms -- the products are exact in fp32, only the accumulation order differs from autograd's"""
    x = q(rnd(n, cin, h, h, seed=81))
    wt = (q(rnd(cout, cin, 3, 3, seed=82, scale=0.05)) if pad == 0 else q(rnd(cin, cout, 3, 3, seed=82, scale=0.05))).requires_grad_(True)
    y = F.conv2d(x, wt) if pad == 0 else F.conv_transpose2d(x, wt)
    gy = q(rnd(*y.shape, seed=83))
    y.backward(gy)
    hy = y.shape[2]
    dw = torch.zeros(9 * cin * cout, dtype=torch.float32, device="cuda")
    d = _hip.ConvDesc()
    xd, gyd = to_nhwc(x, BF), to_nhwc(gy, BF)
    for k_, v in dict(dtype=BF, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=h, Cin=cin, Cout=cout, src0=xd.data_ptr(),
                      src0_H=h, src0_W=h, src0_C=cin).items():
        setattr(d, k_, v)
    _hip.check(_hip.lib().uncl_conv_wgrad(C.byref(d), gyd.data_ptr(), dw.data_ptr(), _hip.stream_ptr()), "wgrad")
    torch.cuda.synchronize()
    got = _unpack(dw, cout, cin, 3, pad == 2, pad == 2)
    assert_elementwise(got, wt.grad, "fp16", "wgrad %d->%d %d pad %d" % (cin, cout, h, pad))
