"""GPU parity of the implicit-GEMM convolution kernel (every loader / epilogue mode) against plain torch CPU
fp32 convolutions of the same op, through the C ABI."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hip_util import from_nhwc, pack_weight, rel_l2, run_conv, run_pipe, to_nhwc
from uncltmo_amd import _hip

pytestmark = pytest.mark.gpu

TOL = {_hip.F32: 2e-5, _hip.BF16: 1.5e-2}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def q(t, code):
    """Round to the compute dtype so that the reference sees the same operands the kernel does."""
    return t.to(_hip.torch_dtype(code)).float()


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
@pytest.mark.parametrize("cin,cout,h,w", [(32, 32, 20, 37), (64, 128, 11, 9), (32, 64, 40, 70), (256, 256, 12, 12)])
def test_conv3x3_valid(code, cin, cout, h, w):
    x, wt, b = q(rnd(2, cin, h, w, seed=1), code), q(rnd(cout, cin, 3, 3, seed=2, scale=0.1), code), rnd(cout, seed=3)
    ref = F.relu(F.conv2d(x, wt, b))
    out = torch.empty(2, h - 2, w - 2, cout, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=3, pad=0, src_mode=_hip.SRC_PLAIN, N=2, H=h, W=w, Cin=cin, Cout=cout,
             src0=to_nhwc(x, code), src0_H=h, src0_W=w, src0_C=cin, weight=pack_weight(wt, code), bias=b.cuda(),
             act=_hip.ACT_RELU, out=out, out_H=h - 2, out_W=w - 2, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[code]


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv_transposed3x3_leaky(code):
    cin, cout, h, w = 64, 32, 13, 33
    x, wt, b = q(rnd(1, cin, h, w, seed=4), code), q(rnd(cin, cout, 3, 3, seed=5, scale=0.1), code), rnd(cout, seed=6)
    ref = F.leaky_relu(F.conv_transpose2d(x, wt, b), 0.2)
    out = torch.empty(1, h + 2, w + 2, cout, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=1, H=h, W=w, Cin=cin, Cout=cout,
             src0=to_nhwc(x, code), src0_H=h, src0_W=w, src0_C=cin,
             weight=pack_weight(wt, code, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_LRELU, out=out,
             out_H=h + 2, out_W=w + 2, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[code]


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv_maxpool_loader(code):
    cin, cout, h, w = 32, 64, 23, 41          # odd sizes: floor pooling drops the last row / column
    x, wt, b = q(rnd(2, cin, h, w, seed=7), code), q(rnd(cout, cin, 3, 3, seed=8, scale=0.1), code), rnd(cout, seed=9)
    ref = F.relu(F.conv2d(F.max_pool2d(x, 2), wt, b))
    hp, wp = h // 2, w // 2
    out = torch.empty(2, hp - 2, wp - 2, cout, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=3, pad=0, src_mode=_hip.SRC_MAXPOOL2, N=2, H=hp, W=wp, Cin=cin, Cout=cout,
             src0=to_nhwc(x, code), src0_H=h, src0_W=w, src0_C=cin, weight=pack_weight(wt, code), bias=b.cuda(),
             act=_hip.ACT_RELU, out=out, out_H=hp - 2, out_W=wp - 2, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[code]


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv_concat_ssr_loader_with_replicate_pad(code):
    c, cout, h, w = 32, 32, 19, 21
    x2 = q(rnd(1, c, h, w, seed=10).abs(), code)              # skip features are post-ReLU, i.e. >= 0
    x1 = q(rnd(1, c, h - 1, w - 1, seed=11), code)            # upsampled map is one short (56 vs 57 upstream)
    wt, b = q(rnd(4 * c, cout, 3, 3, seed=12, scale=0.05), code), rnd(cout, seed=13)
    x1p = F.pad(x1, (0, 1, 0, 1), mode="replicate")
    cat = torch.cat([x2, x1p, x2 ** 2, (x2 + 1e-8) ** 0.5], 1)
    ref = F.relu(F.conv_transpose2d(cat, wt, b))
    out = torch.empty(1, h + 2, w + 2, cout, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=1, H=h, W=w, Cin=4 * c, Cout=cout,
             src0=to_nhwc(x2, code), src0_H=h, src0_W=w, src0_C=c, src1=to_nhwc(x1, code), src1_H=h - 1, src1_W=w - 1,
             src1_C=c, weight=pack_weight(wt, code, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_RELU,
             out=out, out_H=h + 2, out_W=w + 2, out_C=cout)
    # the kernel rounds x2^2 and sqrt(x2) to the compute dtype before the MFMA; bf16 gets the looser bound
    assert rel_l2(from_nhwc(out), ref) < (TOL[code] if code == _hip.F32 else 2e-2)


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv1x1_gelu_residual_dropscale(code):
    n, cin, cout = 3, 256, 256
    x, wt, b = q(rnd(n, cin, 12, 12, seed=14), code), q(rnd(cout, cin, 1, 1, seed=15, scale=0.1), code), rnd(cout, seed=16)
    res = q(rnd(n, cout, 12, 12, seed=17), code)
    sc = torch.tensor([1.0 / 0.95, 0.0, 1.0 / 0.95])
    ref = F.gelu(F.conv2d(x, wt, b)) * sc.reshape(n, 1, 1, 1) + res
    out = torch.empty(n, 12, 12, cout, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=1, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=12, W=12, Cin=cin, Cout=cout,
             src0=to_nhwc(x, code), src0_H=12, src0_W=12, src0_C=cin, weight=pack_weight(wt, code), bias=b.cuda(),
             act=_hip.ACT_GELU, scale_n=sc.cuda(), res=to_nhwc(res, code), out=out, out_H=12, out_W=12, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[code]


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv1x1_grouped(code):
    n, cin, cout, g = 2, 512, 512, 4
    x, wt, b = q(rnd(n, cin, 144, 1, seed=18), code), q(rnd(cout, cin // g, 1, 1, seed=19, scale=0.1), code), rnd(cout, seed=20)
    ref = F.conv2d(x, wt, b, groups=g)
    out = torch.empty(n, 144, 1, cout, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=1, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=144, W=1, Cin=cin // g, Cout=cout // g,
             src0=to_nhwc(x, code), src0_H=144, src0_W=1, src0_C=cin, weight=pack_weight(wt, code), bias=b.cuda(),
             act=_hip.ACT_NONE, out=out, out_H=144, out_W=1, out_C=cout, z_mode=_hip.Z_GROUPS, groups=g)
    assert rel_l2(from_nhwc(out), ref) < TOL[code]


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv_transposed2x2_stride2(code):
    n, c, h, w = 2, 64, 9, 13
    x, wt, b = q(rnd(n, c, h, w, seed=21), code), q(rnd(c, c, 2, 2, seed=22, scale=0.1), code), rnd(c, seed=23)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    out = torch.empty(n, 2 * h, 2 * w, c, dtype=_hip.torch_dtype(code), device="cuda")
    run_conv(dtype=code, ksize=1, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=c, Cout=c,
             src0=to_nhwc(x, code), src0_H=h, src0_W=w, src0_C=c, weight=pack_weight(wt, code, transposed=True),
             bias=b.cuda(), act=_hip.ACT_NONE, out=out, out_H=2 * h, out_W=2 * w, out_C=c, z_mode=_hip.Z_UP2X2)
    assert rel_l2(from_nhwc(out), ref) < TOL[code]


@pytest.mark.parametrize("code", [_hip.F32, _hip.BF16])
def test_conv_fused_outc_sigmoid_and_broadcast_residual(code):
    cin, h, w = 32, 14, 40
    x, wt, b = q(rnd(2, cin, h, w, seed=24), code), q(rnd(cin, 32, 3, 3, seed=25, scale=0.1), code), rnd(32, seed=26)
    w1, b1 = rnd(1, 32, 1, 1, seed=27), rnd(1, seed=28)
    up = F.relu(F.conv_transpose2d(x, wt, b))
    ref1 = torch.sigmoid(F.conv2d(up, w1, b1))
    out = torch.empty(2, h + 2, w + 2, 32, dtype=_hip.torch_dtype(code), device="cuda")
    out1 = torch.empty(2, h + 2, w + 2, dtype=torch.float32, device="cuda")
    run_conv(dtype=code, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=2, H=h, W=w, Cin=cin, Cout=32,
             src0=to_nhwc(x, code), src0_H=h, src0_W=w, src0_C=cin,
             weight=pack_weight(wt, code, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_RELU, out=out,
             out_H=h + 2, out_W=w + 2, out_C=32, out1_w=w1.reshape(32).cuda(), out1_b=b1.cuda(),
             out1_act=_hip.ACT_SIGMOID, out1=out1)
    assert rel_l2(from_nhwc(out), up) < TOL[code]
    assert rel_l2(out1.cpu().unsqueeze(1), ref1) < TOL[code]


# ---- pipelined bf16 kernel (the hot one): every mode, ragged sizes, many tiles per workgroup
BF = _hip.BF16


@pytest.mark.parametrize("cin,cout,h,w,n", [(32, 32, 52, 70, 3), (64, 64, 21, 37, 2), (128, 128, 11, 9, 2),
                                            (32, 64, 40, 33, 1), (256, 256, 12, 12, 5), (32, 32, 256, 256, 2)])
def test_pipe_valid_with_fused_pool(cin, cout, h, w, n):
    x, wt, b = q(rnd(n, cin, h, w, seed=31), BF), q(rnd(cout, cin, 3, 3, seed=32, scale=0.1), BF), rnd(cout, seed=33)
    ref = F.relu(F.conv2d(x, wt, b))
    out = torch.zeros(n, h - 2, w - 2, cout, dtype=torch.bfloat16, device="cuda")
    pool = torch.zeros(n, (h - 2) // 2, (w - 2) // 2, cout, dtype=torch.bfloat16, device="cuda")
    run_pipe(pool_out=pool, dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout,
             src0=to_nhwc(x, BF), src0_H=h, src0_W=w, src0_C=cin, weight=pack_weight(wt, BF), bias=b.cuda(),
             act=_hip.ACT_RELU, out=out, out_H=h - 2, out_W=w - 2, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[BF]
    # the pooled copy must be exactly the max-pool of what was stored (bf16 max is a selection)
    assert torch.equal(from_nhwc(pool), F.max_pool2d(from_nhwc(out), 2))


def test_pipe_transposed_concat_ssr_and_tail():
    c, h, w = 32, 37, 45
    x2 = q(rnd(2, c, h, w, seed=34).abs(), BF)
    x1 = q(rnd(2, c, h - 1, w - 1, seed=35), BF)
    wt, b = q(rnd(4 * c, 32, 3, 3, seed=36, scale=0.05), BF), rnd(32, seed=37)
    cat = torch.cat([x2, F.pad(x1, (0, 1, 0, 1), mode="replicate"), x2 ** 2, (x2 + 1e-8) ** 0.5], 1)
    ref = F.relu(F.conv_transpose2d(cat, wt, b))
    w1, b1 = rnd(1, 32, 1, 1, seed=38), rnd(1, seed=39)
    out = torch.zeros(2, h + 2, w + 2, 32, dtype=torch.bfloat16, device="cuda")
    out1 = torch.zeros(2, h + 2, w + 2, dtype=torch.float32, device="cuda")
    run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=2, H=h, W=w, Cin=4 * c, Cout=32,
             src0=to_nhwc(x2, BF), src0_H=h, src0_W=w, src0_C=c, src1=to_nhwc(x1, BF), src1_H=h - 1, src1_W=w - 1,
             src1_C=c, weight=pack_weight(wt, BF, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_RELU,
             out=out, out_H=h + 2, out_W=w + 2, out_C=32, out1_w=w1.reshape(32).cuda(), out1_b=b1.cuda(),
             out1_act=_hip.ACT_SIGMOID, out1=out1)
    assert rel_l2(from_nhwc(out), ref) < 2e-2
    ref1 = torch.sigmoid(F.conv2d(from_nhwc(out), w1, b1))       # the tail reads the stored (bf16) features
    assert rel_l2(out1.cpu().unsqueeze(1), ref1) < 1e-4


@pytest.mark.parametrize("h,w,n", [(9, 13, 3), (23, 40, 2), (126, 126, 2)])
def test_pipe_concat_ssr_with_upconv_recomputed_in_the_loader(h, w, n):
    """UNCL_SRC_CONCAT_SSR_UP: up() (ConvTranspose2d k2 s2, unet_parts.py:269,288) of the coarse map is rebuilt per halo tile
    inside the concat layer's loader.  Same result as running uncl_upconv2x2 and then the UNCL_SRC_CONCAT_SSR layer, up to the
    bf16 rounding of the up-sampled map (the fused path starts its fp32 accumulation from the bias instead of adding it last)."""
    c = 32
    H, W = 2 * h, 2 * w
    x2 = q(rnd(n, c, H, W, seed=101).abs(), BF)
    xs = q(rnd(n, c, h, w, seed=102), BF)
    wu, bu = q(rnd(c, c, 2, 2, seed=103, scale=0.1), BF), rnd(c, seed=104)
    wt, b = q(rnd(4 * c, 32, 3, 3, seed=105, scale=0.05), BF), rnd(32, seed=106)
    x1 = q(F.conv_transpose2d(xs, wu, bu, stride=2), BF)
    cat = torch.cat([x2, x1, x2 ** 2, (x2 + 1e-8) ** 0.5], 1)
    ref = F.relu(F.conv_transpose2d(cat, wt, b))
    x2d, xsd, wud, bud = to_nhwc(x2, BF), to_nhwc(xs, BF), pack_weight(wu, BF, transposed=True), bu.cuda()
    wd, bd = pack_weight(wt, BF, transposed=True, flip=True), b.cuda()
    common = dict(dtype=BF, ksize=3, pad=2, N=n, H=H, W=W, Cin=4 * c, Cout=32, src0=x2d, src0_H=H, src0_W=W, src0_C=c,
                  weight=wd, bias=bd, act=_hip.ACT_RELU, out_H=H + 2, out_W=W + 2, out_C=32)
    fused = torch.zeros(n, H + 2, W + 2, 32, dtype=torch.bfloat16, device="cuda")
    run_pipe(src_mode=_hip.SRC_CONCAT_SSR_UP, src1=xsd, src1_H=h, src1_W=w, src1_C=c, up_w=wud, up_b=bud, out=fused, **common)
    assert rel_l2(from_nhwc(fused), ref) < 2e-2
    up = torch.zeros(n, H, W, c, dtype=torch.bfloat16, device="cuda")
    _hip.check(_hip.lib().uncl_upconv2x2(xsd.data_ptr(), None, 0, wud.data_ptr(), bud.data_ptr(), up.data_ptr(), n, h, w, c, c,
                                         _hip.stream_ptr()), "upconv")
    two = torch.zeros_like(fused)
    run_pipe(src_mode=_hip.SRC_CONCAT_SSR, src1=up, src1_H=H, src1_W=W, src1_C=c, out=two, **common)
    assert rel_l2(fused.float().cpu(), two.float().cpu()) < 1e-3
    assert (fused != two).float().mean().item() < 0.05
    # the fused mode is the 32-channel last decoder level only, and the up-sampled map must have the skip's extent
    d = _hip.ConvDesc()
    for k_, v in dict(common, src_mode=_hip.SRC_CONCAT_SSR_UP, src0=x2d.data_ptr(), weight=wd.data_ptr(), bias=bd.data_ptr(),
                      src1=xsd.data_ptr(), src1_H=h - 1, src1_W=w, src1_C=c, up_w=wud.data_ptr(), out=fused.data_ptr()).items():
        setattr(d, k_, v)
    assert _hip.lib().uncl_conv3x3_pipe(C.byref(d), None, _hip.stream_ptr()) != 0


@pytest.mark.parametrize("cin,cout,h,w,n,pad,res", [(256, 256, 12, 12, 5, 0, None), (64, 128, 11, 15, 7, 0, "per_sample"),
                                                    (128, 64, 6, 9, 9, 2, None), (256, 256, 10, 10, 3, 2, "broadcast"),
                                                    (64, 64, 3, 3, 11, 0, None)])
def test_pipe_small_maps_whole_samples_per_tile(cin, cout, h, w, n, pad, res):
    """Maps of <= 256 output pixels run with whole samples per tile (the kernel's FLAT form: several samples share the 256
    accumulator rows, each lane reads its own 3x3 window); odd batch sizes leave the last tile partly empty."""
    x = q(rnd(n, cin, h, w, seed=111), BF)
    b = rnd(cout, seed=113)
    if pad == 0:
        wt = q(rnd(cout, cin, 3, 3, seed=112, scale=0.04), BF)
        y, packed = F.conv2d(x, wt, b), pack_weight(wt, BF)
    else:
        wt = q(rnd(cin, cout, 3, 3, seed=112, scale=0.04), BF)
        y, packed = F.conv_transpose2d(x, wt, b), pack_weight(wt, BF, transposed=True, flip=True)
    ref = F.relu(y)
    ho, wo = ref.shape[2], ref.shape[3]
    kw = {}
    if res == "broadcast":
        r = q(rnd(1, cout, ho, wo, seed=114), BF)
        ref = ref + r
        kw = dict(res=to_nhwc(r, BF), res_batch_stride0=1)
    elif res == "per_sample":
        r = q(rnd(n, cout, ho, wo, seed=114), BF)
        ref = ref + r
        kw = dict(res=to_nhwc(r, BF), res_batch_stride0=0)
    out = torch.full((n + 1, ho, wo, cout), 7.0, dtype=torch.bfloat16, device="cuda")      # one guard sample behind the batch
    run_pipe(dtype=BF, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout, src0=to_nhwc(x, BF),
             src0_H=h, src0_W=w, src0_C=cin, weight=packed, bias=b.cuda(), act=_hip.ACT_RELU, out=out, out_H=ho, out_W=wo,
             out_C=cout, **kw)
    assert rel_l2(from_nhwc(out[:n]), ref) < TOL[BF]
    assert (out[n] == 7.0).all()                                                        # nothing written past the batch


def test_pipe_random_shapes_plain_layers():
    """Seeded sweep over channel counts, odd map sizes (3..44), batch sizes and both paddings: rectangular tiles with ragged
    borders, 8-row tiles, 64-channel tiles and the whole-samples-per-tile form all have to agree with the torch reference and
    leave the sample behind the batch untouched."""
    import random
    rng = random.Random(20240807)
    for it in range(24):
        cin, cout = rng.choice([32, 64, 128, 256]), rng.choice([32, 64, 128, 256])
        pad, h, w, n = rng.choice([0, 2]), rng.randint(3, 44), rng.randint(3, 44), rng.randint(1, 6)
        x, b = q(rnd(n, cin, h, w, seed=300 + it), BF), rnd(cout, seed=400 + it)
        if pad == 0:
            wt = q(rnd(cout, cin, 3, 3, seed=500 + it, scale=0.05), BF)
            y, packed = F.conv2d(x, wt, b), pack_weight(wt, BF)
        else:
            wt = q(rnd(cin, cout, 3, 3, seed=500 + it, scale=0.05), BF)
            y, packed = F.conv_transpose2d(x, wt, b), pack_weight(wt, BF, transposed=True, flip=True)
        ref = F.relu(y)
        ho, wo = ref.shape[2], ref.shape[3]
        out = torch.full((n + 1, ho, wo, cout), 7.0, dtype=torch.bfloat16, device="cuda")
        run_pipe(dtype=BF, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout, src0=to_nhwc(x, BF),
                 src0_H=h, src0_W=w, src0_C=cin, weight=packed, bias=b.cuda(), act=_hip.ACT_RELU, out=out, out_H=ho, out_W=wo,
                 out_C=cout)
        assert rel_l2(from_nhwc(out[:n]), ref) < TOL[BF], (cin, cout, pad, h, w, n)
        assert (out[n] == 7.0).all(), (cin, cout, pad, h, w, n)


def test_pipe_random_shapes_concat_loaders():
    """Seeded sweep of the skip-concat loaders (square / square-root operator and plain two-way concat): channel counts, odd
    sizes, both paddings, the up-sampled operand 0..2 pixels smaller than the skip (replicate padding, unet_parts.py:292-298)."""
    import random
    rng = random.Random(19690720)
    for it in range(16):
        c, cout = rng.choice([32, 64, 128]), rng.choice([32, 64, 128])
        pad, h, w, n = rng.choice([0, 2, 2]), rng.randint(5, 40), rng.randint(5, 40), rng.randint(1, 4)
        dy, dx, mode = rng.choice([0, 1, 2]), rng.choice([0, 1, 2]), rng.choice(["ssr", "ssr", "cat2"])
        x2, x1 = q(rnd(n, c, h, w, seed=600 + it).abs(), BF), q(rnd(n, c, h - dy, w - dx, seed=700 + it), BF)
        x1p = F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2), mode="replicate")
        groups = 4 if mode == "ssr" else 2
        cat = torch.cat([x2, x1p, q(x2 ** 2, BF), q((x2 + 1e-8) ** 0.5, BF)], 1) if mode == "ssr" else torch.cat([x2, x1p], 1)
        b = rnd(cout, seed=800 + it)
        if pad == 0:
            wt = q(rnd(cout, groups * c, 3, 3, seed=900 + it, scale=0.03), BF)
            y, packed = F.conv2d(cat, wt, b), pack_weight(wt, BF)
        else:
            wt = q(rnd(groups * c, cout, 3, 3, seed=900 + it, scale=0.03), BF)
            y, packed = F.conv_transpose2d(cat, wt, b), pack_weight(wt, BF, transposed=True, flip=True)
        ref = F.relu(y)
        ho, wo = ref.shape[2], ref.shape[3]
        out = torch.full((n + 1, ho, wo, cout), 7.0, dtype=torch.bfloat16, device="cuda")
        run_pipe(dtype=BF, ksize=3, pad=pad, src_mode=_hip.SRC_CONCAT_SSR if mode == "ssr" else _hip.SRC_CONCAT2, N=n, H=h, W=w,
                 Cin=groups * c, Cout=cout, src0=to_nhwc(x2, BF), src0_H=h, src0_W=w, src0_C=c, src1=to_nhwc(x1, BF),
                 src1_H=h - dy, src1_W=w - dx, src1_C=c, weight=packed, bias=b.cuda(), act=_hip.ACT_RELU, out=out, out_H=ho,
                 out_W=wo, out_C=cout)
        assert rel_l2(from_nhwc(out[:n]), ref) < 2e-2, (mode, c, cout, pad, h, w, n, dy, dx)
        assert (out[n] == 7.0).all(), (mode, c, cout, pad, h, w, n, dy, dx)


def test_pipe_broadcast_residual_and_skip_store():
    cin, cout, h = 256, 256, 10
    x, wt, b = q(rnd(3, cin, h, h, seed=40), BF), q(rnd(cin, cout, 3, 3, seed=41, scale=0.05), BF), rnd(cout, seed=42)
    pe = q(rnd(1, cout, h + 2, h + 2, seed=43), BF)
    ref = F.relu(F.conv_transpose2d(x, wt, b)) + pe
    out = torch.zeros(3, h + 2, h + 2, cout, dtype=torch.bfloat16, device="cuda")
    run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=3, H=h, W=h, Cin=cin, Cout=cout,
             src0=to_nhwc(x, BF), src0_H=h, src0_W=h, src0_C=cin,
             weight=pack_weight(wt, BF, transposed=True, flip=True), bias=b.cuda(), act=_hip.ACT_RELU,
             res=to_nhwc(pe, BF), res_batch_stride0=1, out=out, out_H=h + 2, out_W=h + 2, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[BF]


@pytest.mark.parametrize("c,h,w,n", [(32, 9, 13, 3), (64, 61, 61, 2), (128, 28, 28, 2), (256, 12, 12, 3)])
def test_upconv2x2_bf16(c, h, w, n):
    x, wt, b = q(rnd(n, c, h, w, seed=51), BF), q(rnd(c, c, 2, 2, seed=52, scale=0.1), BF), rnd(c, seed=53)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    out = torch.zeros(n, 2 * h, 2 * w, c, dtype=torch.bfloat16, device="cuda")
    xd, wd, bd = to_nhwc(x, BF), pack_weight(wt, BF, transposed=True), b.cuda()      # keep the operands alive
    _hip.check(_hip.lib().uncl_upconv2x2(xd.data_ptr(), None, 0, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, w,
                                         c, c, _hip.stream_ptr()), "upconv")
    torch.cuda.synchronize()
    assert rel_l2(from_nhwc(out), ref) < TOL[BF]


def test_bad_arguments_are_refused():
    d = _hip.ConvDesc()
    import ctypes as C
    assert _hip.lib().uncl_conv_igemm(C.byref(d), None) == -1
    d.dtype, d.ksize, d.Cin, d.Cout = _hip.BF16, 3, 24, 32     # Cin not a multiple of the 32-channel K-chunk
    assert _hip.lib().uncl_conv_igemm(C.byref(d), None) == -1


@pytest.mark.parametrize("code,h,w", [(_hip.F32, 37, 45), (BF, 37, 45), (BF, 256, 256), (BF, 10, 70)])
def test_first_layer_one_channel_conv(code, h, w):
    """inc.conv.conv (Cin = 1): fp32 VALU kernel is the parity path; the bf16 path runs on the matrix cores with the fp32
    input split into a bf16 head + tail, so only the weights and the stored result are bf16-rounded."""
    n = 3
    x = rnd(n, 1, h, w, seed=71).abs()
    wt, b = rnd(32, 1, 3, 3, seed=72, scale=0.3), rnd(32, seed=73)
    out = torch.zeros(n, h - 2, w - 2, 32, dtype=_hip.torch_dtype(code), device="cuda")
    xd, wd, bd = x.reshape(n, h, w).cuda().contiguous(), wt.cuda().contiguous(), b.cuda()
    _hip.check(_hip.lib().uncl_conv_in_c1(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), code, n, h, w, 32,
                                          _hip.ACT_RELU, _hip.stream_ptr()), "uncl_conv_in_c1")
    torch.cuda.synchronize()
    if code == _hip.F32:
        assert rel_l2(from_nhwc(out), F.relu(F.conv2d(x, wt, b))) < 1e-6
    else:
        ref = F.relu(F.conv2d(x, q(wt, BF), b))                     # bf16 weights, fp32-class input
        assert rel_l2(from_nhwc(out), ref) < 3e-3                   # one bf16 rounding of the stored activations
        # the input is NOT rounded to bf16: against a reference that does round it the error is visibly larger
        assert rel_l2(from_nhwc(out), ref) < 0.7 * rel_l2(F.relu(F.conv2d(q(x, BF), q(wt, BF), b)), ref) + 3e-3


@pytest.mark.parametrize("h,w", [(40, 70), (256, 256)])
def test_pipe_first_layer_recomputed_in_the_loader(h, w):
    """UNCL_SRC_IMAGE1: inc.conv.conv (1 -> 32) is rebuilt from the fp32 image inside inc.conv.conv1's loader; the result
    must equal running the two layers one after the other (with the first layer's output rounded to bf16 in between)."""
    n = 2
    x = rnd(n, 1, h, w, seed=81).abs()
    w0, b0 = rnd(32, 1, 3, 3, seed=82, scale=0.3), rnd(32, seed=83)
    w1, b1 = q(rnd(32, 32, 3, 3, seed=84, scale=0.06), BF), rnd(32, seed=85)
    mid = q(F.relu(F.conv2d(x, q(w0, BF), b0)), BF)
    ref = F.relu(F.conv2d(mid, w1, b1))
    ho, wo = h - 4, w - 4
    out = torch.zeros(n, ho, wo, 32, dtype=torch.bfloat16, device="cuda")
    pooled = torch.zeros(n, ho // 2, wo // 2, 32, dtype=torch.bfloat16, device="cuda")
    run_pipe(pool_out=pooled, dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_IMAGE1, N=n, H=h - 2, W=w - 2, Cin=32, Cout=32,
             src0=x.reshape(n, h, w).cuda().contiguous(), src0_H=h, src0_W=w, src0_C=1, pre_w=w0.cuda().contiguous(),
             pre_b=b0.cuda(), weight=pack_weight(w1, BF), bias=b1.cuda(), act=_hip.ACT_RELU, out=out, out_H=ho, out_W=wo,
             out_C=32)
    assert rel_l2(from_nhwc(out), ref) < TOL[BF]
    assert rel_l2(from_nhwc(pooled), F.max_pool2d(q(ref, BF), 2)) < TOL[BF]
    # wrong shapes for this mode are refused
    d = _hip.ConvDesc()
    for k_, v in dict(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_IMAGE1, N=n, H=h - 2, W=w - 2, Cin=32, Cout=32,
                      src0=out.data_ptr(), src0_H=h, src0_W=w, src0_C=1, weight=out.data_ptr(), out=out.data_ptr()).items():
        setattr(d, k_, v)
    assert _hip.lib().uncl_conv3x3_pipe(C.byref(d), None, _hip.stream_ptr()) != 0


@pytest.mark.parametrize("cin,cout,h,w,n,act", [(64, 64, 26, 30, 2, "relu"), (128, 128, 28, 28, 3, "lrelu"), (96, 64, 9, 41, 1, "none"),
                                                (256, 256, 10, 10, 4, "relu")])
def test_pipe_transposed_plain_layers_zero_borders(cin, cout, h, w, n, act):
    """Plain-source transposed (pad-2) layers with 64-channel output tiles, every activation the epilogue supports."""
    x, wt, b = q(rnd(n, cin, h, w, seed=91), BF), q(rnd(cin, cout, 3, 3, seed=92, scale=0.05), BF), rnd(cout, seed=93)
    y = F.conv_transpose2d(x, wt, b)
    code = {"relu": _hip.ACT_RELU, "lrelu": _hip.ACT_LRELU, "none": _hip.ACT_NONE}[act]
    ref = {"relu": F.relu(y), "lrelu": F.leaky_relu(y, 0.2), "none": y}[act]
    out = torch.zeros(n, h + 2, w + 2, cout, dtype=torch.bfloat16, device="cuda")
    run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout,
             src0=to_nhwc(x, BF), src0_H=h, src0_W=w, src0_C=cin, weight=pack_weight(wt, BF, transposed=True, flip=True),
             bias=b.cuda(), act=code, out=out, out_H=h + 2, out_W=w + 2, out_C=cout)
    assert rel_l2(from_nhwc(out), ref) < TOL[BF]


# ---- producer / consumer kernel (csrc/conv3x3_pc.hip) against the four-wave kernel: same tiles, same MFMA order per
# ---- accumulator, so every stored element must be IDENTICAL (the torch comparisons above already run through it by default)
def _both_structures(fn, pc=2):
    lib = _hip.lib()
    old = lib.uncl_conv3x3_set_pc(0)
    try:
        ref = fn()
        lib.uncl_conv3x3_set_pc(pc)
        got = fn()
    finally:
        lib.uncl_conv3x3_set_pc(old)
    return ref, got


@pytest.mark.parametrize("cin,cout,h,w,n,pad,pool,act", [
    (64, 64, 124, 124, 3, 0, True, _hip.ACT_RELU),      # down_path.0 second conv: two chunks, pooled copy
    (64, 128, 61, 61, 2, 0, False, _hip.ACT_RELU),      # two cout tiles, ragged 59 x 59 output
    (128, 128, 59, 59, 2, 0, True, _hip.ACT_LRELU),     # odd output 57: floor pooling
    (256, 256, 26, 26, 5, 0, True, _hip.ACT_RELU),      # eight chunks, four cout tiles
    (128, 128, 26, 26, 3, 2, False, _hip.ACT_RELU),     # transposed (pad 2): zero borders
    (64, 32, 33, 70, 2, 2, False, _hip.ACT_NONE),       # 32-channel tiles (16 x 32), plain source
    (96, 64, 17, 40, 1, 0, False, _hip.ACT_RELU),       # three chunks: odd chunk count flips the stage parity per tile
])
def test_pc_plain_layers_bitwise(cin, cout, h, w, n, pad, pool, act):
    x, wt, b = q(rnd(n, cin, h, w, seed=201), BF), q(rnd(cout, cin, 3, 3, seed=202, scale=0.1), BF), rnd(cout, seed=203)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    xd, wd, bd = to_nhwc(x, BF), pack_weight(wt, BF), b.cuda()

    def run():
        out = torch.zeros(n, ho, wo, cout, dtype=torch.bfloat16, device="cuda")
        pl = torch.zeros(n, ho // 2, wo // 2, cout, dtype=torch.bfloat16, device="cuda") if pool else None
        run_pipe(pool_out=pl, dtype=BF, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout, src0=xd,
                 src0_H=h, src0_W=w, src0_C=cin, weight=wd, bias=bd, act=act, out=out, out_H=ho, out_W=wo, out_C=cout)
        return out, pl

    (ro, rp), (go, gp) = _both_structures(run)
    assert torch.equal(ro, go)
    if pool:
        assert torch.equal(rp, gp)
    ref = F.conv2d(F.pad(x, (pad,) * 4), wt, b)
    ref = {_hip.ACT_RELU: F.relu, _hip.ACT_LRELU: lambda t: F.leaky_relu(t, 0.2), _hip.ACT_NONE: lambda t: t}[act](ref)
    assert rel_l2(from_nhwc(go), ref) < TOL[BF]


@pytest.mark.parametrize("cin,h,w,n,skip,act1", [(32, 30, 45, 2, False, _hip.ACT_SIGMOID), (32, 126, 126, 2, True, _hip.ACT_SIGMOID),
                                                 (64, 40, 70, 2, True, _hip.ACT_TANH), (32, 17, 33, 3, False, _hip.ACT_NONE)])
def test_pc_fused_outc_tail_bitwise(cin, h, w, n, skip, act1):
    """The fused 1x1 tail (outconv + last activation) under both kernel structures, with the 32-channel map stored (training) and
    skipped (inference).  The 32-channel map is bit-identical; the 1-channel map is the same 32 products of the same ROUNDED
    channels summed in a different order (four-wave kernel, the default for this layer: one channel-order chain; producer /
    consumer kernel, set_pc(3): 16 products per lane, then the two half-waves; same-box A/B 0.325 vs 0.330 ms, so the default stays) -- equal to fp32 reassociation, 1e-6, and both within the stated bf16
    tolerance of the torch reference."""
    x, wt, b = q(rnd(n, cin, h, w, seed=261), BF), q(rnd(cin, 32, 3, 3, seed=262, scale=0.1), BF), rnd(32, seed=263)
    w1, b1 = rnd(32, seed=264).cuda(), rnd(1, seed=265).cuda()
    xd, wd, bd = to_nhwc(x, BF), pack_weight(wt, BF, transposed=True, flip=True), b.cuda()

    def run():
        out = None if skip else torch.zeros(n, h + 2, w + 2, 32, dtype=torch.bfloat16, device="cuda")
        out1 = torch.zeros(n, h + 2, w + 2, dtype=torch.float32, device="cuda")
        run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=32, src0=xd, src0_H=h, src0_W=w,
                 src0_C=cin, weight=wd, bias=bd, act=_hip.ACT_RELU, out=out, out_H=h + 2, out_W=w + 2, out_C=32, out1_w=w1, out1_b=b1,
                 out1_act=act1, out1=out1, skip_main_store=1 if skip else 0)
        return out, out1

    (ro, r1), (go, g1) = _both_structures(run, pc=3)       # 3: the producer / consumer kernel takes the fused tail as well
    assert (r1 - g1).abs().max().item() < 2e-6 * max(1.0, r1.abs().max().item())
    if not skip:
        assert torch.equal(ro, go)
    up = F.relu(F.conv_transpose2d(x, wt, b))
    ref1 = F.conv2d(q(up, BF), w1.cpu().reshape(1, 32, 1, 1), b1.cpu())
    ref1 = {_hip.ACT_SIGMOID: torch.sigmoid, _hip.ACT_TANH: torch.tanh, _hip.ACT_NONE: lambda t: t}[act1](ref1)
    assert rel_l2(g1.cpu().unsqueeze(1), ref1) < TOL[BF]


@pytest.mark.parametrize("code", [_hip.BF16, _hip.F16])
@pytest.mark.parametrize("h,w,n,act1", [(45, 61, 3, _hip.ACT_SIGMOID),        # 47 x 63 outputs: ragged in both directions
                                        (14, 30, 5, _hip.ACT_SIGMOID),        # 16 x 32: one tile per sample, exact
                                        (15, 31, 4, _hip.ACT_SIGMOID),        # 17 x 33: one row / one column past a tile border
                                        (254, 254, 1, _hip.ACT_SIGMOID),      # the generator's own last layer
                                        (30, 40, 2, _hip.ACT_TANH), (30, 40, 2, _hip.ACT_NONE), (30, 40, 2, _hip.ACT_MSIG)])
def test_last_layer_one_channel_form_two_accumulator_sets(code, h, w, n, act1):
    """Round 6: inference's last layer (32 -> 32 transposed 3x3 + ReLU + outconv + last activation, only the one-channel map stored) on
    the producer / consumer structure with the 1x1 tail computed from the accumulators (`O1C`: two accumulator sets for the sigmoid,
    one for the other activations) against the four-wave `O1D` form (same rounded channels, the three partial dot products summed
    in the same order: 2e-6) and against torch; ragged extents, every pixel of the NaN-initialised output written, an odd tile count
    per workgroup (the two sets alternate per tile: last tile in either set)."""
    lib = _hip.lib()
    tdt = _hip.torch_dtype(code)
    x, wt, b = q(rnd(n, 32, h, w, seed=271), code), q(rnd(32, 32, 3, 3, seed=272, scale=0.1), code), rnd(32, seed=273)
    w1, b1 = rnd(32, seed=274).cuda(), rnd(1, seed=275).cuda()
    xd, wd, bd = to_nhwc(x, code), pack_weight(wt, code, transposed=True, flip=True), b.cuda()

    def run():
        out1 = torch.full((n, h + 2, w + 2), float("nan"), dtype=torch.float32, device="cuda")
        run_pipe(dtype=code, ksize=3, pad=2, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=32, Cout=32, src0=xd, src0_H=h, src0_W=w,
                 src0_C=32, weight=wd, bias=bd, act=_hip.ACT_RELU, out=None, out_H=h + 2, out_W=w + 2, out_C=32, out1_w=w1, out1_b=b1,
                 out1_act=act1, out1=out1, skip_main_store=1)
        return out1

    old = lib.uncl_conv3x3_set_pc(0)
    try:
        r1 = run()                                  # four-wave kernel
        lib.uncl_conv3x3_set_pc(2)
        g1 = run()                                  # the default: O1C
        g2 = run()
    finally:
        lib.uncl_conv3x3_set_pc(old)
    assert torch.isfinite(g1).all() and torch.isfinite(r1).all()
    assert torch.equal(g1, g2)                      # run to run
    assert (r1 - g1).abs().max().item() < 2e-6 * max(1.0, r1.abs().max().item())
    up = F.relu(F.conv_transpose2d(x, wt, b))
    ref1 = F.conv2d(q(up, code), w1.cpu().reshape(1, 32, 1, 1), b1.cpu())
    ref1 = {_hip.ACT_SIGMOID: torch.sigmoid, _hip.ACT_TANH: torch.tanh, _hip.ACT_NONE: lambda t: t,
            _hip.ACT_MSIG: lambda t: torch.sigmoid(3.0 * t)}[act1](ref1)
    assert rel_l2(g1.cpu().unsqueeze(1), ref1) < (1.5e-2 if code == _hip.BF16 else 2e-3)


@pytest.mark.parametrize("kind,c,cout,h,n,pool", [("plain", 64, 64, 124, 24, True),      # 24 x 4 x 8 = 768 tiles of 16 x 32 x 64
                                                   ("plain", 128, 128, 59, 48, True),    # 57 x 57 output, two cout tiles
                                                   ("ssr", 128, 64, 57, 96, False)])     # concat source, 59 x 59 output
def test_pc_tall_64_channel_tiles_bitwise(kind, c, cout, h, n, pool):
    """Launches large enough for the 16-row 64-channel tiles (>= 768 of them) under both kernel structures."""
    if kind == "plain":
        x, wt, b = q(rnd(n, c, h, h, seed=251), BF), q(rnd(cout, c, 3, 3, seed=252, scale=0.1), BF), rnd(cout, seed=253)
        ho = h - 2
        xd, wd, bd = to_nhwc(x, BF), pack_weight(wt, BF), b.cuda()

        def run():
            out = torch.zeros(n, ho, ho, cout, dtype=torch.bfloat16, device="cuda")
            pl = torch.zeros(n, ho // 2, ho // 2, cout, dtype=torch.bfloat16, device="cuda") if pool else None
            run_pipe(pool_out=pl, dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=h, Cin=c, Cout=cout, src0=xd,
                     src0_H=h, src0_W=h, src0_C=c, weight=wd, bias=bd, act=_hip.ACT_RELU, out=out, out_H=ho, out_W=ho, out_C=cout)
            return out, pl
    else:
        x2 = q(rnd(n, c, h, h, seed=254).abs(), BF)
        x1 = q(rnd(n, c, h - 1, h - 1, seed=255), BF)
        wt, b = q(rnd(4 * c, cout, 3, 3, seed=256, scale=0.05), BF), rnd(cout, seed=257)
        x2d, x1d, wd, bd = to_nhwc(x2, BF), to_nhwc(x1, BF), pack_weight(wt, BF, transposed=True, flip=True), b.cuda()

        def run():
            out = torch.zeros(n, h + 2, h + 2, cout, dtype=torch.bfloat16, device="cuda")
            run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=n, H=h, W=h, Cin=4 * c, Cout=cout, src0=x2d, src0_H=h,
                     src0_W=h, src0_C=c, src1=x1d, src1_H=h - 1, src1_W=h - 1, src1_C=c, weight=wd, bias=bd, act=_hip.ACT_RELU,
                     out=out, out_H=h + 2, out_W=h + 2, out_C=cout)
            return out, None

    (ro, rp), (go, gp) = _both_structures(run)
    assert torch.equal(ro, go)
    if pool:
        assert torch.equal(rp, gp)


@pytest.mark.parametrize("h,w,n,pool", [(40, 70, 2, True), (256, 256, 2, True), (37, 51, 3, False)])
def test_pc_first_layer_fused_bitwise(h, w, n, pool):
    """UNCL_SRC_IMAGE1 under both kernel structures (producer / consumer: the staging waves rebuild the halo tile from image
    patches parked in LDS one step earlier), with the pooled copy beside 32-channel tiles."""
    x = rnd(n, 1, h, w, seed=231).abs().reshape(n, h, w).cuda().contiguous()
    w0, b0 = rnd(32, 1, 3, 3, seed=232, scale=0.3).cuda().contiguous(), rnd(32, seed=233).cuda()
    w1, b1 = pack_weight(q(rnd(32, 32, 3, 3, seed=234, scale=0.06), BF), BF), rnd(32, seed=235).cuda()
    ho, wo = h - 4, w - 4

    def run():
        out = torch.zeros(n, ho, wo, 32, dtype=torch.bfloat16, device="cuda")
        pl = torch.zeros(n, ho // 2, wo // 2, 32, dtype=torch.bfloat16, device="cuda") if pool else None
        run_pipe(pool_out=pl, dtype=BF, ksize=3, pad=0, src_mode=_hip.SRC_IMAGE1, N=n, H=h - 2, W=w - 2, Cin=32, Cout=32, src0=x,
                 src0_H=h, src0_W=w, src0_C=1, pre_w=w0, pre_b=b0, weight=w1, bias=b1, act=_hip.ACT_RELU, out=out, out_H=ho,
                 out_W=wo, out_C=32)
        return out, pl

    (ro, rp), (go, gp) = _both_structures(run)
    assert torch.equal(ro, go)
    if pool:
        assert torch.equal(rp, gp)


@pytest.mark.parametrize("cin,cout,h,w,n,pad,pool", [(32, 32, 45, 70, 2, 0, True), (32, 32, 126, 126, 2, 2, False),
                                                     (32, 64, 63, 63, 2, 0, False), (64, 32, 40, 44, 2, 0, True)])
def test_pc_single_chunk_and_pooled_32_bitwise(cin, cout, h, w, n, pad, pool):
    x, wt, b = q(rnd(n, cin, h, w, seed=241), BF), q(rnd(cout, cin, 3, 3, seed=242, scale=0.1), BF), rnd(cout, seed=243)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    xd, wd, bd = to_nhwc(x, BF), pack_weight(wt, BF), b.cuda()

    def run():
        out = torch.zeros(n, ho, wo, cout, dtype=torch.bfloat16, device="cuda")
        pl = torch.zeros(n, ho // 2, wo // 2, cout, dtype=torch.bfloat16, device="cuda") if pool else None
        run_pipe(pool_out=pl, dtype=BF, ksize=3, pad=pad, src_mode=_hip.SRC_PLAIN, N=n, H=h, W=w, Cin=cin, Cout=cout, src0=xd,
                 src0_H=h, src0_W=w, src0_C=cin, weight=wd, bias=bd, act=_hip.ACT_RELU, out=out, out_H=ho, out_W=wo, out_C=cout)
        return out, pl

    (ro, rp), (go, gp) = _both_structures(run)
    assert torch.equal(ro, go)
    if pool:
        assert torch.equal(rp, gp)
    assert rel_l2(from_nhwc(go), F.relu(F.conv2d(F.pad(x, (pad,) * 4), wt, b))) < TOL[BF]


@pytest.mark.parametrize("c,cout,h,w,n,short", [(32, 32, 37, 45, 2, 1), (64, 32, 124, 124, 2, 0), (128, 64, 57, 57, 2, 1),
                                                (256, 128, 24, 24, 3, 0)])
def test_pc_concat_ssr_bitwise(c, cout, h, w, n, short):
    x2 = q(rnd(n, c, h, w, seed=211).abs(), BF)
    x1 = q(rnd(n, c, h - short, w - short, seed=212), BF)
    wt, b = q(rnd(4 * c, cout, 3, 3, seed=213, scale=0.05), BF), rnd(cout, seed=214)
    x2d, x1d, wd, bd = to_nhwc(x2, BF), to_nhwc(x1, BF), pack_weight(wt, BF, transposed=True, flip=True), b.cuda()

    def run():
        out = torch.zeros(n, h + 2, w + 2, cout, dtype=torch.bfloat16, device="cuda")
        run_pipe(dtype=BF, ksize=3, pad=2, src_mode=_hip.SRC_CONCAT_SSR, N=n, H=h, W=w, Cin=4 * c, Cout=cout, src0=x2d, src0_H=h,
                 src0_W=w, src0_C=c, src1=x1d, src1_H=h - short, src1_W=w - short, src1_C=c, weight=wd, bias=bd,
                 act=_hip.ACT_RELU, out=out, out_H=h + 2, out_W=w + 2, out_C=cout)
        return out

    ro, go = _both_structures(run)
    assert torch.equal(ro, go)
    cat = torch.cat([x2, F.pad(x1, (0, short, 0, short), mode="replicate"), x2 ** 2, (x2 + 1e-8) ** 0.5], 1)
    assert rel_l2(from_nhwc(go), F.relu(F.conv_transpose2d(cat, wt, b))) < 2e-2


@pytest.mark.parametrize("h,w,n", [(9, 13, 3), (23, 40, 2), (126, 126, 2)])
def test_pc_fused_upconv_bitwise(h, w, n):
    c = 32
    H, W = 2 * h, 2 * w
    x2d = to_nhwc(q(rnd(n, c, H, W, seed=221).abs(), BF), BF)
    xsd = to_nhwc(q(rnd(n, c, h, w, seed=222), BF), BF)
    wud, bud = pack_weight(q(rnd(c, c, 2, 2, seed=223, scale=0.1), BF), BF, transposed=True), rnd(c, seed=224).cuda()
    wd, bd = pack_weight(q(rnd(4 * c, 32, 3, 3, seed=225, scale=0.05), BF), BF, transposed=True, flip=True), rnd(32, seed=226).cuda()

    def run():
        out = torch.zeros(n, H + 2, W + 2, 32, dtype=torch.bfloat16, device="cuda")
        run_pipe(dtype=BF, ksize=3, pad=2, N=n, H=H, W=W, Cin=4 * c, Cout=32, src0=x2d, src0_H=H, src0_W=W, src0_C=c, weight=wd,
                 bias=bd, act=_hip.ACT_RELU, out_H=H + 2, out_W=W + 2, out_C=32, src_mode=_hip.SRC_CONCAT_SSR_UP, src1=xsd,
                 src1_H=h, src1_W=w, src1_C=c, up_w=wud, up_b=bud, out=out)
        return out

    ro, go = _both_structures(run)
    assert torch.equal(ro, go)


@pytest.mark.parametrize("gc,cd,gh,pad_d,n,accumulate,use_mask", [(64, 64, 122, 2, 2, 0, True), (128, 64, 59, 2, 2, 1, True),
                                                                  (32, 128, 254, 0, 1, 0, False), (256, 256, 24, 2, 2, 1, False)])
def test_pc_dgrad_store_bitwise(gc, cd, gh, pad_d, n, accumulate, use_mask):
    """Gradient mode of the epilogue (ReLU mask of the producing layer, accumulation into an existing gradient)."""
    oh = gh + 2 * pad_d - 2
    gy = to_nhwc(q(rnd(n, gc, gh, gh, seed=231), BF), BF)
    wd = pack_weight(q(rnd(cd, gc, 3, 3, seed=232, scale=0.1), BF), BF)
    mask = to_nhwc(q(rnd(n, cd, oh, oh, seed=233), BF), BF)
    init = to_nhwc(q(rnd(n, cd, oh, oh, seed=234), BF), BF)

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

    ro, go = _both_structures(run)
    assert torch.equal(ro, go)
