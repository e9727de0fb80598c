"""The fused last decoder stage (csrc/conv3x3_pc.hip, TAIL form; uncl_conv_desc.tail_w): concat-ssr with the 2x2 up-conv recomputed
in the loader -> ConvTranspose2d 3x3 -> ReLU -> ConvTranspose2d 3x3 -> ReLU -> outconv + last activation as ONE launch whose two
32-channel maps stay in LDS (unet_parts.py:149-162, 269, 288, 319-322, 338-345; Unet_singleFrame.py:200-209).

Gates: (1) against a torch fp32 evaluation of the same rounded operands, element by element; (2) bit for bit against the two-launch
path of the same library (whose rounding points it shares); (3) every output element is written (NaN-poisoned buffer); (4) the whole
generator, fused on / off, on batch sizes that put one, a few and many (strip, row tile) steps on a workgroup -- shares that start
inside a strip re-run the tile above them for its last two rows."""
import pytest
import torch
import torch.nn.functional as F

from hip_util import assert_elementwise, pack_weight, rel_l2, run_pipe
from uncltmo_amd import _hip, synth
from uncltmo_amd.generator import UNet

pytestmark = pytest.mark.gpu


def _rt(t, dt):
    return t.to(dt).float()


def _stage_inputs(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    skip = torch.rand(n, 32, h, w, generator=g) * (torch.rand(n, 32, h, w, generator=g) > 0.3)      # ReLU-like: exact zeros
    src1 = torch.rand(n, 32, h // 2, w // 2, generator=g) * (torch.rand(n, 32, h // 2, w // 2, generator=g) > 0.3)
    up_w = (torch.rand(32, 32, 2, 2, generator=g) - 0.5) * 0.35
    up_b = (torch.rand(32, generator=g) - 0.5) * 0.2
    w0 = (torch.rand(128, 32, 3, 3, generator=g) - 0.5) * 0.12
    b0 = (torch.rand(32, generator=g) - 0.5) * 0.2
    w1 = (torch.rand(32, 32, 3, 3, generator=g) - 0.5) * 0.25
    b1 = (torch.rand(32, generator=g) - 0.5) * 0.2
    ow = (torch.rand(1, 32, 1, 1, generator=g) - 0.5) * 0.5
    ob = (torch.rand(1, generator=g) - 0.5) * 0.2
    return skip, src1, up_w, up_b, w0, b0, w1, b1, ow, ob


def _reference(dt, skip, src1, up_w, up_b, w0, b0, w1, b1, ow, ob):
    """fp32 arithmetic on the operands the kernel sees, rounded where the kernel rounds (every staged / stored 16-bit value)."""
    x2 = _rt(skip, dt).double()
    up = _rt(F.conv_transpose2d(_rt(src1, dt).double(), _rt(up_w, dt).double(), up_b.double(), stride=2).float(), dt).double()
    cat = torch.cat([x2, up, _rt((x2 * x2).float(), dt).double(), _rt(torch.sqrt(x2.float() + 1e-8), dt).double()], 1)
    mid_f = F.relu(F.conv_transpose2d(cat, _rt(w0, dt).double(), b0.double()))
    mid = _rt(mid_f.float(), dt).double()
    t_f = F.relu(F.conv_transpose2d(mid, _rt(w1, dt).double(), b1.double()))
    t = _rt(t_f.float(), dt).double()
    return torch.sigmoid(F.conv2d(t, ow.double(), ob.double())).float(), mid_f.float()


def _run_stage(code, dt, n, h, w, ins, fused, poison=True):
    skip, src1, up_w, up_b, w0, b0, w1, b1, ow, ob = ins
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    xs, x1 = nhwc(skip), nhwc(src1)
    upw = pack_weight(up_w, code, transposed=True, flip=False)
    pw0 = pack_weight(w0, code, transposed=True, flip=True)
    pw1 = pack_weight(w1, code, transposed=True, flip=True)
    dev = lambda t: t.float().contiguous().cuda()
    upb, b0d, b1d, owd, obd = dev(up_b), dev(b0), dev(b1), dev(ow.reshape(32)), dev(ob)
    out = torch.full((n, h + 4, w + 4), float("nan"), device="cuda") if poison else torch.empty(n, h + 4, w + 4, device="cuda")
    common = dict(dtype=code, ksize=3, pad=2, N=n, H=h, W=w, src0=xs, src0_H=h, src0_W=w, src0_C=32)
    if fused:
        run_pipe(None, **common, src_mode=5, Cin=128, Cout=32, src1=x1, src1_H=h // 2, src1_W=w // 2, src1_C=32, weight=pw0,
                 bias=b0d, act=_hip.ACT_RELU, up_w=upw, up_b=upb, out=None, out_H=h + 2, out_W=w + 2, out_C=32, out1_w=owd,
                 out1_b=obd, out1_act=_hip.ACT_SIGMOID, out1=out, skip_main_store=1, tail_w=pw1, tail_b=b1d)
        return out, None
    mid = torch.full((n, h + 2, w + 2, 32), float("nan"), dtype=dt, device="cuda")
    run_pipe(None, **common, src_mode=5, Cin=128, Cout=32, src1=x1, src1_H=h // 2, src1_W=w // 2, src1_C=32, weight=pw0, bias=b0d,
             act=_hip.ACT_RELU, up_w=upw, up_b=upb, out=mid, out_H=h + 2, out_W=w + 2, out_C=32)
    old = _hip.lib().uncl_conv3x3_set_pc(3)           # the fused 1x1 tail in the producer / consumer epilogue: the same arithmetic
    try:
        run_pipe(None, dtype=code, ksize=3, pad=2, N=n, H=h + 2, W=w + 2, src0=mid, src0_H=h + 2, src0_W=w + 2, src0_C=32,
                 src_mode=0, Cin=32, Cout=32, weight=pw1, bias=b1d, act=_hip.ACT_RELU, out=None, out_H=h + 4, out_W=w + 4,
                 out_C=32, out1_w=owd, out1_b=obd, out1_act=_hip.ACT_SIGMOID, out1=out, skip_main_store=1)
    finally:
        _hip.lib().uncl_conv3x3_set_pc(old)
    return out, mid


@pytest.mark.parametrize("dtype,n,h,w", [("bf16", 2, 252, 252), ("bf16", 3, 40, 44), ("bf16", 1, 28, 92), ("bf16", 5, 64, 60),
                                         ("fp16", 2, 40, 44), ("fp16", 1, 252, 252)])
def test_fused_stage_matches_reference_and_two_launches(dtype, n, h, w):
    code = _hip.dtype_code(dtype)
    dt = _hip.torch_dtype(code)
    ins = _stage_inputs(n, h, w, 1234 + h)
    fused, _ = _run_stage(code, dt, n, h, w, ins, True)
    assert torch.isfinite(fused).all(), "an element of the poisoned output was not written"
    two, mid = _run_stage(code, dt, n, h, w, ins, False)
    assert torch.isfinite(two).all() and torch.isfinite(mid.float()).all()
    assert torch.equal(fused, two), "fused stage != two launches: max |d| %.3e" % (fused - two).abs().max().item()
    ref, mid_ref = _reference(dt, *ins)
    # the intermediate map of the two-launch path, element by element
    assert_elementwise(mid.float().cpu().permute(0, 3, 1, 2), mid_ref, dtype, "intermediate map")
    # the result: a few flipped roundings of the two intermediate maps (fp32 vs fp64 accumulation) move the logit by ~1e-3
    err = (fused.cpu().unsqueeze(1) - ref).abs()
    assert err.max().item() < (6e-3 if dtype == "bf16" else 1e-3), err.max().item()
    assert rel_l2(fused.cpu().unsqueeze(1), ref) < (1.5e-3 if dtype == "bf16" else 2.5e-4)


def _make_g(dtype):
    g = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0,
             compute_dtype=dtype)
    synth.fill_state_dict(g, "g0")
    return g.cuda().eval()


@pytest.mark.parametrize("dtype,n", [("bf16", 1), ("bf16", 3), ("bf16", 37), ("bf16", 200), ("fp16", 3)])
def test_generator_inference_fused_tail_is_bitwise_the_two_launch_path(dtype, n):
    lib = _hip.lib()
    net = _make_g(dtype)
    x = synth.hdr_frames(n, 256, 256, salt="tail%d" % n).cuda()
    old_t = lib.uncl_gen_set_fused_tail(1)
    old_p = lib.uncl_conv3x3_set_pc(2)
    try:
        y_f = net.infer(x).clone()
        lib.uncl_gen_set_fused_tail(0)
        lib.uncl_conv3x3_set_pc(3)
        y_2 = net.infer(x).clone()
        lib.uncl_conv3x3_set_pc(2)
        y_4 = net.infer(x).clone()             # the shipped two-launch form (four-wave kernel, sequential 1x1 chain)
    finally:
        lib.uncl_gen_set_fused_tail(old_t)
        lib.uncl_conv3x3_set_pc(old_p)
    assert torch.isfinite(y_f).all()
    assert torch.equal(y_f, y_2), (y_f - y_2).abs().max().item()
    # against the four-wave kernel only the outconv's 32-term dot product differs: an fp32 fma chain there, six MFMAs over three
    # 16-bit pieces of the fp32 weights here (exact products, fp32 accumulation in another order)
    assert (y_f - y_4).abs().max().item() < 2e-6
