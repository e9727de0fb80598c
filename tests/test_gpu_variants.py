"""Generator configurations outside the published one that run on the published kernels: skip operators that are sub-sets of
[x2, x1, x2^2, sqrt(x2 + eps)] (unet_parts.py:311-332: original_unet, square, square_root -- zero weights for the members they leave
out), the bilinear decoder path (nn.Upsample + 1x1 convolution, unet_parts.py:256-259 = a 2x2 stride-2 transposed convolution
with one weight on all four taps) and the parameter-free zero-insertion upsampling (`up_mode`, unet_parts.py:284-288 = the same
convolution with an identity on one tap, no bias).  Against the reference's own outputs and parameter gradients (tests/golden/generator_variants.npz)."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from generator_variants import GENERATOR_VARIANTS
from uncltmo_amd import params, state_spec, synth
from uncltmo_amd.generator import UNet

pytestmark = pytest.mark.gpu


def make(op, bil, dtype, upm=0):
    net = UNet(1, 1, "sigmoid", 4, params.get_layer_factor(op), op, 32, bil, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, upm,
               compute_dtype=dtype)
    synth.fill_state_dict(net, "g0")
    return net.cuda()


def inputs():
    return torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("tag,op,bil,upm", GENERATOR_VARIANTS)
def test_forward_matches_reference_golden(golden, tag, op, bil, upm):
    g = golden("generator_variants")
    net = make(op, bil, "fp32", upm).eval()
    assert [(k, ",".join(str(d) for d in v.shape)) for k, v in net.state_dict().items()] == list(zip(g[tag + ".keys"], g[tag + ".shapes"]))
    with torch.no_grad():
        y, up = net(inputs().cuda())
        check_summary(y.cpu(), g, tag + ".x_out", rtol=3e-4, atol=2e-6)
        check_summary(up.float().cpu(), g, tag + ".up_x", rtol=3e-4, atol=2e-5)
        # the 16-bit inference paths (fused decoder loaders included) against the fp32 run
        for dtype, tol in (("bf16", 3e-2), ("fp16", 7e-3)):
            y16, _ = make(op, bil, dtype, upm).eval()(inputs().cuda())
            assert rel(y16.float().cpu(), y.cpu()) < tol, (dtype, rel(y16.float().cpu(), y.cpu()))


@pytest.mark.parametrize("tag,op,bil,upm", GENERATOR_VARIANTS)
def test_fp32_gradients_match_reference_golden(golden, tag, op, bil, upm):
    """fp32 parity mode: every parameter's gradient against the reference's norm and 64 sampled elements.  The gates are those of
    tests/test_gpu_backward.py's conditioning note: decoder / graph tensors 2e-3, encoder tensors (behind sqrt(x2 + 1e-8) where the
    operator has it) 3e-2"""
    g = golden("generator_variants")
    net = make(op, bil, "fp32", upm).train()
    net.drop_path_prob = 0.0
    wy = 0.5 + synth.smooth_hdr_frames(2, salt="bwy")
    y, up = net(inputs().cuda())
    ((y * wy.cuda()).sum() + 1e-3 * up.sum()).backward()
    bad = {}
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None and tuple(p.grad.shape) == tuple(p.shape), k
        gr = p.grad.double().reshape(-1).cpu()
        ref_n = float(g["%s.grad.%s" % (tag, k)])
        idx = torch.from_numpy(g["%s.gradpos.%s" % (tag, k)])
        rms = ref_n / max(gr.numel(), 1) ** 0.5
        e_samp = (gr[idx] - torch.from_numpy(g["%s.gradval.%s" % (tag, k)])).norm().item() / (len(idx) ** 0.5 * rms + 1e-30)
        gate = 2e-3 if (k.startswith("up_path") or k.startswith("outc") or k.startswith("gcn")) else 3e-2
        if abs(gr.norm().item() - ref_n) > gate * ref_n or e_samp > 3 * gate:
            bad[k] = (gr.norm().item(), ref_n, e_samp)
    assert not bad, bad


def test_bf16_training_step_of_a_variant_runs_and_descends():
    """bf16 fast path, square operator with the bilinear decoder: gradients land on the variant's own parameter shapes and an
    SGD step along them lowers the loss"""
    net = make("square", 1, "bf16").train()
    net.drop_path_prob = 0.0
    x = inputs().cuda()
    wy = (0.5 + synth.smooth_hdr_frames(2, salt="bwy")).cuda()

    def loss():
        y, up = net(x)
        return (y * wy).sum() + 1e-3 * up.float().sum()

    l0 = loss()
    l0.backward()
    with torch.no_grad():
        for p in net.parameters():
            if not p.requires_grad:
                continue
            assert p.grad is not None and p.grad.shape == p.shape
            p.add_(p.grad, alpha=-1e-7)
    assert loss().item() < l0.item()


@pytest.mark.parametrize("op,bil,upm", [("square", 1, 0), ("original_unet", 0, 0), ("square_root", 0, 1)])
def test_variant_gradients_through_the_data_parallel_reducer_equal_the_plain_pass(op, bil, upm):
    """Round 6: variant generators train data-parallel.  The reducer is handed the PUBLISHED-layout buffers and applies the variant's
    re-layout (slices of the skip-concat weights, tap sums of a bilinear `up`) after the collectives (distributed.GradReducer.keep(...,
    post)).  On one GPU over RCCL (world size 1, forced data-parallel path) the gradients that arrive in .grad after
    DistributedOptimizer.synchronize() must equal the plain backward pass's bit for bit (fp32 mode: deterministic)."""
    import os
    import torch.distributed as td
    from uncltmo_amd.distributed import DistributedOptimizer
    x = inputs().cuda()
    wy = (0.5 + synth.smooth_hdr_frames(2, salt="bwy")).cuda()
    grads = []
    os.environ["UNCL_FORCE_DIST"] = "1"
    td.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29900 + os.getpid() % 90), rank=0, world_size=1,
                          device_id=torch.device("cuda", 0))
    try:
        for wrap in (False, True):
            net = make(op, bil, "fp32", upm).train()
            net.drop_path_prob = 0.0
            opt = torch.optim.SGD([q for q in net.parameters() if q.requires_grad], lr=0.0)
            if wrap:
                opt = DistributedOptimizer(opt, module=net)
                assert net._grad_reducer.active()
            y, up = net(x)
            ((y * wy).sum() + 1e-3 * up.float().sum()).backward()
            if wrap:
                assert all(q.grad is None for q in net.parameters())          # the buffers belong to the reducer until step()
                opt.synchronize()
            torch.cuda.synchronize()
            grads.append({k: q.grad.clone() for k, q in net.named_parameters() if q.requires_grad})
            net = opt = None
        assert grads[0].keys() == grads[1].keys()
        for k in grads[0]:
            assert grads[0][k].shape == grads[1][k].shape and torch.equal(grads[0][k], grads[1][k]), k
    finally:
        import gc
        gc.collect()
        torch.cuda.synchronize()
        td.destroy_process_group()
        os.environ.pop("UNCL_FORCE_DIST", None)


def test_unsupported_variants_are_still_refused():
    with pytest.raises(NotImplementedError):
        UNet(1, 1, "sigmoid", 4, 3, "gamma", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)
    with pytest.raises(NotImplementedError):      # layer_factor must be the operator's member count (the reference would fail at the concat)
        UNet(1, 1, "sigmoid", 4, 4, "square", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)


def test_video_generator_variant_forward_and_backward_through_time_vs_oracle():
    """the recurrent generator (Unet.py:213-289) shares the decoder: `square` operator + bilinear path, fp32 mode, a two-frame clip
    of batch 2 against the oracle's autograd (forward 1e-4; decoder / graph gradients 2e-3, encoder 3e-2 as above)"""
    from oracle import generator as OG
    from uncltmo_amd.generator import UNetVideo
    op, bil = "square", 1
    net = UNetVideo(1, 1, "sigmoid", 4, params.get_layer_factor(op), op, 32, bil, "unet", 0, 0, "none", "none", "relu", 1, "replicate",
                    2, 0, compute_dtype="fp32")
    synth.fill_state_dict(net, "g0")
    net = net.cuda().train()
    net.drop_path_prob = 0.0
    x = torch.stack([torch.cat([synth.smooth_hdr_frames(1, salt="vv%d_%d" % (b, t)) for t in range(2)], 0) for b in range(2)], 0)
    wy = 0.5 + synth.smooth_hdr_frames(4, salt="vvw").reshape(2, 2, 1, 256, 256)
    y, f = net(x.cuda())
    ((y * wy.cuda()).sum() + 1e-2 * f.sum()).backward()
    sd = {k: v.detach().cpu().double().requires_grad_(v.requires_grad) for k, v in net.state_dict().items()}
    for k, p in net.named_parameters():
        sd[k].requires_grad_(p.requires_grad)
    yo, fo = OG.unet_video_forward(sd, x.double(), con_operator=op)       # (DropPath off on both sides: nothing else differs in training mode)
    assert rel(y.detach().cpu(), yo.detach()) < 1e-4 and rel(f.detach().cpu(), fo.detach()) < 1e-3
    ((yo * wy.double()).sum() + 1e-2 * fo.sum()).backward()
    bad = {}
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None and p.grad.shape == p.shape, k
        gate = 2e-3 if (k.startswith("up_path") or k.startswith("outc") or k.startswith("gcn")) else 3e-2
        e = rel(p.grad.cpu(), sd[k].grad)
        if e > gate:
            bad[k] = e
    assert not bad, bad
