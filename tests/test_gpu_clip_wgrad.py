"""Clip layout of the video generator's training pass (uncl_gen_run.clip_T / uncl_gen_bwd.clip_T, uncltmo_amd/autograd.py): the
frames of a clip in ONE workspace, the 3x3 / 2x2 weight and bias gradients taken once per clip over all T * B samples instead of
once per frame.  The reference computes them per frame through autograd (Unet.py:213-289 under GanTrainer.py:338,460); the sums
are the same sums in another order, so the two forms must agree to summation-order rounding -- and both are gated against the
oracle's autograd in tests/test_gpu_backward.py (bf16 takes the clip form by default there)."""
import pytest
import torch

from hip_util import rel_l2
from uncltmo_amd import synth
from uncltmo_amd.generator import UNetVideo

pytestmark = pytest.mark.gpu


def _net(dtype):
    net = UNetVideo(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
                    "replicate", 2, 0, compute_dtype=dtype)
    synth.fill_state_dict(net, "g0")
    return net.cuda().eval()


def _pass(net, clip, B, T, last_only=False, twice=False):
    net.clip_wgrad = clip
    net.zero_grad()
    x = synth.smooth_hdr_frames(B * T, salt="cw").reshape(B, T, 1, 256, 256)
    wy = (0.5 + synth.smooth_hdr_frames(B * T, salt="cwy")).reshape(B, T, 1, 256, 256)
    wf = torch.from_numpy(synth.hash_uniform("cwf", 64).copy()).reshape(1, 1, 64, 1, 1) * 10.0
    y, ft = net(x.cuda())
    if last_only:
        loss = (y[:, T - 1] * wy[:, T - 1].cuda()).sum()
    else:
        loss = (y * wy.cuda()).sum() + (ft * wf.cuda()).sum()
    if twice:       # the reference's two backward calls on one graph (GanTrainer.py:338 retain_graph=True, :460)
        loss.backward(retain_graph=True)
        loss.backward()
    else:
        loss.backward()
    torch.cuda.synchronize()
    return y.detach().clone(), ft.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dtype,B,T,tol", [("fp32", 1, 3, 2e-5), ("bf16", 2, 3, 1e-4), ("bf16", 1, 5, 1e-4), ("bf16", 3, 2, 1e-4)])
def test_clip_weight_gradients_equal_per_frame_weight_gradients(dtype, B, T, tol):
    net = _net(dtype)
    y0, f0, g0 = _pass(net, False, B, T)
    y1, f1, g1 = _pass(net, True, B, T)
    # the forward runs the same kernels on the same values at other addresses
    assert torch.equal(y0, y1) and torch.equal(f0, f1)
    assert len(g0) == len(g1) == 57
    errs = {k: rel_l2(g1[k].cpu(), g0[k].cpu()) for k in g0}
    for k, r in errs.items():
        print("%-45s %.2e" % (k, r))
    bad = {k: r for k, r in errs.items() if not r < tol}
    assert not bad, bad


def test_clip_form_last_frame_loss_and_two_backward_calls():
    """loss on the last frame only (the earlier frames' gradients arrive through the carries, their g_out is zero), and the
    reference's two backward() calls on one graph: the second pass must not see anything the first left in the arenas"""
    net = _net("bf16")
    _, _, g0 = _pass(net, False, 1, 3, last_only=True)
    _, _, g1 = _pass(net, True, 1, 3, last_only=True)
    bad = {k: rel_l2(g1[k].cpu(), g0[k].cpu()) for k in g0}
    assert all(r < 1e-4 for r in bad.values()), bad
    _, _, g2 = _pass(net, True, 1, 3, last_only=True, twice=True)
    bad = {k: rel_l2(g2[k].cpu(), 2.0 * g1[k].cpu()) for k in g0}
    assert all(r < 1e-4 for r in bad.values()), bad


def test_clip_layout_refused_where_it_does_not_apply():
    from uncltmo_amd import _hip
    net = _net("bf16")
    x = synth.smooth_hdr_frames(1, salt="cw").reshape(1, 256, 256).cuda()
    with pytest.raises(_hip.HipError):
        net._run(x, need_feat=True, keep_act=False, clip=(3, 0))
    with pytest.raises(_hip.HipError):       # frame index out of range: refused by the library
        net._run(x, need_feat=True, keep_act=True, save_preact=True, clip=(3, 3))


def test_clip_form_under_the_data_parallel_reducer_and_in_a_captured_step():
    """The video trainer's step with the clip form: (a) under DistributedOptimizer(module=G) over RCCL at world size 1 -- the
    decoder-done event is recorded by the DEFERRED pass (after the clip's decoder weight gradients, not inside frame 0's
    data-gradient chain), the decoder half is unpacked and reduced on the side stream from there -- the parameters after two
    steps must equal the plain steps' exactly (fp32 mode: deterministic); (b) the step replays as one hipGraph with the clip
    workspace / arena at fixed addresses, bit for bit.  (The per-frame form is compared gradient by gradient above.)"""
    import os
    import torch.distributed as td
    from test_gpu_trainer import _fp32_trainer, step_inputs
    from uncltmo_amd.distributed import DistributedOptimizer
    from uncltmo_amd.step_graph import StepGraph
    hdr, pos, neg = step_inputs()

    def run(clip, wrap, graph=False):
        tr, G, D = _fp32_trainer(True)
        G.clip_wgrad = clip
        if wrap:
            tr.optimizerG = DistributedOptimizer(tr.optimizerG, module=G)
            tr.optimizerD = DistributedOptimizer(tr.optimizerD)
        sg = None
        if graph:
            sg = StepGraph(tr, hdr, hdr.clone(), pos, neg, 0, warmup=2)
            for _ in range(2):
                sg.replay()
        else:
            for _ in range(2):
                tr.train_D(hdr, pos, neg, 0)
                tr.train_G(hdr, hdr.clone(), pos, neg, 0)
        torch.cuda.synchronize()
        out = {k: v.detach().clone() for k, v in G.state_dict().items()}
        sg = None
        return out

    plain = run(True, False)
    replayed = run(True, False, graph=True)
    for k in plain:
        assert torch.equal(plain[k], replayed[k]), k
    os.environ["UNCL_FORCE_DIST"] = "1"
    td.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29300 + os.getpid() % 200), rank=0, world_size=1,
                          device_id=torch.device("cuda", 0))
    try:
        reduced = run(True, True)
        for k in plain:
            assert torch.equal(plain[k], reduced[k]), k
    finally:
        import gc
        gc.collect()
        torch.cuda.synchronize()
        td.destroy_process_group()
        os.environ.pop("UNCL_FORCE_DIST", None)
