"""GPU parity of the whole generator forward and of the tiler against the oracle and the reference's goldens."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from hip_util import rel_l2
from oracle import generator as OG
from oracle import tiler as OT
from uncltmo_amd import synth, tiler
from uncltmo_amd.generator import UNet, UNetVideo, gauss_stats

pytestmark = pytest.mark.gpu


def make_g(dtype, chunk=0):
    g = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
             "replicate", 2, 0, compute_dtype=dtype, chunk=chunk)
    synth.fill_state_dict(g, "g0")
    return g.cuda().eval()


def cpu_sd(g):
    return {k: v.detach().cpu() for k, v in g.state_dict().items()}


def golden_input():
    return torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB")], 0)


def test_generator_fp32_matches_reference_golden(golden):
    g = golden("generator")
    net = make_g("fp32")
    x = golden_input().cuda()
    with torch.no_grad():
        y, up = net(x)
        out2, knn = net.infer(x, want_knn=True)
    assert y.shape == (2, 1, 256, 256) and up.shape == (2, 32, 256, 256)
    # tolerance stated by the north star: 1e-4 rel-L2 against the reference's CPU output
    assert rel_l2(y.cpu(), torch.from_numpy(g["g_eval.x_out"])) < 1e-4
    assert torch.equal(out2, y)
    np.testing.assert_array_equal(knn.cpu().numpy().astype(np.int64), g["g_eval.knn_idx"])   # index work: exact
    check_summary(up.float().cpu(), g, "g_eval.up_x", rtol=2e-3, atol=2e-4)


def test_generator_fp32_vs_oracle_other_inputs():
    net = make_g("fp32", chunk=2)
    x = torch.cat([synth.smooth_hdr_frames(2, salt="o1"), synth.hdr_frames(1, salt="o2"), torch.zeros(1, 1, 256, 256),
                   torch.ones(1, 1, 256, 256)], 0)
    want = {}
    with torch.no_grad():
        y, up = net(x.cuda())
        _, knn = net.infer(x.cuda(), want_knn=True)
        y_ref, up_ref = OG.unet_image_forward(cpu_sd(net), x, want=want)
    assert rel_l2(y.cpu(), y_ref) < 1e-4
    assert rel_l2(up.float().cpu(), up_ref) < 1e-4
    # constant frames make every node's features position-only; the remaining ties are broken like torch.topk
    assert (knn.cpu().long() == want["knn_idx"]).float().mean().item() > 0.999


def test_generator_bf16_vs_oracle():
    net = make_g("bf16")
    x = golden_input()
    with torch.no_grad():
        y, up = net(x.cuda())
        y_ref, up_ref = OG.unet_image_forward(cpu_sd(net), x)
    # bf16 activations/weights with fp32 accumulation through 27 layers: stated tolerance 3e-2 rel-L2
    assert rel_l2(y.cpu(), y_ref) < 3e-2
    assert rel_l2(up.float().cpu(), up_ref) < 5e-2


def test_generator_fp16_vs_oracle_and_training_is_refused():
    """UNCL_F16 (BASELINE configs[4], "tiled UNet forward fp16"): the same kernels on v_mfma_f32_32x32x16_f16.  Ten mantissa
    bits instead of bf16's seven: the stated tolerance is 4x tighter.  fp16 is an inference dtype: asking for gradients raises."""
    net = make_g("fp16")
    x = golden_input()
    with torch.no_grad():
        y, up = net(x.cuda())
        y_ref, up_ref = OG.unet_image_forward(cpu_sd(net), x)
    assert up.dtype == torch.float16
    assert rel_l2(y.cpu(), y_ref) < 7e-3
    assert rel_l2(up.float().cpu(), up_ref) < 1.2e-2
    net.train()
    with pytest.raises(NotImplementedError):
        net(x.cuda())


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_large_batch_on_several_streams_equals_one_stream(dtype):
    """uncl_gen_forward spreads an un-chunked batch of >= 64 tiles over up to four streams (contiguous parts, forked and joined
    with events).  Per-tile results do not depend on the split, and the call keeps the caller's stream semantics."""
    from uncltmo_amd import _hip
    net = make_g(dtype)
    n = 135 if dtype == "bf16" else 65
    x = synth.hdr_frames(n, 256, 256, salt="two-streams").cuda()
    lib = _hip.lib()
    try:
        with torch.no_grad():
            _hip.check(lib.uncl_gen_set_streams(1), "set_streams")
            y1, k1 = net.infer(x, want_knn=True)
            y1 = y1.clone()
            _hip.check(lib.uncl_gen_set_streams(4), "set_streams")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # a non-default caller stream
                y2, k2 = net.infer(x, want_knn=True)
                y2 = y2.clone()
            torch.cuda.current_stream().wait_stream(side)
            _hip.check(lib.uncl_gen_set_streams(3), "set_streams")
            y3, _ = net.infer(x, want_knn=True)
    finally:
        lib.uncl_gen_set_streams(2)
    assert torch.equal(y1, y2) and torch.equal(k1, k2) and torch.equal(y1, y3)
    assert lib.uncl_gen_set_streams(0) != 0 and lib.uncl_gen_set_streams(5) != 0


def test_generator_train_mode_with_injected_droppath(golden):
    g = golden("generator")
    net = make_g("fp32")
    net.train()
    net.forced_drop_keep = [[1.0, 0.0], [1.0, 0.0]]
    with torch.no_grad():
        y, up = net(golden_input().cuda())
    check_summary(y.cpu(), g, "g_train.x_out", rtol=1e-3, atol=1e-5)


def test_generator_rejects_other_sizes_and_host_tensors():
    net = make_g("fp32")
    for hw in [(268, 268), (512, 512), (256, 512)]:
        with pytest.raises(ValueError):
            net(torch.zeros(1, 1, *hw, device="cuda"))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 256, 256))


def make_gv(dtype):
    g = UNetVideo(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
                  "replicate", 2, 0, compute_dtype=dtype)
    synth.fill_state_dict(g, "g0")
    return g.cuda().eval()


def test_knn_graph_many_samples_unsplit_rows():
    """DenseDilatedKnnGraph (gcn_lib/torch_edge.py:54-86,150-158) on more samples than a third of the CUs: the kernel then keeps
    a whole sample per workgroup.  Same indices as the split form on the same samples (each dot product is one sequential fma
    chain either way), and the neighbours are the nine smallest entries of the fp64 distance matrix."""
    import ctypes as C
    from uncltmo_amd import _hip
    lib = _hip.lib()
    n_s, nodes, ch, k = 100, 144, 256, 9
    x = torch.randn(n_s, nodes, ch, generator=torch.Generator().manual_seed(7)).cuda()
    rel = (torch.rand(nodes, nodes, generator=torch.Generator().manual_seed(8)) * 0.1).cuda()

    def run(xs, want_dist=False):
        idx = torch.empty(xs.shape[0], nodes, k, dtype=torch.int32, device="cuda")
        dist = torch.empty(xs.shape[0], nodes, nodes, device="cuda") if want_dist else None
        _hip.check(lib.uncl_gcn_knn(xs.data_ptr(), _hip.F32, rel.data_ptr(), idx.data_ptr(), dist.data_ptr() if want_dist else None,
                                    xs.shape[0], nodes, ch, k, None, _hip.stream_ptr()), "uncl_gcn_knn")
        torch.cuda.synchronize()
        return idx, dist

    whole, dist = run(x, True)
    parts = torch.cat([run(x[i:i + 25].contiguous())[0] for i in range(0, n_s, 25)], 0)
    assert torch.equal(whole, parts)
    xn = torch.nn.functional.normalize(x.double().cpu(), dim=-1)
    sq = (xn * xn).sum(-1, keepdim=True)
    ref = sq - 2 * xn @ xn.transpose(1, 2) + sq.transpose(1, 2) + rel.double().cpu()
    assert (dist.double().cpu() - ref).abs().max().item() < 1e-5
    got = torch.gather(ref, 2, whole.long().cpu())
    want = torch.topk(ref, k, dim=2, largest=False).values
    assert (got - want).abs().max().item() < 1e-5          # the same nine distances (ties / last-ulp swaps aside)


def test_gauss_stats_vs_torch():
    import torch.nn.functional as F
    x = synth.ldr_frames(3, 62, 70, salt="gs")                      # (3,1,62,70)
    win = OG.gauss_window()
    mu = F.conv2d(x, win)
    var = F.conv2d(x * x, win) - mu ** 2
    st = gauss_stats(x.reshape(3, 62, 70).cuda(), 3, 62, 70, 1).cpu()
    np.testing.assert_allclose(st[:, 0, 0].numpy(), x.mean(dim=(1, 2, 3)).numpy(), rtol=1e-5)
    np.testing.assert_allclose(st[:, 1, 0].numpy(), var.mean(dim=(1, 2, 3)).numpy(), rtol=2e-4)
    xc = torch.rand(2, 16, 40, 256, generator=torch.Generator().manual_seed(3))
    mu = F.conv2d(xc.reshape(32, 1, 40, 256), win)
    var = (F.conv2d(xc.reshape(32, 1, 40, 256) ** 2, win) - mu ** 2).reshape(2, 16, -1).mean(-1)
    st = gauss_stats(xc.permute(0, 2, 3, 1).contiguous().cuda(), 2, 40, 256, 16).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), xc.mean(dim=(2, 3)).numpy(), rtol=1e-5)
    np.testing.assert_allclose(st[:, 1].numpy(), var.numpy(), rtol=2e-4)


@pytest.mark.parametrize("n,h,w,c", [(2, 256, 256, 32), (3, 40, 200, 32), (2, 11, 11, 8), (1, 27, 12, 16), (8, 75, 256, 32), (1, 23, 13, 8)])
def test_gauss_stats_16bit_nhwc_vs_torch_fp64(n, h, w, c):
    """the 16-bit kernel (pointwise x^2 term, unrolled ring): every band length / trailing-row case, one-window maps, against fp64
    on the same bf16 values (Unet.py:112-123: compute_contrast + adaptive_avg_pool2d)"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n * 1000 + h + w)
    # (per-channel spread 0.2 ... 1 around 0.3: mean^2 / variance <= ~50; a flatter channel is ill-conditioned in fp32 for the
    # reference's own conv2d(x * x) - mu ** 2 as well -- the absolute term below is that conditioning, 16 eps32 E[x^2])
    xc = (0.3 + torch.rand(n, c, h, w, generator=g) * (0.2 + 0.8 * torch.rand(n, c, 1, 1, generator=g))).to(torch.bfloat16)
    win = OG.gauss_window().double()
    xd = xc.double().reshape(n * c, 1, h, w)
    mu = F.conv2d(xd, win)
    var = (F.conv2d(xd * xd, win) - mu ** 2).reshape(n, c, -1).mean(-1)
    st = gauss_stats(xc.permute(0, 2, 3, 1).contiguous().cuda(), n, h, w, c).cpu().double()
    np.testing.assert_allclose(st[:, 0].numpy(), xd.reshape(n, c, -1).mean(-1).numpy(), rtol=1e-5)
    e2 = (xd * xd).reshape(n, c, -1).mean(-1).numpy()
    err = np.abs(st[:, 1].numpy() - var.numpy())
    assert (err <= 2e-4 * np.abs(var.numpy()) + 16 * 5.96e-8 * e2).all(), (err / np.abs(var.numpy())).max()


def test_video_generator_fp32_matches_reference_golden(golden):
    g = golden("video")
    net = make_gv("fp32")
    x = torch.cat([synth.smooth_hdr_frames(1, salt="v%d" % t) for t in range(3)], 0).unsqueeze(0).cuda()
    with torch.no_grad():
        y, f = net(x)
    assert y.shape == (1, 3, 1, 256, 256) and f.shape == (1, 3, 64, 1, 1)
    for t in range(3):
        check_summary(y[:, t].cpu(), g, "v_eval.frame%d" % t, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(f.cpu().numpy(), g["v_eval.feats"], rtol=2e-3, atol=1e-6)


def test_video_generator_bf16_vs_oracle_batch2():
    net = make_gv("bf16")
    x = torch.stack([torch.cat([synth.smooth_hdr_frames(1, salt="w%d_%d" % (b, t)) for t in range(2)], 0)
                     for b in range(2)], 0)
    with torch.no_grad():
        y, f = net(x.cuda())
        y_ref, f_ref = OG.unet_video_forward(cpu_sd(net), x)
    assert rel_l2(y.cpu(), y_ref) < 3e-2
    assert rel_l2(f.cpu(), f_ref) < 5e-2


def _standin(p, **kw):
    yy = torch.arange(256.0, device=p.device).reshape(1, 1, 256, 1) / 255.0
    xx = torch.arange(256.0, device=p.device).reshape(1, 1, 1, 256) / 255.0
    return p * (0.5 + xx + 2.0 * yy), None


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_tiles_read_in_place_equal_the_gathered_ones(dtype, monkeypatch):
    """the tiler's gather folded into the first layer's loader (uncl_gen_run.x_tile_off): same tiles, same kernels, so the tone-mapped
    frames must be IDENTICAL to the gather-then-run path; odd frame sizes (unaligned tile origins, a last tile that overlaps its
    neighbour almost completely), several frames, and the two-stream split (>= 64 tiles)"""
    from uncltmo_amd import tiler
    net = make_g(dtype)
    for (F, H, W) in ((1, 300, 517), (3, 600, 1024), (8, 1024, 1024)):
        frames = synth.hdr_frames(F, H, W, salt="tip%d" % H).cuda()
        monkeypatch.setenv("UNCL_TILES_IN_PLACE", "0")
        ref = tiler.test_big_size_image2(frames, net, 0, 0, 0)
        monkeypatch.setenv("UNCL_TILES_IN_PLACE", "1")
        assert net.infer_frames(frames.reshape(F, H, W)) is not None
        got = tiler.test_big_size_image2(frames, net, 0, 0, 0)
        assert torch.equal(ref, got), (F, H, W)
    assert make_g("fp32").infer_frames(synth.hdr_frames(1, 300, 300, salt="tipf").cuda().reshape(1, 300, 300)) is None   # fp32: gathered


def test_tiler_blend_weights_exact(golden):
    g = golden("tiler")
    x = synth.hdr_frames(1, 272, 272, salt="tile272").cuda()
    y = tiler.test_big_size_image2(x, _standin, 0, 0, 0)
    np.testing.assert_allclose(y.cpu().numpy(), g["tiler.standin.272x272"], rtol=1e-6, atol=1e-7)
    x = synth.hdr_frames(1, 400, 528, salt="tile400").cuda()
    check_summary(tiler.test_big_size_image2(x, _standin, 0, 0, 0).cpu(), g, "tiler.standin.400x528", rtol=1e-6, atol=1e-7)
    assert tiler.tile_count(1024, 1024) == 25 and tiler.tile_count(2160, 3840) == 220
    with pytest.raises(ValueError):
        tiler.tile_count(256, 300)


def test_tiler_real_generator(golden):
    g = golden("tiler")
    net = make_g("fp32")
    x = synth.smooth_hdr_frames(1, 272, 272, salt="tileG").cuda()
    y = tiler.test_big_size_image2(x, net, 0, 0, 0)
    assert rel_l2(y.cpu(), torch.from_numpy(g["tiler.realG.272"])) < 1e-4


def test_tiled_forward_as_one_graph():
    """tiler.TiledGraph: gather -> generator -> cross-fade captured once as a hipGraph (incl. the multi-stream fork / join of
    large batches) and replayed on new frames: bit-identical to the eager path."""
    net = make_g("bf16")
    for n_frames, h, w in ((1, 528, 784), (3, 1040, 1040)):             # 6 and 3 x 36 tiles (>= 64: several streams)
        tg = tiler.TiledGraph(net, n_frames, h, w)
        for salt in ("g1", "g2"):
            fr = synth.hdr_frames(n_frames, h, w, salt=salt).cuda()
            want = tiler.test_big_size_image2(fr, net, 0, 0, 0)
            got = tg(fr)
            assert torch.equal(got, want)
        with pytest.raises(ValueError):
            tg(torch.zeros(n_frames, 1, h + 1, w, device="cuda"))


def test_tiler_full_size_properties():
    """1024^2 (25 tiles): with an identity 'model' the cross-fade must reproduce the frame exactly (the blend
    weights of every pixel sum to one), and tiling must commute with a per-pixel affine map."""
    x = synth.hdr_frames(2, 1024, 1024, salt="big").cuda()
    ident = lambda p, **kw: (p, None)
    y = tiler.test_big_size_image2(x, ident, 0, 0, 0)
    assert (y - x).abs().max().item() < 2e-6
    aff = lambda p, **kw: (0.25 * p + 0.5, None)
    y2 = tiler.test_big_size_image2(x, aff, 0, 0, 0)
    assert (y2 - (0.25 * x + 0.5)).abs().max().item() < 2e-6


@pytest.mark.parametrize("last_layer,activation", [("tanh", "relu"), ("msig", "relu"), ("none", "relu")])
def test_generator_other_last_layers(last_layer, activation):
    """The other heads the reference's constructor accepts with the published topology (Unet_singleFrame.py:207-212,
    models/Blocks.py:85-91): fp32 parity forward, bf16 backward.  (activation='leakyrelu' is accepted too and covered at
    the layer level; with the published skip operator it feeds negative values to sqrt(x + 1e-8) and both the reference
    and this path produce NaN.)"""
    from uncltmo_amd.generator import UNet
    args = (1, 1, last_layer, 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", activation, 1, "replicate", 2, 0)
    net = UNet(*args, compute_dtype="fp32")
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    x = synth.smooth_hdr_frames(2, salt="ll")
    with torch.no_grad():
        y, up = net(x.cuda())
        y_ref, up_ref = OG.unet_image_forward(cpu_sd(net), x, last_layer=last_layer, activation=activation)
    assert rel_l2(y.cpu(), y_ref) < 1e-4 and rel_l2(up.float().cpu(), up_ref) < 1e-4
    # backward through the head (bf16 path): gradient of the 1x1 head's parameters against the oracle's autograd
    netb = UNet(*args, compute_dtype="bf16")
    synth.fill_state_dict(netb, "g0")
    netb = netb.cuda().eval()
    wy = 0.5 + synth.smooth_hdr_frames(2, salt="llw")
    yb, _ = netb(x.cuda())
    (yb * wy.cuda()).sum().backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(not k.endswith("relative_pos")) for k, v in netb.state_dict().items()}
    yo, _ = OG.unet_image_forward(sd, x, last_layer=last_layer, activation=activation)
    (yo * wy).sum().backward()
    named = dict(netb.named_parameters())
    for k in ("outc.conv.weight", "outc.conv.bias", "up_path.3.conv.conv1.weight", "inc.conv.conv.weight"):
        assert rel_l2(named[k].grad.cpu(), sd[k].grad) < 6e-2, k


def test_fused_graph_tail_matches_separate_kernels():
    """uncl_gcn_tail (max-relative gather + grouped conv + fc2 + FFN in one launch, intermediates in LDS) against the
    gather kernel and the four 1x1 convolutions it replaces: same rounding points, so the generator's output must agree
    to 16-bit rounding noise, for bf16 and fp16 and for a batch that is not a multiple of anything."""
    from uncltmo_amd import _hip
    lib = _hip.lib()
    for dt in ("bf16", "fp16"):
        net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0,
                   compute_dtype=dt)
        synth.fill_state_dict(net, "g0")
        net = net.cuda().eval()
        x = synth.hdr_frames(7, 256, 256, salt="fused-gcn").cuda()
        old = lib.uncl_gen_set_fused_graph(0)
        try:
            with torch.no_grad():
                y0, k0 = net.infer(x, want_knn=True)
                y0, k0 = y0.clone(), k0.clone()
                for mode in (2, 1):          # 2: fc1 and kNN as launches + fused tail; 1: the whole block in one launch
                    lib.uncl_gen_set_fused_graph(mode)
                    y1, k1 = net.infer(x, want_knn=True)
                    assert torch.equal(k0, k1), (dt, mode)
                    assert rel_l2(y1.float().cpu(), y0.float().cpu()) < 2e-3, (dt, mode)
                    assert (y1.float() - y0.float()).abs().max().item() < 2e-2, (dt, mode)
        finally:
            lib.uncl_gen_set_fused_graph(old)
