"""Latency of ONE 1024x1024 frame (25 tiles) and of one 2160x3840 frame (220 tiles) through the tiled generator, bf16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncltmo_amd import synth, tiler
from uncltmo_amd.generator import UNet
net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0,
           compute_dtype="bf16")
synth.fill_state_dict(net, "g0")
net = net.cuda().eval()
for (h, w) in ((1024, 1024), (2160, 3840)):
    fr = synth.hdr_frames(1, h, w, salt="lat").cuda()
    for _ in range(3):
        out = tiler.test_big_size_image2(fr, net, 0, 0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = tiler.test_big_size_image2(fr, net, 0, 0, 0)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%dx%d: %.2f ms per frame (synchronous), %s" % (h, w, dt * 1e3, tuple(out.shape)))
    tg = tiler.TiledGraph(net, 1, h, w)
    ref = tiler.test_big_size_image2(fr, net, 0, 0, 0)
    got = tg(fr)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    t0 = time.perf_counter()
    for _ in range(n):
        out = tg(fr)
        torch.cuda.synchronize()
    print("%dx%d: %.2f ms per frame as one hipGraph replay" % (h, w, (time.perf_counter() - t0) / n * 1e3))
