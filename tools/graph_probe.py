"""Eager tiled forward vs hipGraph replay (tiler.TiledGraph) at the bench shape: python tools/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncltmo_amd import synth, tiler
from uncltmo_amd.generator import UNet
net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0,
           compute_dtype="bf16")
synth.fill_state_dict(net, "g0")
net = net.cuda().eval()
frames = synth.hdr_frames(8, 1024, 1024, salt="bench0").cuda()
def timed(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3, out
e, o1 = timed(lambda: tiler.test_big_size_image2(frames, net, 0, 0, 0))
g = tiler.TiledGraph(net, 8, 1024, 1024)
r, o2 = timed(lambda: g(frames))
print("eager %.3f ms, graph replay %.3f ms, identical %s" % (e, r, torch.equal(o1, o2)))
