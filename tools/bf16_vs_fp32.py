"""bf16 fast path vs fp32 parity path of the tiled generator on two 1024x1024 frames (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from uncltmo_amd import synth, tiler
from uncltmo_amd.generator import UNet
args=(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)
outs={}
for dt in ("fp32","bf16"):
    net=UNet(*args, compute_dtype=dt); synth.fill_state_dict(net,"g0"); net=net.cuda().eval()
    x=synth.smooth_hdr_frames(2, 1024, 1024, salt="num").cuda()
    outs[dt]=tiler.test_big_size_image2(x, net, 0, 0, 0).double()
d=(outs["bf16"]-outs["fp32"])
print("rel-L2 %.3e  max|d| %.3e  mean|d| %.3e  (8-bit LSB = %.3e)" % ((d.norm()/outs["fp32"].norm()).item(), d.abs().max().item(), d.abs().mean().item(), 1/255))
