"""Per-layer launch time of the generator forward (bf16, 200 tiles): HIP events around every launch of one packed-weight layer
at a time (uncl_prof_enable), a few steps each.  Run on the GPU box:  python tools/layer_times.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from uncltmo_amd import _hip, synth, tiler  # noqa: E402
from uncltmo_amd.generator import UNet  # noqa: E402

# (Cin, Cout, output H) per packed layer; FLOP = 2 * taps * Cin * Cout * Hout^2
SHAPES = {"inc.conv.conv1": (32, 32, 252, 9), "down_path.0.mpconv.1.conv": (32, 64, 124, 9), "down_path.0.mpconv.1.conv1": (64, 64, 122, 9),
          "down_path.1.mpconv.1.conv": (64, 128, 59, 9), "down_path.1.mpconv.1.conv1": (128, 128, 57, 9),
          "down_path.2.mpconv.1.conv": (128, 256, 26, 9), "down_path.2.mpconv.1.conv1": (256, 256, 24, 9),
          "down_path.3.mpconv.1.conv": (256, 256, 10, 9), "down_path.3.mpconv.1.conv1": (256, 256, 12, 9),
          "gcn.module.0.0.fc1.0": (256, 256, 12, 1), "gcn.module.0.0.graph_conv.gconv.nn.0": (128, 512, 12, 1),
          "gcn.module.0.0.fc2.0": (512, 256, 12, 1), "gcn.module.0.1.fc1.0": (256, 256, 12, 1), "gcn.module.0.1.fc2.0": (256, 256, 12, 1),
          "up_path.0.up": (256, 256, 24, 1), "up_path.0.conv.conv": (1024, 128, 26, 9), "up_path.0.conv.conv1": (128, 128, 28, 9),
          "up_path.1.up": (128, 128, 56, 1), "up_path.1.conv.conv": (512, 64, 59, 9), "up_path.1.conv.conv1": (64, 64, 61, 9),
          "up_path.2.up": (64, 64, 122, 1), "up_path.2.conv.conv": (256, 32, 124, 9), "up_path.2.conv.conv1": (32, 32, 126, 9),
          "up_path.3.up": (32, 32, 252, 1), "up_path.3.conv.conv": (128, 32, 254, 9), "up_path.3.conv.conv1": (32, 32, 256, 9)}


def main():
    if len(sys.argv) > 1:          # A/B on the same box: python tools/layer_times.py tools/_ab/libX.so
        _hip.LIB_PATH = os.path.abspath(sys.argv[1])
    lib = _hip.lib()
    lib.uncl_gen_set_streams(int(os.environ.get("UNCL_STREAMS", "1")))   # per-layer times are meaningful on one stream
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
               "replicate", 2, 0, compute_dtype="bf16")
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    nf = int(os.environ.get("UNCL_FRAMES", "8"))       # 4: the 100 tiles one of the two streams of the bench forward sees per launch
    frames = synth.hdr_frames(nf, 1024, 1024, salt="bench0").cuda()
    for _ in range(3):
        tiler.test_big_size_image2(frames, net, 0, 0, 0)
    buf = (ctypes.c_float * 64)()
    tot = 0.0
    for i in range(_hip.G_NUM_WEIGHTS):
        name = lib.uncl_gen_layer_name(i).decode()
        lib.uncl_prof_enable(i, 64)
        for _ in range(4):
            tiler.test_big_size_image2(frames, net, 0, 0, 0)
        torch.cuda.synchronize()
        n = lib.uncl_prof_read(buf, 64)
        ms = sum(buf[j] for j in range(n)) / max(n, 1)
        cin, cout, ho, taps = SHAPES[name]
        gf = 2.0 * taps * cin * cout * ho * ho * 25 * nf / 1e9
        tot += ms
        print("%2d %-42s %7.3f ms  %8.1f GFLOP  %7.1f TFLOP/s" % (i, name, ms, gf, gf / ms if ms > 0 else 0.0))
    lib.uncl_prof_enable(-1, 0)
    print("sum of the 26 packed layers: %.3f ms" % tot)


if __name__ == "__main__":
    main()
