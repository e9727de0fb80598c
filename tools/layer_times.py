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

import bench  # noqa: E402   (the layer table: SURVEY 8(d)'s algorithmic GFLOP per layer and what the implicit GEMM multiplies)


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
    table = bench.layer_gflops(25 * nf)       # GFLOP: the survey's convention (transposed 3x3 layers over their input pixels)
    for i in range(_hip.G_NUM_WEIGHTS):
        name = lib.uncl_gen_layer_name(i).decode()
        lib.uncl_prof_enable(i, 64)
        for _ in range(4):
            tiler.test_big_size_image2(frames, net, 0, 0, 0)
        torch.cuda.synchronize()
        n = lib.uncl_prof_read(buf, 64)
        ms = sum(buf[j] for j in range(n)) / max(n, 1)
        row = table[i]
        assert row[0] == name, (row[0], name)
        gf = row[1]
        tot += ms
        print("%2d %-42s %7.3f ms  %8.1f GFLOP  %7.1f TFLOP/s  (%.3f of 2.5 PF; executed %.1f GFLOP)" % (
            i, name, ms, gf, gf / ms if ms > 0 else 0.0, gf / ms / 2500.0 if ms > 0 else 0.0, row[2]))
    lib.uncl_prof_enable(-1, 0)
    print("sum of the 26 packed layers: %.3f ms" % tot)


if __name__ == "__main__":
    main()
