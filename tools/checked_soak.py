"""The hot path under the CHECKED library (`python -c "import __graft_entry__ as g; g.build_checked()"` first):
every global access of the 3x3 convolution, weight-gradient and 2x2 up-conv kernels is looked up in the table of tensors its
launch was given.  Runs the headline forward (8 x 1024^2, two streams), a 4K fp16 frame, the fused last stage, and `steps`
image / video optimisation steps (replayed and eager), then prints uncl_checked_report(): violations must be 0.
  python tools/checked_soak.py [steps]
  UNCL_CHECKED_SHRINK=64 python tools/checked_soak.py 2      # positive control: every tensor registered 64 bytes short -> violations"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UNCL_HIP_LIB", os.path.join(ROOT, "uncltmo_amd", "libuncltmo_hip_checked.so"))
os.environ["UNCL_BENCH_WGRAD"] = "0"
import torch  # noqa: E402
import bench  # noqa: E402
from uncltmo_amd import _hip, synth, tiler  # noqa: E402
from uncltmo_amd.generator import UNet  # noqa: E402


def report(tag):
    out = (C.c_ulonglong * 4)()
    rc = _hip.lib().uncl_checked_report(out, 0)
    assert rc == 0, "not a checked library: %s" % _hip.LIB_PATH
    print("%-34s violations %d  first address 0x%x  source line %d  bytes %d" % (tag, out[0], out[1], out[2], out[3]), flush=True)
    return out[0]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    lib = _hip.lib()
    print("library:", _hip.LIB_PATH, flush=True)
    report("start")
    for dt, hw, salt in (("bf16", (1024, 1024), "bench0"), ("fp16", (2160, 3840), "k4")):
        net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
                   "replicate", 2, 0, compute_dtype=dt)
        synth.fill_state_dict(net, "g0")
        net = net.cuda().eval()
        frames = synth.hdr_frames(8 if dt == "bf16" else 1, hw[0], hw[1], salt=salt).cuda()
        for fused in (0, 1):
            lib.uncl_gen_set_fused_tail(fused)
            for _ in range(2):
                out = tiler.test_big_size_image2(frames, net, 0, 0, 0)
            assert torch.isfinite(out).all()
            report("forward %s %dx%d fused_tail=%d" % (dt, hw[0], hw[1], fused))
        lib.uncl_gen_set_fused_tail(0)
        del net, frames, out
    a = bench.parse([])
    rk = bench.Ranks(a)
    for video in (False, True):
        tr, step, n = bench.make_trainer(rk, video)
        for mode, fn in (("replay", step), ("eager", tr._eager_step)):
            if fn is None:
                continue
            for i in range(steps if not video else max(steps // 3, 5)):
                fn()
            vals = [float(tr.errD), float(tr.errG_d), float(tr.errG_struct)]
            assert all(v == v and abs(v) < 1e6 for v in vals), (mode, vals)
            report("%s step %s" % ("video" if video else "image", mode))
        del tr, step
    bad = report("end")
    if os.environ.get("UNCL_CHECKED_SHRINK"):
        print("control run (tensors registered %s bytes short): %d violations reported -- the checks are live" % (os.environ["UNCL_CHECKED_SHRINK"], bad)
              if bad else "CONTROL FAILED: nothing reported", flush=True)
        return 0 if bad else 1
    print("OK: no access outside its launch's tensors" if bad == 0 else "VIOLATIONS", flush=True)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
