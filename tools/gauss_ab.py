"""Time uncl_gauss_stats / uncl_gauss_stats_backward on the generator's feature map (N, 256, 256, 32) bf16, the per-frame calls of
the video step (N = 8) and the image batch (N = 32).  UNCL_GAUSS_H16=0 selects the generic forward kernel (A/B)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from uncltmo_amd import _hip  # noqa: E402
from uncltmo_amd.generator import gauss_stats  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    lib = _hip.lib()
    for n in ([int(a) for a in sys.argv[1:]] or [8, 32]):
        x = (0.3 + torch.rand(n, 256, 256, 32, device="cuda")).to(torch.bfloat16)
        gst = torch.randn(n, 2, 32, device="cuda")
        gx = torch.empty_like(x)
        fwd = timed(lambda: gauss_stats(x, n, 256, 256, 32))
        bwd = timed(lambda: _hip.check(lib.uncl_gauss_stats_backward(x.data_ptr(), _hip.BF16, gst.data_ptr(), gx.data_ptr(), n, 256,
                                                                       256, 32, 0, _hip.stream_ptr()), "bwd"))
        mb = x.numel() * 2 / 1e6
        print("N=%2d  forward %.1f us (%.2f TB/s of the %.0f MB read once)   backward %.1f us (%.2f TB/s read + write)"
              % (n, fwd, mb / fwd / 1e0 * 1e-6 * 1e6 / 1e6, mb, bwd, 2 * mb / bwd / 1e6 * 1e6 / 1e6), flush=True)


if __name__ == "__main__":
    main()
