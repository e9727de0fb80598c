"""HBM ceilings of this box as PyTorch's own element-wise kernels see them: write-only (fill), read-only (sum), read+write
(copy) on 2 GiB buffers -- the yardsticks for the store-heavy layers.  python tools/mem_probe.py"""
import torch

n = 1 << 30          # bf16 elements: 2 GiB
x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
y = torch.empty(n, dtype=torch.bfloat16, device="cuda")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 2 / 1e9
t = timed(lambda: x.fill_(1.0))
print("fill  (write only): %.3f ms  %.2f TB/s" % (t, gb / t))
t = timed(lambda: y.copy_(x))
print("copy  (read+write): %.3f ms  %.2f TB/s (bytes moved: 2x)" % (t, 2 * gb / t))
xs = x.view(torch.int16)
t = timed(lambda: xs.sum())
print("sum   (read only) : %.3f ms  %.2f TB/s" % (t, gb / t))
xf = x.view(torch.float32)
t = timed(lambda: xf.fill_(1.0))
print("fill fp32         : %.3f ms  %.2f TB/s" % (t, gb / t))
