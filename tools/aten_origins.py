"""Where the PyTorch-launched kernels of one optimisation step come from: one eager image (or --video) step under torch.profiler
with python stacks; every device kernel that is not one of this library's is printed with the aten operator that launched it and
the innermost frames of uncltmo_amd / autograd that called it.  Run on the GPU box:  python tools/aten_origins.py [--video]"""
import collections
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from torch.profiler import ProfilerActivity, profile
    import bench
    os.environ["UNCL_TRAIN_GRAPH"] = "0"
    a = bench.parse(["--mode", "train"])
    rk = bench.Ranks(a)
    tr, step, _ = bench.make_trainer(rk, "--video" in sys.argv)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    ev = prof.events()
    # cpu ops that launched kernels: walk each kernel's launching op
    rows = collections.OrderedDict()
    n_k = 0
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        # skip ops whose parent also reports the same kernels (keep the innermost op)
        if any(c.kernels for c in e.cpu_children):
            continue
        for k in e.kernels:
            if "anonymous namespace" in k.name and "at::native" not in k.name:
                continue
            if not ("at::native" in k.name or "rocclr" in k.name or "Memcpy" in k.name or "Memset" in k.name):
                continue
            n_k += 1
            st = [s for s in (e.stack or []) if "uncltmo_amd" in s or "bench.py" in s][:3]
            key = (e.name, str(e.input_shapes)[:60], tuple(st))
            r = rows.setdefault(key, [0, 0.0, k.name[:70]])
            r[0] += 1
            r[1] += k.duration
    print("%d PyTorch-launched kernels in the step" % n_k)
    for (name, shapes, st), (cnt, dur, kn) in rows.items():
        print("%2d x %-28s %-60s %6.1f us  %s" % (cnt, name[:28], shapes, dur, kn))
        for s in st:
            print("        " + s.replace(ROOT + "/", "")[:150])


if __name__ == "__main__":
    main()
