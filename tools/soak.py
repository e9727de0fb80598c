"""Soak: many optimisation steps in one process (replayed graph and eager), finite losses throughout.
  python tools/soak.py [steps] [video]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
video = len(sys.argv) > 2 and sys.argv[2] == "video"
a = bench.parse([])
rk = bench.Ranks(a)
tr, step, n = bench.make_trainer(rk, video)
print("graph replay:", tr._step_graph is not None, tr._step_graph_error, flush=True)
for mode, fn in (("replay", step), ("eager", tr._eager_step)):
    if fn is None:
        continue
    t0 = time.time()
    for i in range(steps):
        fn()
        if i % 100 == 99:
            vals = [float(tr.errD), float(tr.errG_d), float(tr.errG_struct)]
            assert all(v == v and abs(v) < 1e6 for v in vals), (mode, i, vals)
            print("%s step %d  errD %.4f errG_d %.4f errG_struct %.4f" % (mode, i + 1, *vals), flush=True)
    torch.cuda.synchronize()
    print("%s: %d steps, %.2f ms per step (wall)" % (mode, steps, (time.time() - t0) * 1e3 / steps), flush=True)
bad = [k for k, p in tr.netG.named_parameters() if not torch.isfinite(p).all()]
assert not bad, bad
print("OK: all generator parameters finite", flush=True)
