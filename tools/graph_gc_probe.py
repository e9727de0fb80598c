"""Does destroying one captured optimisation step (StepGraph -> CUDAGraph and its private pool) while ANOTHER one is replaying
disturb the running one?  bench.py runs the image step and then the video step in one process; the first trainer sits in a
reference cycle, so the garbage collector used to pick the moment.  python tools/graph_gc_probe.py [when]  (when = replay
index at which gc.collect() runs without a device synchronisation, default 10; -1 = never)"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

when = int(sys.argv[1]) if len(sys.argv) > 1 else 10
a = bench.parse([])
rk = bench.Ranks(a)
gc.disable()
tr1, step1, _ = bench.make_trainer(rk, False)
for _ in range(10):
    step1()
torch.cuda.synchronize()
print("image step: graph", tr1._step_graph is not None, tr1._step_graph_error, flush=True)
del tr1, step1
torch.cuda.empty_cache()
tr2, step2, _ = bench.make_trainer(rk, True)
print("video step: graph", tr2._step_graph is not None, tr2._step_graph_error, flush=True)
for i in range(40):
    step2()
    if i == when:
        n = gc.collect()
        print("gc.collect() at replay %d freed %d objects" % (i, n), flush=True)
torch.cuda.synchronize()
print("OK errD %.4f" % float(tr2.errD), flush=True)
